"""Developer check of row f4 (volpath / BSSRDF) against the reference fixtures and the oracle.  Runs against whatever
PBRT_AMD_DEVICE_LIB points at (the host emulation here, the real library on a GPU box)."""
import os, sys, importlib, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pa = importlib.import_module("pbrt-v3-distributed_amd")
import oracle_lib as ol
import edge_scenes
G = os.path.join(ROOT, "tests", "golden")
names = sys.argv[1:] or (edge_scenes.VOL_NAMES + edge_scenes.SSS_NAMES)
bad = 0
for name in names:
    t0 = time.time()
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    rng = np.random.default_rng(7)
    n = 1500
    xy = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n)], axis=1).astype(np.int32)
    s = rng.integers(0, sc.info["spp"], n).astype(np.int32)
    dev, ref = ctx.li(xy, s), ol.li(sc, xy, s)
    err = np.linalg.norm(dev - ref, axis=1)
    ok = err <= 1e-4 * (1 + np.linalg.norm(ref, axis=1))
    exact = (dev == ref).all(axis=1).mean()
    ctx.render()
    img = sc.film_image(ctx.film())
    fx = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    frac, relmse = ol.image_metrics(img, fx)
    cnt = ctx.counters()
    print("%-10s li within tol %.4f (bit-identical %.4f, max err %.3g) | image vs reference fixture: pixels within tol %.4f relMSE %.3g | guard %d | %.1fs" %
          (name, ok.mean(), exact, err.max(), frac, relmse, cnt["trace_guard_trips"], time.time() - t0), flush=True)
    if ok.mean() < 0.995 or frac < 0.995 or relmse > 1e-4: bad += 1
    if ok.mean() < 0.995:
        w = np.where(~ok)[0][:5]
        for k in w: print("   sample", xy[k], s[k], "dev", dev[k], "ref", ref[k])
    ctx.close()
sys.exit(1 if bad else 0)
