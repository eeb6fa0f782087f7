#!/usr/bin/env python3
"""Share of the interior-node visits served from the LDS copy of the hot nodes, per traversal kernel (closest hit / any hit / MIS), from one counting frame (GPU box).
Usage: hot_share.py [--tris N] [--spp S] [--textured] [--leafmask]"""
import argparse, importlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000); ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--config", default="sanmiguel")
args = ap.parse_args()
pa = importlib.import_module("pbrt-v3-distributed_amd")
d = tempfile.mkdtemp(prefix="hot_share_", dir="/tmp")
f = os.path.join(d, "s.pbrt")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), args.config, "--tris", str(args.tris), "--res", "1920", "1080", "--spp", str(args.spp), "--out", f], stdout=subprocess.DEVNULL)
sc = pa.Scene(f, strict=True)
ctx = pa.Context(sc, device=0)
ctx.counters_reset()
ctx.render(count_work=True)
c = ctx.counters()
for kind in ("closest", "any", "mis"):
    n, h = c["nodes_" + kind], c["nodes_hot_" + kind]
    rays = {"closest": c["closest_rays"] - c["mis_rays"], "any": c["shadow_rays"], "mis": c["mis_rays"]}[kind]
    print("%-8s rays %12d  node visits per ray %6.2f  from LDS %5.1f %%  triangle tests per ray %5.2f" % (kind, rays, n / max(1, rays), 100.0 * h / max(1, n), c["tris_" + kind] / max(1, rays)))
ctx.close()
