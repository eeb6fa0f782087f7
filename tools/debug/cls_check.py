"""Developer check (round 5): the material-class instances of k_shade (PBRT_AMD_SHADE_CLASSES, default on) against the one generic launch (=0) -- films compared BIT FOR BIT on
a set of scenes, plus the counters (same rays, same segments).  Runs against whatever PBRT_AMD_DEVICE_LIB selects (the GPU library on a GPU box, tools/hostemu otherwise).
usage: cls_check.py [scene.pbrt ...]   (default: cornell, materials, generated San-Miguel-class / bathroom-class stand-ins at small sizes, with and without --subsurface)"""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pa = importlib.import_module("pbrt-v3-distributed_amd")

def render(scene_file, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sc = pa.Scene(filename=scene_file)
        ctx = pa.Context(sc, device=0)
        ctx.render()
        film = np.array(ctx.film(), copy=True)
        cnt = ctx.counters()
        ctx.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return film, cnt

def gen(kind, out, *args):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), kind, "--out", out] + list(args), stdout=subprocess.DEVNULL)
    return out

scenes = sys.argv[1:]
tmp = tempfile.mkdtemp(prefix="cls_check_")
if not scenes:
    scenes = [os.path.join(ROOT, "scenes", "cornell.pbrt"), os.path.join(ROOT, "scenes", "materials.pbrt"),
              gen("sanmiguel", os.path.join(tmp, "sm.pbrt"), "--tris", "60000", "--res", "96", "64", "--spp", "8"),
              gen("sanmiguel", os.path.join(tmp, "sm_sss.pbrt"), "--tris", "60000", "--res", "96", "64", "--spp", "4", "--subsurface"),
              gen("bathroom", os.path.join(tmp, "bath.pbrt"), "--tris", "20000", "--res", "96", "64", "--spp", "8", "--maxdepth", "12"),
              gen("sanmiguel", os.path.join(tmp, "sm_tex.pbrt"), "--tris", "60000", "--res", "96", "64", "--spp", "4", "--textured", "--leafmask")]
bad = 0
for s in scenes:
    a, ca = render(s, {"PBRT_AMD_SHADE_CLASSES": "0"})
    b, cb = render(s, {"PBRT_AMD_SHADE_CLASSES": "1", "PBRT_AMD_VERBOSE": "1"})
    same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    keys = ("camera_rays", "closest_rays", "shadow_rays", "path_segments")
    cs = all(ca.get(k) == cb.get(k) for k in keys)
    print("%-40s film bit-identical: %s   counters equal: %s   %s" % (os.path.basename(s), same, cs, {k: cb.get(k) for k in keys}))
    if not same:
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        print("    max abs diff %.3g, differing values %d of %d" % (d.max(), int((a.view(np.uint32) != b.view(np.uint32)).sum()), a.size))
    bad += (not same) or (not cs)
sys.exit(1 if bad else 0)
