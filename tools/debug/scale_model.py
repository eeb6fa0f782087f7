"""What one rank of an N-GPU tile-sharded frame costs, measured on ONE GPU: mi_render(rank, world = N) renders exactly the tiles rank r of N owns
(t % N == r), so its time on this device is the per-rank time of the N-GPU job (scene replicated, no collective on the data path); the film
reduction that follows (33 MB at 1080p) is a sub-millisecond RCCL reduce.  Predicted strong-scaling speed-up = T(1) / max_r T(r, N)."""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pa = importlib.import_module("pbrt-v3-distributed_amd")
d = "/tmp/pbrt_amd_scale"; f = os.path.join(d, "s.pbrt")
if not os.path.exists(f):
    os.makedirs(d, exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", "10000000", "--res", "1920", "1080", "--spp", "64", "--out", f], stdout=subprocess.DEVNULL)
sc = pa.Scene(f); ctx = pa.Context(sc)
def frame(rank, world):
    ctx.film_clear(); ctx.sync()
    t0 = time.perf_counter(); ctx.render(rank=rank, world=world, sync=True); return (time.perf_counter() - t0) * 1e3
frame(0, 1)
out = {}
for world in (1, 2, 4, 8):
    ts = []
    for rank in sorted(set([0, world // 2, world - 1])):
        frame(rank, world)                      # re-uploads the tile list / re-sizes the path pool
        ts.append(min(frame(rank, world), frame(rank, world)))
    out[world] = max(ts)
t1 = out[1]
res = {"ms_per_rank_frame": {str(k): round(v, 2) for k, v in out.items()}, "predicted_speedup": {str(k): round(t1 / v, 2) for k, v in out.items()},
       "note": "C3 frame (10 M triangles, 1080p, 64 spp); slowest of ranks {0, N/2, N-1}; excludes the film reduce (33 MB, < 1 ms over xGMI) and per-rank scene load"}
print(json.dumps(res))
