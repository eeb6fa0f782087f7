#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X wavefront path tracer (BASELINE.json metric:
Msamples/sec (+ Mrays/sec), San Miguel 1080p; the scene itself is not available offline, so the
workload is the San-Miguel-class synthetic stand-in of SURVEY.md s.8(d), generated deterministically).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

A "step" = one full pass of the hot path (SamplerIntegrator::Render: 1920x1080 x 64 spp, path maxdepth 5)
over the frame with the scene already resident in HBM.  With N>1 the 16x16 image tiles are sharded
round-robin over the ranks (scene replicated, no collective on the data path) and the FilmTile buffers
are combined on rank 0 with one RCCL reduction per step (disjoint tiles: sum == gather).
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec)
NODE_BYTES, TRI_BYTES, RAY_BYTES = 128, 48, 48   # algorithmic bytes: BVH4 node, triangle record, ray in + hit out


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU box shows 256 hardware
    threads but grants a container 16 CPUs' worth of time -- more threads than that only throttle each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--res", type=int, nargs=2, default=[1920, 1080])
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--scene", default=None, help="render this .pbrt instead of the generated stand-in")
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4"],
                    help="BASELINE.json config: c3 = San-Miguel-class 1080p 64spp (default, the metric's workload); "
                         "c2 = killeroo 1080p 128spp; c4 = bathroom-class 1080p 256spp maxdepth 30")
    ap.add_argument("--textured", action="store_true", help="c3 only: the stand-in with image-mapped / bump-mapped materials (SURVEY.md s.8 row f2)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target length of the CPU baseline sample (0 = skip)")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 path on a 1-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="testing aid: every rank uses GPU 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    # ---- scene (generated once per node by local rank 0)
    bench_dir = os.environ.get("PBRT_AMD_BENCH_DIR", "/tmp/pbrt_amd_bench")
    if args.scene:
        scene_file = args.scene
        workload = os.path.basename(scene_file)
    elif args.config == "c2":
        d = os.path.join(bench_dir, "killeroo_%dx%d" % (args.res[0], args.res[1]))
        spp = args.spp if args.spp != 64 else 128
        scene_file = os.path.join(d, "killeroo_%dspp.pbrt" % spp)
        if local_rank == 0 and not os.path.exists(scene_file):
            os.makedirs(d, exist_ok=True)
            text = open(os.path.join(ROOT, "scenes", "killeroo.pbrt")).read()
            text = text.replace('[700] "integer yresolution" [700]', '[%d] "integer yresolution" [%d]' % (args.res[0], args.res[1]))
            text = text.replace('"integer pixelsamples" [8]', '"integer pixelsamples" [%d]' % spp)
            text = text.replace("killeroo_geo/", os.path.join(ROOT, "scenes", "killeroo_geo") + "/")
            open(scene_file + ".tmp", "w").write(text)
            os.rename(scene_file + ".tmp", scene_file)
        while not os.path.exists(scene_file):
            time.sleep(0.2)
        workload = "killeroo-simple.pbrt as the reference ships it (Sphere area light, Halton sampler; geometry pre-subdivided to PLY): 66.5 k triangles, %dx%d, %d spp, path maxdepth 5" % (args.res[0], args.res[1], spp)
    elif args.config == "c4":
        spp = args.spp if args.spp != 64 else 256
        tris = args.tris if args.tris != 10_000_000 else 600_000
        key = "bathroom_synth_%dk_%dx%d_%dspp" % (tris // 1000, args.res[0], args.res[1], spp)
        d = os.path.join(bench_dir, key)
        scene_file = os.path.join(d, "bathroom_synth.pbrt")
        marker = os.path.join(d, ".done")
        if local_rank == 0 and not os.path.exists(marker):
            os.makedirs(d, exist_ok=True)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "bathroom", "--tris", str(tris),
                                   "--res", str(args.res[0]), str(args.res[1]), "--spp", str(spp), "--out", scene_file], stdout=sys.stderr)
            open(marker, "w").write("ok")
        while not os.path.exists(marker):
            time.sleep(0.2)
        workload = "Contemporary-Bathroom-class synthetic stand-in: glass/mirror/metal, %dx%d, %d spp, path maxdepth 30" % (args.res[0], args.res[1], spp)
    else:
        key = "sanmiguel_synth_%dk_%dx%d_%dspp%s" % (args.tris // 1000, args.res[0], args.res[1], args.spp, "_tex" if args.textured else "")
        d = os.path.join(bench_dir, key)
        scene_file = os.path.join(d, "sanmiguel_synth.pbrt")
        marker = os.path.join(d, ".done")
        if local_rank == 0 and not os.path.exists(marker):
            os.makedirs(d, exist_ok=True)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", str(args.tris),
                                   "--res", str(args.res[0]), str(args.res[1]), "--spp", str(args.spp), "--out", scene_file] + (["--textured"] if args.textured else []),
                                  stdout=sys.stderr)
            open(marker, "w").write("ok")
        while not os.path.exists(marker):
            time.sleep(0.2)
        workload = "San-Miguel-class synthetic stand-in (SURVEY.md s.8d): %d triangles, %dx%d, %d spp, path maxdepth 5, sobol, box filter" % (
            args.tris, args.res[0], args.res[1], args.spp)
        if args.textured:
            workload += "; TEXTURED variant (row f2): image-mapped Kd (EWA) on every material, roughness / bump maps"

    pa = importlib.import_module("pbrt-v3-distributed_amd")
    par = importlib.import_module("pbrt-v3-distributed_amd.parallel")
    # every rank parses the scene and builds the (reference's) BVH itself: share the usable CPUs between the ranks of this node
    os.environ.setdefault("PBRT_AMD_NTHREADS", str(max(1, host_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    t0 = time.time()
    sc = pa.Scene(scene_file)
    t_load = time.time() - t0
    t0 = time.time()
    ctx = pa.Context(sc, device=local_rank)
    t_upload = time.time() - t0
    if rank == 0:
        log("[bench] scene %s: %d tris, %d BVH2 nodes, %d materials, %d lights; parse+BVH %.1f s, upload+BVH4 %.1f s" %
            (workload, sc.info["n_tris"], sc.info["n_bvh_nodes"], sc.info["n_materials"], sc.info["n_lights"], t_load, t_upload))

    film_t = None
    if world > 1:
        film_t = torch.zeros(sc.height * sc.width * 4, dtype=torch.float32, device="cuda")
        ctx.film_bind(film_t.data_ptr())

    def sync_all():
        ctx.sync()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step(count=False):
        ctx.film_clear()
        ctx.render(rank=rank, world=world, count_work=count, max_paths=args.max_paths, sync=False)
        if world > 1:
            ctx.sync()   # the ctx stream is not torch's current stream
            par.combine_films(film_t, dst=0)
            torch.cuda.synchronize()   # the reduction reads film_t: it must be done before the next step clears the film

    # ---- one counting pass (deterministic work: node / triangle fetch counts feed the roofline), then warm-up
    ctx.counters_reset()
    step(count=True)
    sync_all()
    work = ctx.counters()
    for _ in range(max(0, args.warmup - 1)):
        step()
    sync_all()

    # ---- timed region: exactly K steps
    ctx.counters_reset()
    ctx.timing_enable(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    timing = ctx.timing()
    cnt = ctx.counters()
    ctx.timing_enable(False)

    # whole-job unit counts (all ranks)
    samples = np.array([cnt["camera_rays"], cnt["closest_rays"] + cnt["shadow_rays"]], dtype=np.float64)
    if world > 1:
        st = torch.tensor(samples, dtype=torch.float64, device="cuda")
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        samples = st.cpu().numpy()

    if rank == 0:
        msamples = samples[0] / elapsed * 1e-6
        mrays = samples[1] / elapsed * 1e-6
        # ---- roofline of the dominant kernel (closest-hit traversal of path-extension rays), this rank
        n_launch = max(1, timing["closest"][1])
        ext_rays = work["closest_rays"] - work["mis_rays"]
        alg_bytes = ext_rays * RAY_BYTES + work["nodes_closest"] * NODE_BYTES + work["tris_closest"] * TRI_BYTES   # per step
        t_closest_ms = timing["closest"][0] / args.steps                                                           # per step
        achieved = alg_bytes / (t_closest_ms * 1e-3) * 1e-9 if t_closest_ms > 0 else 0.0
        launches_per_step = n_launch / args.steps
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_closest.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "k_trace<0,false,false> (BVH4 closest hit, path-extension rays; all-triangle instance)", "bound": "hbm",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "alg_bytes_per_launch": alg_bytes / max(1.0, launches_per_step),
                    "avg_launch_ms": t_closest_ms / max(1.0, launches_per_step), "launches_per_step": launches_per_step,
                    "nodes_per_ray": work["nodes_closest"] / max(1, ext_rays), "tris_per_ray": work["tris_closest"] / max(1, ext_rays)}
        try:
            roofline["hbm_stream_read_GBps"] = round(ctx.stream_read_gbps(4 << 30), 1)   # achievable ceiling, beside the spec peak
        except Exception as e:   # measurement aid only
            log("[bench] stream-read ceiling not measured: %s" % e)
        if traffic:
            # measured HBM-side bytes (PMC pass of the same command, profiles/traffic_closest.json) over the live launch time:
            # the algorithmic figure counts every node fetch, most of which the XCD L2s serve
            roofline["hbm_traffic_GBps"] = round(traffic / (roofline["avg_launch_ms"] * 1e-3) * 1e-9, 1)
            roofline["hbm_traffic_frac"] = round(roofline["hbm_traffic_GBps"] / HBM_PEAK_GBS, 4)
        kernel_ms = {k: round(v[0] / args.steps, 3) for k, v in timing.items() if v[1]}

        # ---- CPU baseline: the oracle port on the host cores, on a bounded centre crop of the same frame
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            import oracle_lib as ol
            ncores = host_cpus()
            ntx, nty = (sc.width + 15) // 16, (sc.height + 15) // 16
            cx, cy = ntx // 2, nty // 2
            spp_cpu = int(min(sc.info["spp"], 8))
            side = 8   # centre box of side x side tiles, grown until the sample takes long enough
            while True:
                bx0, by0 = max(0, cx - side // 2), max(0, cy - side // 2)
                box = [bx0, by0, min(ntx, bx0 + side), min(nty, by0 + side)]
                rgbw_cpu, cc, secs = ol.render(sc, 0, spp_cpu, ncores, tiles=box)
                if secs >= args.cpu_seconds * 0.5 or (box[2] - box[0] >= ntx and box[3] - box[1] >= nty):
                    break
                side = int(side * max(1.5, min(4.0, (args.cpu_seconds / max(secs, 1e-3)) ** 0.5)))
            cpu = {"value": round(cc["camera_rays"] / secs * 1e-6, 4), "unit": "Msamples/s", "cores": ncores, "kind": "port",
                   "mrays_per_s": round((cc["closest_rays"] + cc["shadow_rays"]) / secs * 1e-6, 3),
                   "sample": "oracle/pt_oracle.cpp (CPU restatement, pinned to pbrt_ref) on tiles [%d,%d)x[%d,%d) of the same frame, %d of %d spp, %d threads "
                             "(= usable CPUs: affinity / cgroup quota; %d hardware threads visible), %.1f s"
                             % (box[0], box[2], box[1], box[3], spp_cpu, sc.info["spp"], ncores, os.cpu_count() or 1, secs)}
            # parity spot check on that crop: GPU render of the same samples vs the oracle
            ctx.film_clear()
            ctx.render(spp_begin=0, spp_end=spp_cpu)
            g = ctx.film()
            ys, ye, xs, xe = box[1] * 16, min(sc.height, box[3] * 16), box[0] * 16, min(sc.width, box[2] * 16)
            gi = sc.film_image(g)[ys:ye, xs:xe]
            ci = sc.film_image(rgbw_cpu)[ys:ye, xs:xe]
            frac, relmse = ol.image_metrics(gi, ci)
            cpu["parity_crop"] = {"pixels_within_tol": round(frac, 5), "relMSE": relmse}

        out = {"metric": "Msamples/sec (whole node), San Miguel 1080p", "value": round(msamples, 3), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "tiles": "16x16 round-robin over ranks", "parallelism": "tile-sharded x%d" % world},
               "mrays_per_s": round(mrays, 2), "rays_per_sample": round(samples[1] / max(1.0, samples[0]), 3),
               "roofline": roofline, "cpu_baseline": cpu, "kernel_ms_per_step": kernel_ms,
               "setup_s": {"parse_and_bvh_build": round(t_load, 2), "upload_and_bvh4": round(t_upload, 2)}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
