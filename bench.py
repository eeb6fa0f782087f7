#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X wavefront path tracer (BASELINE.json metric:
Msamples/sec (+ Mrays/sec), San Miguel 1080p; the scene itself is not available offline, so the
workload is the San-Miguel-class synthetic stand-in of SURVEY.md s.8(d), generated deterministically).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

A "step" = one full pass of the hot path (SamplerIntegrator::Render: 1920x1080 x 64 spp, path maxdepth 5)
over the frame with the scene already resident in HBM.  With N>1 the 16x16 image tiles are sharded over the
ranks on a skewed 2-D lattice (mi_tile_owner; scene replicated, no collective on the data path) and every rank's
reachable FilmTile pixels are added into rank 0's film with one grouped RCCL send / recv per step, overlapped with
the next step's rendering (parallel.FilmExchange).
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
# tools/valu_probe/valu_probe2 on the MI355X (profiles/r04_a_valu_probe2_issue_rate_and_clock.txt), 8 waves per SIMD, host-event time x measured clock:
PROBE_MIX_CYCLES = 3.39   # cycles per wave instruction per SIMD of the box test's instruction mix (kind 3: 1.430 ns at 2.372 GHz)
PROBE_FMA_CYCLES = 2.22   # ... of independent v_fma_f32 (kind 0: 1.049 ns at 2.114 GHz)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_T0 = time.time()
_RANK = int(os.environ.get("RANK", "0"))
_TRACE = int(os.environ.get("WORLD_SIZE", "1")) > 1 or bool(os.environ.get("PBRT_AMD_BENCH_TRACE"))


def phase(name):
    """one stderr line per rank and phase of a multi-rank run (scene / ctx / init_pg / step / exchange / barrier): a stalled job says where it stalled"""
    if _TRACE:
        log("[bench r%d +%.1fs] %s" % (_RANK, time.time() - _T0, name))


def wait_for(path, timeout_s, what):
    """bounded wait for a file another local rank publishes"""
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            raise SystemExit("bench.py (rank %d): %s did not appear within %.0f s (%s)" % (_RANK, path, timeout_s, what))
        time.sleep(0.1)


def publish_dir(d, make, leader, timeout_s, what):
    """A generated scene directory, published complete or not at all: the node's leader (local rank 0) has `make(tmpdir)` fill a fresh private directory,
    stamps it and renames it to `d`; every rank then waits (bounded) for d/.done.  Nobody ever reads a file that is still being written."""
    import shutil
    marker = os.path.join(d, ".done")
    if leader and not os.path.exists(marker):
        tmp = "%s.tmp.%d" % (d, os.getpid())
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        make(tmp)
        with open(os.path.join(tmp, ".done"), "w") as f:
            f.write("ok")
        try:
            os.rename(tmp, d)
        except OSError:
            if os.path.exists(marker):          # another job published the same scene meanwhile: identical by construction
                shutil.rmtree(tmp, ignore_errors=True)
            else:                                # an unstamped leftover of an interrupted run
                shutil.rmtree(d, ignore_errors=True)
                os.rename(tmp, d)
    wait_for(marker, timeout_s, what)


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


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_closest.json")


def _lib_id():
    """identifies the kernel build a PMC figure belongs to: hash of the device library"""
    import hashlib
    f = os.path.join(ROOT, "pbrt-v3-distributed_amd", "lib", "libpbrt_amd.so")
    try:
        return hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def traffic_live(wl_args, workload, pmc_steps=2):
    """HBM-side bytes per launch of the closest-hit kernel from a separate `rocprofv3 --pmc` pass over the same workload (child process, plain frames only): the L2's
    memory-side read requests BY SIZE, bytes = 32 n32 + 64 n64 + 128 n128 (TCC_EA0_RDREQ_32B / _64B / _128B).  Calibrated in round 5 on known byte counts
    (profiles/r05_a_fetch_size_calibration.txt, r05_b_request_sizes.txt): a streaming read of 8.59 GB shows 6.711e7 128-byte requests = 8.59 GB exactly; FETCH_SIZE tallies
    every request at 64 B (hence the guide's x2 on gfx950); chains of random 64-byte records fetch a whole 128-byte line per record (x2 of the useful bytes) -- every request
    of the traversal kernels is a 128-byte one.  Infinity-Cache hits are included (a 64 MiB gather that never leaves the MALL shows the same counts): an upper bound on HBM."""
    import csv, glob, shutil, tempfile, collections
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        log("[bench] no rocprofv3: HBM traffic not measured live")
        return None
    d = tempfile.mkdtemp(prefix="pbrt_amd_pmc_", dir="/tmp")
    names = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
    cmd = [exe, "--pmc"] + names + ["-d", d, "-o", "c", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__)] + wl_args + \
          ["--steps", str(pmc_steps), "--warmup", "0", "--cpu-seconds", "0", "--traffic", "none", "--pmc-child"]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    except Exception as e:
        log("[bench] PMC pass failed to run: %s" % e)
        return None
    if r.returncode != 0:
        log("[bench] PMC pass failed (rc %d): %s" % (r.returncode, r.stdout[-600:]))
        return None
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") in names and row["Kernel_Name"].startswith("void k_trace<0, false"):
                cnt[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[row["Kernel_Name"]].add(row["Dispatch_Id"])
    shutil.rmtree(d, ignore_errors=True)
    if not cnt:
        log("[bench] PMC pass produced no TCC_EA0_RDREQ rows for k_trace<0, false, ...>")
        return None
    k = max(cnt, key=lambda n: cnt[n]["TCC_EA0_RDREQ_sum"])
    n = len(disp[k])
    c = {a: b / n for a, b in cnt[k].items()}
    per = 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128.0 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
    log("[bench] PMC pass: %s, %d launches, %.3f GB per launch by request size (%.4g requests, %.4g of them 128-byte) (%.0f s)" %
        (k.split("(")[0], n, per * 1e-9, c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_128B_sum", 0), time.time() - t0))
    return {"kernel": k.split("(")[0].replace("void ", ""), "launches": n, "requests_per_launch": {a.replace("TCC_EA0_", "").replace("_sum", ""): b for a, b in c.items()},
            "FETCH_SIZE_KiB_per_launch": c["TCC_EA0_RDREQ_sum"] * 64.0 / 1024.0, "bytes_per_launch": per,
            "source": "live: rocprofv3 --pmc TCC_EA0_RDREQ{,_32B,_64B,_128B} pass of this workload inside this bench run; bytes by request size (calibrated: profiles/r05_a_fetch_size_calibration.txt, r05_b_request_sizes.txt; FETCH_SIZE = 64 B x requests would read half)",
            "workload": workload, "lib": _lib_id()}


def valu_live(wl_args, pmc_steps=1):
    """What bounds the traversal kernels: VALU issue.  A separate `rocprofv3 --pmc` pass over the same workload with the SQ's
    instruction counters -> per launch of the closest-hit kernel: VALU instructions, active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU /
    SQ_ACTIVE_INST_VALU), waves.  None when the pass cannot run."""
    import csv, glob, shutil, tempfile, collections
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="pbrt_amd_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "-d", d, "-o", "c", "--output-format", "csv", "--",
           sys.executable, os.path.abspath(__file__)] + wl_args + ["--steps", str(pmc_steps), "--warmup", "0", "--cpu-seconds", "0", "--traffic", "none", "--pmc-child"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    except Exception as e:
        log("[bench] SQ counter pass failed to run: %s" % e)
        return None
    if r.returncode != 0:
        log("[bench] SQ counter pass failed (rc %d): %s" % (r.returncode, r.stdout[-400:]))
        shutil.rmtree(d, ignore_errors=True)
        return None
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("void k_trace<0, false") or row["Kernel_Name"].startswith("void k_shade<"):
                agg[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[row["Kernel_Name"]].add(row["Dispatch_Id"])
    shutil.rmtree(d, ignore_errors=True)

    def figures(prefix):
        ks = [n for n in agg if n.startswith(prefix)]
        if not ks:
            return None
        k = max(ks, key=lambda n: agg[n].get("SQ_INSTS_VALU", 0))
        n = max(1, len(disp[k]))
        c = {a: b / n for a, b in agg[k].items()}
        if not c.get("SQ_ACTIVE_INST_VALU") or not c.get("SQ_WAVES"):
            return None
        return {"kernel": k.split("(")[0].replace("void ", ""), "launches_in_pass": n,
                "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "waves_per_launch": c["SQ_WAVES"], "lanes_active_per_valu_inst": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_ACTIVE_INST_VALU"], 2),
                "valu_share_of_wave_lifetime": round(c["SQ_ACTIVE_INST_VALU"] / max(1.0, c.get("SQ_WAVE_CYCLES", 0)), 4)}

    res = figures("void k_trace<0, false")
    if res is None:
        return None
    res.pop("kernel"); res.pop("launches_in_pass")
    shade = figures("void k_shade<")   # the same pass also saw the shading kernel: bench.py's roofline_shade
    if shade:
        res["_shade"] = shade
    return res


def secondary_line(extra):
    """bench.py itself on a variant workload, 3 steps after 1 warm-up, with a short pbrt_ref crop at full spp; returns the fields of its JSON line worth keeping"""
    cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--steps", "3", "--warmup", "1", "--traffic", "none", "--cpu-seconds", "8", "--cpu-port-seconds", "0", "--secondary", "off"]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
        line = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
    except Exception as e:
        log("[bench] secondary line failed to run: %s" % e)
        return {"error": str(e)}
    if not line:
        log("[bench] secondary line failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-600:]))
        return {"error": "rc %d" % r.returncode}
    log("[bench] secondary line (%s): %.1f Msamples/s, %.1f ms per step (%.0f s)" % (" ".join(extra[:2]), line["value"], line["ms_per_step"], time.time() - t0))
    return {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"], "kernel_ms_per_step": line.get("kernel_ms_per_step"),
            "workload": line["config"]["workload"], "rays_per_sample": line.get("rays_per_sample"),
            "parity_crop": (line.get("cpu_baseline") or {}).get("parity_crop"), "cpu_baseline_value": (line.get("cpu_baseline") or {}).get("value")}


def traffic_from_file(workload):
    try:
        ent = json.load(open(TRAFFIC_FILE)).get("entries", {}).get(workload)
    except Exception:
        return None
    if not ent:
        return None   # never apply another workload's bytes
    ent = dict(ent)
    stale = ent.get("lib") != _lib_id()
    ent["source"] = "file: profiles/traffic_closest.json (%s)%s" % (ent.get("profile", "committed PMC pass"), "; STALE: collected with a different kernel build" if stale else "")
    return ent


def traffic_save(workload, t):
    try:
        doc = json.load(open(TRAFFIC_FILE))
        if "entries" not in doc:
            doc = {"entries": {}}
    except Exception:
        doc = {"entries": {}}
    e = {k: t[k] for k in ("kernel", "launches", "FETCH_SIZE_KiB_per_launch", "bytes_per_launch", "lib")}
    e["profile"] = "rocprofv3 --pmc TCC_EA0_RDREQ{,_32B,_64B,_128B}, own pass of `bench.py` on this workload; bytes by request size"
    doc["entries"][workload] = e
    json.dump(doc, open(TRAFFIC_FILE, "w"), indent=1, sort_keys=True)


def cpu_reference(sc, scene_file, ncores, seconds, ol):
    """pbrt_ref --nthreads <usable> --cropwindow <centre window> on the bench scene; None when the binary is not there."""
    import re, tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
    if not os.access(ref, os.X_OK):
        log("[bench] oracle/_ref/pbrt_ref not present: CPU baseline falls back to the oracle port")
        return None
    W, H, spp = sc.width, sc.height, sc.info["spp"]
    est = 0.06 * ncores * 1e6                      # samples/s the reference manages on this class of scene (measured ~0.9 M/s on 16 threads)
    frac = min(1.0, seconds * est / (float(W) * H * spp))
    side = frac ** 0.5
    x0, x1, y0, y1 = 0.5 - side / 2, 0.5 + side / 2, 0.5 - side / 2, 0.5 + side / 2
    out = tempfile.mktemp(prefix="pbrt_ref_crop_", suffix=".pfm", dir="/tmp")
    cmd = [ref, "--nthreads", str(ncores), "--cropwindow", "%.6f" % x0, "%.6f" % x1, "%.6f" % y0, "%.6f" % y1, "--outfile", out, scene_file]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=dict(os.environ, PBRT_REF_RENDER_TIMES="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    except Exception as e:
        log("[bench] pbrt_ref did not run: %s" % e)
        return None
    wall = time.time() - t0
    m = re.search(r"Integrator::Render seconds ([0-9.]+)", r.stderr)
    def stat(name):
        mm = re.search(name + r"\s+(\d+)", r.stdout)
        return int(mm.group(1)) if mm else 0
    cam, reg, shd = stat("Camera rays traced"), stat("Regular ray intersection tests"), stat("Shadow ray intersection tests")
    if r.returncode != 0 or not m or not cam:
        log("[bench] pbrt_ref failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        return None
    secs = float(m.group(1))
    cpu = {"value": round(cam / secs * 1e-6, 4), "unit": "Msamples/s", "cores": ncores, "kind": "reference",
           "mrays_per_s": round((reg + shd) / secs * 1e-6, 3),
           "sample": "oracle/_ref/pbrt_ref (pbrt-v3 built from the unmodified reference sources) --nthreads %d --cropwindow %.4f %.4f %.4f %.4f on the bench scene, "
                     "all %d spp: %d camera samples, Integrator::Render tile loop %.1f s (whole process incl. parse + BVH build %.0f s); "
                     "%d hardware threads visible, %d usable (affinity / cgroup quota)" % (ncores, x0, x1, y0, y1, spp, cam, secs, wall, os.cpu_count() or 1, ncores)}
    # parity against the reference itself at full size: the SAME crop rendered on the GPU (the crop window moves the sampler's
    # sample bounds -- sobol.cpp:65-70 builds the sampler over film->GetSampleBounds() -- so the window has to be part of the
    # scene on both sides; pixels of the uncropped frame carry different Sobol' points)
    try:
        pa = importlib.import_module("pbrt-v3-distributed_amd")
        ci = pa.read_pfm(out)
        cw = [float("%.6f" % v) for v in (x0, x1, y0, y1)]
        sc2 = pa.Scene(scene_file, cropwindow=cw)
        ctx2 = pa.Context(sc2, device=0)
        ctx2.render()
        gi = sc2.film_image(ctx2.film())
        ctx2.close(); sc2.close()
        if gi.shape == ci.shape:
            f, relmse = ol.image_metrics(gi, ci)
            cpu["parity_crop"] = {"against": "pbrt_ref", "pixels": int(gi.shape[0] * gi.shape[1]), "spp": spp, "pixels_within_tol": round(f, 5), "relMSE": relmse,
                                  "criterion": "per-pixel L2 <= 1e-3 (1 + |ref|) for >= 99.5 % of the pixels, relMSE <= 1e-4"}
        else:
            cpu["parity_crop"] = {"error": "crop shapes differ: %s vs %s" % (gi.shape, ci.shape)}
    except Exception as e:
        cpu["parity_crop"] = {"error": str(e)}
    finally:
        try:
            os.remove(out)
        except OSError:
            pass
    return cpu


def cpu_port(sc, ctx, ncores, seconds, ol):
    """the oracle port (oracle/pt_oracle.cpp) on centre tiles of the same frame, 8 spp; + parity of those samples against the GPU"""
    ntx, nty = (sc.width + 15) // 16, (sc.height + 15) // 16
    cx, cy = ntx // 2, nty // 2
    spp_cpu = int(min(sc.info["spp"], 8))
    side = 8   # centre box of side x side tiles, grown until the sample takes long enough
    while True:
        bx0, by0 = max(0, cx - side // 2), max(0, cy - side // 2)
        box = [bx0, by0, min(ntx, bx0 + side), min(nty, by0 + side)]
        rgbw_cpu, cc, secs = ol.render(sc, 0, spp_cpu, ncores, tiles=box)
        if secs >= seconds * 0.5 or (box[2] - box[0] >= ntx and box[3] - box[1] >= nty):
            break
        side = int(side * max(1.5, min(4.0, (seconds / max(secs, 1e-3)) ** 0.5)))
    cpu = {"value": round(cc["camera_rays"] / secs * 1e-6, 4), "unit": "Msamples/s", "cores": ncores, "kind": "port",
           "mrays_per_s": round((cc["closest_rays"] + cc["shadow_rays"]) / secs * 1e-6, 3),
           "sample": "oracle/pt_oracle.cpp (CPU restatement, pinned to pbrt_ref) on tiles [%d,%d)x[%d,%d) of the same frame, %d of %d spp, %d threads, %.1f s"
                     % (box[0], box[2], box[1], box[3], spp_cpu, sc.info["spp"], ncores, secs)}
    ctx.film_clear()
    ctx.render(spp_begin=0, spp_end=spp_cpu)
    g = ctx.film()
    ys, ye, xs, xe = box[1] * 16, min(sc.height, box[3] * 16), box[0] * 16, min(sc.width, box[2] * 16)
    f, relmse = ol.image_metrics(sc.film_image(g)[ys:ye, xs:xe], sc.film_image(rgbw_cpu)[ys:ye, xs:xe])
    cpu["parity_crop"] = {"against": "oracle port", "pixels_within_tol": round(f, 5), "relMSE": relmse}
    return cpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--res", type=int, nargs=2, default=[1920, 1080])
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--scene", default=None, help="render this .pbrt instead of the generated stand-in")
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c3 = San-Miguel-class 1080p 64spp (default, the metric's workload); "
                         "c2 = killeroo 1080p 128spp; c4 = bathroom-class 1080p 256spp maxdepth 30; "
                         "c5 = the San-Miguel-class scene at 3840x2160, 512 spp (33-bit Sobol' indices, 32 passes of 2^27 paths; tile-sharded with --gpus N)")
    ap.add_argument("--textured", action="store_true", help="c3 only: the stand-in with image-mapped / bump-mapped materials (SURVEY.md s.8 row f2)")
    ap.add_argument("--smokebox", action="store_true", help="c3 only: a heterogeneous (grid) medium behind a BSDF-less box in the far half of the stand-in, VolPathIntegrator (split form of k_shade_vol: ratio tracking in the walked transmittance queries); PBRT_AMD_VOL_SPLIT=0 = the general form")
    ap.add_argument("--subsurface", action="store_true", help="c3 only: three of the stand-in's 24 materials become kdsubsurface (BSSRDF probe chains walked through the queues; SURVEY.md s.8 row f4); PBRT_AMD_VOL_INLINE=1 = the per-lane form")
    ap.add_argument("--leafmask", action="store_true", help="c3 only: the stand-in's leaf quads as alpha-masked meshes (a disc cut out of each by a float texture; SURVEY.md s.8 row f2 alpha masks), combinable with --volpath / --fogbox")
    ap.add_argument("--fogbox", action="store_true", help="c3 only: a bank of fog behind a BSDF-less box in the far half of the stand-in, Integrator \"volpath\" (medium interfaces: k_vol_tr)")
    ap.add_argument("--volpath", action="store_true", help="c3 only: the stand-in inside a homogeneous medium, Integrator \"volpath\" (SURVEY.md s.8 row f4; k_shade_vol)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target length of the CPU baseline sample: the reference binary oracle/_ref/pbrt_ref on a centre crop of the same frame (0 = skip)")
    ap.add_argument("--cpu-port-seconds", type=float, default=6.0, help="length of the second CPU sample, the oracle port (0 = skip)")
    ap.add_argument("--traffic", default="live", choices=["live", "file", "none"],
                    help="HBM bytes per launch of the dominant kernel: live = an extra rocprofv3 --pmc pass (memory-side read requests by size) of this workload run as a child process; "
                         "file = the committed profiles/traffic_closest.json entry of exactly this workload and kernel; none = null")
    ap.add_argument("--save-traffic", action="store_true", help="write the live PMC result into profiles/traffic_closest.json")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--secondary", default="auto", choices=["auto", "on", "off"],
                    help="also render the San-Miguel-like variant of the default workload (--textured --leafmask: image-mapped / bump-mapped materials, alpha-masked leaf quads) "
                         "for 3 steps in a child process and report it as secondary.textured_leafmask inside the one JSON line; auto = with the default C3 workload on one GPU")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 path on a 1-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="testing aid: every rank uses GPU 0 (device ordinal only: local rank 0 still builds the scene alone, the others map it)")
    ap.add_argument("--dump-film", default=None, help="rank 0 saves the combined FilmTilePixel array (H, W, 4 float32: contribSum rgb, filterWeightSum) of the last timed step as .npy")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, the
        # same launch line the driver uses) and hand their exit code back; rank 0 of the child job prints the JSON line
        par0 = importlib.import_module("pbrt-v3-distributed_amd.parallel")

        def sweep(launcher_pid):   # whatever node_scene left under this job's name (a '.failed' note for ranks that were still waiting, a blob of a job that was killed)
            import glob
            for d in {os.environ.get("PBRT_AMD_BLOB_DIR") or "/dev/shm", os.environ.get("PBRT_AMD_BENCH_DIR", "/tmp/pbrt_amd_bench")}:
                for f in glob.glob(os.path.join(d, "pbrt_amd_scene_%d.blob*" % launcher_pid)):
                    try:
                        os.remove(f)
                    except OSError:
                        pass
        raise SystemExit(par0.launch_ranks(args.gpus, __file__, sys.argv[1:], after=sweep))
    if args.config == "c5":   # configs[4]: the C3 scene at 4K / 512 spp
        if args.res == [1920, 1080]:
            args.res = [3840, 2160]
        if args.spp == 64:
            args.spp = 512
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # --one-device moves every rank's DEVICE ORDINAL to GPU 0 and nothing else: local rank 0 alone generates / builds / publishes the scene, the others map it
    dev_ordinal = 0 if (world > 1 and args.one_device) else local_rank
    if world > 1:
        import faulthandler
        importlib.import_module("pbrt-v3-distributed_amd.parallel").die_with_parent()   # the launcher starts ranks in sessions of their own: a killed launcher must not leave them on the GPU
        if os.getppid() == 1:
            raise SystemExit("bench.py (rank %d): the launcher is already gone" % rank)
        faulthandler.enable()
        faulthandler.dump_traceback_later(float(os.environ.get("PBRT_AMD_BENCH_STACKS_S", "120")), repeat=True)   # a stalled rank shows its stack
    wait_s = float(os.environ.get("PBRT_AMD_BENCH_WAIT_S", "1800"))
    phase("start: rank %d of %d, local rank %d, device %d, pid %d" % (rank, world, local_rank, dev_ordinal, os.getpid()))

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
        wait_for(scene_file, wait_s, "the c2 scene file, written by local rank 0")
        workload = "killeroo-simple.pbrt as the reference ships it (Sphere area light, Halton sampler; geometry pre-subdivided to PLY): 66.5 k triangles, %dx%d, %d spp, path maxdepth 5" % (args.res[0], args.res[1], spp)
    elif args.config == "c4":
        spp = args.spp if args.spp != 64 else 256
        tris = args.tris if args.tris != 10_000_000 else 600_000
        key = "bathroom_synth_%dk_%dx%d_%dspp" % (tris // 1000, args.res[0], args.res[1], spp)
        d = os.path.join(bench_dir, key)
        scene_file = os.path.join(d, "bathroom_synth.pbrt")
        publish_dir(d, lambda tmp: subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "bathroom", "--tris", str(tris),
                                                          "--res", str(args.res[0]), str(args.res[1]), "--spp", str(spp), "--out", os.path.join(tmp, "bathroom_synth.pbrt")], stdout=sys.stderr),
                    local_rank == 0, wait_s, "the c4 scene, generated by local rank 0")
        workload = "Contemporary-Bathroom-class synthetic stand-in: glass/mirror/metal, %dx%d, %d spp, path maxdepth 30" % (args.res[0], args.res[1], spp)
    else:
        key = "sanmiguel_synth_%dk_%dx%d_%dspp%s" % (args.tris // 1000, args.res[0], args.res[1], args.spp, ("_tex" if args.textured else "") + ("_haze" if args.volpath else "") + ("_fogbox" if args.fogbox else "") + ("_leafmask" if args.leafmask else "") + ("_sss" if args.subsurface else "") + ("_smokebox" if args.smokebox else ""))
        d = os.path.join(bench_dir, key)
        scene_file = os.path.join(d, "sanmiguel_synth.pbrt")
        variant = (["--textured"] if args.textured else []) + (["--haze"] if args.volpath else []) + (["--fogbox"] if args.fogbox else []) + (["--leafmask"] if args.leafmask else []) + \
                  (["--subsurface"] if args.subsurface else []) + (["--smokebox"] if args.smokebox else [])
        publish_dir(d, lambda tmp: subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", str(args.tris),
                                                          "--res", str(args.res[0]), str(args.res[1]), "--spp", str(args.spp), "--out", os.path.join(tmp, "sanmiguel_synth.pbrt")] + variant, stdout=sys.stderr),
                    local_rank == 0, wait_s, "the c3 scene, generated by local rank 0")
        workload = "San-Miguel-class synthetic stand-in (SURVEY.md s.8d): %d triangles, %dx%d, %d spp, path maxdepth 5, sobol, box filter" % (
            args.tris, args.res[0], args.res[1], args.spp)
        if args.textured:
            workload += "; TEXTURED variant (row f2): image-mapped Kd (EWA) on every material, roughness / bump maps"
        if args.volpath:
            workload = workload.replace("path maxdepth 5", "VOLPATH maxdepth 5 (row f4): camera and scene inside a homogeneous medium")
        if args.fogbox:
            workload = workload.replace("path maxdepth 5", "VOLPATH maxdepth 5 (row f4): a homogeneous medium behind a BSDF-less box in the far half of the scene")
        if args.smokebox:
            workload = workload.replace("path maxdepth 5", "VOLPATH maxdepth 5 (row f4): a heterogeneous (16 x 8 x 12 grid) medium behind a BSDF-less box in the far half of the scene")
        if args.subsurface:
            workload += "; SUBSURFACE variant (row f4): three of the 24 materials are kdsubsurface (BSSRDF probe chains)"
        if args.leafmask:
            workload += "; LEAFMASK variant (row f2): the leaf quads (about half of the triangles) are alpha-masked meshes"

    pa = importlib.import_module("pbrt-v3-distributed_amd")
    par = importlib.import_module("pbrt-v3-distributed_amd.parallel")
    # one scene build per node (parallel.node_scene): local rank 0 parses the scene and builds the reference's BVH with every usable CPU and
    # publishes the flattened scene as one file; the other ranks of the node map it (no N-fold parse / build / host copy of the geometry)
    os.environ.setdefault("PBRT_AMD_NTHREADS", str(max(1, host_cpus())))
    blob_dir = os.environ.get("PBRT_AMD_BLOB_DIR") or ("/dev/shm" if os.access("/dev/shm", os.W_OK) else bench_dir)   # tmpfs: the blob is a memcpy, not a disk write
    blob = os.path.join(blob_dir, "pbrt_amd_scene_%d.blob" % os.getppid())   # the launcher's pid: unique per job
    phase("scene files in place: %s" % scene_file)
    if world > 1:
        sc, t_load, scene_source = par.node_scene(lambda: pa.Scene(scene_file, strict=True), blob, local_rank, timeout_s=wait_s)
    else:
        t0 = time.time()
        sc = pa.Scene(scene_file, strict=True)
        t_load, scene_source = time.time() - t0, "built"
    t0 = time.time()
    phase("scene %s in %.1f s: %d tris" % (scene_source, t_load, sc.info["n_tris"]))
    ctx = pa.Context(sc, device=dev_ordinal)
    t_upload = time.time() - t0
    phase("device context on GPU %d in %.1f s" % (dev_ordinal, t_upload))
    if rank == 0:
        log("[bench] scene %s: %d tris, %d BVH2 nodes, %d materials, %d lights; parse+BVH %.1f s, upload+BVH4 %.1f s" %
            (workload, sc.info["n_tris"], sc.info["n_bvh_nodes"], sc.info["n_materials"], sc.info["n_lights"], t_load, t_upload))

    # one rank of the tile-sharded frame (pbrt-v3-distributed_amd/parallel.py): film in a torch tensor, RCCL reduce onto rank 0 per step
    frame = par.ShardedFrame(ctx, sc, rank, world, dev_ordinal, backend=args.backend, trace=phase)
    phase("process group up (%s)" % args.backend)
    scene_sources = None
    if world > 1:   # how every rank got its scene ("built" on local rank 0, "mapped" elsewhere): part of the line, so a test can tell the blob path really ran
        import torch.distributed as dist
        scene_sources = [None] * world
        dist.all_gather_object(scene_sources, scene_source)
    sync_all = frame.sync_all

    def step(count=False):
        frame.step(count_work=count, max_paths=args.max_paths)

    if args.pmc_child:   # under rocprofv3 --pmc: plain (non-counting) frames only, nothing printed
        for _ in range(max(1, args.steps)):
            step()
        sync_all()
        ctx.close()
        return

    # ---- one counting pass (deterministic work: node / triangle fetch counts feed the roofline), then warm-up
    ctx.counters_reset()
    step(count=True)
    sync_all()
    work = ctx.counters()
    phase("counting pass done")
    try:
        trace_clk = ctx.trace_clock()   # shader clock inside the traversal launches of the counting pass (s_memtime / s_memrealtime per wave)
    except Exception:
        trace_clk = None
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
    elapsed = frame.max_over_ranks(time.perf_counter() - t0)
    phase("%d timed step(s) done: %.1f ms per step" % (args.steps, elapsed / max(1, args.steps) * 1e3))
    if args.dump_film and rank == 0:
        np.save(args.dump_film, frame.root_film())
    timing = ctx.timing()
    cnt = ctx.counters()
    if cnt.get("trace_guard_trips", 0):
        raise SystemExit("bench.py: %d traversal waves hit the non-termination guard -- the frame is invalid" % cnt["trace_guard_trips"])
    ctx.timing_enable(False)

    # whole-job unit counts (all ranks)
    samples = frame.sum_over_ranks([cnt["camera_rays"], cnt["closest_rays"] + cnt["shadow_rays"]])

    if rank == 0:
        msamples = samples[0] / elapsed * 1e-6
        mrays = samples[1] / elapsed * 1e-6
        # ---- roofline of the dominant kernel (closest-hit traversal of path-extension rays), this rank.
        # `achieved` = HBM-side bytes per launch (rocprofv3 --pmc: the L2's memory-side read requests by size, traffic_live -- what the guide's
        # "FETCH_SIZE x 2" amounts to on gfx950, calibrated in round 5) / the live average launch time (HIP events on the ctx stream): a true fraction of the
        # 8 TB/s peak.  The ALGORITHMIC byte rate of SURVEY.md s.8(d) (every node / triangle fetch counted, most of them
        # served by the XCD L2s) is reported beside it as l2_served_GBps -- it exceeds the HBM peak and bounds nothing.
        n_launch = max(1, timing["closest"][1])
        ext_rays = work["closest_rays"] - work["mis_rays"]
        tinfo = ctx.trace_info()
        alg_bytes = ext_rays * RAY_BYTES + work["nodes_closest"] * tinfo["node_bytes"] + work["tris_closest"] * TRI_BYTES   # per step
        t_closest_ms = timing["closest"][0] / args.steps                                                           # per step
        launches_per_step = n_launch / args.steps
        avg_launch_ms = t_closest_ms / max(1.0, launches_per_step)
        alg_rate = alg_bytes / (t_closest_ms * 1e-3) * 1e-9 if t_closest_ms > 0 else 0.0
        wl_args = ["--config", args.config, "--tris", str(args.tris), "--res", str(args.res[0]), str(args.res[1]), "--spp", str(args.spp)]
        if args.scene:
            wl_args += ["--scene", args.scene]
        if args.textured:
            wl_args += ["--textured"]
        if args.volpath:
            wl_args += ["--volpath"]
        if args.fogbox:
            wl_args += ["--fogbox"]
        if args.leafmask:
            wl_args += ["--leafmask"]
        if args.subsurface:
            wl_args += ["--subsurface"]
        if args.smokebox:
            wl_args += ["--smokebox"]
        if args.max_paths:
            wl_args += ["--max-paths", str(args.max_paths)]
        traffic = None
        roofline_shade = None
        if world == 1 and args.traffic == "live":
            traffic = traffic_live(wl_args, workload, 1 if args.config == "c5" else 2)
        if world == 1 and traffic is None and args.traffic in ("live", "file"):
            traffic = traffic_from_file(workload)
        if traffic and args.save_traffic and traffic.get("source", "").startswith("live"):
            traffic_save(workload, traffic)
        roofline = {"kernel": (traffic or {}).get("kernel", "k_trace<0, ...> (BVH4 closest hit, path-extension rays)"), "bound": "hbm",
                    "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                    "avg_launch_ms": avg_launch_ms, "launches_per_step": launches_per_step,
                    "alg_bytes_per_launch": alg_bytes / max(1.0, launches_per_step), "l2_served_GBps": round(alg_rate, 1),
                    "nodes_per_ray": work["nodes_closest"] / max(1, ext_rays), "tris_per_ray": work["tris_closest"] / max(1, ext_rays),
                    "layout": "%s: %d nodes x %d B" % (tinfo["name"], tinfo["nodes"], tinfo["node_bytes"]),
                    # interior steps served from the traversal blocks' LDS copy of the scene's most visited nodes (hot-node probe at upload)
                    "hot_nodes_in_lds": tinfo.get("hot_nodes", 0), "hot_share_of_node_visits": round(work.get("nodes_hot_closest", 0) / max(1, work["nodes_closest"]), 4),
                    "launch_shape": "%d threads x %d block(s) per CU" % (tinfo.get("block_threads", 256), tinfo.get("blocks_per_cu", 6))}
        if traffic:
            roofline["achieved"] = round(traffic["bytes_per_launch"] / (avg_launch_ms * 1e-3) * 1e-9, 1)
            roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 4)
            roofline["alg_over_hbm_bytes"] = round(roofline["alg_bytes_per_launch"] / traffic["bytes_per_launch"], 2)
        try:
            roofline["hbm_stream_read_GBps"] = round(ctx.stream_read_gbps(4 << 30), 1)   # achievable ceiling, beside the spec peak
            if roofline["achieved"]:
                roofline["frac_of_stream_ceiling"] = round(roofline["achieved"] / roofline["hbm_stream_read_GBps"], 4)
        except Exception as e:   # measurement aid only
            log("[bench] stream-read ceiling not measured: %s" % e)
        # What the traversal IS bound by: the rate at which a CU serves incoherent 16-byte loads.  achieved = the kernel's own
        # lane requests (node words + triangle vertices + the ray record, from the counting pass) / its launch time; ceiling = mi_gather_rate:
        # chains of dependent random fetches of records of the node size from a buffer of the node array's size, same launch shape, no arithmetic.
        try:
            loads = max(1, min(4, tinfo["node_bytes"] // 16))
            # vector-memory lane requests: the node steps NOT served from LDS, the triangle vertices, the ray record
            req_per_step = (work["nodes_closest"] - work.get("nodes_hot_closest", 0)) * (tinfo["node_bytes"] // 16) + work["tris_closest"] * 3 + ext_rays * 3
            ach = req_per_step / (t_closest_ms * 1e-3) * 1e-9 if t_closest_ms > 0 else 0.0
            node_mb = max(1, (tinfo["nodes"] * tinfo["node_bytes"]) >> 20)
            ladder = {}   # the same chain of dependent record fetches over working sets that sit in the L1s, the L2s, the Infinity Cache, HBM
            for name, nbytes in (("L1_32KB", 32 << 10), ("L2_2MB", 2 << 20), ("MALL_64MB", 64 << 20), ("HBM_%dMB_node_array" % node_mb, node_mb << 20)):
                ladder[name] = round(ctx.gather_rate(nbytes, loads), 1)
            roofline["request_rate"] = {"achieved_Greq_per_s": round(ach, 1), "requests_per_ray": round(req_per_step / max(1, ext_rays), 1),
                                        "ceilings_Greq_per_s": ladder, "frac_of_L2_resident_ceiling": round(ach / ladder["L2_2MB"], 4) if ladder["L2_2MB"] > 0 else None,
                                        "ceilings_are": "mi_gather_rate: chains of dependent random %d x 16 B record fetches per lane over a buffer of the stated size, "
                                                        "traversal launch shape, no arithmetic; the traversal's own fetches are a MIX of these levels (upper tree levels are shared)" % loads}
        except Exception as e:   # measurement aid only
            log("[bench] request-rate ceiling not measured: %s" % e)
        # ... and what does bound it: VALU issue.  Everything here is measured: the launch's SIMD cycles = live launch time x the shader
        # clock the kernel itself reports (s_memtime / s_memrealtime per wave, mi_trace_clock); its VALU instructions per SIMD from a separate rocprofv3 --pmc
        # pass; the ceiling from tools/valu_probe (profiles/r04_a_valu_probe2_issue_rate_and_clock.txt: a SIMD issues the box test's instruction mix at
        # 3.39 cycles per wave instruction with 8 waves resident, plain v_fma_f32 at 2.2; profiles/r04_c_issue_probe_instruction_costs.txt per instruction).
        clk_ghz = (trace_clk or {}).get("closest_GHz") or 0.0
        if clk_ghz:
            roofline["shader_clock_GHz"] = {"closest": round(trace_clk["closest_GHz"], 3), "anyhit": round(trace_clk["anyhit_GHz"], 3),
                                            "source": "s_memtime / s_memrealtime ticks per wave inside the launches of the counting pass (mi_trace_clock)"}
        if world == 1 and args.traffic == "live":
            try:
                vi = valu_live(wl_args, 1)
                if vi and clk_ghz:
                    # CUs from the persistent launch itself: waves per launch = CUs x blocks per CU x waves per block
                    simds = 4 * max(1, int(round(vi["waves_per_launch"] / (tinfo.get("blocks_per_cu", 6) * tinfo.get("block_threads", 256) / 64.0))))
                    launch_cycles = avg_launch_ms * 1e-3 * clk_ghz * 1e9
                    per_simd = vi["valu_insts_per_launch"] / simds
                    vi["simd_cycles_per_valu_inst"] = round(launch_cycles / per_simd, 3)      # what one VALU instruction of this kernel costs its SIMD, all stalls included
                    vi["probe_cycles_per_valu_inst"] = PROBE_MIX_CYCLES                        # the same SIMD saturated with the box test's mix (tools/valu_probe)
                    vi["issue_frac"] = round(PROBE_MIX_CYCLES / vi["simd_cycles_per_valu_inst"], 4)
                    vi["note"] = ("SQ counters of a separate rocprofv3 --pmc pass of this workload; launch time from the unprofiled run x the clock measured inside the kernel; "
                                  "ceiling = tools/valu_probe2 kind 3 at 8 waves per SIMD (%.2f cycles per wave instruction; plain v_fma_f32 %.2f)" % (PROBE_MIX_CYCLES, PROBE_FMA_CYCLES))
                    shade_sq = vi.pop("_shade", None)
                    roofline["valu_issue"] = vi
                    roofline["frac_valu_lane_throughput"] = round(vi["issue_frac"] * vi["lanes_active_per_valu_inst"] / 64.0, 4)
                    # ... and the same figures for the shading kernel, from the same counter pass: k_shade has no memory roof worth the name (its
                    # tables sit in the L2 / scalar cache); what bounds it is the rate at which its SIMDs issue VALU instructions, and how many lanes each one carries
                    if shade_sq and timing.get("shade") and timing["shade"][1]:
                        sh_ms = timing["shade"][0] / timing["shade"][1]          # live average launch time (HIP events on the ctx stream)
                        sh_cycles = sh_ms * 1e-3 * clk_ghz * 1e9                  # (the clock measured inside the traversal launches of the same frame)
                        per_simd_sh = shade_sq["valu_insts_per_launch"] / simds
                        shade_sq["avg_launch_ms"] = round(sh_ms, 3)
                        shade_sq["launches_per_step"] = timing["shade"][1] / args.steps
                        shade_sq["simd_cycles_per_valu_inst"] = round(sh_cycles / per_simd_sh, 3)
                        shade_sq["probe_cycles_per_valu_inst"] = PROBE_FMA_CYCLES   # a SIMD saturated with independent v_fma_f32 (tools/valu_probe2 kind 0): the arithmetic of shading is fma / mul / add
                        shade_sq["issue_frac"] = round(PROBE_FMA_CYCLES / shade_sq["simd_cycles_per_valu_inst"], 4)
                        shade_sq["frac_valu_lane_throughput"] = round(shade_sq["issue_frac"] * shade_sq["lanes_active_per_valu_inst"] / 64.0, 4)
                        shade_sq["bound"] = "valu_issue"
                        shade_sq["note"] = ("SQ counters of the same rocprofv3 --pmc pass as valu_issue; launch time live from this run; issue_frac = the measured issue rate of plain v_fma_f32 "
                                            "(%.2f cycles per wave instruction per SIMD, profiles/r04_a_valu_probe2_issue_rate_and_clock.txt) / the SIMD cycles one VALU instruction of this kernel "
                                            "costs, all stalls included; memory side (not measured in this run): 132.7 GB of read requests per 64 spp C3 frame = 1.03 TB/s, 0.13 of the 8 TB/s peak "
                                            "(profiles/r06_w_kernel_traffic.txt, the same TCC_EA0_RDREQ pass as roofline.traffic, per kernel)" % PROBE_FMA_CYCLES)
                        roofline_shade = shade_sq
            except Exception as e:   # measurement aid only
                log("[bench] VALU issue figure not measured: %s" % e)
        # the three named fractions: SURVEY.md s.8(d)'s algorithmic bytes against the HBM peak (> 1: served by LDS / L2 / MALL, not an HBM
        # quantity), the counter-measured HBM fraction (= `frac`; the north star's >= 0.40 target is NOT met and HBM is not the binding roof), and the roof
        # that binds: VALU issue x lane occupancy
        roofline["frac_alg_8d"] = round(roofline["alg_bytes_per_launch"] / (avg_launch_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4) if avg_launch_ms > 0 else None
        roofline["frac_hbm_counter"] = roofline["frac"]
        roofline["binding_roof"] = "VALU issue (frac_valu_lane_throughput = issue_frac x lanes_active / 64); HBM target of the north star (>= 0.40 of 8 TB/s in this kernel) unmet"
        # the line itself says what is true: `bound` names the roof `frac` is quoted against (the contract's HBM figure, from calibrated counters),
        # `bound_actual` the roof that binds, and the north star's >= 0.40 HBM target is reported as a boolean
        roofline["bound_actual"] = "valu_issue"
        roofline["hbm_target"] = {"north_star": ">= 0.40 of the 8 TB/s HBM peak in the traversal kernel", "met": bool(roofline.get("frac") is not None and roofline["frac"] >= 0.40),
                                  "note": "frac counts the L2's memory-side requests by size; Infinity-Cache (MALL, 256 MiB) hits are included -- no counter of this tool separates them "
                                          "(TCC_EA0_RDREQ_DRAM equals TCC_EA0_RDREQ) -- so true HBM utilisation is <= frac"}
        kernel_ms = {k: round(v[0] / args.steps, 3) for k, v in timing.items() if v[1]}

        # ---- CPU baseline beside it (rank 0, N = 1): the REFERENCE's own multithreaded path -- oracle/_ref/pbrt_ref, built from the
        # unmodified sources -- on a centre crop of the same frame at full spp; time = SamplerIntegrator::Render's tile loop
        # (stamped by the build shim, PBRT_REF_RENDER_TIMES), samples / rays from its own statistics.  The oracle port is
        # timed as a second sample.  Reported baselines, not targets.
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            import oracle_lib as ol
            ncores = host_cpus()
            cpu = cpu_reference(sc, scene_file, ncores, args.cpu_seconds, ol)
            port = cpu_port(sc, ctx, ncores, args.cpu_port_seconds if cpu else max(args.cpu_port_seconds, args.cpu_seconds), ol) \
                if (args.cpu_port_seconds > 0 or cpu is None) else None
            if cpu is None:
                cpu = port
            elif port:
                cpu["port"] = port

        out = {"metric": "Msamples/sec (whole node), San Miguel 1080p", "value": round(msamples, 3), "unit": "Msamples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "tiles": "16x16, 2-D lattice over ranks (mi_tile_owner)", "parallelism": "tile-sharded x%d" % world},
               "mrays_per_s": round(mrays, 2), "rays_per_sample": round(samples[1] / max(1.0, samples[0]), 3),
               "roofline": roofline, "roofline_shade": roofline_shade, "cpu_baseline": cpu, "kernel_ms_per_step": kernel_ms,
               "setup_s": {"parse_and_bvh_build": round(t_load, 2), "upload_and_bvh4": round(t_upload, 2),
                           "scene": "rank 0 of the node builds it once, the other ranks map the published blob" if world > 1 else "built"}}
        if world > 1:
            out["setup_s"]["scene_by_rank"] = scene_sources
    phase("closing")
    frame.close()
    ctx.close()
    phase("closed")
    if rank == 0:
        # ---- the San-Miguel-like variant beside the headline: what the real scene is -- textured, bump-mapped materials and alpha-masked foliage --
        # rendered by a child process once this process has released the GPU memory of the headline frame; the headline workload itself is unchanged
        plain_c3 = args.config == "c3" and not (args.textured or args.leafmask or args.volpath or args.fogbox or args.subsurface or args.smokebox or args.scene)
        if world == 1 and (args.secondary == "on" or (args.secondary == "auto" and plain_c3 and args.spp == 64 and args.tris == 10_000_000 and args.cpu_seconds > 0)):
            out["secondary"] = {"textured_leafmask": secondary_line(["--textured", "--leafmask", "--tris", str(args.tris), "--res", str(args.res[0]), str(args.res[1]), "--spp", str(args.spp)])}
        print(json.dumps(out), flush=True)
        # parity at full size is part of the measurement: a frame that does not match the reference's is not a result
        bad = []
        for name, line in [("headline", out)] + [("secondary." + k, v) for k, v in (out.get("secondary") or {}).items()]:
            pc = ((line or {}).get("cpu_baseline") or {}).get("parity_crop") if name == "headline" else (line or {}).get("parity_crop")
            if pc and "pixels_within_tol" in pc and (pc["pixels_within_tol"] < 0.995 or not (pc.get("relMSE", 0) <= 1e-4)):
                bad.append("%s: %.5f of the crop's pixels within tolerance, relMSE %.3g" % (name, pc["pixels_within_tol"], pc.get("relMSE", float("nan"))))
        if bad:
            raise SystemExit("bench.py: the frame does NOT match pbrt_ref's crop -- " + "; ".join(bad))
    if world > 1 and local_rank == 0:   # every rank has passed the final barrier of frame.close(): nobody still needs the file (mappings stay valid anyway)
        par.node_scene_cleanup()


if __name__ == "__main__":
    main()
