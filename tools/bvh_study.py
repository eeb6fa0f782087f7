#!/usr/bin/env python3
"""Offline study for the next traversal design (NOT product code): how many nodes / triangles a ray visits under different BVH
layouts of the SAME reference tree -- BVH2 as the reference traverses it, the device's BVH4 collapse with and without dropping
popped entries beyond the current hit, and a BVH8 collapse -- on camera rays and on incoherent diffuse-bounce rays of the
San-Miguel-class stand-in.  Prints visits per ray and the bytes they cost with 128-byte BVH4 / 256-byte BVH8 / 80-byte
compressed BVH8 nodes (Ylitie et al. 2017) and 48-byte triangles.

  python tools/bvh_study.py [--tris 1000000]
"""
import argparse, ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--rays", type=int, default=200000)
    ap.add_argument("--rebuild", action="store_true", help="tree study: the same triangles under an own SAH tree (3 axes, binned / exact sweep, leaf sizes) instead of the reference's")
    ap.add_argument("--wavesim", action="store_true", help="wave-scheduling simulator: SIMT efficiency of k_trace's node / leaf phases under different policies")
    ap.add_argument("--ray-order", default="random", help="--wavesim: order of the bounce rays in the queue: random | morton (origin cell) | morton_oct (direction octant, then origin cell) | cellN (N^3 grid of origin cells x octant, counting-sort-like: stable inside a cell)")
    ap.add_argument("--reinsert", action="store_true", help="the product's own topology over the reference's leaves with and without the insertion-based optimisation pass (pt_treebuild.h)")
    ap.add_argument("--hot", action="store_true", help="hot-node study: share of the BVH4 node visits on the K nodes a block could hold in LDS")
    args = ap.parse_args()
    import oracle_lib as ol
    pa = importlib.import_module("pbrt-v3-distributed_amd")
    so = "/tmp/libbvhstudy.so"
    L0 = None
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-shared", "-fPIC", os.path.join(ROOT, "tools", "bvh_study.cpp"), "-I" + os.path.join(ROOT, "include"), "-o", so])
    L = C.CDLL(so)
    L.bvh_study.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.bvh_study_collapse.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int]
    f = "/tmp/bvh_study_scene.pbrt"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", str(args.tris), "--res", "640", "360", "--spp", "1", "--out", f], stdout=subprocess.DEVNULL)
    sc = pa.Scene(f)
    rng = np.random.RandomState(1)
    xy = np.stack([rng.randint(0, sc.width, args.rays), rng.randint(0, sc.height, args.rays)], 1).astype(np.int32)
    cam, _ = ol.camera_rays(sc, xy, np.zeros(args.rays, np.int32))
    hits, _cnt = ol.intersect(sc, cam)
    ok = hits["prim"] >= 0
    p = cam["o"][ok] + cam["d"][ok] * hits["t"][ok][:, None]
    n = hits["n"][ok]
    n = n * np.sign(-(n * cam["d"][ok]).sum(1))[:, None]          # towards the camera side
    # cosine-distributed bounce directions about n
    u1, u2 = rng.rand(len(p)), rng.rand(len(p))
    r, ph = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, 0:1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])
    t1 = np.cross(n, a); t1 /= np.linalg.norm(t1, axis=1)[:, None]; t2 = np.cross(n, t1)
    d = t1 * (r * np.cos(ph))[:, None] + t2 * (r * np.sin(ph))[:, None] + n * np.sqrt(np.maximum(0, 1 - u1))[:, None]
    sec = np.zeros(len(p), dtype=pa.RAY_DTYPE)
    sec["o"] = (p + n * 1e-3).astype(np.float32); sec["d"] = d.astype(np.float32); sec["tmax"] = np.inf
    print("scene: %d triangles, %d BVH2 nodes; %d camera rays, %d bounce rays" % (sc.info["n_tris"], sc.info["n_bvh_nodes"], len(cam), len(sec)))
    if args.reinsert:
        L.bvh_study_reinsert.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.bvh_study_collapse(0, C.c_float(1), C.c_float(1), 0)
        print("BVH4Q (greedy-area collapse, cull on pop) over the reference's leaves: the reference's tree, the product's top-down SAH topology, and that topology after reinsertion passes")
        print("%-8s %-52s %9s %9s %9s %9s %9s %7s" % ("rays", "tree", "nodes/ray", "tris/ray", "req/ray", "SAH rel.", "moved", "s"))
        sh = sec.copy(); sh["tmax"] = 3.0
        for name, rays, anyhit in (("camera", cam, 0), ("bounce", sec, 0), ("shadow", sh, 1)):
            rays = np.ascontiguousarray(rays)
            out = np.zeros(8)
            L.bvh_study(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), 4, 1, anyhit, out.ctypes.data_as(C.c_void_p))
            nr, tr = out[0] / len(rays), out[1] / len(rays)
            print("%-8s %-52s %9.2f %9.2f %9.1f" % (name, "reference", nr, tr, 3 + 4 * nr + 3 * tr))
            base = None
            for label, frac, passes in (("product: 3 axes x 32 bins, sweep below 2048", 0.0, 0), ("+ reinsertion: 2 % largest subtrees, 1 pass", 0.02, 1), ("+ reinsertion: 5 %, 2 passes", 0.05, 2),
                                        ("+ reinsertion: 20 %, 2 passes", 0.2, 2), ("+ reinsertion: 100 %, 2 passes", 1.0, 2), ("+ reinsertion: 100 %, 5 passes", 1.0, 5)):
                out = np.zeros(8)
                L.bvh_study_reinsert(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), 1, anyhit, frac, passes, out.ctypes.data_as(C.c_void_p))
                nr, tr = out[0] / len(rays), out[1] / len(rays)
                req = 3 + 4 * nr + 3 * tr
                if base is None: base = req
                print("%-8s %-52s %9.2f %9.2f %9.1f %9.3f %9d %7.1f   (%+.1f %% requests)" % (name, label, nr, tr, req, out[4], out[5], out[6], 100 * (req / base - 1)))
        return
    if args.hot:
        L.bvh_study_hot.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        Ks = np.array([64, 128, 256, 512, 768, 1024, 1536, 2048, 4096], dtype=np.int32)
        print("share of the interior-node visits of the device's BVH4 (greedy-area collapse, cull on pop) that land on K nodes")
        print("%-8s %-8s %9s | %s" % ("rays", "kind", "nodes/ray", "  ".join("K=%-5d" % k for k in Ks)))
        sh = sec.copy(); sh["tmax"] = 3.0   # short any-hit rays, shadow-ray-like
        for name, rays, anyhit in (("camera", cam, 0), ("bounce", sec, 0), ("bounce", sec, 1)):
            rays = np.ascontiguousarray(rays)
            out = np.zeros(2 * len(Ks) + 1)
            L.bvh_study_hot(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), anyhit, Ks.ctypes.data_as(C.c_void_p), len(Ks), out.ctypes.data_as(C.c_void_p))
            print("%-8s %-8s %9.2f | %s   <- the K most visited nodes (needs the rays)" % (name, "any" if anyhit else "closest", out[-1], "  ".join("%6.1f%%" % (100 * out[2 * i]) for i in range(len(Ks)))))
            print("%-8s %-8s %9s | %s   <- first K of a largest-area-first expansion from the root (static)" % ("", "", "", "  ".join("%6.1f%%" % (100 * out[2 * i + 1]) for i in range(len(Ks)))))
        return
    if args.wavesim:
        L.bvh_study_wavesim.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        rng2 = np.random.RandomState(5)
        rays = np.ascontiguousarray(sec[rng2.permutation(len(sec))])
        if args.ray_order != "random":   # round 6: what would a spatial sort of the extension rays buy the wave scheduler?
            o = rays["o"].astype(np.float64); lo = o.min(0); ext = (o.max(0) - lo).max()
            octant = ((rays["d"][:, 0] < 0).astype(np.int64) | ((rays["d"][:, 1] < 0).astype(np.int64) << 1) | ((rays["d"][:, 2] < 0).astype(np.int64) << 2))
            if args.ray_order.startswith("cell"):
                N = int(args.ray_order[4:]); c = np.minimum(N - 1, ((o - lo) / ext * N).astype(np.int64))
                key = ((c[:, 2] * N + c[:, 1]) * N + c[:, 0]) * 8 + octant
            else:
                q = np.minimum(1023, ((o - lo) / ext * 1024).astype(np.int64))
                def spread(v):
                    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
                key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
                if args.ray_order == "morton_oct": key = key | (octant << 30)
            rays = np.ascontiguousarray(rays[np.argsort(key, kind="stable")])
            print("ray order: %s (%d distinct keys)" % (args.ray_order, len(np.unique(key))))
        CN, CL = 143.0, 163.0   # VALU instructions of one node phase / one triangle phase of k_trace<0, QN> (round 4: profiles/r04_h_isa_node_step.txt)
        print("64-lane waves over %d incoherent bounce rays; cost = node phases x %.0f + triangle phases x %.0f instructions" % (len(rays), CN, CL))
        print("%-64s %9s %9s %8s %9s %9s %8s %10s" % ("policy", "nodePh/ray", "lanes/ph", "eff", "triPh/ray", "lanes/ph", "eff", "instr/ray"))
        base = None
        for label, pol, ns, lm, rf in (("k_trace today: <= 8 node phases, leaf phase at 24 lanes, 1 triangle, refill 16", 0, 8, 24, 16),
                                        ("same, refill at 8 idle lanes", 0, 8, 24, 8), ("same, refill at 32", 0, 8, 24, 32),
                                        ("<= 4 node phases", 0, 4, 24, 16), ("<= 16 node phases, leaf at 32", 0, 16, 32, 16), ("<= 32 node phases, leaf at 40", 0, 32, 40, 16),
                                        ("<= 64 node phases, leaf at 48", 0, 64, 48, 16), ("<= 64 node phases, leaf at 56", 0, 64, 56, 16),
                                        ("whole leaf per leaf phase, <= 8 / 24", 1, 8, 24, 16), ("whole leaf, <= 32 / 40", 1, 32, 40, 16), ("whole leaf, <= 64 / 56", 1, 64, 56, 16),
                                        ("pending leaf (1 tri per phase) + node steps go on, <= 8 / 24", 3, 8, 24, 16), ("pending leaf, <= 4 / 24", 3, 4, 24, 16), ("pending leaf, <= 4 / 32", 3, 4, 32, 16),
                                        ("pending leaf, <= 2 / 32", 3, 2, 32, 16), ("pending leaf, <= 8 / 32", 3, 8, 32, 16), ("pending leaf, <= 8 / 40", 3, 8, 40, 16), ("pending leaf, <= 16 / 48", 3, 16, 48, 16),
                                        ("pending leaf, <= 4 / 24, refill 8", 3, 4, 24, 8), ("pending leaf, <= 8 / 24, refill 8", 3, 8, 24, 8), ("pending leaf, <= 4 / 24, refill 24", 3, 4, 24, 24),
                                        ("2 pending leaves, <= 8 / 24", 4, 8, 24, 16), ("2 pending leaves, <= 4 / 24, refill 8", 4, 4, 24, 8), ("3 pending leaves, <= 8 / 24", 5, 8, 24, 16),
                                        ("pending leaf, <= 8 / 24 + entries beyond the hit dropped at pop", 3, 1008, 24, 16), ("today's policy + dropped at pop", 0, 1008, 24, 16),
                                        ("TWO rays per lane, each with a pending leaf, <= 8 / 24, refill 16 lanes' worth", 6, 8, 24, 16), ("two rays per lane, <= 8 / 32", 6, 8, 32, 16), ("two rays per lane, <= 8 / 40", 6, 8, 40, 16),
                                        ("two rays per lane, <= 8 / 48", 6, 8, 48, 16), ("two rays per lane, <= 16 / 48", 6, 16, 48, 16), ("two rays per lane, <= 8 / 40, refill 8", 6, 8, 40, 8), ("two rays per lane, <= 8 / 56", 6, 8, 56, 16),
                                        ("SHARED triangle phases (parked triangles over all 64 lanes), <= 8 / 24", 7, 8, 24, 16), ("shared triangle phases, <= 8 / 32", 7, 8, 32, 16), ("shared triangle phases, <= 8 / 16", 7, 8, 16, 16),
                                        ("postponed leaves, <= 8 / 24", 2, 8, 24, 16), ("postponed leaves, <= 16 / 32", 2, 16, 32, 16), ("postponed leaves, <= 32 / 48", 2, 32, 48, 16)):
            out = np.zeros(8)
            L.bvh_study_wavesim(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), pol, ns, lm, rf, out.ctypes.data_as(C.c_void_p))
            nr = out[0]
            cost = (out[1] * CN + out[3] * CL) / (nr / 64.0) / 64.0
            if base is None: base = cost
            print("%-64s %9.2f %9.1f %7.1f%% %9.2f %9.1f %7.1f%% %10.0f (%+.1f %%)  nodes/ray %.2f tris/ray %.2f" % (label, out[1] * 64 / nr, out[2] / max(1, out[1]), 100 * out[2] / max(1, out[1]) / 64,
                  out[3] * 64 / nr, out[4] / max(1, out[3]), 100 * out[4] / max(1, out[3]) / 64, cost, 100 * (cost / base - 1), out[2] / nr, out[4] / nr))
        return
    if args.rebuild:
        L.bvh_study_rebuilt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.bvh_study_collapse(0, C.c_float(1), C.c_float(1), 0)
        print("BVH4Q (greedy-area collapse, cull on pop) over different BVH2 trees of the same triangles; requests/ray = 3 + 4 nodes + 3 triangles")
        print("%-8s %-58s %9s %9s %9s %9s %9s" % ("rays", "tree", "nodes/ray", "tris/ray", "req/ray", "SAH rel.", "tris/leaf"))
        for name, rays in (("camera", cam), ("bounce", sec)):
            rays = np.ascontiguousarray(rays)
            out = np.zeros(8)
            L.bvh_study(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), 4, 1, 0, out.ctypes.data_as(C.c_void_p))
            nr, tr = out[0] / len(rays), out[1] / len(rays)
            base = 3 + 4 * nr + 3 * tr
            print("%-8s %-58s %9.2f %9.2f %9.1f %9s %9s" % (name, "reference: 12 buckets on the widest centroid axis, leaves <= 4", nr, tr, base, "1.000", "-"))
            for label, bins, sweep, leaf, cn, ct in (("3 axes x 16 bins, leaves <= 4", 16, 0, 4, 1, 1), ("3 axes x 32 bins, sweep below 256, leaves <= 4", 32, 256, 4, 1, 1),
                                                     ("3 axes x 32 bins, sweep below 4096, leaves <= 4", 32, 4096, 4, 1, 1),
                                                     ("same, leaves <= 2", 32, 4096, 2, 1, 1), ("same, leaves <= 8", 32, 4096, 8, 1, 1),
                                                     ("same, leaves <= 8, cTri 0.5", 32, 4096, 8, 1, 0.5), ("same, leaves <= 16, cTri 0.5", 32, 4096, 16, 1, 0.5),
                                                     ("same, leaves <= 4, cTri 2", 32, 4096, 4, 1, 2)):
                out = np.zeros(8)
                L.bvh_study_rebuilt(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), 4, 1, 0, bins, sweep, leaf, C.c_float(cn), C.c_float(ct), out.ctypes.data_as(C.c_void_p))
                nr, tr = out[0] / len(rays), out[1] / len(rays)
                req = 3 + 4 * nr + 3 * tr
                print("%-8s %-58s %9.2f %9.2f %9.1f %9.3f %9.2f   (%+.1f %% requests)" % (name, label, nr, tr, req, out[4], out[6], 100 * (req / base - 1)))
            # round 6: spatial splits (SBVH); the "SAH rel." column shows references per triangle instead
            L.bvh_study_sbvh.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
            for label, bins, leaf, cn, ct, alpha in (("own leaves, 3 x 32 bins, no spatial splits, leaves <= 4", 32, 4, 1, 1, -1), ("SBVH alpha 1e-5, leaves <= 4", 32, 4, 1, 1, 1e-5),
                                                     ("SBVH alpha 1e-6, leaves <= 4", 32, 4, 1, 1, 1e-6), ("SBVH alpha 1e-4, leaves <= 4", 32, 4, 1, 1, 1e-4),
                                                     ("SBVH alpha 1e-5, leaves <= 4, cTri 2", 32, 4, 1, 2, 1e-5), ("SBVH alpha 1e-5, leaves <= 2", 32, 2, 1, 1, 1e-5),
                                                     ("SBVH alpha 0 (everywhere), leaves <= 4", 32, 4, 1, 1, 0.0)):
                out = np.zeros(8)
                L.bvh_study_sbvh(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), 4, 1, 0, bins, leaf, C.c_float(cn), C.c_float(ct), C.c_float(alpha), out.ctypes.data_as(C.c_void_p))
                nr, tr = out[0] / len(rays), out[1] / len(rays)
                req = 3 + 4 * nr + 3 * tr
                print("%-8s %-58s %9.2f %9.2f %9.1f %9.3f %9.2f   (%+.1f %% requests; refs/tri in col 6)" % (name, label, nr, tr, req, out[4], out[6], 100 * (req / base - 1)))
        return
    print("%-8s %-36s %10s %10s %12s" % ("rays", "layout", "nodes/ray", "tris/ray", "KB/ray"))
    for name, rays in (("camera", cam), ("bounce", sec)):
        rays = np.ascontiguousarray(rays)
        for label, width, cull, nodeB, dp in (("BVH2 reference order", 2, 0, 32, None), ("BVH4 (device layout)", 4, 0, 128, None), ("BVH4 + cull on pop", 4, 1, 128, None),
                                              ("BVH4Q greedy-area collapse + cull", 4, 1, 64, None),
                                              ("BVH4Q SAH-optimal collapse", 4, 1, 64, (1.0, 0.75, 0)), ("BVH4Q SAH-opt, leaves <= 4", 4, 1, 64, (1.0, 0.75, 4)),
                                              ("BVH4Q SAH-opt, leaves <= 8", 4, 1, 64, (1.0, 0.75, 8)), ("BVH4Q SAH-opt, leaves <= 16", 4, 1, 64, (1.0, 0.75, 16)),
                                              ("BVH4Q SAH-opt cTri .4, leaves <= 8", 4, 1, 64, (1.0, 0.4, 8)),
                                              ("BVH8", 8, 0, 256, None), ("BVH8 + cull on pop", 8, 1, 256, None), ("BVH8 compressed 80 B + cull", 8, 1, 80, None),
                                              ("BVH8c SAH-optimal collapse", 8, 1, 80, (1.0, 0.6, 0))):
            out = np.zeros(4)
            if dp: L.bvh_study_collapse(1, C.c_float(dp[0]), C.c_float(dp[1]), dp[2])
            else: L.bvh_study_collapse(0, C.c_float(1), C.c_float(1), 0)
            L.bvh_study(sc.desc, rays.ctypes.data_as(C.c_void_p), len(rays), width, cull, 0, out.ctypes.data_as(C.c_void_p))
            nr, tr = out[0] / len(rays), out[1] / len(rays)
            print("%-8s %-36s %10.2f %10.2f %12.2f  req/ray %7.1f  nodes %d" % (name, label, nr, tr, (48 + nr * nodeB + tr * 48) / 1024, 3 + nr * nodeB / 16 + tr * 3, out[3]))


if __name__ == "__main__":
    main()
