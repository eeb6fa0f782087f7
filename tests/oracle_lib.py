"""ctypes access to oracle/liboracle.so (the CPU restatement) and oracle/_ref/pbrt_ref (the real
reference, when built).  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
pa = importlib.import_module("pbrt-v3-distributed_amd")

ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
PBRT_REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(ORACLE_SO)
        L.oracle_sobol_interval_to_index.restype = C.c_uint64
        L.oracle_sobol_interval_to_index.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        L.oracle_sobol_sample_float.restype = C.c_float
        L.oracle_sobol_sample_float.argtypes = [C.c_int64, C.c_int]
        L.oracle_sobol.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_intersect_p.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_triangle_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_li.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_sphere_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_bxdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.oracle_texture_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_render_sharded.restype = C.c_double
        L.oracle_render_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_sample_discrete.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_void_p]
        L.oracle_distribution1d.restype = C.c_float
        L.oracle_distribution1d.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_render.restype = C.c_double
        L.oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def render(scene, spp_begin=0, spp_end=-1, nthreads=0, tiles=None, rank=0, world=1):
    """-> (rgbw (H,W,4), counters dict, seconds); rank/world: only the tiles mi_render gives that rank"""
    rgbw = np.zeros((scene.height, scene.width, 4), dtype=np.float32)
    cnt = np.zeros(8, dtype=np.uint64)
    t = None if tiles is None else np.asarray(tiles, dtype=np.int32)
    secs = lib().oracle_render_sharded(scene.desc, _p(rgbw), spp_begin, spp_end, nthreads, _p(cnt), _p(t) if t is not None else None, rank, world)
    names = ["camera_rays", "closest_rays", "shadow_rays", "nodes_closest", "tris_closest", "nodes_any", "tris_any"]
    return rgbw, dict(zip(names, [int(v) for v in cnt[:7]])), secs


def sobol(scene, px, py, n_samples, n_dims):
    out = np.zeros((n_samples, n_dims), dtype=np.float32)
    idx = np.zeros(n_samples, dtype=np.uint64)
    lib().oracle_sobol(scene.desc, px, py, n_samples, n_dims, _p(out), _p(idx))
    return out, idx


def camera_rays(scene, pixels_xy, sample_num):
    pixels_xy = np.ascontiguousarray(pixels_xy, dtype=np.int32)
    sample_num = np.ascontiguousarray(sample_num, dtype=np.int32)
    n = len(sample_num)
    rays = np.zeros(n, dtype=pa.RAY_DTYPE)
    pf = np.zeros((n, 2), dtype=np.float32)
    lib().oracle_camera_rays(scene.desc, _p(pixels_xy), _p(sample_num), n, _p(rays), _p(pf))
    return rays, pf


def intersect(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=pa.RAY_DTYPE)
    hits = np.zeros(len(rays), dtype=pa.HIT_DTYPE)
    cnt = np.zeros(2, dtype=np.uint64)
    lib().oracle_intersect(scene.desc, _p(rays), len(rays), _p(hits), _p(cnt))
    return hits, cnt


def intersect_p(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=pa.RAY_DTYPE)
    occ = np.zeros(len(rays), dtype=np.uint8)
    cnt = np.zeros(2, dtype=np.uint64)
    lib().oracle_intersect_p(scene.desc, _p(rays), len(rays), _p(occ), _p(cnt))
    return occ, cnt


def li(scene, pixels_xy, sample_num):
    pixels_xy = np.ascontiguousarray(pixels_xy, dtype=np.int32)
    sample_num = np.ascontiguousarray(sample_num, dtype=np.int32)
    out = np.zeros((len(sample_num), 3), dtype=np.float32)
    lib().oracle_li(scene.desc, _p(pixels_xy), _p(sample_num), len(sample_num), _p(out))
    return out


def triangle_intersect(p0, p1, p2, o, d, tmax=np.inf):
    ray = np.zeros(1, dtype=pa.RAY_DTYPE)
    ray["o"][0] = o; ray["d"][0] = d; ray["tmax"][0] = tmax
    t = C.c_float(0)
    b = np.zeros(3, dtype=np.float32)
    v = [np.ascontiguousarray(x, dtype=np.float32) for x in (p0, p1, p2)]
    hit = lib().oracle_triangle_intersect(_p(v[0]), _p(v[1]), _p(v[2]), _p(ray), C.byref(t), _p(b))
    return bool(hit), t.value, b


def distribution1d(func):
    func = np.ascontiguousarray(func, dtype=np.float32)
    cdf = np.zeros(len(func) + 1, dtype=np.float32)
    fi = lib().oracle_distribution1d(_p(func), len(func), _p(cdf))
    return cdf, float(fi)


def sample_discrete(func, cdf, func_int, u):
    func = np.ascontiguousarray(func, dtype=np.float32); cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    pdf = C.c_float(0)
    idx = lib().oracle_sample_discrete(_p(func), _p(cdf), func_int, len(func), u, C.byref(pdf))
    return idx, pdf.value


def have_ref():
    return os.path.exists(PBRT_REF)


def run_ref(scene_file, outfile, nthreads=None, extra=()):
    """Render with the real reference binary; returns the image (H,W,3)."""
    cmd = [PBRT_REF, "--quiet", "--outfile", outfile]
    if nthreads:
        cmd += ["--nthreads", str(nthreads)]
    cmd += list(extra) + [scene_file]
    subprocess.check_call(cmd)
    return pa.read_pfm(outfile)


def sphere_records(rows):
    """ref_vectors.npz 'spheres' rows -> (mi_sphere array, ray array)"""
    sp = np.zeros(len(rows), dtype=pa.SPHERE_DTYPE)
    for k in ("o2w", "w2o", "radius", "zmin", "zmax", "theta_min", "theta_max", "phi_max", "area"):
        sp[k] = rows[k]
    sp["flags"] = rows["flags"].astype(np.uint32)
    rays = np.zeros(len(rows), dtype=pa.RAY_DTYPE)
    rays["o"] = rows["o"]; rays["d"] = rows["d"]; rays["tmax"] = rows["tmax"]
    return sp, rays


def bxdf_eval(rows):
    """ref_vectors.npz 'bxdfs' rows -> dict of the oracle's f, pdf, Sample_f results"""
    n = len(rows)
    b = np.ascontiguousarray(rows["bxdf"]); wo = np.ascontiguousarray(rows["wo"]); wi = np.ascontiguousarray(rows["wi"]); u = np.ascontiguousarray(rows["u"])
    out = {"f": np.zeros((n, 3), np.float32), "pdf": np.zeros(n, np.float32), "wi_s": np.zeros((n, 3), np.float32), "pdf_s": np.zeros(n, np.float32),
           "f_s": np.zeros((n, 3), np.float32), "type_s": np.zeros(n, np.int32)}
    lib().oracle_bxdf(_p(b), _p(wo), _p(wi), _p(u), n, _p(out["f"]), _p(out["pdf"]), _p(out["wi_s"]), _p(out["pdf_s"]), _p(out["f_s"]), _p(out["type_s"]))
    return out


def texture_eval(scene, node, queries):
    queries = np.ascontiguousarray(queries, dtype=pa.TEX_QUERY_DTYPE)
    out = np.zeros((len(queries), 3), dtype=np.float32)
    lib().oracle_texture_eval(scene.desc, int(node), _p(queries), len(queries), _p(out))
    return out


def light_sample(scene, queries):
    """oracle_light_sample: Light::Sample_Li / Pdf_Li at explicit reference points (records of pa.LIGHT_QUERY_DTYPE)"""
    q = np.ascontiguousarray(queries, dtype=pa.LIGHT_QUERY_DTYPE)
    out = np.zeros(len(q), dtype=pa.LIGHT_RESULT_DTYPE)
    L = lib()
    L.oracle_light_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.oracle_light_sample.restype = None
    L.oracle_light_sample(scene.desc, _p(q), len(q), _p(out))
    return out


def next_float_array(v, direction):
    """NextFloatUp (direction > 0) / NextFloatDown of every element (core/pbrt.h:237-263 as the oracle restates them)"""
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros_like(v)
    L = lib()
    L.oracle_next_float.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.oracle_next_float.restype = None
    L.oracle_next_float(_p(v), len(v), int(direction), _p(out))
    return out


def next_float(v, direction):
    return next_float_array(np.array([v], dtype=np.float32), direction)[0]


def sphere_intersect(spheres, rays):
    hits = np.zeros(len(rays), dtype=pa.SPHERE_HIT_DTYPE)
    lib().oracle_sphere_intersect(_p(spheres), _p(rays), len(rays), _p(hits))
    return hits


def image_metrics(img, ref):
    """per-pixel L2 criterion of SURVEY.md s.8(c): fraction of pixels with |d|_2 <= 1e-3 (1+|ref|_2), and relMSE."""
    d = np.linalg.norm(img.astype(np.float64) - ref.astype(np.float64), axis=-1)
    r = np.linalg.norm(ref.astype(np.float64), axis=-1)
    frac = float(np.mean(d <= 1e-3 * (1 + r)))
    relmse = float(np.mean(d ** 2) / max(1e-30, np.mean(r ** 2)))
    return frac, relmse


class _BssrdfTable(C.Structure):
    _fields_ = [("n_rho", C.c_int32), ("n_radius", C.c_int32), ("rho_samples", C.c_void_p), ("radius_samples", C.c_void_p), ("profile", C.c_void_p),
                ("rho_eff", C.c_void_p), ("profile_cdf", C.c_void_p)]


def bssrdf_table(rec):
    """mi_bssrdf_table over the arrays of one record of tests/golden/bssrdf_tables.npz 'tables' (the arrays are kept alive on the returned object)"""
    keep = [np.ascontiguousarray(rec[k], dtype=np.float32) for k in ("rho_samples", "radius_samples", "profile", "rho_eff", "profile_cdf")]
    t = _BssrdfTable(100, 64, *[a.ctypes.data for a in keep])
    t._keep = keep
    return t


def bssrdf_radial(table, eta, recs):
    """TabulatedBSSRDF::Sr / Sample_Sr / Pdf_Sr of the oracle for the coefficient records -> (Sr rgb, Sample_Sr, Pdf_Sr)"""
    n = len(recs)
    sa, ss = np.ascontiguousarray(recs["sigma_a"], np.float32), np.ascontiguousarray(recs["sigma_s"], np.float32)
    ch, r, u = np.ascontiguousarray(recs["ch"], np.int32), np.ascontiguousarray(recs["r"], np.float32), np.ascontiguousarray(recs["u"], np.float32)
    sr, smp, pdf = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    L = lib()
    L.oracle_bssrdf_radial.argtypes = [C.c_void_p, C.c_float] + [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 3
    L.oracle_bssrdf_radial.restype = None
    L.oracle_bssrdf_radial(C.byref(table), eta, _p(sa), _p(ss), _p(ch), _p(r), _p(u), n, _p(sr), _p(smp), _p(pdf))
    return sr, smp, pdf


def subsurface_from_diffuse(table, kd, mfp):
    kd, mfp = np.ascontiguousarray(kd, np.float32), np.ascontiguousarray(mfp, np.float32)
    sa, ss = np.zeros_like(kd), np.zeros_like(kd)
    L = lib()
    L.oracle_subsurface_from_diffuse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.oracle_subsurface_from_diffuse.restype = None
    L.oracle_subsurface_from_diffuse(C.byref(table), _p(kd), _p(mfp), len(kd), _p(sa), _p(ss))
    return sa, ss


def hg(recs):
    n = len(recs)
    g, wo, wi, u = [np.ascontiguousarray(recs[k], np.float32) for k in ("g", "wo", "wi", "u")]
    p, ws, ps = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    L = lib()
    L.oracle_hg.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3
    L.oracle_hg.restype = None
    L.oracle_hg(_p(g), _p(wo), _p(wi), _p(u), n, _p(p), _p(ws), _p(ps))
    return p, ws, ps
