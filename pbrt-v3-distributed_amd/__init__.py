"""pbrt-v3-distributed_amd: ctypes binding over the C ABI of include/pbrt_amd.h.

Python here is plumbing only: it loads
  lib/libpbrt_amd_host.so  (C++ host: .pbrt parser, pbrt API state machine, SAH BVH build, Film)
  lib/libpbrt_amd.so       (hand-written HIP kernels for gfx950 behind extern "C" mi_* entry points)
and passes POD pointers between them.  There is no Python or CPU rendering path: every render /
intersect / sample call goes to the HIP library and raises if it (or a GPU) is unavailable.

Import with:  importlib.import_module("pbrt-v3-distributed_amd")   (the directory name is not an identifier)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HOST_LIB = os.path.join(LIB_DIR, "libpbrt_amd_host.so")
DEVICE_LIB = os.environ.get("PBRT_AMD_DEVICE_LIB", os.path.join(LIB_DIR, "libpbrt_amd.so"))   # env override: kernel-variant A/B runs

MI_CNT_COUNT = 16
MI_K_COUNT = 8
COUNTER_NAMES = ["camera_rays", "closest_rays", "shadow_rays", "nodes_closest", "tris_closest", "nodes_any",
                 "tris_any", "path_segments", "mis_rays", "nodes_mis", "tris_mis", "nodes_hot_closest", "nodes_hot_any", "nodes_hot_mis", "film_gather_builds",
                 "trace_guard_trips"]
KERNEL_NAMES = ["raygen", "closest", "sort", "shade", "anyhit", "mis_closest", "film", "other"]


class MiRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("tmax", C.c_float), ("d", C.c_float * 3), ("time", C.c_float)]


class MiHit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
                ("n", C.c_float * 3)]


class MiRenderParams(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("spp_begin", C.c_int32), ("spp_end", C.c_int32),
                ("count_work", C.c_int32), ("max_paths_in_flight", C.c_int32)]


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("tmax", np.float32), ("d", np.float32, 3), ("time", np.float32)])
HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("b0", np.float32), ("b1", np.float32),
                      ("b2", np.float32), ("n", np.float32, 3)])

# Every symbol include/pbrt_amd.h declares (checked by tests/test_abi.py against the header text)
DEVICE_SYMBOLS = [
    "mi_last_error", "mi_abi_version", "mi_ctx_create", "mi_ctx_destroy", "mi_scene_upload", "mi_render", "mi_sync",
    "mi_film_clear", "mi_film_download", "mi_film_device_ptr", "mi_film_bind", "mi_film_pixel_count", "mi_counters",
    "mi_counters_reset", "mi_timing_enable", "mi_timing_get", "mi_stream_read_gbps", "mi_gather_rate", "mi_trace_clock", "mi_owned_tiles", "mi_bvh4_validate", "mi_bvh4q_validate", "mi_trace_info", "mi_film_gather", "mi_rccl_probe", "mi_bxdf_eval", "mi_light_sample", "mi_bssrdf_eval", "mi_phase_hg", "mi_libm_eval", "mi_intersect", "mi_triangle_intersect", "mi_sphere_intersect", "mi_texture_eval", "mi_intersect_p", "mi_sobol",
    "mi_camera_rays", "mi_camera_differentials", "mi_li",
]

_host = None
_dev = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("%s missing: run __graft_entry__.build() / make -C pbrt-v3-distributed_amd" % HOST_LIB)
        L = C.CDLL(HOST_LIB, mode=C.RTLD_GLOBAL)
        L.pbrt_amd_scene_load.restype = C.c_void_p
        L.pbrt_amd_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
        L.pbrt_amd_scene_load_crop.restype = C.c_void_p
        L.pbrt_amd_scene_load_crop.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_float)]
        L.pbrt_amd_scene_free.argtypes = [C.c_void_p]
        L.pbrt_amd_scene_save_blob.argtypes = [C.c_void_p, C.c_char_p]
        L.pbrt_amd_scene_map_blob.restype = C.c_void_p
        L.pbrt_amd_scene_map_blob.argtypes = [C.c_char_p]
        L.pbrt_amd_scene_desc.restype = C.c_void_p
        L.pbrt_amd_scene_desc.argtypes = [C.c_void_p]
        L.pbrt_amd_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pbrt_amd_scene_texture_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pbrt_amd_scene_media_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pbrt_amd_scene_light.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        for f in ("pbrt_amd_film_merge", "pbrt_amd_film_rgb"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.pbrt_amd_film_clear.argtypes = [C.c_void_p]
        L.pbrt_amd_film_write.argtypes = [C.c_void_p, C.c_char_p]
        _host = L
    return _host


def device_lib():
    """The HIP library.  Fails loudly: there is no fallback path."""
    global _dev
    if _dev is None:
        if not os.path.exists(DEVICE_LIB):
            raise RuntimeError("%s missing: the HIP extension was not built (hipcc --offload-arch=gfx950)" % DEVICE_LIB)
        L = C.CDLL(DEVICE_LIB, mode=C.RTLD_GLOBAL)
        L.mi_last_error.restype = C.c_char_p
        L.mi_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mi_ctx_destroy.argtypes = [C.c_void_p]
        L.mi_scene_upload.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_render.argtypes = [C.c_void_p, C.POINTER(MiRenderParams)]
        L.mi_sync.argtypes = [C.c_void_p]
        L.mi_film_clear.argtypes = [C.c_void_p]
        L.mi_film_download.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_film_device_ptr.restype = C.c_void_p
        L.mi_film_device_ptr.argtypes = [C.c_void_p]
        L.mi_film_bind.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_film_pixel_count.restype = C.c_int64
        L.mi_film_pixel_count.argtypes = [C.c_void_p]
        L.mi_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_counters_reset.argtypes = [C.c_void_p]
        L.mi_timing_enable.argtypes = [C.c_void_p, C.c_int]
        L.mi_timing_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mi_bvh4_validate.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_bvh4q_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.mi_trace_info.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_film_gather.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mi_rccl_probe.argtypes = [C.c_void_p, C.c_int64]
        L.mi_texture_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_stream_read_gbps.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
        L.mi_trace_clock.argtypes = [C.c_void_p, C.c_void_p]
        L.mi_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_intersect_p.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_triangle_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_sphere_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_sobol.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mi_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.mi_camera_differentials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.mi_li.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _dev = L
    return _dev


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """A parsed .pbrt scene, flattened to the POD mi_scene_desc (host side only; no GPU needed)."""

    INFO_FIELDS = ["n_verts", "n_tris", "n_meshes", "n_bvh_nodes", "n_materials", "n_lights", "xres", "yres",
                   "crop_x0", "crop_y0", "crop_x1", "crop_y1", "spp", "max_depth", "sobol_resolution",
                   "sobol_log2_resolution"]

    def __init__(self, filename=None, text=None, quiet=True, outfile=None, cropwindow=None, blob=None, strict=False):
        """cropwindow = (x0, x1, y0, y1): the command line's --cropwindow (overrides the Film's own, as in the reference).
        strict = True: any Error() the loader logged while reading the scene (a missing or cut-off PLY file, a bad parameter -- things the reference logs and
        renders WITHOUT, main/pbrt.cpp keeps going) raises instead: a benchmark or a rank of a sharded job must never render a different scene.  `errors` = their count.
        blob = a file written by save_blob(): the mi_scene_desc of a scene ANOTHER process parsed and built, mapped read-only (shared page
        cache; one scene build per node instead of one per rank).  A mapped scene has no Film: film_image / write_image belong to the builder."""
        L = host_lib()
        src = text if text is not None else filename
        self.mapped = blob is not None
        L.pbrt_amd_error_count.restype = C.c_int
        errors0 = L.pbrt_amd_error_count()
        if blob is not None:
            self._h = L.pbrt_amd_scene_map_blob(blob.encode())
            if not self._h:
                raise RuntimeError("scene blob %r: missing, truncated or written by another ABI version" % blob)
        elif cropwindow is not None:
            cw = (C.c_float * 4)(*[float(v) for v in cropwindow])
            self._h = L.pbrt_amd_scene_load_crop(src.encode(), 1 if text is not None else 0, 1 if quiet else 0,
                                                 outfile.encode() if outfile else None, cw)
        else:
            self._h = L.pbrt_amd_scene_load(src.encode(), 1 if text is not None else 0, 1 if quiet else 0,
                                            outfile.encode() if outfile else None)
        if not self._h:
            raise RuntimeError("scene load failed: %r" % (filename or "<text>"))
        self.errors = L.pbrt_amd_error_count() - errors0
        if strict and self.errors:
            self.close()
            raise RuntimeError("scene %r: the loader reported %d error(s) (see stderr); strict mode does not render a scene that differs from the file" % (filename or "<text>", self.errors))
        self._finish_init()

    def _finish_init(self):
        L = host_lib()
        self.desc = L.pbrt_amd_scene_desc(self._h)
        info = (C.c_int64 * len(self.INFO_FIELDS))()
        L.pbrt_amd_scene_info(self._h, info)
        self.info = dict(zip(self.INFO_FIELDS, [int(v) for v in info]))
        tinfo = (C.c_int64 * 4)()
        L.pbrt_amd_scene_texture_info(self._h, tinfo)
        self.info.update(n_textures=int(tinfo[0]), n_images=int(tinfo[1]), n_textured_materials=int(tinfo[2]), n_masked_meshes=int(tinfo[3]))
        L.pbrt_amd_scene_media_info(self._h, tinfo)
        self.info.update(n_media=int(tinfo[0]), n_medium_transitions=int(tinfo[1]), camera_medium=int(tinfo[2]), integrator=("path", "volpath")[int(tinfo[3])])
        finfo = (C.c_double * 6)()
        L.pbrt_amd_scene_film_info.argtypes = [C.c_void_p, C.c_void_p]
        L.pbrt_amd_scene_film_info(self._h, finfo)
        self.info.update(filter_radius=(float(finfo[0]), float(finfo[1])), sample_bounds=(int(finfo[2]), int(finfo[3]), int(finfo[4]), int(finfo[5])))
        self.width = self.info["crop_x1"] - self.info["crop_x0"]
        self.height = self.info["crop_y1"] - self.info["crop_y0"]

    def save_blob(self, path):
        """Write the flattened scene (mi_scene_desc + every array it points to) to `path`, published atomically (rename): other ranks of the
        node map it with Scene(blob=path) instead of parsing and building the scene themselves."""
        rc = host_lib().pbrt_amd_scene_save_blob(self._h, path.encode())
        if rc != 0:
            raise RuntimeError("save_blob(%r) failed (%d)" % (path, rc))

    def light(self, i):
        """(type, emitted rgb) of light i of the flattened scene, as handed to mi_scene_upload"""
        t, rgb = C.c_int(), (C.c_float * 3)()
        if host_lib().pbrt_amd_scene_light(self._h, int(i), C.byref(t), rgb) != 0:
            raise IndexError(i)
        return t.value, np.array(rgb[:], dtype=np.float32)

    def close(self):
        if self._h:
            host_lib().pbrt_amd_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host Film (MergeFilmTile / WriteImage semantics, reference film.cpp:117-130,168-210)
    def film_image(self, rgbw):
        """rgbw: (H, W, 4) float32 {contribSum rgb, filterWeightSum} -> final (H, W, 3) image."""
        L = host_lib()
        if self.mapped:
            raise RuntimeError("a scene mapped from a blob has no Film: the process that built the scene owns the image")
        rgbw = np.ascontiguousarray(rgbw, dtype=np.float32)
        L.pbrt_amd_film_clear(self._h)
        L.pbrt_amd_film_merge(self._h, _ptr(rgbw))
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        L.pbrt_amd_film_rgb(self._h, _ptr(out))
        return out

    def write_image(self, rgbw, filename):
        self.film_image(rgbw)
        host_lib().pbrt_amd_film_write(self._h, filename.encode())


class Context:
    """One GPU context of the HIP path tracer with an uploaded scene."""

    def __init__(self, scene, device=0, stream=None):
        L = device_lib()
        self.scene = scene
        self._ctx = C.c_void_p()
        if L.mi_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._ctx)) != 0:
            raise RuntimeError("mi_ctx_create: %s" % L.mi_last_error().decode())
        if L.mi_scene_upload(self._ctx, scene.desc) != 0:
            raise RuntimeError("mi_scene_upload: %s" % L.mi_last_error().decode())

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, device_lib().mi_last_error().decode()))

    def close(self):
        if self._ctx:
            device_lib().mi_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, rank=0, world=1, spp_begin=0, spp_end=-1, count_work=False, max_paths=0, sync=True):
        p = MiRenderParams(rank, world, spp_begin, spp_end, 1 if count_work else 0, max_paths)
        self._chk(device_lib().mi_render(self._ctx, C.byref(p)), "mi_render")
        if sync:
            self.sync()

    def sync(self):
        self._chk(device_lib().mi_sync(self._ctx), "mi_sync")

    def rccl_probe(self, n_pixels=1 << 16):
        """mi_rccl_probe: mi_film_gather's RCCL step (communicator, one grouped ncclSend / ncclRecv of packed pixels, the add kernel) on this context's GPU alone"""
        self._chk(device_lib().mi_rccl_probe(self._ctx, int(n_pixels)), "mi_rccl_probe")

    def film_clear(self):
        self._chk(device_lib().mi_film_clear(self._ctx), "mi_film_clear")

    def film(self):
        out = np.empty((self.scene.height, self.scene.width, 4), dtype=np.float32)
        self._chk(device_lib().mi_film_download(self._ctx, _ptr(out)), "mi_film_download")
        return out

    def film_device_ptr(self):
        return device_lib().mi_film_device_ptr(self._ctx)

    def film_bind(self, device_ptr):
        """Render into a caller-owned device buffer of height*width float4 (e.g. a torch tensor's data_ptr())."""
        self._chk(device_lib().mi_film_bind(self._ctx, C.c_void_p(device_ptr) if device_ptr else None), "mi_film_bind")

    def trace_info(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(device_lib().mi_trace_info(self._ctx, _ptr(out)), "mi_trace_info")
        names = {0: "general steps over the 128-byte BVH4", 4: "two-level BVH4 (instanced scene)", 5: "general steps over the 64-byte quantised BVH4"}
        return {"mode": int(out[0]), "name": names[int(out[0])], "node_bytes": int(out[1]), "nodes": int(out[2]), "lds_stack_entries": int(out[3]),
                "hot_nodes": int(out[4]), "hot_probe_share": out[5] * 1e-6, "block_threads": int(out[6]), "blocks_per_cu": int(out[7])}

    def counters(self):
        out = np.zeros(MI_CNT_COUNT, dtype=np.uint64)
        self._chk(device_lib().mi_counters(self._ctx, _ptr(out)), "mi_counters")
        return dict(zip(COUNTER_NAMES, [int(v) for v in out[:len(COUNTER_NAMES)]]))

    def counters_reset(self):
        self._chk(device_lib().mi_counters_reset(self._ctx), "mi_counters_reset")

    def timing_enable(self, on=True):
        self._chk(device_lib().mi_timing_enable(self._ctx, 1 if on else 0), "mi_timing_enable")

    def timing(self):
        ms = np.zeros(MI_K_COUNT, dtype=np.float64)
        n = np.zeros(MI_K_COUNT, dtype=np.uint64)
        self._chk(device_lib().mi_timing_get(self._ctx, _ptr(ms), _ptr(n)), "mi_timing_get")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KERNEL_NAMES)}

    def trace_clock(self):
        """shader clock (GHz) inside the closest-hit / any-hit launches of the counting passes since the last counters_reset"""
        out = np.zeros(4, dtype=np.float64)
        self._chk(device_lib().mi_trace_clock(self._ctx, _ptr(out)), "mi_trace_clock")
        return {"closest_GHz": float(out[0]), "anyhit_GHz": float(out[1]), "closest_wave_cycles": float(out[2]), "anyhit_wave_cycles": float(out[3])}

    def stream_read_gbps(self, nbytes=4 << 30):
        """achievable HBM read rate (streaming read of nbytes), GB/s"""
        v = C.c_double(0)
        self._chk(device_lib().mi_stream_read_gbps(self._ctx, C.c_uint64(nbytes), C.byref(v)), "mi_stream_read_gbps")
        return float(v.value)

    def light_sample(self, queries):
        """mi_light_sample: Light::Sample_Li / Pdf_Li of the uploaded scene's lights at explicit reference points (LIGHT_QUERY_DTYPE records)"""
        q = np.ascontiguousarray(queries, dtype=LIGHT_QUERY_DTYPE)
        out = np.zeros(len(q), dtype=LIGHT_RESULT_DTYPE)
        L = device_lib()
        L.mi_light_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        if L.mi_light_sample(self._ctx, _ptr(q), len(q), _ptr(out)) != 0:
            raise RuntimeError("mi_light_sample: %s" % L.mi_last_error().decode())
        return out

    def gather_rate(self, nbytes, loads_per_record=4):
        """rate of dependent random 64-byte record fetches (loads_per_record x 16 B per lane) over a buffer of nbytes, 1e9 lane requests / s"""
        v = C.c_double(0)
        L = device_lib()
        L.mi_gather_rate.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
        self._chk(L.mi_gather_rate(self._ctx, C.c_uint64(int(nbytes)), int(loads_per_record), C.byref(v)), "mi_gather_rate")
        return float(v.value)

    # ---- stage-level entry points
    def texture_eval(self, node, queries):
        """Texture::Evaluate of node `node` of the scene's texture table at the recorded interactions -> (n, 3)"""
        queries = np.ascontiguousarray(queries, dtype=TEX_QUERY_DTYPE)
        out = np.zeros((len(queries), 3), dtype=np.float32)
        self._chk(device_lib().mi_texture_eval(self._ctx, int(node), _ptr(queries), len(queries), _ptr(out)), "mi_texture_eval")
        return out

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        self._chk(device_lib().mi_intersect(self._ctx, _ptr(rays), len(rays), _ptr(hits)), "mi_intersect")
        return hits

    def intersect_p(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        occ = np.zeros(len(rays), dtype=np.uint8)
        self._chk(device_lib().mi_intersect_p(self._ctx, _ptr(rays), len(rays), _ptr(occ)), "mi_intersect_p")
        return occ

    def sobol(self, px, py, n_samples, n_dims):
        out = np.zeros((n_samples, n_dims), dtype=np.float32)
        idx = np.zeros(n_samples, dtype=np.uint64)
        self._chk(device_lib().mi_sobol(self._ctx, px, py, n_samples, n_dims, _ptr(out), _ptr(idx)), "mi_sobol")
        return out, idx

    def camera_rays(self, pixels_xy, sample_num):
        pixels_xy = np.ascontiguousarray(pixels_xy, dtype=np.int32)
        sample_num = np.ascontiguousarray(sample_num, dtype=np.int32)
        n = len(sample_num)
        rays = np.zeros(n, dtype=RAY_DTYPE)
        pfilm = np.zeros((n, 2), dtype=np.float32)
        self._chk(device_lib().mi_camera_rays(self._ctx, _ptr(pixels_xy), _ptr(sample_num), n, _ptr(rays), _ptr(pfilm)),
                  "mi_camera_rays")
        return rays, pfilm

    def camera_differentials(self, pixels_xy, sample_num):
        """rx / ry of the camera samples (scaled by 1 / sqrt(spp)): (n, 4, 3) = rxOrigin, rxDirection, ryOrigin, ryDirection"""
        pixels_xy = np.ascontiguousarray(pixels_xy, dtype=np.int32)
        sample_num = np.ascontiguousarray(sample_num, dtype=np.int32)
        n = len(sample_num)
        out = np.zeros((n, 4, 3), dtype=np.float32)
        self._chk(device_lib().mi_camera_differentials(self._ctx, _ptr(pixels_xy), _ptr(sample_num), n, _ptr(out)), "mi_camera_differentials")
        return out

    def li(self, pixels_xy, sample_num):
        pixels_xy = np.ascontiguousarray(pixels_xy, dtype=np.int32)
        sample_num = np.ascontiguousarray(sample_num, dtype=np.int32)
        n = len(sample_num)
        out = np.zeros((n, 3), dtype=np.float32)
        self._chk(device_lib().mi_li(self._ctx, _ptr(pixels_xy), _ptr(sample_num), n, _ptr(out)), "mi_li")
        return out


def triangle_intersect(tri9, rays, device=0):
    """Device Triangle::Intersect for independent (triangle, ray) pairs (no scene)."""
    tri9 = np.ascontiguousarray(tri9, dtype=np.float32).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    hits = np.zeros(len(rays), dtype=HIT_DTYPE)
    L = device_lib()
    if L.mi_triangle_intersect(device, _ptr(tri9), _ptr(rays), len(rays), _ptr(hits)) != 0:
        raise RuntimeError("mi_triangle_intersect: %s" % L.mi_last_error().decode())
    return hits


SPHERE_DTYPE = np.dtype([("o2w", np.float32, 16), ("w2o", np.float32, 16), ("radius", np.float32), ("zmin", np.float32), ("zmax", np.float32),
                         ("theta_min", np.float32), ("theta_max", np.float32), ("phi_max", np.float32), ("flags", np.uint32), ("area", np.float32)])   # mi_sphere
TEX_QUERY_DTYPE = np.dtype([("p", np.float32, 3), ("uv", np.float32, 2), ("dpdx", np.float32, 3), ("dpdy", np.float32, 3), ("dudx", np.float32),
                            ("dvdx", np.float32), ("dudy", np.float32), ("dvdy", np.float32)])   # mi_tex_query
SPHERE_HIT_DTYPE = np.dtype([("hit", np.int32), ("t", np.float32), ("p", np.float32, 3), ("p_error", np.float32, 3), ("n", np.float32, 3)])   # mi_sphere_hit


def bvh4q_validate(scene, rays=None, any_hit=False, want_hits=True):
    """Host-only: build the 64-byte quantised BVH4 of csrc/pt_bvh4q.h (what the default traversal kernels walk), check it, and run
    the kernel's per-ray state machine on the host for `rays` -> (hits or None, dict of statistics).  No GPU needed."""
    st = np.zeros(8, dtype=np.int64)
    L = device_lib()
    n = 0 if rays is None else len(rays)
    r = np.ascontiguousarray(rays, dtype=RAY_DTYPE) if n else None
    hits = np.zeros(n, dtype=HIT_DTYPE) if (n and want_hits) else None
    if L.mi_bvh4q_validate(scene.desc, _ptr(r) if n else None, n, 1 if any_hit else 0, _ptr(hits) if hits is not None else None, _ptr(st)) != 0:
        raise RuntimeError("mi_bvh4q_validate: %s" % L.mi_last_error().decode())
    keys = ["nodes", "leaf_refs", "depth", "max_stack", "prims", "nodes_visited", "prims_tested", "rays_hit"]
    return hits, dict(zip(keys, [int(v) for v in st]))


LIGHT_QUERY_DTYPE = np.dtype([("light", "<i4"), ("p", "<f4", 3), ("n", "<f4", 3), ("u", "<f4", 2), ("wi", "<f4", 3)])
LIGHT_RESULT_DTYPE = np.dtype([("wi", "<f4", 3), ("pdf", "<f4"), ("Li", "<f4", 3), ("ray_o", "<f4", 3), ("ray_d", "<f4", 3), ("ray_tmax", "<f4"), ("pdf_wi", "<f4"), ("delta", "<i4"), ("le_wi", "<f4", 3)])


BSSRDF_QUERY_DTYPE = np.dtype([("sigma_a", "<f4", 3), ("sigma_s", "<f4", 3), ("ch", "<i4"), ("r", "<f4"), ("u", "<f4"), ("kd", "<f4", 3), ("mfp", "<f4", 3)])
BSSRDF_RESULT_DTYPE = np.dtype([("sr", "<f4", 3), ("sample_sr", "<f4"), ("pdf_sr", "<f4"), ("sigma_a", "<f4", 3), ("sigma_s", "<f4", 3)])
HG_QUERY_DTYPE = np.dtype([("g", "<f4"), ("wo", "<f4", 3), ("wi", "<f4", 3), ("u", "<f4", 2)])
HG_RESULT_DTYPE = np.dtype([("p", "<f4"), ("wi_s", "<f4", 3), ("p_s", "<f4")])


class _BssrdfTable(C.Structure):
    _fields_ = [("n_rho", C.c_int32), ("n_radius", C.c_int32), ("rho_samples", C.c_void_p), ("radius_samples", C.c_void_p), ("profile", C.c_void_p),
                ("rho_eff", C.c_void_p), ("profile_cdf", C.c_void_p)]


def bssrdf_eval(table_rec, eta, queries, device=0):
    """mi_bssrdf_eval: TabulatedBSSRDF::Sr / Sample_Sr / Pdf_Sr + SubsurfaceFromDiffuse on the device for a table given as a record with the arrays
    rho_samples [100], radius_samples [64], profile [6400], rho_eff [100], profile_cdf [6400]"""
    keep = [np.ascontiguousarray(table_rec[k], dtype=np.float32) for k in ("rho_samples", "radius_samples", "profile", "rho_eff", "profile_cdf")]
    t = _BssrdfTable(len(keep[0]), len(keep[1]), *[a.ctypes.data for a in keep])
    q = np.ascontiguousarray(queries, dtype=BSSRDF_QUERY_DTYPE)
    out = np.zeros(len(q), dtype=BSSRDF_RESULT_DTYPE)
    L = device_lib()
    L.mi_bssrdf_eval.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    if L.mi_bssrdf_eval(device, C.byref(t), eta, _ptr(q), len(q), _ptr(out)) != 0:
        raise RuntimeError("mi_bssrdf_eval: %s" % L.mi_last_error().decode())
    return out


def phase_hg(queries, device=0):
    """mi_phase_hg: HenyeyGreenstein::p / Sample_p on the device"""
    q = np.ascontiguousarray(queries, dtype=HG_QUERY_DTYPE)
    out = np.zeros(len(q), dtype=HG_RESULT_DTYPE)
    L = device_lib()
    L.mi_phase_hg.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    if L.mi_phase_hg(device, _ptr(q), len(q), _ptr(out)) != 0:
        raise RuntimeError("mi_phase_hg: %s" % L.mi_last_error().decode())
    return out


LIBM_FUNCS = {"sinf": 0, "cosf": 1, "sincosf": 2, "expf": 3, "logf": 4, "acosf": 5, "atanf": 6, "atan2f": 7}


def libm_eval(fn, a, b=None, device=0):
    """mi_libm_eval: routine `fn` of csrc/pt_libm.h on the device over float32 arrays -> out (sincosf: (sin, cos))"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros(len(a), dtype=np.float32)
    out2 = np.zeros(len(a), dtype=np.float32) if fn == "sincosf" else None
    L = device_lib()
    L.mi_libm_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    if L.mi_libm_eval(device, LIBM_FUNCS[fn], _ptr(a), None if b is None else _ptr(b), len(a), _ptr(out), None if out2 is None else _ptr(out2)) != 0:
        raise RuntimeError("mi_libm_eval: %s" % L.mi_last_error().decode())
    return (out, out2) if out2 is not None else out


def bxdf_eval(rows, device=0):
    """mi_bxdf_eval on records shaped like tests/golden/ref_vectors.npz 'bxdfs' -> dict of the device's f, pdf, Sample_f results"""
    n = len(rows)
    b = np.ascontiguousarray(rows["bxdf"]); wo = np.ascontiguousarray(rows["wo"]); wi = np.ascontiguousarray(rows["wi"]); u = np.ascontiguousarray(rows["u"])
    out = {"f": np.zeros((n, 3), np.float32), "pdf": np.zeros(n, np.float32), "wi_s": np.zeros((n, 3), np.float32), "pdf_s": np.zeros(n, np.float32),
           "f_s": np.zeros((n, 3), np.float32), "type_s": np.zeros(n, np.int32)}
    L = device_lib()
    L.mi_bxdf_eval.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 6
    if L.mi_bxdf_eval(device, _ptr(b), _ptr(wo), _ptr(wi), _ptr(u), n, _ptr(out["f"]), _ptr(out["pdf"]), _ptr(out["wi_s"]), _ptr(out["pdf_s"]), _ptr(out["f_s"]),
                      _ptr(out["type_s"])) != 0:
        raise RuntimeError("mi_bxdf_eval: %s" % L.mi_last_error().decode())
    return out


def film_gather(ctxs, root=0):
    """mi_film_gather: sum the films of `ctxs` (one context per GPU, rendered with rank = i, world = len(ctxs)) into ctxs[root]'s
    film -- one grouped ncclReduce over RCCL when the contexts sit on distinct GPUs, a device-side sum when they share one."""
    arr = (C.c_void_p * len(ctxs))(*[c._ctx for c in ctxs])
    if device_lib().mi_film_gather(arr, len(ctxs), root) != 0:
        raise RuntimeError("mi_film_gather: %s" % device_lib().mi_last_error().decode())


def bvh4_validate(scene):
    """Host-only self check of the BVH2 -> BVH4 collapse (no GPU needed): dict of tree statistics, raises on a broken invariant."""
    st = np.zeros(8, dtype=np.int64)
    L = device_lib()
    if L.mi_bvh4_validate(scene.desc, _ptr(st)) != 0:
        raise RuntimeError("mi_bvh4_validate: %s" % L.mi_last_error().decode())
    return {"nodes": int(st[0]), "leaf_refs": int(st[1]), "depth": int(st[2]), "stack_need": int(st[3]), "prims": int(st[4]), "objects": int(st[5]),
            "own_topology": bool(st[6])}   # the library's own topology over the reference's leaves (the default for single-level scenes; PBRT_AMD_TREE=reference: the tree as handed over)


def sphere_intersect(spheres, rays, device=0):
    """Device Sphere::Intersect for independent (sphere, ray) pairs (no scene)."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    hits = np.zeros(len(rays), dtype=SPHERE_HIT_DTYPE)
    L = device_lib()
    if L.mi_sphere_intersect(device, _ptr(spheres), _ptr(rays), len(rays), _ptr(hits)) != 0:
        raise RuntimeError("mi_sphere_intersect: %s" % L.mi_last_error().decode())
    return hits


def read_image(path, max_pixels=1 << 24):
    """the host's ReadImage (.pfm / .png / .tga) -> (H, W, 3) float32, row 0 = top"""
    L = host_lib()
    L.pbrt_amd_read_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    buf = np.zeros(3 * max_pixels, dtype=np.float32)
    w, h = C.c_int(0), C.c_int(0)
    if L.pbrt_amd_read_image(path.encode(), _ptr(buf), len(buf), C.byref(w), C.byref(h)) != 0:
        raise RuntimeError("ReadImage failed: %s" % path)
    return buf[:3 * w.value * h.value].reshape(h.value, w.value, 3).copy()


def read_pfm(path):
    with open(path, "rb") as f:
        tag = f.readline().strip()
        if tag != b"PF":
            raise ValueError("not a 3-channel PFM: %s" % path)
        w, h = [int(v) for v in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)
