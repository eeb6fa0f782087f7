// libpbrt_amd.so -- wavefront path tracer for MI355X (gfx950), hand-written HIP behind the C ABI of
// include/pbrt_amd.h.  One context per GPU.  Reference hot path: SamplerIntegrator::Render
// (core/integrator.cpp:228-339) -> PathIntegrator::Li (integrators/path.cpp:64-188).
//
// Wavefront organisation (per pass over a chunk of (pixel, sample) pairs; all queues live in HBM):
//   k_raygen     Sobol' index + camera sample + PerspectiveCamera ray           -> extension queue
//   per bounce:
//   k_closest    BVH4 closest hit, LDS-resident per-lane stack                  -> hit records + material key counts
//   k_scan/k_scatter   counting sort of the hit paths by material (wave ballots) -> material-sorted queue
//   k_shade      emission, BSDF build, UniformSampleOneLight/EstimateDirect, BSDF::Sample_f, RR
//                                                                               -> shadow / MIS / next extension queues
//   k_anyhit     BVH4 any-hit for shadow rays, adds the light-sampled term
//   k_closest<MIS>  closest hit for the BSDF-sampled MIS ray, adds its term
//   k_film       FilmTile::AddSample into the device film
// All kernels are persistent grid-stride loops whose trip counts come from device-side queue
// counters (no host round trips inside a pass) with an XCD-contiguous chunk mapping.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "pt_shade.h"
#include "pt_material.h"
#include "pt_volume.h"
#include "pt_bvh4q.h"
#include "sobol_tables.inc"

__constant__ DevTex c_tex;
__constant__ const DevInstance *c_instances;   // instance table of the scene being rendered (two-level scenes only)   // texture tables of the scene being rendered (set by mi_render for textured scenes only)

// ============================================================================ device side
// Per-path state is array-of-structures, one 128-byte cache line per path (+ one for the two NEE rays): after the
// material sort a wave's paths are scattered over the slot range, and with a structure-of-arrays layout every 16-byte
// field access dragged in a full line of which 7/8 went unused (measured: k_shade missed L2 on 68 % of its requests
// and moved ~4 TB/s for ~0.6 TB/s of payload).  With AoS a lane's loads all land in the one or two lines it owns.
struct __attribute__((aligned(128))) PathRec {
    float4 ray_o, ray_d;       // o.xyz,tMax | d.xyz,-
    float4 beta;               // rgb, etaScale
    float4 L;                  // rgb, -
    uint4 smp;                 // sobol index lo, hi, dimension, bounces | specularBounce << 16 | (textured scenes) noDifferentials << 17
    uint2 hit;                 // prim (0xffffffff = miss), t bits
    float2 pfilm;
    uint32_t pixel;            // sample-space pixel x | y << 16, 0xffffffff = inactive
    uint32_t pad0;
    float2 lens;               // textured scenes: the camera sample's pLens (k_shade rebuilds the ray differentials from pfilm + lens)
    float4 pad2;
};
struct __attribute__((aligned(128))) NeeRec {
    float4 sh_o, sh_d, sh_c;   // shadow ray: o.xyz,tMax | d | contribution rgb
    float4 mi_o, mi_d, mi_c;   // MIS ray: o | d.xyz,lightNum | contribution rgb
    float4 pad[2];
};
// state of a shadow ([0]) / MIS ([1]) ray that is walked through BSDF-less medium interfaces segment by segment (DevVol::tr_queues)
struct __attribute__((aligned(64))) TrState {
    float4 acc[2];   // transmittance so far (rgb)
    uint4 hit[2];    // the segment's closest hit: prim, t bits, instance, -
};
struct SssRec;   // pt_volpath.h
struct PathState {
    PathRec *rec;
    NeeRec *nee;
    TrState *trs;              // DevVol::tr_queues / sss_wave only, else null
    SssRec *sss;               // DevVol::sss_wave only: parked subsurface paths (BSSRDF at po, probe chain)
    uint32_t *q_sss;           // ... and the queue of the paths whose chain arrived at its chosen hit (row QC_SSS)
    uint32_t *q_cont;          // DevVol::tr_dims only: the vertices that wait for their transmittances (k_vol_continue's queue, row QC_CONT)
    uint32_t *q_probe[2];      // ... and the two queues the probe walk ping-pongs between (rows QC_PROBE0 / QC_PROBE1; the direct-lighting walk keeps q_tr)
    uint32_t *q_tr[2];         // second shadow / MIS queues (the walk ping-pongs between q_shadow / q_mis and these)
    uint32_t qrow_shadow, qrow_mis;   // counter rows of the queues k_trace<2> / <1> read (QC_SHADOW / QC_MIS unless a walk swapped them)
    uint2 *keyrank;            // per QUEUE POSITION of the extension queue: .x = the path's material key, written by k_trace<0> when the ray at that position finishes (round 6:
                               // it used to go to a per-path array that k_keycount then gathered through the queue, one 128-byte line per key: 1.2 of its 1.25 ms per launch);
                               // k_keycount reads it in queue order, adds the rank within the block (.y), k_scatter walks the same positions
    uint32_t *q_ext[2], *q_shadow, *q_mis, *q_sorted;
    // Queues are cut into QSEG segments (one per XCD-aligned block class, blockIdx & 7), each with its own fill counter in its own
    // 128-byte line: qcount[QCI(queue, seg)].  A single counter per queue made every wave of the chip hit ONE word -- the L2 serialises
    // same-address atomics at ~88 per microsecond, which is exactly what k_raygen cost (2.07 M wave appends = 24 ms per frame) and most
    // of k_shade's launch time.  Segment `seg` of a queue occupies [seg * seg_cap, seg * seg_cap + count): producers append to the segment
    // of their own block class, consumers walk (k_keycount / k_scatter) or pull from (k_trace) the segments.
    uint32_t *qcount;          // rows: [0],[1] extension queues, [2] shadow, [3] mis; [4] the material-sorted total
    uint32_t seg_cap;          // entries per queue segment
    uint32_t *keycount, *keyoffset;
    uint32_t *blockhist;       // [gridBlocks][nkeys]
    uint32_t *cursor;          // [QSEG * QC_STRIDE] per-segment fetch cursors of k_trace, one 128-byte line each
    unsigned long long *counters;
    uint32_t *spill;
    int spill_per_thread;
    uint32_t cap;
    uint32_t vol_tr;           // k_trace<1> only: "volpath" scenes in wavefront form -- the MIS term is attenuated by the homogeneous medium's transmittance over the hit distance (NeeRec::pad[0] = sigma_t)
    // The material-sorted queue is shaded in PARTS, one launch each (mi_ctx::shadeParts): in scenes under Integrator "path" whose only reason for k_shade_vol are BSSRDF
    // materials (mi_ctx::sssRoute) the sort puts those materials' keys LAST (key_remap, applied by k_keycount), k_shade takes the first part, k_shade_vol the second.  A
    // launch walks the keys [shade_key_lo, shade_key_hi) of the sorted queue (ShadeRange).  (Round 5 also built parts per material CLASS with in-line instances -- matte,
    // diffuse + glossy, specular-only -- and measured them against one launch over the interleaved DynIter partition: +0.9 % / -3.4 % / 0 on C3 / C2 / C4,
    // profiles/r05_cde_*, r05_fg_*: removed.)
    float4 *sss_log_o, *sss_log_d;   // k_sss_probe_tail's lists of counted hits (SssLog, pt_volpath.h): sss_log_cap entries per thread for the first sss_log_threads threads of the launch; null: off
    uint32_t *sss_log_inst;
    uint32_t sss_log_threads, sss_log_cap;
    const uint32_t *key_remap;   // [nkeys] or null
    uint32_t shade_key_lo, shade_key_hi;   // the launch's part of the sorted queue in (remapped) keys; hi = 0xffffffff: to the end
};
enum { QC_EXT0 = 0, QC_EXT1 = 1, QC_SHADOW = 2, QC_MIS = 3, QC_SORTED = 4, QC_BINNED = 5, QC_SHADOW2 = 6, QC_MIS2 = 7, QC_SSS = 8, QC_PROBE0 = 9, QC_PROBE1 = 10, QC_CONT = 11, QC_ROWS = 12 };
#define QSEG 8u
#define QC_STRIDE 32u   /* words between counters: one 128-byte line each */
#define QCI(q, seg) (((uint32_t)(q) * QSEG + (uint32_t)(seg)) * QC_STRIDE)
#define QC_WORDS (QC_ROWS * QSEG * QC_STRIDE)
// the items of one segmented queue, walked by the blocks of the segment's class (XCD x = blockIdx & 7 takes segment x, 256 items per block step)
struct SegIter {
    uint32_t n, c, bpx, base;
    PT_DEV SegIter(const uint32_t *qcount, uint32_t q, uint32_t seg_cap) {
        uint32_t x = blockIdx.x & 7;
        n = qcount[QCI(q, x)];
        base = x * seg_cap;
        c = blockIdx.x >> 3;
        bpx = gridDim.x >> 3;
    }
    PT_DEV bool more() const { return c * PT_BLOCK < n; }
    PT_DEV bool valid() const { return c * PT_BLOCK + threadIdx.x < n; }
    PT_DEV uint32_t item() const { return base + c * PT_BLOCK + threadIdx.x; }   // position in the queue array (also indexes keyrank)
    PT_DEV void next() { c += bpx; }
};

struct PassInfo {
    const uint32_t *tiles;     // owned tile ids (tile = ty * nTilesX + tx)
    uint32_t n_tiles_x;
    uint32_t pix0, npix;       // owned-pixel range of this pass (16x16 tile-major pixel numbering)
    uint32_t s0, ns;           // sample numbers [s0, s0+ns)
    const int32_t *list_xy;    // explicit (pixel, sample) list mode (mi_li / mi_camera_rays), else null
    const int32_t *list_s;
    uint32_t serial = 0, w = 0;   // tile-serial round (MI_SAMPLER_RANDOM / STRATIFIED / ZEROTWO): path i = sample s0 of pixel w (0..255, row major) of owned tile i; npix = owned tiles, ns = 1
};

#define MISS_PRIM 0xffffffffu
#define INACTIVE_PIXEL 0xffffffffu

// ---- XCD-aware persistent loop: the queue is cut into 256-item chunks; XCD x (= blockIdx.x % 8, the
// observed dispatch placement) owns one contiguous eighth of the chunks so that its private 4 MiB L2
// sees a spatially coherent band of rays; correctness never depends on the placement.
struct ChunkIter {
    uint32_t nch, cpx, c, xcd, bpx;
    PT_DEV ChunkIter(uint32_t n) {
        nch = (n + PT_BLOCK - 1) / PT_BLOCK;
        cpx = (nch + 7) / 8;
        xcd = blockIdx.x & 7;
        bpx = gridDim.x >> 3;
        c = blockIdx.x >> 3;
    }
    PT_DEV bool more() const { return c < cpx; }
    PT_DEV uint32_t item() const { return (xcd * cpx + c) * PT_BLOCK + threadIdx.x; }
    PT_DEV void next() { c += bpx; }
};

// The same item range handed out DYNAMICALLY, per wave: block class x (= XCD x) owns one contiguous eighth (as ChunkIter), and a wave takes
// PT_DYN_GRAIN items of its class's eighth at a time with one atomic on that eighth's cursor (its own 128-byte line).  No stealing across
// classes: a class appends to ITS segment of the output queues, whose capacity is one eighth of the items (ensure_state).
// Static partitions leave the machine part-empty while the slowest blocks finish (k_shade: 2.2 of 3 resident waves per SIMD on average, SQ counters).
#ifndef PT_DYN_GRAIN
#define PT_DYN_GRAIN 64u   /* one wave's worth per grab.  Re-measured in round 5 over the interleaved partition (profiles/r05_h_*): 64 / 128 / 256 / 512 -> shade 31.2 / 31.1 / 31.1 / 31.4 ms on C3 (16 spp), 28.6 / 28.4 / 29.4 / 28.7 on C2 (32 spp), 79.6 / 82.2 / 89.2 / 103.4 on C4 (32 spp, maxdepth 30: the short queues of deep bounces end in a tail of one grab); round 2 over contiguous eighths: flat between 64 and 256 */
#endif
#ifndef PT_WAVE_SIZE
#define PT_WAVE_SIZE 64u
#endif
// the part of the material-sorted queue a shading launch walks (PathState::shade_key_lo / _hi): [base, base + n).  Every launch deals ITS part out to the eight block
// classes grain by grain (DynIter), so a class appends up to n_part / 8 + one grain of entries per part to its queue segment: PathState::seg_cap carries that headroom
// for PT_SHADE_PARTS_MAX parts (ensure_state).
#define PT_SHADE_PARTS_MAX 2
PT_DEV uint32_t ShadeKeyPos(const PathState &ps, uint32_t key, uint32_t total) { return key == 0xffffffffu ? total : (key ? ps.keyoffset[key] : 0u); }   // first position of `key` in the sorted queue
PT_DEV void ShadeRange(const PathState &ps, uint32_t *base, uint32_t *n) {
    const uint32_t total = ps.qcount[QCI(QC_SORTED, 0)];
    uint32_t lo = ShadeKeyPos(ps, ps.shade_key_lo, total), hi = ShadeKeyPos(ps, ps.shade_key_hi, total);
    *base = lo;
    *n = hi - lo;
}
struct DynIter {
    // Round 5: the eighths are INTERLEAVED, grain by grain.  The range is the material-sorted queue, and a contiguous eighth handed XCD x the vertices of three or four
    // materials -- whose cost differs by 2x between matte and the microfacet materials: the XCDs with the expensive eighths finished last (no stealing across classes) and
    // identical runs differed by 17 % (profiles/r05_e_*, r05_f_*: equal wave cycles, equal instruction-cache hit rates, 38.2-44.7 ms in one launch against 33.6 when the queue
    // was shaded class by class).  Grain g goes to class g mod 8, so every XCD shades the same mix of materials.  Capacity: a class takes ceil(grains / 8) grains <= n / 8 + one
    // grain of items, as before (PathState::seg_cap).
    uint32_t n, seg, cur, end;
    uint32_t *cursor;
    PT_DEV DynIter(uint32_t n_, uint32_t *cursor_) {
        seg = blockIdx.x & 7;
        n = n_;
        cursor = cursor_ + seg * QC_STRIDE;
        cur = end = 0;
    }
    PT_DEV bool more() {
        if (cur < end) return true;
        uint32_t base = 0;
        if (__lane_id() == 0) base = atomicAdd(cursor, PT_DYN_GRAIN);
        base = __shfl(base, 0);
        const uint32_t first = (base * 8u) + seg * PT_DYN_GRAIN;   // grain (base / GRAIN) * 8 + seg
        if (base >= 0x1fffffffu - PT_DYN_GRAIN || first >= n) return false;
        cur = first;
        end = cur + PT_DYN_GRAIN < n ? cur + PT_DYN_GRAIN : n;
        return true;
    }
    PT_DEV uint32_t item() const { return cur + __lane_id(); }
    PT_DEV bool valid() const { return cur + __lane_id() < end; }
    PT_DEV void next() { cur += PT_WAVE_SIZE; }
};
PT_DEV uint32_t lane_id() { return __lane_id(); }
// wave-aggregated queue append: one atomic per wave (ballot + popcount), order-preserving inside the wave
PT_DEV uint32_t wave_append(uint32_t *counter, bool active) {
    unsigned long long mask = __ballot(active);
    if (mask == 0) return 0;
    uint32_t lane = lane_id();
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// three appends at once (k_shade's extension / shadow / MIS queues): lanes 0..2 issue the three atomics in ONE
// instruction, so the wave pays one atomic round trip instead of three dependent ones
PT_DEV void wave_append3(uint32_t *c0, uint32_t *c1, uint32_t *c2, bool a0, bool a1, bool a2, uint32_t *p0, uint32_t *p1, uint32_t *p2) {
    unsigned long long m0 = __ballot(a0), m1 = __ballot(a1), m2 = __ballot(a2);
    uint32_t lane = lane_id();
    uint32_t n = lane == 0 ? (uint32_t)__popcll(m0) : (lane == 1 ? (uint32_t)__popcll(m1) : (uint32_t)__popcll(m2));
    uint32_t *ctr = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
    uint32_t base = 0;
    if (lane < 3 && n) base = atomicAdd(ctr, n);
    uint32_t b0 = __shfl(base, 0), b1 = __shfl(base, 1), b2 = __shfl(base, 2);
    unsigned long long lt = (1ull << lane) - 1ull;
    *p0 = b0 + (uint32_t)__popcll(m0 & lt);
    *p1 = b1 + (uint32_t)__popcll(m1 & lt);
    *p2 = b2 + (uint32_t)__popcll(m2 & lt);
}
#ifndef PT_WAVE_APPEND3
#define PT_WAVE_APPEND3 wave_append3
#endif
// wave-aggregated histogram slot: lanes with equal key share one atomic (match-any built from ballots)
PT_DEV uint32_t wave_key_rank(uint32_t *keycount, uint32_t key, bool active) {
    uint32_t lane = lane_id();
    uint32_t rank = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        int first = __ffsll((long long)todo) - 1;
        uint32_t k0 = __shfl(key, first);
        unsigned long long same = __ballot(active && key == k0) & todo;
        if (active && key == k0) {
            uint32_t base = 0;
            if ((int)lane == first) base = atomicAdd(&keycount[k0], (uint32_t)__popcll(same));
            base = __shfl(base, first);
            rank = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        }
        todo &= ~same;
    }
    return rank;
}
// Wave-wide vote on a PREDICATE: the lane mask straight from the compare.  HIP's __ballot(int) / __any(int) first materialise the predicate as a 0 / 1
// integer and compare it again (v_cndmask + v_cmp_ne: two more VALU instructions per call, in a kernel bound by VALU issue).
PT_DEV unsigned long long PtBallot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
PT_DEV bool PtAny(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// slots 64..67 (counting passes only, mi_trace_clock): per-wave s_memtime / s_memrealtime ticks spent inside the closest-hit and any-hit kernels
#define PT_CNT_CLK 64
PT_DEV void wave_count(unsigned long long *counter, uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane_id() == 0 && v) atomicAdd(counter, (unsigned long long)v);
}

// ---- camera: Sampler::GetCameraSample (core/sampler.cpp:46-52) + PerspectiveCamera::GenerateRayDifferential
// main ray (cameras/perspective.cpp:95-144) + Transform::operator()(Ray) (core/transform.h:252-264)
PT_DEV V3 XfPoint(const float *m, const V3 &p) {   // core/transform.h:223-234
    Float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    Float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    Float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
// the camera half of GenerateCameraRay: film position + lens sample -> world-space ray (cameras/perspective.cpp:95-144, transform.h:252-264)
PT_DEV void CameraRayFromFilm(const DevScene &sc, Float pFilmX, Float pFilmY, Float l0, Float l1, V3 *o, V3 *d, Float *tMax) {
    const mi_camera &cam = sc.camera;
    V3 pCamera = XfPoint(cam.raster_to_camera, V3(pFilmX, pFilmY, 0));
    V3 ro(0, 0, 0), rd = Normalize(V3(pCamera.x, pCamera.y, pCamera.z));
    if (cam.lens_radius > 0) {
        Float dx, dy;
        ConcentricSampleDisk(l0, l1, &dx, &dy);
        Float lx = cam.lens_radius * dx, ly = cam.lens_radius * dy;
        Float ft = cam.focal_distance / rd.z;
        V3 pFocus = ro + rd * ft;
        ro = V3(lx, ly, 0);
        rd = Normalize(pFocus - ro);
    }
    const float *m = cam.camera_to_world;
    V3 wo_ = XfPoint(m, ro);   // w == 1 for the affine camera-to-world matrix; XfPoint handles the general case
    Float xAbs = (absf(m[0] * ro.x) + absf(m[1] * ro.y) + absf(m[2] * ro.z) + absf(m[3]));
    Float yAbs = (absf(m[4] * ro.x) + absf(m[5] * ro.y) + absf(m[6] * ro.z) + absf(m[7]));
    Float zAbs = (absf(m[8] * ro.x) + absf(m[9] * ro.y) + absf(m[10] * ro.z) + absf(m[11]));
    V3 oError = gamma_n(3) * V3(xAbs, yAbs, zAbs);
    V3 wd(m[0] * rd.x + m[1] * rd.y + m[2] * rd.z, m[4] * rd.x + m[5] * rd.y + m[6] * rd.z, m[8] * rd.x + m[9] * rd.y + m[10] * rd.z);
    Float lengthSquared = wd.LengthSquared();
    Float tm = PT_INFINITY;
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(wd), oError) / lengthSquared;
        wo_ = wo_ + wd * dt;
        tm -= dt;
    }
    *o = wo_; *d = wd; *tMax = tm;
}
PT_DEV void GenerateCameraRay(const DevScene &sc, Sampler &smp, V3 *o, V3 *d, Float *tMax, Float *pfx, Float *pfy, Float *lens0, Float *lens1) {
    Float u[5];
    if (Sampler::Pix(sc)) {   // Sampler::GetCameraSample core/sampler.cpp:44-50: pFilm = Get2D(), time = Get1D(), pLens = Get2D()
        smp.PixGet2D(sc, &u[0], &u[1]);
        u[2] = smp.PixGet1D(sc);
        smp.PixGet2D(sc, &u[3], &u[4]);
    } else
        SamplerBatch<5>(sc, smp.index, 0, u);   // dims 0,1 film offset, 2 time, 3,4 lens
    Float u0 = u[0], u1 = u[1];
    if (sc.sampler_type == MI_SAMPLER_SOBOL) {
        // SobolSampler::SampleDimension remaps the two pixel dimensions (samplers/sobol.cpp:54-57); Halton's are in-pixel already
        u0 = clampf((u[0] * sc.sobol_resolution + sc.sample_min[0]) - smp.px, (Float)0, PT_ONE_MINUS_EPS);
        u1 = clampf((u[1] * sc.sobol_resolution + sc.sample_min[1]) - smp.py, (Float)0, PT_ONE_MINUS_EPS);
    }
    Float l0 = u[3], l1 = u[4];
    if (!Sampler::Pix(sc)) smp.dimension = 5;
    Float pFilmX = (Float)smp.px + u0, pFilmY = (Float)smp.py + u1;   // static scene: the time sample (dim 2) is never read
    CameraRayFromFilm(sc, pFilmX, pFilmY, l0, l1, o, d, tMax);
    *pfx = pFilmX; *pfy = pFilmY; *lens0 = l0; *lens1 = l1;
}

// ---- the tile-serial samplers (ABI v11).  `sampler->Clone(seed = tile.y * nTiles.x + tile.x)` gives every 16x16 tile ONE PCG32 stream
// (integrator.cpp:246-248, random.cpp:54-58, stratified.cpp:72-76, zerotwosequence.cpp:70-74); k_pix_seed starts the streams of a frame,
// k_pix_start_pixel is <Sampler>::StartPixel for pixel w of every owned tile -- one lane per tile, the tile's numbers drawn one after the other.
__global__ void __launch_bounds__(PT_BLOCK) k_pix_seed(DevScene sc, const uint32_t *tiles, uint32_t nTiles) {
    for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < nTiles; i += gridDim.x * PT_BLOCK) {
        const uint32_t tile = tiles[i];
        unsigned long long state = 0u, inc = ((unsigned long long)(long long)(int)tile << 1u) | 1u;   // RNG::SetSequence rng.h:128-134
        Pcg32Next(state, inc);
        state += 0x853c49e6748fea9bULL;
        Pcg32Next(state, inc);
        sc.pix_rng[2 * (size_t)tile] = state; sc.pix_rng[2 * (size_t)tile + 1] = inc;
    }
}
template <int DIM> PT_DEV void PixShuffle(float *samp, int count, unsigned long long &state, unsigned long long inc) {   // Shuffle core/sampling.h:150-157
    for (int i = 0; i < count; ++i) {
        int other = i + (int)Pcg32Bounded(state, inc, (uint32_t)(count - i));
        for (int j = 0; j < DIM; ++j) { float t = samp[DIM * i + j]; samp[DIM * i + j] = samp[DIM * other + j]; samp[DIM * other + j] = t; }
    }
}
__global__ void __launch_bounds__(PT_BLOCK) k_pix_start_pixel(DevScene sc, const uint32_t *tiles, uint32_t nTiles, uint32_t nTilesX, uint32_t w) {
    for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < nTiles; i += gridDim.x * PT_BLOCK) {
        const uint32_t tile = tiles[i];
        const int x = sc.sample_min[0] + (int)(tile % nTilesX) * 16 + (int)(w & 15), y = sc.sample_min[1] + (int)(tile / nTilesX) * 16 + (int)(w >> 4);
        if (x >= sc.sample_max[0] || y >= sc.sample_max[1]) continue;   // not a pixel of the (clipped) tile; pixels outside pixelBounds DO start (integrator.cpp:262-273)
        unsigned long long state = sc.pix_rng[2 * (size_t)tile];
        const unsigned long long inc = sc.pix_rng[2 * (size_t)tile + 1];
        const int nd = sc.pix_nd, spp = sc.spp;
        float *s1 = sc.pix_s1 + (size_t)tile * nd * spp, *s2 = sc.pix_s2 + 2 * (size_t)tile * nd * spp;
        if (sc.sampler_type == MI_SAMPLER_STRATIFIED) {   // StratifiedSampler::StartPixel stratified.cpp:43-56
            const int nx = sc.strat_nx, ny = sc.strat_ny, n = nx * ny;
            const bool jitter = sc.strat_jitter != 0;
            for (int d = 0; d < nd; ++d) {
                float *p = s1 + (size_t)d * spp;
                const Float invN = (Float)1 / n;   // StratifiedSample1D core/sampling.cpp:42-48
                for (int k = 0; k < n; ++k) {
                    Float delta = jitter ? Pcg32Float(state, inc) : 0.5f;
                    Float v = (k + delta) * invN;
                    p[k] = v < PT_ONE_MINUS_EPS ? v : PT_ONE_MINUS_EPS;
                }
                PixShuffle<1>(p, n, state, inc);
            }
            for (int d = 0; d < nd; ++d) {
                float *p = s2 + 2 * (size_t)d * spp, *q = p;
                const Float dx = (Float)1 / nx, dy = (Float)1 / ny;   // StratifiedSample2D core/sampling.cpp:50-60
                for (int yy = 0; yy < ny; ++yy)
                    for (int xx = 0; xx < nx; ++xx) {
                        Float jx = jitter ? Pcg32Float(state, inc) : 0.5f;
                        Float jy = jitter ? Pcg32Float(state, inc) : 0.5f;
                        Float vx = (xx + jx) * dx, vy = (yy + jy) * dy;
                        q[0] = vx < PT_ONE_MINUS_EPS ? vx : PT_ONE_MINUS_EPS;
                        q[1] = vy < PT_ONE_MINUS_EPS ? vy : PT_ONE_MINUS_EPS;
                        q += 2;
                    }
                PixShuffle<2>(p, n, state, inc);
            }
        } else if (sc.sampler_type == MI_SAMPLER_ZEROTWO || sc.sampler_type == MI_SAMPLER_MAXMIN) {   // ZeroTwoSequenceSampler::StartPixel zerotwosequence.cpp:53-60: VanDerCorput / Sobol2D (lowdiscrepancy.h:144-226), one value per pixel sample
            const bool maxmin = sc.sampler_type == MI_SAMPLER_MAXMIN;
            if (maxmin) {   // MaxMinDistSampler::StartPixel maxmin.cpp:41-47: samples2D[0][i] = (i / spp, SampleGeneratorMatrix(CPixel, i)), shuffled FIRST; then as above from 2D dimension 1
                const Float invSPP = (Float)1 / spp;
                for (int k = 0; k < spp; ++k) {
                    uint32_t v = 0;   // MultiplyGenerator lowdiscrepancy.h:128-133
                    for (uint32_t a = (uint32_t)k, c = 0; a != 0; ++c, a >>= 1) if (a & 1u) v ^= sc.pix_maxmin[c];
                    const Float f = v * (Float)0x1p-32;
                    s2[2 * k] = k * invSPP;
                    s2[2 * k + 1] = f < PT_ONE_MINUS_EPS ? f : PT_ONE_MINUS_EPS;
                }
                PixShuffle<2>(s2, spp, state, inc);
            }
            for (int d = 0; d < nd; ++d) {
                float *p = s1 + (size_t)d * spp;
                uint32_t v = Pcg32Next(state, inc);   // scramble; GrayCodeSample (lowdiscrepancy.h:113-126) over CVanDerCorput[k] = 1 << (31 - k)
                for (uint32_t k = 0; k < (uint32_t)spp; ++k) {
                    Float f = v * (Float)0x1p-32;
                    p[k] = f < PT_ONE_MINUS_EPS ? f : PT_ONE_MINUS_EPS;
                    v ^= 0x80000000u >> __builtin_ctz(k + 1);
                }
                for (int k = 0; k < spp; ++k) PixShuffle<1>(p + k, 1, state, inc);   // Shuffle(samples + i, 1, 1, rng): one number each
                PixShuffle<1>(p, spp, state, inc);
            }
            for (int d = maxmin ? 1 : 0; d < nd; ++d) {
                float *p = s2 + 2 * (size_t)d * spp;
                uint32_t v0 = Pcg32Next(state, inc), v1 = Pcg32Next(state, inc);
                for (uint32_t k = 0; k < (uint32_t)spp; ++k) {
                    Float f0 = v0 * (Float)0x1p-32, f1 = v1 * (Float)0x1p-32;
                    p[2 * k] = f0 < PT_ONE_MINUS_EPS ? f0 : PT_ONE_MINUS_EPS;
                    p[2 * k + 1] = f1 < PT_ONE_MINUS_EPS ? f1 : PT_ONE_MINUS_EPS;
                    const int c = __builtin_ctz(k + 1);
                    v0 ^= 0x80000000u >> c;
                    uint32_t c1 = 0x80000000u;   // CSobol[1][c] (lowdiscrepancy.h:215-220): column k = column k-1 ^ (column k-1 >> 1) from 0x80000000
                    for (int j = 0; j < c; ++j) c1 ^= c1 >> 1;
                    v1 ^= c1;
                }
                for (int k = 0; k < spp; ++k) PixShuffle<2>(p + 2 * k, 1, state, inc);
                PixShuffle<2>(p, spp, state, inc);
            }
        }   // RandomSampler::StartPixel: nothing to draw without requested sample arrays (random.cpp:63-73)
        sc.pix_rng[2 * (size_t)tile] = state;
    }
}

#ifndef PT_RAYGEN_LDS
#if defined(PT_HOST_EMU) && PT_HOST_EMU
#define PT_RAYGEN_LDS 0   /* (one-lane waves of the x86 emulator) */
#else
#define PT_RAYGEN_LDS 1   /* 0: every lane stores its own record's fields (rounds 1-5; the A/B partner) */
#endif
#endif
// TEX: the scene has textured materials -- keep the lens sample with the path
template <bool TEX>
__global__ void __launch_bounds__(PT_BLOCK) k_raygen(DevScene sc, PathState ps, PassInfo pass, uint32_t qout) {
    uint32_t n = pass.list_xy ? pass.npix : pass.npix * pass.ns;
    uint32_t ncam = 0;
#if PT_RAYGEN_LDS
    __shared__ float4 s_rec[PT_BLOCK / 64][7][65];   // per wave: the seven written fields of its 64 records, field-major (+1: the read-out's bank spread)
#endif
    for (ChunkIter it(n); it.more(); it.next()) {
        uint32_t i = it.item();
        bool active = i < n;
        int x = 0, y = 0;
        uint32_t s = 0, tileId = 0;
        if (active) {
            if (pass.list_xy) { x = pass.list_xy[2 * i]; y = pass.list_xy[2 * i + 1]; s = (uint32_t)pass.list_s[i]; }
            else if (pass.serial) {
                tileId = pass.tiles[i];
                s = pass.s0;
                x = sc.sample_min[0] + (int)(tileId % pass.n_tiles_x) * 16 + (int)(pass.w & 15);
                y = sc.sample_min[1] + (int)(tileId / pass.n_tiles_x) * 16 + (int)(pass.w >> 4);
                active = x < sc.sample_max[0] && y < sc.sample_max[1] && x >= sc.pixel_min[0] && x < sc.pixel_max[0] &&
                         y >= sc.pixel_min[1] && y < sc.pixel_max[1];
            } else {
                uint32_t p = i % pass.npix;
                s = pass.s0 + i / pass.npix;
                uint32_t k = pass.pix0 + p, tile = pass.tiles[k >> 8], w = k & 255;
                x = sc.sample_min[0] + (int)(tile % pass.n_tiles_x) * 16 + (int)(w & 15);
                y = sc.sample_min[1] + (int)(tile / pass.n_tiles_x) * 16 + (int)(w >> 4);
                // tile clipped to the sample bounds (integrator.cpp:251-255) and the pixelBounds test (:273)
                active = x < sc.sample_max[0] && y < sc.sample_max[1] && x >= sc.pixel_min[0] && x < sc.pixel_max[0] &&
                         y >= sc.pixel_min[1] && y < sc.pixel_max[1];
            }
        }
        // The wave's 64 path records are 8 KiB of consecutive memory.  Written field by field from the lanes that own them, every store instruction touched 64
        // different 128-byte lines with 16 bytes each (seven partial-line requests per path at the L2); through LDS the wave writes its records as whole lines
        // (PT_RAYGEN_LDS, round 6: eight fully coalesced 1 KiB stores per wave and chunk; the record's unused words are written as zeros).
        float4 f_o = make_float4(0, 0, 0, 0), f_d = f_o, f_beta = f_o, f_smp = f_o, f_5 = f_o, f_6 = f_o;
        f_6.x = __uint_as_float(active ? ((uint32_t)(x - sc.sample_min[0]) | ((uint32_t)(y - sc.sample_min[1]) << 16)) : INACTIVE_PIXEL);
        if (active) {
            Sampler smp;
            if (pass.serial) smp.PixStart(tileId, s, x, y);
            else smp.Start(sc, x, y, s);
            V3 o, d;
            Float tMax, pfx, pfy, lens0, lens1;
            GenerateCameraRay(sc, smp, &o, &d, &tMax, &pfx, &pfy, &lens0, &lens1);
            if (TEX) { f_6.z = lens0; f_6.w = lens1; }
            f_o = make_float4(o.x, o.y, o.z, tMax);
            f_d = make_float4(d.x, d.y, d.z, 0);
            f_beta = make_float4(1, 1, 1, 1);
            f_smp = make_float4(__uint_as_float((uint32_t)smp.index), __uint_as_float((uint32_t)(smp.index >> 32)), __uint_as_float((uint32_t)smp.dimension), 0);
            f_5.z = pfx; f_5.w = pfy;
            ++ncam;
        }
#if PT_RAYGEN_LDS
        {
            static_assert(sizeof(PathRec) == 128, "eight 16-byte fields per record");
            float4 (*S)[65] = s_rec[threadIdx.x >> 6];
            const uint32_t ln = threadIdx.x & 63;
            S[0][ln] = f_o; S[1][ln] = f_d; S[2][ln] = f_beta; S[3][ln] = make_float4(0, 0, 0, 0) /* L */; S[4][ln] = f_smp; S[5][ln] = f_5 /* hit, pfilm */; S[6][ln] = f_6 /* pixel, pad0, lens */;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t i0 = i - ln;   // the wave's first path
            float4 *dst = reinterpret_cast<float4 *>(ps.rec + i0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t q = (uint32_t)j * 64u + ln, r = q >> 3, f = q & 7u;
                if (i0 + r < n) dst[q] = f == 7u ? make_float4(0, 0, 0, 0) : S[f][r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // the next chunk's LDS writes stay behind these reads
        }
#else
        if (i < n) {
            ps.rec[i].L = make_float4(0, 0, 0, 0);
            ps.rec[i].pixel = __float_as_uint(f_6.x);
        }
        if (active) {
            if (TEX) ps.rec[i].lens = make_float2(f_6.z, f_6.w);
            ps.rec[i].ray_o = f_o;
            ps.rec[i].ray_d = f_d;
            ps.rec[i].beta = f_beta;
            ps.rec[i].smp = make_uint4(__float_as_uint(f_smp.x), __float_as_uint(f_smp.y), __float_as_uint(f_smp.z), 0);
            ps.rec[i].pfilm = make_float2(f_5.z, f_5.w);
        }
#endif
        uint32_t pos = wave_append(&ps.qcount[QCI(qout, blockIdx.x & 7)], active);   // this block class's segment of the queue
        if (active) ps.q_ext[qout][(blockIdx.x & 7) * ps.seg_cap + pos] = i;
    }
    wave_count(&ps.counters[MI_CNT_CAMERA_RAYS], ncam);
}

// ---- ray traversal over a queue, persistent lanes with dynamic ray fetch.
// Incoherent rays take wildly different numbers of BVH steps (measured: a plain one-ray-per-lane loop
// keeps 9 % of the VALU lanes busy on the 10 M-triangle scene, profiles/r01_a_*).  Here a lane that
// finishes its ray immediately takes the next one, so a wave only idles at the very end of the queue:
//   * the queue is cut into 8 XCD segments; a wave pulls batches of TRACE_BATCH rays from the segment of the
//     XCD it (most likely) runs on -- one atomic per batch -- and steals from the other segments afterwards;
//   * lanes re-fill when at least TRACE_REFILL of the 64 are idle; between re-fills every lane advances its
//     own ray: up to TRACE_NODE_STEPS interior-node steps, then one leaf step (while-while traversal).
// MODE 0: path-extension rays -> hit record.  MODE 1: MIS rays of EstimateDirect (core/integrator.cpp:167-213)
// -> adds f*Li*weight/scatteringPdf.  MODE 2: shadow rays, VisibilityTester::Unoccluded (core/light.cpp:59-61).
#ifndef TRACE_BATCH
#define TRACE_BATCH 64u
#endif
#ifndef TRACE_REFILL
#define TRACE_REFILL 16
#endif
#ifndef TRACE_NODE_STEPS
#define TRACE_NODE_STEPS 8     /* at most this many node steps between two leaf phases */
#endif
#ifndef TRACE_LEAF_MIN
#define TRACE_LEAF_MIN 24      /* run the leaf phase as soon as this many lanes wait at a leaf */
#endif
#ifndef PT_BATCH_FINALIZE
#define PT_BATCH_FINALIZE 1
#endif
#ifndef PT_TRACE_GUARD_ITERS
#define PT_TRACE_GUARD_ITERS (1u << 22)
#endif
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 1   /* __launch_bounds__ second argument: minimum waves per SIMD the register allocator must allow */
#endif
#ifndef PT_GRID_PER_CU
#define PT_GRID_PER_CU 6   /* persistent blocks per CU (6 x 24 KiB LDS stacks fit the 160 KiB LDS) */
#endif
// Quantised-node instances (round 3): the block keeps the scene's most visited nodes in LDS (TravNodeStepQ) beside its stacks.  Measured on the
// 10 M-triangle frame (profiles/r03_b_*, r03_c_*): the kernels need their 24 waves per CU (one 1024-thread block per CU with 1024 hot nodes: 69 % of
// the node steps off the vector-memory path and still 22 % SLOWER than 6 x 256 threads without any), so the closest-hit and any-hit instances
// of all-triangle scenes without alpha masks (80 VGPRs) run as TWO 768-thread blocks per CU: 16 stack entries x 768 lanes x 4 B = 48 KiB + 512 nodes x 64 B = 32 KiB
// per block, 160 KiB per CU; deeper stack entries go to the per-thread HBM slice as before.  The MIS instance and the instances with spheres or
// alpha masks in the leaf step need 125-190 VGPRs: they keep the 256-thread shape of round 2 and read every node through the vector-memory path.
// PT_HOT_NODES 0 = that shape everywhere.
#ifndef PT_HOT_NODES
#define PT_HOT_NODES 512
#endif
// (round 4: the quantised-node instances of k_trace take the interior step with the shorter tail, TravNodeStepQ2 / TravStackB, pt_scene.h; round 3's step TravNodeStepQ serves
// the hot-node probe and k_shade_vol's per-lane tracer)
#ifndef PT_TRACEQ_BLOCK
#define PT_TRACEQ_BLOCK 768
#endif
#ifndef PT_TRACEQ_LDS_STACK
#define PT_TRACEQ_LDS_STACK 16
#endif
#ifndef PT_TRACEQ_WAVES
#define PT_TRACEQ_WAVES 6   /* waves per SIMD the register allocator must leave room for (2 blocks x 12 waves on 4 SIMDs) */
#endif
// Round 5, MID: the closest-hit / any-hit instances of all-triangle scenes WITH alpha masks.  The mask interpreter behind the wave-wide alpha phase needs ~100 VGPRs beside the
// traversal's own state: at round 2's shape (6 x 256 threads, no register cap) the instance took 149 VGPRs = 3 waves per SIMD and had no hot nodes.  MID = two 512-thread
// blocks per CU (16 stack entries x 512 x 4 B = 32 KiB + the 512 hot nodes = 64 KiB per block), 4 waves per SIMD: a 128-VGPR cap under which the compiler spills what lives
// across the (rare) interpreter call, and every node step of the hot nodes leaves the vector-memory path as in the plain instances.
#ifndef PT_TRACE_MID
#define PT_TRACE_MID 1   /* 0: round 2's shape for the masked instances (A/B) */
#endif
#ifndef PT_TRACE_MID_MIS
#define PT_TRACE_MID_MIS 1   /* the MIS instance of sphere scenes takes the MID shape too (round 6, C2: k_trace<1> 36.7 -> 30.3 ms per frame, profiles/r06_q_*); 0: the 256-thread shape */
#endif
#ifndef PT_TRACE_MID_BLOCK
#define PT_TRACE_MID_BLOCK 512
#endif
#ifndef PT_TRACE_MID_WAVES
#define PT_TRACE_MID_WAVES 4
#endif
template <int MODE, bool SPHERES, bool ALPHA, bool QN> struct TraceShape {
    static constexpr bool BIG = QN && PT_HOT_NODES > 0 && MODE != 1 && !ALPHA && !SPHERES;
    static constexpr bool MID = QN && PT_HOT_NODES > 0 && (MODE != 1 || (PT_TRACE_MID_MIS && SPHERES)) && (ALPHA != SPHERES) && PT_TRACE_MID;   // masks, or (round 6) spheres -- not both: 128 VGPRs hold one of the two leaf extensions
    static constexpr int BLOCK = BIG ? PT_TRACEQ_BLOCK : (MID ? PT_TRACE_MID_BLOCK : PT_BLOCK);
    static constexpr int HOT = (BIG || MID) ? PT_HOT_NODES : 0;
    static constexpr int NLDS = (BIG || MID) ? PT_TRACEQ_LDS_STACK : PT_LDS_STACK;
    static constexpr int WAVES = BIG ? PT_TRACEQ_WAVES : (MID ? PT_TRACE_MID_WAVES : PT_TRACE_WAVES);
    static constexpr int LDS_BYTES = NLDS * BLOCK * (int)sizeof(StackEntry) + HOT * 64;
    static constexpr int PER_CU = (BIG || MID) ? (160 * 1024) / LDS_BYTES : PT_GRID_PER_CU;
    static constexpr bool STEP2 = QN && PT_PEND_LEAF;   // every quantised-node instance (round 4; BIG or the 256-thread shape alike)
    static_assert(NLDS >= PT_LDS_STACK_MIN, "the spill slices are sized for stack_need - PT_LDS_STACK_MIN entries");
    static_assert(PER_CU >= 1, "stacks + hot nodes of one block exceed the CU's 160 KiB of LDS");
    static_assert((size_t)PER_CU * BLOCK <= (size_t)PT_GRID_PER_CU * PT_BLOCK, "the spill slices are sized for gridBlocks x PT_BLOCK threads");
};
// SPHERES: the scene has Sphere primitives (separate instances keep the all-triangle traversal free of the call)
// ALPHA: some mesh has an alpha / shadow-alpha mask (the leaf step then evaluates the mask texture at candidate hits)
// INST: two-level scenes (TransformedPrimitive leaves, see TravStateI); BVH4 only
template <bool INST> struct TravTypes { typedef TravState State; typedef TravStack Stack; typedef StackEntry Entry; typedef LdsStackEntry LdsEntry; enum { LDS = PT_LDS_STACK }; };
template <> struct TravTypes<true> { typedef TravStateI State; typedef TravStack Stack; typedef StackEntry Entry; typedef LdsStackEntry LdsEntry; enum { LDS = PT_LDS_STACK }; };
template <int BLOCK, int NLDS, bool STEP2 = false> struct TravTypesQ { typedef TravStateQ State; typedef TravStackT<BLOCK, NLDS> Stack; typedef StackEntry Entry; typedef LdsStackEntry LdsEntry; enum { LDS = NLDS }; };
template <int BLOCK, int NLDS> struct TravTypesQ<BLOCK, NLDS, true> { typedef TravStateQ State; typedef TravStackB<BLOCK, NLDS> Stack; typedef StackEntry Entry; typedef LdsStackEntry LdsEntry; enum { LDS = NLDS }; };
// QN: the general steps over the 64-byte quantised BVH4 nodes (pt_bvh4q.h): four vector-memory requests per interior step instead of seven
// (the default for single-level scenes; the full-precision 128-byte nodes serve two-level scenes and PBRT_AMD_TRACE=general)
template <bool PEND, class TS> PT_DEV bool TraceDone(const TS &ts) {
    if constexpr (PEND) return ts.cur == TRAV_DONE && ts.pend == TRAV_DONE;
    else return ts.done();
}
// TR (MODE 1 / 2, DevVol::tr_queues): the ray is one SEGMENT of a shadow / MIS ray that is walked through BSDF-less medium interfaces -- closest hit
// (also for shadow segments: the nearest surface decides whether the walk ends or goes on), result into TrState::hit; k_vol_tr_step does the rest
// (Round 4 also built the whole probe chain INTO these lanes -- k_trace<2, ..., TR, WALK> + SssWalkStep, 248 VGPRs; measured in round 5 against the rounds + tail kernel with the
// hit list: 129.2 vs 133.1 Msamples/s on the subsurface C3, profiles/r05_a_* -- and removed.)
template <int MODE, bool COUNT, bool SPHERES, bool ALPHA = false, bool INST = false, bool QN = false, bool TR = false>
__global__ void __launch_bounds__((TraceShape<MODE, SPHERES, ALPHA, QN>::BLOCK), (TraceShape<MODE, SPHERES, ALPHA, QN>::WAVES)) k_trace(DevScene sc, PathState ps, uint32_t qin) {
    static_assert(!QN || !INST, "quantised nodes: single-level BVH4");
    constexpr int BLOCK = TraceShape<MODE, SPHERES, ALPHA, QN>::BLOCK;
    constexpr int HOT = TraceShape<MODE, SPHERES, ALPHA, QN>::HOT;
    constexpr bool PEND = QN && PT_PEND_LEAF;
    constexpr bool ADEFER = PEND && ALPHA && PT_ALPHA_DEFER;   // masks evaluated in wave-wide alpha phases (TravPendStep<..., DEFER>)
    constexpr bool STEP2 = TraceShape<MODE, SPHERES, ALPHA, QN>::STEP2;
    typedef typename std::conditional<QN, TravTypesQ<BLOCK, TraceShape<MODE, SPHERES, ALPHA, QN>::NLDS, STEP2>, TravTypes<INST>>::type TT;
    __shared__ typename TT::Entry lds_stack[TT::LDS * BLOCK];
    __shared__ uint4 lds_hot[HOT ? 4 * HOT : 1];
    if constexpr (ALPHA) NoiseLdsInit();   // alpha masks may be procedural (`dots`, `fbm` ...): the noise table of pt_texture.h in LDS
    if constexpr (HOT > 0) {   // the hot nodes as four word planes [word][node]; coalesced 16-byte reads of nodesq[0 .. n_hot)
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.nodesq);
        const uint32_t nHot = sc.n_hot < (uint32_t)HOT ? sc.n_hot : (uint32_t)HOT;
        for (uint32_t i = threadIdx.x; i < 4u * nHot; i += BLOCK) lds_hot[(i & 3u) * HOT + (i >> 2)] = src[i];
        __syncthreads();   // the only barrier: from here on every wave runs its own loop
    }
    LdsNodeWord *hot = (LdsNodeWord *)lds_hot;
    typename TT::Stack st;
    if constexpr (STEP2) {
        st.base = (typename TT::LdsEntry *)&lds_stack[threadIdx.x];
        *st.base = TRAV_DONE;   // the sentinel under every lane's stack: popped last, ends the ray
        st.reset();
    } else
        st.lds = (typename TT::LdsEntry *)&lds_stack[threadIdx.x];
    st.spill = reinterpret_cast<typename TT::Entry *>(ps.spill) + (size_t)(blockIdx.x * BLOCK + threadIdx.x) * ps.spill_per_thread;
    static_assert(!TR || MODE != 0, "segment walks are for shadow / MIS rays");
    const uint32_t *queue = MODE == 0 ? ps.q_ext[qin] : (MODE == 1 ? ps.q_mis : ps.q_shadow);
    const uint32_t qrow = MODE == 0 ? qin : (MODE == 1 ? ps.qrow_mis : ps.qrow_shadow);
    const uint32_t segLen = ps.seg_cap;
    const uint32_t lane = lane_id();
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    uint32_t seg = blockIdx.x & 7, segsTried = 0;
    uint32_t poolNext = 0, poolEnd = 0;   // wave-uniform
    bool active = false;
    uint32_t slot = 0, lightNum = 0;
    uint32_t qpos = 0;   // MODE 0: the ray's position in the queue (where its sort key goes)
    typename TT::State ts;
    ts.cur = TRAV_DONE;
    if constexpr (PEND) ts.pend = TRAV_DONE;
    TraceCounters tc = {0, 0, 0};
    uint32_t nrays = 0;
    unsigned long long clk0 = 0, rt0 = 0;
    if (COUNT) { clk0 = __builtin_readcyclecounter(); rt0 = wall_clock64(); }   // s_memtime / s_memrealtime: the shader clock this kernel really runs at (mi_trace_clock)
    // hands a finished ray's result over (hit record + sort key / the unoccluded light term / the MIS term)
    auto finalize = [&]() {
        if (active && TraceDone<PEND>(ts)) {
            if constexpr (TR) {   // one segment of a walked shadow / MIS ray: the step kernel reads the hit
                uint32_t hi = TRAV_NO_INSTANCE;
                if constexpr (INST) hi = ts.hitInst;
                ps.trs[slot].hit[MODE == 1 ? 1 : 0] = make_uint4(ts.prim, __float_as_uint(ts.tHit), hi, 0u);
            } else if (MODE == 0) {
                ps.rec[slot].hit = make_uint2(ts.prim, __float_as_uint(ts.tHit));
                if constexpr (INST) ps.rec[slot].pad0 = ts.hitInst;   // which instance the hit primitive was reached through (TRAV_NO_INSTANCE: none)
                uint32_t key = sc.n_materials;                                   // escaped rays
                if (ts.prim != TRAV_MISS) {
                    int mat = (int)sc.tri_info[ts.prim].y;
                    key = mat >= 0 ? (uint32_t)mat : sc.n_materials + 1;         // null-BSDF surfaces: own bucket
                }
                ps.keyrank[qpos].x = key;   // by queue position: k_keycount reads the keys in order
            } else if (MODE == 2) {
                if (ts.prim == TRAV_MISS) {   // unoccluded: add the light-sampled term
                    float4 c = ps.nee[slot].sh_c, L = ps.rec[slot].L;
                    L.x += c.x; L.y += c.y; L.z += c.z;
                    ps.rec[slot].L = L;
                }
            } else {
                const DevLight &light = sc.lights[lightNum];
                RGB Li(0.f);
                const V3 ro = ts.o, rd = ts.d;   // the MIS ray
                if (ts.prim != TRAV_MISS) {
                    if ((int)sc.tri_info[ts.prim].z == (int)lightNum) {   // lightIsect.primitive->GetAreaLight() == &light (integrator.cpp:207)
                        V3 p0, p1, p2;
                        uint32_t tf;
                        LoadTri(sc, ts.prim, &p0, &p1, &p2, &tf);
                        Isect li;
                        if (SPHERES && (tf & TRI_FLAG_SPHERE)) li = SphereIsectToIsect(sc.spheres + __float_as_uint(p0.x), ro, rd, ts.prim);
                        else {
                            TriHit th;
                            TriangleTest(p0, p1, p2, ro, rd, PT_INFINITY, &th);
                            BuildIsect(GeomTables(sc), ts.prim, p0, p1, p2, th, rd, &li);
                        }
                        Li = AreaL(light, li.n, -rd);   // lightIsect.Le(-wi)
                    }
                } else if (light.type == MI_LIGHT_INFINITE)
                    Li = InfiniteLe(&light, rd);   // light.Le(ray)
                if (ps.vol_tr && !Li.IsBlack()) {   // Scene::IntersectTr's Tr for a homogeneous medium up to the first surface (core/scene.cpp:56-70, homogeneous.cpp:41-44)
                    float4 sg = ps.nee[slot].pad[0];
                    if (sg.x != 0 || sg.y != 0 || sg.z != 0) {
                        Float th = PT_INFINITY;
                        if (ts.prim != TRAV_MISS) th = ts.tHit;
                        Li = Li * ExpRGB(-RGB(sg.x, sg.y, sg.z) * mn(th * rd.Length(), PT_MAX_FLOAT));
                    }
                }
                if (!Li.IsBlack()) {
                    float4 c = ps.nee[slot].mi_c, L = ps.rec[slot].L;
                    L.x += c.x * Li.r; L.y += c.y * Li.g; L.z += c.z * Li.b;
                    ps.rec[slot].L = L;
                }
            }
            active = false;
        }
    };
    uint32_t waveIters = 0;
    unsigned long long apend = 0;   // ADEFER: the lanes whose parked triangle waits for its mask (wave-uniform)
    while (true) {
        // PT_BATCH_FINALIZE: the rays that finished since the last refill hand their results over TOGETHER (one pass through the block with all of
        // them, instead of one pass per scheduling round with a lane or two: the kernel is bound by VALU issue and a wave instruction costs the
        // same for 1 lane or 64)
        if (PT_BATCH_FINALIZE) finalize();
        unsigned long long idle = PtBallot(!active);
        int nIdle = __popcll(idle);
        if (nIdle >= TRACE_REFILL && segsTried < 8) {
            while (segsTried < 8 && nIdle > 0) {
                if (poolNext >= poolEnd) {   // take the next batch of this segment (one atomic per wave and batch)
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&ps.cursor[seg * QC_STRIDE], TRACE_BATCH);   // one cache line per segment cursor
                    base = __shfl(base, 0);
                    uint32_t segBeg = seg * segLen, segEnd;
                    segEnd = segBeg + ps.qcount[QCI(qrow, seg)];
                    if (segBeg + base >= segEnd) {
                        seg = (seg + 1) & 7; ++segsTried;
                        poolNext = poolEnd = 0;
                        continue;
                    }
                    poolNext = segBeg + base;
                    poolEnd = poolNext + TRACE_BATCH < segEnd ? poolNext + TRACE_BATCH : segEnd;
                }
                uint32_t avail = poolEnd - poolNext;
                uint32_t rank = (uint32_t)__popcll(idle & ltMask);
                if (!active && rank < avail) {
                    if (MODE == 0 && !TR) qpos = poolNext + rank;
                    slot = queue[poolNext + rank];
                    float4 o4 = MODE == 0 ? ps.rec[slot].ray_o : (MODE == 1 ? ps.nee[slot].mi_o : ps.nee[slot].sh_o);
                    float4 d4 = MODE == 0 ? ps.rec[slot].ray_d : (MODE == 1 ? ps.nee[slot].mi_d : ps.nee[slot].sh_d);
                    if (MODE == 1) lightNum = __float_as_uint(d4.w);
                    ts.init(sc, V3(o4.x, o4.y, o4.z), V3(d4.x, d4.y, d4.z), MODE == 1 ? PT_INFINITY : o4.w, st);
                    active = true;
                    ++nrays;
                }
                poolNext += (uint32_t)nIdle < avail ? (uint32_t)nIdle : avail;
                idle = PtBallot(!active);
                nIdle = __popcll(idle);
            }
            waveIters = 0;   // the guard bounds the rounds BETWEEN two refills (and after the last one), not the whole persistent launch: independent of the frame size
        }
        if (!PtAny(active)) break;
        const bool mayRefill = segsTried < 8;
        while (true) {
            // safety net: a traversal that does not terminate (a corrupted stack / node reference) must not hang the GPU -- after
            // PT_TRACE_GUARD_ITERS scheduling rounds (two orders of magnitude beyond any real frame) the wave drops its rays and
            // reports it (MI_CNT_TRACE_GUARD_TRIPS: checked to be zero by the parity tests, bench.py and smoke())
            if (++waveIters > PT_TRACE_GUARD_ITERS) {
                if (lane == 0) atomicAdd(&ps.counters[MI_CNT_TRACE_GUARD_TRIPS], 1ull);
                active = false; segsTried = 8;
                break;
            }
            // node phase: keep stepping through interior nodes while enough lanes still want one (lanes that reached
            // a leaf or finished wait); then ONE leaf phase for everybody waiting at a leaf.  A leaf step (up to 16
            // watertight triangle tests) costs several node steps, so it should run with many lanes, not for one.
            {
                int guard = 0;
                while (true) {
                    bool wantNode = ts.atNode();   // (a lane without a ray stands at TRAV_DONE with nothing parked: no `active &&` in the votes of this loop)
                    int nWant = __popcll(PtBallot(wantNode));
                    if (nWant == 0) break;
                    if (wantNode) {
                        if constexpr (QN) {
                            if constexpr (STEP2) TravNodeStepQ2<COUNT, HOT>(sc, ts, st, &tc, hot);
                            else TravNodeStepQ<COUNT, HOT>(sc, ts, st, &tc, hot);
                            if constexpr (PEND) TravParkLeaf(ts, st);   // arrived at a leaf: park it, go on with the stack
                        } else TravNodeStep<COUNT, !(MODE == 2 && PT_ANY_NOSORT)>(sc, ts, st, &tc);
                    }
                    int nLeaf;
                    if constexpr (PEND) nLeaf = __popcll(PtBallot(ts.pend != TRAV_DONE) & ~apend);
                    else nLeaf = __popcll(PtBallot(ts.atLeaf()));
                    if (nLeaf >= TRACE_LEAF_MIN || ++guard >= TRACE_NODE_STEPS) break;
                }
            }
            if constexpr (ADEFER) {
                bool cand = false;
                if (active && ts.pend != TRAV_DONE && !((apend >> lane) & 1ull)) TravPendStep<MODE == 2 && !TR, COUNT, SPHERES, ALPHA, typename TT::Stack, true>(sc, ts, st, &tc, &cand);
                apend |= PtBallot(cand);
                if (apend) {
                    const bool waits = (apend >> lane) & 1ull;
                    const int nA = __popcll(apend), nGo = __popcll(PtBallot(active && (ts.atNode() || (ts.pend != TRAV_DONE && !waits))));   // lanes with a step to take without a mask
                    if (nA >= PT_ALPHA_MIN || nGo * PT_ALPHA_GO_MUL <= nA) {   // alpha phase: the same step again, this time through the mask
                        if (waits) TravPendStep<MODE == 2 && !TR, false, SPHERES, ALPHA, typename TT::Stack, false>(sc, ts, st, &tc);
                        apend = 0;
                    }
                }
            } else if constexpr (PEND) {
                if (ts.pend != TRAV_DONE) TravPendStep<MODE == 2 && !TR, COUNT, SPHERES, ALPHA, typename TT::Stack>(sc, ts, st, &tc);
            } else {
                if (active && ts.atLeaf()) TravLeafStep<MODE == 2 && !TR, COUNT, SPHERES, ALPHA, typename TT::State, typename TT::Stack, INST>(sc, ts, st, &tc);
            }
            if (!PT_BATCH_FINALIZE) finalize();
            int nAct = __popcll(PtBallot(!TraceDone<PEND>(ts)));   // (batched: finished lanes keep their result until the next refill, top of the outer loop)
            if (nAct == 0 || (mayRefill && nAct <= 64 - TRACE_REFILL)) break;
        }
    }
    wave_count(&ps.counters[MODE == 2 && !TR ? MI_CNT_SHADOW_RAYS : MI_CNT_CLOSEST_RAYS], nrays);   // (walked segments: closest-hit queries, as the general form of k_shade_vol counts them)
    if (MODE == 1 && !TR) wave_count(&ps.counters[MI_CNT_MIS_RAYS], nrays);
    if (COUNT) {
        wave_count(&ps.counters[MODE == 0 ? MI_CNT_NODES_CLOSEST : (MODE == 1 ? MI_CNT_NODES_MIS : MI_CNT_NODES_ANY)], tc.nodes);
        wave_count(&ps.counters[MODE == 0 ? MI_CNT_TRIS_CLOSEST : (MODE == 1 ? MI_CNT_TRIS_MIS : MI_CNT_TRIS_ANY)], tc.tris);
        if (HOT > 0) wave_count(&ps.counters[MODE == 0 ? MI_CNT_NODES_HOT_CLOSEST : (MODE == 1 ? MI_CNT_NODES_HOT_MIS : MI_CNT_NODES_HOT_ANY)], tc.hot);
        if (MODE != 1 && lane == 0) {
            atomicAdd(&ps.counters[PT_CNT_CLK + (MODE == 2 ? 2 : 0)], (unsigned long long)__builtin_readcyclecounter() - clk0);
            atomicAdd(&ps.counters[PT_CNT_CLK + (MODE == 2 ? 3 : 1)], (unsigned long long)wall_clock64() - rt0);
        }
    }
}

// ---- hot-node probe (round 3).  Which BVH4Q nodes should the traversal blocks keep in LDS?  Not the top of the tree: on the 10 M-triangle frame the
// 1024 nodes with the largest boxes receive 28 % of the node visits, the 1024 most VISITED ones 75 % (tools/bvh_study.py --hot, profiles/r03_b_*) --
// the camera sees a part of the scene and the paths stay near it.  So mi_scene_upload measures: this kernel walks a few thousand probe paths
// (camera samples on a regular pixel grid, then cosine-distributed bounces off the geometric normal with a hash for random numbers) through the
// quantised tree and counts the visits per node; the host renumbers the nodes so that the most visited ones are nodesq[0 .. n_hot).  The choice
// only decides WHERE a node is read from, never what is read: hits are the same for every choice (the parity suite runs with it).
PT_DEV uint32_t ProbeHash(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__global__ void __launch_bounds__(PT_BLOCK) k_hot_probe(DevScene sc, uint32_t nProbe, uint32_t npx, uint32_t npy, int bounces, uint32_t *visits, StackEntry *spill, int spillPerThread) {
    __shared__ StackEntry lds_stack[PT_LDS_STACK * PT_BLOCK];
    const uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= nProbe) return;
    TravStack st;
    st.lds = (LdsStackEntry *)&lds_stack[threadIdx.x];
    st.spill = spill + (size_t)i * spillPerThread;
    const uint32_t W = (uint32_t)(sc.sample_max[0] - sc.sample_min[0]), H = (uint32_t)(sc.sample_max[1] - sc.sample_min[1]);
    const uint32_t gx = i % npx, gy = (i / npx) % npy, s = (i / (npx * npy)) % (uint32_t)sc.spp;
    const int x = sc.sample_min[0] + (int)(((uint64_t)gx * W + W / 2) / npx), y = sc.sample_min[1] + (int)(((uint64_t)gy * H + H / 2) / npy);
    // the probe's own camera samples (a hash of the path number): it must not touch the scene's sampler state -- the tile-serial samplers keep one
    // stream per tile in pix_rng / pix_s1 / pix_s2, not yet initialised at upload time and not indexable per probe lane
    V3 o, d;
    Float tMax;
    const Float h0 = (Float)(ProbeHash(i, 0x51u + s) >> 8) * 0x1p-24f, h1 = (Float)(ProbeHash(i, 0x52u + s) >> 8) * 0x1p-24f;
    const Float h2 = (Float)(ProbeHash(i, 0x53u + s) >> 8) * 0x1p-24f, h3 = (Float)(ProbeHash(i, 0x54u + s) >> 8) * 0x1p-24f;
    CameraRayFromFilm(sc, (Float)x + h0, (Float)y + h1, h2, h3, &o, &d, &tMax);
    TraceCounters tc = {0, 0, 0};
    for (int b = 0; b <= bounces; ++b) {
        TravStateQ ts;
        ts.init(sc, o, d, tMax, st);
        while (!ts.done()) {
            if (ts.atNode()) { atomicAdd(&visits[ts.cur], 1u); TravNodeStepQ<false>(sc, ts, st, &tc); }
            else TravLeafStep<false, false, true, false, TravStateQ, TravStack, false>(sc, ts, st, &tc);
        }
        if (ts.prim == TRAV_MISS) break;
        V3 p0, p1, p2;
        uint32_t fl;
        LoadTri(sc, ts.prim, &p0, &p1, &p2, &fl);
        if (fl & TRI_FLAG_SPHERE) break;
        V3 n = Cross(p1 - p0, p2 - p0);
        const Float len = n.Length();
        if (!(len > 0)) break;
        n = n * (1 / len);
        if (Dot(n, d) > 0) n = -n;
        const V3 p = o + d * ts.tHit;
        const Float u1 = (Float)(ProbeHash(i, 2u * (uint32_t)b) >> 8) * 0x1p-24f, u2 = (Float)(ProbeHash(i, 2u * (uint32_t)b + 1u) >> 8) * 0x1p-24f;
        const Float r = sqrtf(u1), ph = 6.2831853f * u2, cz = sqrtf(mx((Float)0, 1 - u1));
        const V3 a = absf(n.x) > 0.9f ? V3(0, 1, 0) : V3(1, 0, 0);
        const V3 t1 = Normalize(Cross(n, a)), t2 = Cross(n, t1);
        d = t1 * (r * cosf(ph)) + t2 * (r * sinf(ph)) + n * cz;
        o = p + n * (1e-4f * mx(mx(absf(p.x), absf(p.y)), mx(absf(p.z), (Float)1)));
        tMax = PT_INFINITY;
    }
}
// the shading kernels' per-triangle line (DevScene::tri_rec) from the three per-triangle arrays
__global__ void __launch_bounds__(PT_BLOCK) k_build_tri_rec(const float4 *verts, const TriShade *shade, const uint4 *info, float4 *rec, uint32_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < 8ull * n; i += (uint64_t)gridDim.x * PT_BLOCK) {
        const uint32_t t = (uint32_t)(i >> 3), f = (uint32_t)(i & 7u);
        float4 v;
        if (f < 3) v = verts[3 * (size_t)t + f];
        else if (f < 7) v = reinterpret_cast<const float4 *>(shade + t)[f - 3];
        else { const uint4 u = info[t]; v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)); }
        rec[i] = v;
    }
}
// dst[newIdx[i]] = src[i] with the interior child references renumbered the same way (leaf references and empty slots carry the leaf bit)
__global__ void __launch_bounds__(PT_BLOCK) k_renumber_nodes(const BVH4QNode *src, BVH4QNode *dst, const uint32_t *newIdx, uint32_t n) {
    for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += gridDim.x * PT_BLOCK) {
        BVH4QNode nd = src[i];
        for (int k = 0; k < 4; ++k) if (!(nd.child[k] & BVH4_LEAF)) nd.child[k] = newIdx[nd.child[k]];
        dst[newIdx[i]] = nd;
    }
}

// (Round 2 also binned the path-extension rays by origin cell x direction octant before the traversal -- k_raybin_*: -63 % HBM traffic and -48 % L2 misses in k_trace at
// UNCHANGED kernel time, profiles/r02_b_* -- the evidence that the traversal is not bound by memory; it cost 3 % of the frame, stayed off, and was removed in round 5.)
// ---- counting sort of the traced paths by material key, without global atomics:
//   k_keycount : every (persistent) block histograms the keys of ITS chunks in LDS -- wave ballots merge equal
//                keys, so an LDS atomic is issued per (wave, distinct key) -- and records each path's rank inside
//                the block; the block histogram goes to blockhist[block][key]
//   k_scan_keys: offsets[block][key] = keyBase[key] + sum of the histograms of lower blocks (one thread per key)
//   k_scatter  : same chunk->block mapping, so sorted[offsets[block][key] + rank] = path
#ifndef PT_KEYRANK_ATOMIC
#define PT_KEYRANK_ATOMIC 1   /* round 6: sort 7.7 -> 4.4 ms per C3 frame, 29.3 -> 21.2 on C4 (profiles/r06_r_*); 0: the ballot rounds of rounds 1-5 */
#endif
__global__ void __launch_bounds__(PT_BLOCK) k_keycount(DevScene sc, PathState ps, uint32_t qin, uint32_t nkeys) {
    extern __shared__ uint32_t lhist[];
    for (uint32_t k = threadIdx.x; k < nkeys; k += PT_BLOCK) lhist[k] = 0;
    __syncthreads();
    for (SegIter it(ps.qcount, qin, ps.seg_cap); it.more(); it.next()) {
        uint32_t i = it.item();
        bool active = it.valid();
        uint32_t key = 0;
        if (active) {
            key = ps.keyrank[i].x;   // left by k_trace<0> at the ray's queue position
            if (ps.key_remap) key = ps.key_remap[key];
        }
#if PT_KEYRANK_ATOMIC
        uint32_t rank = active ? atomicAdd(&lhist[key], 1u) : 0u;   // one returning LDS atomic per lane: the hardware serialises equal keys (cost ~ the largest group of equal keys in the wave)
#else
        uint32_t rank = wave_key_rank(lhist, key, active);           // one LDS atomic per (wave, distinct key): a ballot round per distinct key
#endif
        if (active) ps.keyrank[i] = make_uint2(key, rank);   // indexed by queue position: k_scatter walks the same chunks
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nkeys; k += PT_BLOCK) ps.blockhist[(size_t)blockIdx.x * nkeys + k] = lhist[k];
}
// (round 5: one BLOCK per key scans that key's column of the block histograms -- rounds 1-4 walked the column with one thread, ~1 500 dependent fetches = 0.3 ms per bounce
// whatever the queue held: 2.4 % of the maxdepth-30 frame -- and a second, one-block launch turns the per-key totals into the key offsets)
__global__ void __launch_bounds__(PT_BLOCK) k_scan_keys(PathState ps, uint32_t nkeys, uint32_t nblocks) {
    __shared__ uint32_t part[PT_BLOCK];
    const uint32_t k = blockIdx.x;
    if (k >= nkeys) return;
    const uint32_t per = (nblocks + PT_BLOCK - 1) / PT_BLOCK, b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += ps.blockhist[(size_t)b * nkeys + k];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {   // exclusive prefix over the 256 partial sums (serial: 256 LDS reads)
        uint32_t run = 0;
        for (uint32_t j = 0; j < PT_BLOCK; ++j) { uint32_t v = part[j]; part[j] = run; run += v; }
        ps.keycount[k] = run;   // this key's total
    }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) { uint32_t h = ps.blockhist[(size_t)b * nkeys + k]; ps.blockhist[(size_t)b * nkeys + k] = acc; acc += h; }
}
__global__ void __launch_bounds__(PT_BLOCK) k_scan_keys_total(PathState ps, uint32_t nkeys) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0;
    for (uint32_t k = 0; k < nkeys; ++k) { ps.keyoffset[k] = run; run += ps.keycount[k]; }
    ps.qcount[QCI(QC_SORTED, 0)] = run;
}
__global__ void __launch_bounds__(PT_BLOCK) k_scatter(PathState ps, uint32_t qin, uint32_t nkeys) {
    for (SegIter it(ps.qcount, qin, ps.seg_cap); it.more(); it.next()) {
        uint32_t i = it.item();
        if (it.valid()) {
            uint32_t slot = ps.q_ext[qin][i];
            uint2 kr = ps.keyrank[i];
            ps.q_sorted[ps.keyoffset[kr.x] + ps.blockhist[(size_t)blockIdx.x * nkeys + kr.x] + kr.y] = slot;
        }
    }
}

// ---- SpatialLightDistribution::ComputeDistribution (core/lightdistrib.cpp:219-300), evaluated for every voxel at upload
// time.  halton: the 128 x {RadicalInverse(0..4, i)} values (identical for every voxel).  One thread per (voxel, light)
// keeps the per-light sum over the samples in the reference's order.
#define PT_SPATIAL_SAMPLES 128
PT_DEV Float LerpB(Float t, Float a, Float b) { return (1 - t) * a + t * b; }   // pbrt::Lerp via Bounds3::Lerp geometry.h:775-779
__global__ void __launch_bounds__(PT_BLOCK) k_spatial_contrib(DevScene sc, float *contrib, const float *halton, uint64_t total) {
    __shared__ float s_h[PT_SPATIAL_SAMPLES * 5];
    for (uint32_t k = threadIdx.x; k < PT_SPATIAL_SAMPLES * 5; k += PT_BLOCK) s_h[k] = halton[k];
    __syncthreads();
    const uint32_t nl = sc.n_lights;
    for (uint64_t i = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < total; i += (uint64_t)gridDim.x * PT_BLOCK) {
        uint32_t vox = (uint32_t)(i / nl), j = (uint32_t)(i - (uint64_t)vox * nl);
        int pi2 = (int)(vox % (uint32_t)sc.sp_nvox[2]);
        uint32_t r = vox / (uint32_t)sc.sp_nvox[2];
        int pi1 = (int)(r % (uint32_t)sc.sp_nvox[1]), pi0 = (int)(r / (uint32_t)sc.sp_nvox[1]);
        V3 bmin = v3(sc.sp_bmin), bmax = v3(sc.sp_bmax);
        V3 t0(Float(pi0) / Float(sc.sp_nvox[0]), Float(pi1) / Float(sc.sp_nvox[1]), Float(pi2) / Float(sc.sp_nvox[2]));
        V3 t1(Float(pi0 + 1) / Float(sc.sp_nvox[0]), Float(pi1 + 1) / Float(sc.sp_nvox[1]), Float(pi2 + 1) / Float(sc.sp_nvox[2]));
        V3 a(LerpB(t0.x, bmin.x, bmax.x), LerpB(t0.y, bmin.y, bmax.y), LerpB(t0.z, bmin.z, bmax.z));
        V3 b(LerpB(t1.x, bmin.x, bmax.x), LerpB(t1.y, bmin.y, bmax.y), LerpB(t1.z, bmin.z, bmax.z));
        V3 vmin(mn(a.x, b.x), mn(a.y, b.y), mn(a.z, b.z)), vmax(mx(a.x, b.x), mx(a.y, b.y), mx(a.z, b.z));   // Bounds3(p1, p2)
        const DevLight &light = sc.lights[j];
        Isect intr;
        intr.pError = V3(); intr.n = V3(); intr.ns = V3(); intr.wo = V3(1, 0, 0);
        Float acc = 0;
        for (int k = 0; k < PT_SPATIAL_SAMPLES; ++k) {
            const float *h = &s_h[5 * k];
            intr.p = V3(LerpB(h[0], vmin.x, vmax.x), LerpB(h[1], vmin.y, vmax.y), LerpB(h[2], vmin.z, vmax.z));
            LightSample ls = SampleLiAny(GeomTables(sc), &light, intr.p, intr.pError, intr.n, h[3], h[4]);
            if (ls.pdf > 0) acc += ls.Li.y() / ls.pdf;
        }
        contrib[i] = acc;
    }
}
// second half of ComputeDistribution + the Distribution1D constructor (core/sampling.h:55-70), one thread per voxel
// (the sums are sequential in the reference; the order is kept)
__global__ void __launch_bounds__(PT_BLOCK) k_spatial_cdf(uint32_t nvox, uint32_t nl, float *func /* in: contrib */, float *cdf, float *funcInt) {
    for (uint32_t v = blockIdx.x * PT_BLOCK + threadIdx.x; v < nvox; v += gridDim.x * PT_BLOCK) {
        float *f = func + (size_t)v * nl, *c = cdf + (size_t)v * (nl + 1);
        Float sum = 0;
        for (uint32_t j = 0; j < nl; ++j) sum = sum + f[j];
        Float avg = sum / (Float)((uint64_t)PT_SPATIAL_SAMPLES * (uint64_t)nl);
        Float minContrib = (avg > 0) ? (Float)(.001 * (double)avg) : 1;
        for (uint32_t j = 0; j < nl; ++j) f[j] = mx(f[j], minContrib);
        int n = (int)nl;
        c[0] = 0;
        for (int i = 1; i < n + 1; ++i) c[i] = c[i - 1] + f[i - 1] / n;
        Float fi = c[n];
        if (fi == 0) for (int i = 1; i < n + 1; ++i) c[i] = Float(i) / Float(n);
        else for (int i = 1; i < n + 1; ++i) c[i] /= fi;
        funcInt[v] = fi;
    }
}

// guide words of SpatialPick (pt_shade.h), one thread per voxel: two merges of the voxel's non-decreasing cdf against the M cell bounds
__global__ void __launch_bounds__(PT_BLOCK) k_spatial_guide(uint32_t nvox, uint32_t nl, uint32_t M, const float *cdf, uint32_t *guide) {
    for (uint32_t v = blockIdx.x * PT_BLOCK + threadIdx.x; v < nvox; v += gridDim.x * PT_BLOCK) {
        const float *c = cdf + (size_t)v * (nl + 1);
        const uint32_t size = nl + 1;
        uint32_t g = 0, h = 0;
        const Float invM = 1 / (Float)M;   // exact: M is a power of two
        for (uint32_t j = 0; j < M; ++j) {
            const Float lo = (Float)j * invM, hi = (Float)(j + 1) * invM;
            while (g < size && c[g] <= lo) ++g;   // G = #{cdf[k] <= j / M}
            while (h < size && c[h] < hi) ++h;    // H = #{cdf[k] < (j + 1) / M}
            guide[(size_t)v * M + j] = g | (h << 16);
        }
    }
}

#define PT_CNT_ALLOC 96

// ---- shading: one path vertex per lane, lanes of a wave share a material (sorted queue)
#ifndef PT_CDF_LDS
#define PT_CDF_LDS 2048
#endif
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 3   /* 168 VGPRs -> 3 waves per SIMD: measured best of 2..5 (profiles/r01 notes) */
#endif
#ifndef PT_SHADE_GRID_PER_CU
#define PT_SHADE_GRID_PER_CU PT_SHADE_WAVES   /* one round of resident blocks: DynIter hands the items out dynamically.  (Rounds 1-4: four rounds, which evened out the STATIC chunk partition; re-measured over the dynamic, interleaved one in round 5, profiles/r05_tux_*: 3 / 6 / 12 blocks per CU within 0.3 % on C3, C2, C4 and the textured C3) */
#endif
// ENV ("rich" scenes): an infinite light with a radiance map (escaped rays look it up) or Sphere primitives; plain scenes run the leaner instance
// TEX: some material has image / procedural textures or a bump map -- every material's lobe list is then a per-lane record
// (built per hit by pt_material.h for the textured ones, copied for the constant ones) and the BSDF code reads it per lane
#ifndef PT_TEX_SHADE_WAVES
#define PT_TEX_SHADE_WAVES 2   /* the textured / volumetric instances: 242-256 VGPRs, 2 waves per SIMD -- measured against 3 (168 VGPRs, more scratch): textured C3 141.8 -> 145.2, volpath C3 144.2 -> 149.9 Msamples/s (profiles/r02_j_*); the plain instances keep 168 */
#endif
// INST: two-level scenes -- the hit primitive may have been reached through an instance (PathRec::pad0): the interaction is
// built in the object's space from the transformed ray and carried back to world space (only instantiated together with ENV and TEX)
// SMP: 0 Sobol' / 1 Halton (the vertex's dimensions drawn in one batch) / 2 the tile-serial samplers (drawn call by call, in the reference's order: the
// values come from the tile's stream)
template <bool ENV, int SMP, bool TEX, bool INST = false>
__global__ void __launch_bounds__(PT_BLOCK, (TEX ? PT_TEX_SHADE_WAVES : PT_SHADE_WAVES)) k_shade(DevScene sc, PathState ps, uint32_t qout) {
    // light-selection CDF in LDS when it fits: Distribution1D::SampleDiscrete is a chain of dependent look-ups
    __shared__ float s_cdf[PT_CDF_LDS];
    if constexpr (TEX) NoiseLdsInit();   // procedural textures / bump maps: the noise table of pt_texture.h in LDS
#if PT_SHADE_PROF
    if ((threadIdx.x & 63) < 24) { s_pacc[threadIdx.x >> 6][threadIdx.x & 63] = 0; s_pcnt[threadIdx.x >> 6][threadIdx.x & 63] = 0; s_plan[threadIdx.x >> 6][threadIdx.x & 63] = 0; }
    if ((threadIdx.x & 63) == 0) s_prof[threadIdx.x >> 6] = clock64();
#endif
    const bool cdfInLds = sc.n_lights + 1 <= PT_CDF_LDS;
    if (cdfInLds) for (uint32_t k = threadIdx.x; k < sc.n_lights + 1; k += PT_BLOCK) s_cdf[k] = sc.light_cdf[k];
    __syncthreads();
    const float *cdf = cdfInLds ? s_cdf : sc.light_cdf;
    uint32_t n, sbase;
    ShadeRange(ps, &sbase, &n);
    uint32_t nseg = 0;
    for (DynIter it(n, ps.cursor); it.more(); it.next()) {
        uint32_t i = it.item();
        PROBE(0)   // loop overhead / queue bookkeeping of the previous item
        bool active = it.valid();
        bool cont = false, wantShadow = false, wantMis = false;
        uint32_t slot = 0;
        if (active) {
            slot = ps.q_sorted[sbase + i];
            uint2 hr = ps.rec[slot].hit;
            float4 o4 = ps.rec[slot].ray_o, d4 = ps.rec[slot].ray_d, b4 = ps.rec[slot].beta, L4 = ps.rec[slot].L;
            uint4 s4 = ps.rec[slot].smp;
            V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            RGB beta(b4.x, b4.y, b4.z), L(L4.x, L4.y, L4.z);
            Float etaScale = b4.w;
            int bounces = (int)(s4.w & 0xffffu);
            bool specularBounce = (s4.w >> 16) & 1u;
            bool noDiff = (s4.w >> 17) & 1u;   // a null-material surface was stepped through: the ray is a plain Ray from then on (read by the textured instances; every instance carries it -- a path may change kernels from vertex to vertex, mi_ctx::sssRoute)
            PROBE(1)   // queue + path record loads
            Sampler smp;
            smp.index = (uint64_t)s4.x | ((uint64_t)s4.y << 32);
            smp.dimension = (int)s4.z;
            smp.px = smp.py = 0;   // only dimensions 0/1 (camera sample) look at the pixel
            // the (at most) 8 sample dimensions this vertex can consume: light pick, uLight, uScattering, BSDF, RR
            Float us[8];
            constexpr bool PIX = SMP == 2;
            if (PIX) {
#pragma unroll
                for (int k = 0; k < 8; ++k) us[k] = 0;
            } else if (SMP == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) us[k] = HaltonSampleDimension(sc, smp.index, smp.dimension + k);
            } else
                SobolBatch<8>(sc, smp.index, smp.dimension, us);
            PROBE(2)   // Sobol batch
            int ui = 0;      // sample dimensions consumed by this vertex (added to the path's dimension at the end)
            int ubase = 0;   // sample dimensions consumed before the BSDF sample: 0, 1 (light pick only) or 5 -- static indices keep us[] in registers
            ++nseg;
            bool found = hr.x != MISS_PRIM;
            // path.cpp:91-101: emitted light at the vertex / from the environment
            Isect isect;
            IsectX ix;
            uint4 tinfo = make_uint4(0, 0, 0, 0);
            if (found) {
                TriShadeRegs tsr;
                V3 p0, p1, p2;
                uint32_t tf;
                LoadHitTriangle(sc, hr.x, &p0, &p1, &p2, &tf, &tsr, &tinfo);   // one 128-byte line (DevScene::tri_rec)
                Pin(tsr.a, tsr.b, tsr.c); Pin(tsr.d); Pin(tinfo);
                V3 iro = ro, ird = rd;   // the ray in the space the primitive lives in
                const DevInstance *hitInst = nullptr;
                if constexpr (INST) {
                    uint32_t hi = ps.rec[slot].pad0;
                    if (hi != TRAV_NO_INSTANCE) {
                        hitInst = c_instances + hi;
                        InstRay ir = InstanceRay(hitInst, ro, rd);
                        iro = ir.o; ird = ir.d;
                    }
                }
                if (ENV && (tf & TRI_FLAG_SPHERE)) {
                    isect = SphereIsectToIsect(sc.spheres + __float_as_uint(p0.x), iro, ird, hr.x);
                    if (TEX) ix = SphereIsectTex(sc.spheres + __float_as_uint(p0.x), iro, ird);
                } else {
                    TriHit th;
                    TriangleTest(p0, p1, p2, iro, ird, PT_INFINITY, &th);   // same code, same inputs as the traversal: same b0,b1,b2,t
                    isect = MakeIsect(BuildIsectPre(tinfo.x, tsr, p0, p1, p2, V3(th.b0, th.b1, th.b2)), ird, hr.x);
                    if (TEX) ix = BuildIsectTex(tinfo.x, tsr, p0, p1, p2, V3(th.b0, th.b1, th.b2));
                }
                if constexpr (INST) {
                    if (hitInst && !hitInst->identity) InstanceToWorld(hitInst, &isect, &ix);
                }
            }
            // The light pick of a spatial table is a chain of dependent fetches (voxel -> funcInt, guide word -> cdf / func entries -> light record): its first link is
            // issued here, as soon as the hit point is known, and flies while the emission and the BSDF set-up run -- same arithmetic on the same values (measured: k_shade
            // 131.1 -> 128.6 ms per C3 frame, profiles/r05_aa_*; issuing the WHOLE pick here gained less, 129.7).  The guide word needs the light-pick dimension:
            // only for the samplers that have drawn it by now (us[0]).
            const bool spatial = sc.light_strategy == MI_LIGHT_STRATEGY_SPATIAL;
            size_t vox = 0;
            Float funcIntE = 0;
            uint32_t guideE = 0;
            const bool pickEarly = spatial && found && bounces < sc.max_depth && (int)tinfo.y >= 0 && sc.n_lights > 0;
            if (pickEarly) {
                vox = SpatialVoxel(sc, isect.p);
                funcIntE = sc.sp_func_int[vox];
                if (SMP != 2 && sc.sp_guide_m) guideE = SpatialGuideWord(sc, vox, us[0]);
            }
            PROBE(3)   // triangle reload + BuildIsect
            if (bounces == 0 || specularBounce) {
                if (found) {
                    int li = (int)tinfo.z;
                    if (li >= 0) { RGB Le = AreaL(sc.lights[li], isect.n, -rd); L = L + beta * Le; }
                } else {
                    for (uint32_t k = 0; k < sc.n_infinite; ++k) {   // infiniteLights Le(ray)
                        const DevLight &il = sc.lights[sc.infinite_lights[k]];
                        L = L + beta * (ENV ? InfiniteLe(&il, rd) : rgb3(il.L));
                    }
                }
            }
            if (found && bounces < sc.max_depth) {
                int matIdx = (int)tinfo.y;
                if (matIdx < 0) {
                    // null BSDF: step through the surface, same bounce count, no sampler use (path.cpp:108-113)
                    V3 no = OffsetRayOrigin(isect.p, isect.pError, isect.n, rd);
                    ps.rec[slot].ray_o = make_float4(no.x, no.y, no.z, PT_INFINITY);
                    cont = true;
                    noDiff = true;
                } else {
                    PROBE(4)   // emission
                    // Lanes of a wave share a material almost always (sorted queue); at the boundary between two materials
                    // the block below runs once per distinct material, each time with a WAVE-UNIFORM material pointer:
                    // lobe counts, types and parameters come through scalar loads and every lobe switch is a scalar branch.
                    bool matTodo = true;
                    while (matTodo) {
                    int matU = UniformInt(matIdx);
                    if (SameAs(matIdx, matU)) {
                    matTodo = false;
                    const mi_material *matPtr = sc.materials + matU;
                    typedef BSDF_T<!TEX> BS;
                    mi_material laneMat;
                    if (TEX) {
                        // isect.ComputeScatteringFunctions(ray, arena, true) path.cpp:107: ComputeDifferentials (camera rays only: every
                        // later ray is a plain Ray, path.cpp:159), then the material's textures / bump map at this hit
                        if (bounces == 0 && !noDiff) {
                            float2 pf = ps.rec[slot].pfilm, ln = ps.rec[slot].lens;
                            RayDiffT rdf = CameraDifferentials(&c_tex.camera, pf.x, pf.y, ln.x, ln.y, c_tex.spp, ro, rd);
                            ComputeDifferentials(isect.p, isect.n, &ix, rdf);
                        }
                        PROBE(14)   // (textured instances) differentials of camera rays
                        ComputeScatteringFunctionsT<true>(sc.materials, matU, &isect, &ix, &laneMat);   // (matU: wave-uniform inside the waterfall)
                        matPtr = &laneMat;
                        PROBE(15)   // (textured instances) the material's textures / bump map -> per-lane lobe list
                    }
                    BS bsdf(isect, matPtr, TEX ? nullptr : sc.mat_pack, matU);
                    // ---- UniformSampleOneLight (core/integrator.cpp:85-106)
                    if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0 && sc.n_lights > 0) {
                        // path.cpp:125-127 lightDistribution->Lookup(isect.p): the one table, or the voxel's (lightdistrib.cpp:139-152)
                        Float funcInt = sc.light_func_int;
                        if (spatial) {
                            if (!pickEarly) vox = SpatialVoxel(sc, isect.p);
                            funcInt = pickEarly ? funcIntE : sc.sp_func_int[vox];
                        }
                        PROBE(5)   // BSDF ctor + NumComponents + voxel lookup
                        Float ul = PIX ? smp.PixGet1D(sc) : us[0];
                        ubase = 1;
                        // Distribution1D::SampleDiscrete (core/sampling.h:90-100) / FindInterval (core/pbrt.h:398-411): `first` = the number of leading cdf
                        // entries <= u (the cdf is non-decreasing, so the reference's bisection finds exactly that count).  Spatial tables: through the
                        // voxel's guide word (SpatialPick); the one power / uniform table: up to 16 independent probes per round on the LDS copy.
                        int lightNum;
                        Float funcAt;
                        if (spatial) SpatialPick(sc, vox, ul, &lightNum, &funcAt, !PIX && pickEarly, guideE);
                        else {
                            const int size = (int)sc.n_lights + 1, first = CdfCountLE(cdf, size, ul);
                            lightNum = first - 1 < 0 ? 0 : (first - 1 > size - 2 ? size - 2 : first - 1);
                            funcAt = sc.light_func[lightNum];
                        }
                        Float selPdf = (funcInt > 0) ? funcAt / (funcInt * (int)sc.n_lights) : 0;
                        if (selPdf != 0) {
                            Float uL0, uL1, uS0, uS1;
                            if (PIX) { smp.PixGet2D(sc, &uL0, &uL1); smp.PixGet2D(sc, &uS0, &uS1); }
                            else { uL0 = us[1]; uL1 = us[2]; uS0 = us[3]; uS1 = us[4]; }
                            ubase = 5;
                            // ---- EstimateDirect (core/integrator.cpp:108-215), handleMedia=false, specular=false
                            const DevLight &light = sc.lights[lightNum];
                            const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
                            PROBE(6)   // light pick
                            LightSample ls = ENV ? SampleLiAny(GeomTables(sc), &light, isect.p, isect.pError, isect.n, uL0, uL1)
                                                 : SampleLi(GeomTables(sc), &light, isect.p, isect.pError, isect.n, uL0, uL1);
                            PROBE(7)   // SampleLi
                            Float lightPdf = ls.pdf, scatteringPdf = 0;
                            if (lightPdf > 0 && !ls.Li.IsBlack()) {
                                RGB f = bsdf.fPdf(isect.wo, ls.wi, bsdfFlags, &scatteringPdf) * AbsDot(ls.wi, isect.ns);
                                if (!f.IsBlack()) {
                                    RGB Ld;
                                    if (ls.delta) Ld = f * ls.Li / lightPdf;
                                    else {
                                        Float weight = PowerHeuristic(lightPdf, scatteringPdf);
                                        Ld = f * ls.Li * weight / lightPdf;
                                    }
                                    RGB c = beta * (Ld / selPdf);   // added by k_anyhit iff the shadow ray is unoccluded
                                    ps.nee[slot].sh_o = make_float4(ls.shadow.o.x, ls.shadow.o.y, ls.shadow.o.z, ls.shadow.tMax);
                                    ps.nee[slot].sh_d = make_float4(ls.shadow.d.x, ls.shadow.d.y, ls.shadow.d.z, 0);
                                    ps.nee[slot].sh_c = make_float4(c.r, c.g, c.b, 0);
                                    wantShadow = true;
                                }
                            }
                            PROBE(8)   // NEE f + Pdf + shadow ray store
                            if (!ls.delta) {   // BSDF-sampling half of the MIS estimator
                                V3 wi;
                                int sampledType;
                                RGB f = bsdf.Sample_f(isect.wo, &wi, uS0, uS1, &scatteringPdf, bsdfFlags, &sampledType);
                                f = f * AbsDot(wi, isect.ns);
                                PROBE(9)   // MIS Sample_f
                                bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
                                if (!f.IsBlack() && scatteringPdf > 0) {
                                    Float weight = 1;
                                    bool ok = true;
                                    if (!sampledSpecular) {
                                        lightPdf = ENV ? PdfLiAny(GeomTables(sc), &light, isect.p, isect.pError, isect.n, wi) : PdfLi(GeomTables(sc), &light, isect.p, isect.pError, isect.n, wi);
                                        if (lightPdf == 0) ok = false;
                                        else weight = PowerHeuristic(scatteringPdf, lightPdf);
                                    }
                                    if (ok) {
                                        V3 mo = OffsetRayOrigin(isect.p, isect.pError, isect.n, wi);   // it.SpawnRay(wi)
                                        RGB c = beta * ((f * weight / scatteringPdf) / selPdf);         // times Li, resolved by k_closest<1>
                                        ps.nee[slot].mi_o = make_float4(mo.x, mo.y, mo.z, 0);
                                        ps.nee[slot].mi_d = make_float4(wi.x, wi.y, wi.z, __uint_as_float((uint32_t)lightNum));
                                        ps.nee[slot].mi_c = make_float4(c.r, c.g, c.b, 0);
                                        wantMis = true;
                                    }
                                }
                            }
                        }
                    }
                    PROBE(10)   // MIS PdfLi + ray store
                    // ---- sample the BSDF for the next path segment (path.cpp:131-150)
                    V3 wo = -rd, wi;
                    Float pdf, u0, u1;
                    int flags;
                    if (PIX) smp.PixGet2D(sc, &u0, &u1);
                    else {
                        u0 = ubase == 0 ? us[0] : (ubase == 1 ? us[1] : us[5]);
                        u1 = ubase == 0 ? us[1] : (ubase == 1 ? us[2] : us[6]);
                        ui = ubase + 2;
                    }
                    RGB f = bsdf.Sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                    PROBE(11)   // path Sample_f
                    if (!(f.IsBlack() || pdf == 0.f)) {
                        beta = beta * (f * AbsDot(wi, isect.ns) / pdf);
                        specularBounce = (flags & BSDF_SPECULAR) != 0;
                        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                            Float eta = bsdf.m->eta;
                            etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                        }
                        V3 no = OffsetRayOrigin(isect.p, isect.pError, isect.n, wi);   // isect.SpawnRay(wi)
                        cont = true;
                        // Russian roulette (path.cpp:176-184)
                        RGB rrBeta = beta * etaScale;
                        if (rrBeta.MaxComponentValue() < sc.rr_threshold && bounces > 3) {
                            Float q = mx((Float).05, 1 - rrBeta.MaxComponentValue());
                            Float urr;
                            if (PIX) urr = smp.PixGet1D(sc);
                            else { urr = ubase == 0 ? us[2] : (ubase == 1 ? us[3] : us[7]); ++ui; }
                            if (urr < q) cont = false;
                            else beta = beta / (1 - q);
                        }
                        if (cont) {
                            ps.rec[slot].ray_o = make_float4(no.x, no.y, no.z, PT_INFINITY);
                            ps.rec[slot].ray_d = make_float4(wi.x, wi.y, wi.z, 0);
                            ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                            ++bounces;
                        }
                    }
                    }   // matIdx == matU
                    }   // material waterfall
                }
            }
            PROBE(12)   // RR + record stores
            ps.rec[slot].L = make_float4(L.r, L.g, L.b, 0);
            if (cont) ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)(smp.dimension + ui), (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17));
        }
        uint32_t posE, posS, posM;
        const uint32_t qseg = blockIdx.x & 7, qbase = qseg * ps.seg_cap;   // this block class's segment of the three queues
        PT_WAVE_APPEND3(&ps.qcount[QCI(qout, qseg)], &ps.qcount[QCI(QC_SHADOW, qseg)], &ps.qcount[QCI(QC_MIS, qseg)], cont, wantShadow, wantMis, &posE, &posS, &posM);
        if (cont) ps.q_ext[qout][qbase + posE] = slot;
        if (wantShadow) ps.q_shadow[qbase + posS] = slot;
        if (wantMis) ps.q_mis[qbase + posM] = slot;
        PROBE(13)   // L store + queue appends
    }
    wave_count(&ps.counters[MI_CNT_PATH_SEGMENTS], nseg);
#if PT_SHADE_PROF
    if ((threadIdx.x & 63) < 24 && s_pcnt[threadIdx.x >> 6][threadIdx.x & 63]) {
        atomicAdd(&ps.counters[16 + (threadIdx.x & 63)], s_pacc[threadIdx.x >> 6][threadIdx.x & 63]);
        atomicAdd(&ps.counters[40 + (threadIdx.x & 63)], s_pcnt[threadIdx.x >> 6][threadIdx.x & 63]);
        atomicAdd(&ps.counters[68 + (threadIdx.x & 63)], s_plan[threadIdx.x >> 6][threadIdx.x & 63]);
    }
#endif
}

#include "pt_volpath.h"   // k_shade_vol: the shading kernel of "volpath" scenes and of scenes with BSSRDF materials

// ---- film: the radiance guards of integrator.cpp:294-315 + FilmTile::AddSample (core/film.h:121-161).
// One lane per owned pixel walks that pixel's samples of the pass in sample order.
//   SPILL == false: only the lane's own pixel, as a plain running sum continued from the film value -- the
//                   reference's accumulation order for a pixel's own samples, no atomics;
//   SPILL == true : every other footprint pixel (wide filters; with the box filter only a sample that lands
//                   exactly on a pixel edge) through atomicAdd, in a second launch so that it cannot race
//                   with the plain stores of the first.
// (round 6) The second launch used to walk every sample of the pass again -- a whole 128-byte path record fetched for its 8 bytes of pFilm, to find that under the box
// filter nothing spills: half of the film time.  Now the first launch LISTS the samples whose footprint holds other pixels (PathState::q_sorted is free once the pass
// has been shaded; counter row QC_BINNED, zeroed at the start of the pass) and the second launch walks that list only.
struct FilmFootprint { int p0x, p0y, p1x, p1y; Float dx, dy; };
PT_DEV FilmFootprint FilmFoot(const DevScene &sc, float2 pf) {   // FilmTile::AddSample core/film.h:131-140, clipped to the crop window
    FilmFootprint f;
    f.dx = pf.x - 0.5f; f.dy = pf.y - 0.5f;
    f.p0x = (int)__builtin_ceilf(f.dx - sc.filter_radius[0]); f.p0y = (int)__builtin_ceilf(f.dy - sc.filter_radius[1]);
    f.p1x = (int)__builtin_floorf(f.dx + sc.filter_radius[0]) + 1; f.p1y = (int)__builtin_floorf(f.dy + sc.filter_radius[1]) + 1;
    f.p0x = f.p0x > sc.crop_min[0] ? f.p0x : sc.crop_min[0]; f.p0y = f.p0y > sc.crop_min[1] ? f.p0y : sc.crop_min[1];
    f.p1x = f.p1x < sc.crop_max[0] ? f.p1x : sc.crop_max[0]; f.p1y = f.p1y < sc.crop_max[1] ? f.p1y : sc.crop_max[1];
    return f;
}
PT_DEV RGB FilmRadiance(const DevScene &sc, float4 L4) {   // the guards of integrator.cpp:294-315 and the maxSampleLuminance clamp (film.h:128-130)
    RGB L(L4.x, L4.y, L4.z);
    if (L.HasNaNs()) L = RGB(0.f);
    else if ((double)L.y() < -1e-5) L = RGB(0.f);
    else if (__builtin_isinf(L.y())) L = RGB(0.f);
    if (L.y() > sc.max_sample_luminance) L = L * (sc.max_sample_luminance / L.y());
    return L;
}
// the footprint pixels of one sample: OWN = only the lane's own pixel into `acc` (plain sum), else every other pixel through atomicAdd
template <bool OWN>
PT_DEV void FilmSplat(const DevScene &sc, const FilmFootprint &fp, const RGB &L, int ownx, int owny, float4 *film, int cw, float4 *acc) {
    const int W = MI_FILTER_TABLE_WIDTH;
    Float invRx = 1 / sc.filter_radius[0], invRy = 1 / sc.filter_radius[1];
    for (int y = fp.p0y; y < fp.p1y; ++y) {
        Float fy = absf((y - fp.dy) * invRy * W);
        int iy = mni((int)__builtin_floorf(fy), W - 1);
        for (int x = fp.p0x; x < fp.p1x; ++x) {
            bool isOwn = x == ownx && y == owny;
            if (isOwn != OWN) continue;
            Float fx = absf((x - fp.dx) * invRx * W);
            int ix = mni((int)__builtin_floorf(fx), W - 1);
            Float fw = sc.filter_table[iy * W + ix];
            RGB c = L * 1.f * fw;   // L * sampleWeight * filterWeight
            if (OWN) { acc->x += c.r; acc->y += c.g; acc->z += c.b; acc->w += fw; }
            else {
                float *o = reinterpret_cast<float *>(&film[(size_t)(y - sc.crop_min[1]) * cw + (x - sc.crop_min[0])]);
                atomicAdd(o, c.r); atomicAdd(o + 1, c.g); atomicAdd(o + 2, c.b); atomicAdd(o + 3, fw);
            }
        }
    }
}
template <bool SPILL>
__global__ void __launch_bounds__(PT_BLOCK) k_film(DevScene sc, PathState ps, PassInfo pass, float4 *film) {
    int cw = sc.crop_max[0] - sc.crop_min[0];
    if constexpr (SPILL) {   // the listed samples: every footprint pixel but the own one
        const uint32_t count = ps.qcount[QCI(QC_BINNED, 0)];
        for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < count; i += gridDim.x * PT_BLOCK) {
            const uint32_t slot = ps.q_sorted[i];
            const uint32_t pix = ps.rec[slot].pixel;
            const int ownx = sc.sample_min[0] + (int)(pix & 0xffffu), owny = sc.sample_min[1] + (int)(pix >> 16);
            const FilmFootprint fp = FilmFoot(sc, ps.rec[slot].pfilm);
            FilmSplat<false>(sc, fp, FilmRadiance(sc, ps.rec[slot].L), ownx, owny, film, cw, nullptr);
        }
        return;
    }
    for (ChunkIter it(pass.npix); it.more(); it.next()) {
        uint32_t p = it.item();
        if (p >= pass.npix) continue;
        uint32_t pix = ps.rec[p].pixel;
        if (pix == INACTIVE_PIXEL) continue;
        int ownx = sc.sample_min[0] + (int)(pix & 0xffffu), owny = sc.sample_min[1] + (int)(pix >> 16);
        bool ownInside = ownx >= sc.crop_min[0] && ownx < sc.crop_max[0] && owny >= sc.crop_min[1] && owny < sc.crop_max[1];
        float4 *own = &film[(size_t)(owny - sc.crop_min[1]) * cw + (ownx - sc.crop_min[0])];
        float4 acc = ownInside ? *own : make_float4(0, 0, 0, 0);
        for (uint32_t s = 0; s < pass.ns; ++s) {
            uint32_t slot = s * pass.npix + p;
            const FilmFootprint fp = FilmFoot(sc, ps.rec[slot].pfilm);
            const bool any = fp.p0x < fp.p1x && fp.p0y < fp.p1y;
            const bool others = any && !(fp.p0x == ownx && fp.p1x == ownx + 1 && fp.p0y == owny && fp.p1y == owny + 1);   // pixels besides the own one: the second launch's
            const uint32_t pos = wave_append(&ps.qcount[QCI(QC_BINNED, 0)], others);
            if (others) ps.q_sorted[pos] = slot;
            if (any) FilmSplat<true>(sc, fp, FilmRadiance(sc, ps.rec[slot].L), ownx, owny, film, cw, &acc);
        }
        if (ownInside) *own = acc;
    }
}

// ---- stage-level kernels (parity tests): one lane per input record
// mi_intersect / mi_intersect_p run the SHIPPED traversal kernels (k_trace<0> / k_trace<2>, whichever instance the uploaded scene selects):
// k_stage_fill_rays writes the rays into the path records and the segmented queue, k_stage_collect_hits turns PathRec::hit back into mi_hit
// (barycentrics and normal recomputed from the hit primitive, as k_shade does).
__global__ void __launch_bounds__(PT_BLOCK) k_stage_fill_rays(PathState ps, const mi_ray *rays, int64_t n, int anyHit) {
    for (int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT_BLOCK) {
        const mi_ray r = rays[i];
        const float4 o4 = make_float4(r.o[0], r.o[1], r.o[2], r.tmax), d4 = make_float4(r.d[0], r.d[1], r.d[2], 0);
        ps.rec[i].L = make_float4(0, 0, 0, 0);
        ps.rec[i].hit = make_uint2(TRAV_MISS, 0u);
        ps.rec[i].pad0 = TRAV_NO_INSTANCE;
        if (anyHit) { ps.nee[i].sh_o = o4; ps.nee[i].sh_d = d4; ps.nee[i].sh_c = make_float4(1, 1, 1, 0); }
        else { ps.rec[i].ray_o = o4; ps.rec[i].ray_d = d4; }
        const uint32_t seg = (uint32_t)(i & 7), pos = (uint32_t)(i >> 3);
        (anyHit ? ps.q_shadow : ps.q_ext[0])[(size_t)seg * ps.seg_cap + pos] = (uint32_t)i;
    }
}
__global__ void __launch_bounds__(PT_BLOCK) k_stage_collect_hits(DevScene sc, PathState ps, const mi_ray *rays, int64_t n, mi_hit *hits, uint8_t *occluded) {
    for (int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT_BLOCK) {
        if (occluded) { occluded[i] = ps.rec[i].L.x == 0 ? 1 : 0; continue; }   // k_trace<2> adds sh_c = 1 to L when the segment is unoccluded
        mi_hit h;
        h.prim = -1; h.t = 0; h.b0 = h.b1 = h.b2 = 0; h.n[0] = h.n[1] = h.n[2] = 0;
        const uint2 hr = ps.rec[i].hit;
        if (hr.x != TRAV_MISS) {
            const uint32_t prim = hr.x;
            h.prim = (int32_t)prim; h.t = __uint_as_float(hr.y);
            if (ps.rec[i].pad0 == TRAV_NO_INSTANCE) {   // (hits inside an instance: primitive and distance only)
                const mi_ray r = rays[i];
                V3 o(r.o[0], r.o[1], r.o[2]), d(r.d[0], r.d[1], r.d[2]);
                V3 p0, p1, p2;
                uint32_t tf;
                LoadTri(sc, prim, &p0, &p1, &p2, &tf);
                TriHit th;
                Isect is;
                if (tf & TRI_FLAG_SPHERE) { is = SphereIsectToIsect(sc.spheres + __float_as_uint(p0.x), o, d, prim); th.t = h.t; th.b0 = th.b1 = th.b2 = 0; }
                else {
                    TriangleTest(p0, p1, p2, o, d, PT_INFINITY, &th);
                    BuildIsect(GeomTables(sc), prim, p0, p1, p2, th, d, &is);
                }
                h.t = th.t; h.b0 = th.b0; h.b1 = th.b1; h.b2 = th.b2;
                h.n[0] = is.n.x; h.n[1] = is.n.y; h.n[2] = is.n.z;
            }
        }
        hits[i] = h;
    }
}
__global__ void k_stage_triangles(const float *tri9, const mi_ray *rays, int64_t n, mi_hit *hits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = tri9 + 9 * i;
    V3 p0(t[0], t[1], t[2]), p1(t[3], t[4], t[5]), p2(t[6], t[7], t[8]);
    mi_ray r = rays[i];
    V3 o(r.o[0], r.o[1], r.o[2]), d(r.d[0], r.d[1], r.d[2]);
    mi_hit h;
    h.prim = -1; h.t = 0; h.b0 = h.b1 = h.b2 = 0; h.n[0] = h.n[1] = h.n[2] = 0;
    TriHit th;
    // + the per-triangle degeneracy rejection (triangle.cpp:308-315; default uvs) the scene path precomputes
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    V3 dpdu = (-1.f * dp02 - -1.f * dp12) * 1.f, dpdv = (-0.f * dp02 + -1.f * dp12) * 1.f;
    bool reject = Cross(dpdu, dpdv).LengthSquared() == 0 && Cross(p2 - p0, p1 - p0).LengthSquared() == 0;
    if (!reject && TriangleTest(p0, p1, p2, o, d, r.tmax, &th)) {
        h.prim = 0; h.t = th.t; h.b0 = th.b0; h.b1 = th.b1; h.b2 = th.b2;
        V3 n = Normalize(Cross(dp02, dp12));
        h.n[0] = n.x; h.n[1] = n.y; h.n[2] = n.z;
    }
    hits[i] = h;
}
__global__ void k_stage_sobol(DevScene sc, int px, int py, int n_samples, int n_dims, float *out, unsigned long long *index_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples * n_dims) return;
    int s = i / n_dims, d = i % n_dims;
    Sampler smp;
    smp.Start(sc, px, py, (uint64_t)s);
    out[i] = smp.SampleDimension(sc, d);
    if (d == 0 && index_out) index_out[s] = smp.index;
}
__global__ void k_stage_export_rays(PathState ps, int64_t n, mi_ray *rays, float *pfilm) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 o = ps.rec[i].ray_o, d = ps.rec[i].ray_d;
    mi_ray r;
    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.tmax = o.w; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.time = 0;
    rays[i] = r;
    float2 pf = ps.rec[i].pfilm;
    pfilm[2 * i] = pf.x; pfilm[2 * i + 1] = pf.y;
}
// camera-ray differentials as the shading kernels rebuild them at the first hit of a textured scene (CameraDifferentials, pt_material.h:
// PerspectiveCamera::GenerateRayDifferential's rx / ry, perspective.cpp:118-141, after RayDifferential::ScaleDifferentials(1 / sqrt(spp)),
// integrator.cpp:262-263): out[12 i ..] = rxOrigin, rxDirection, ryOrigin, ryDirection
__global__ void k_stage_export_diffs(PathState ps, mi_camera cam, int spp, int64_t n, float *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 o = ps.rec[i].ray_o, d = ps.rec[i].ray_d;
    float2 pf = ps.rec[i].pfilm, ln = ps.rec[i].lens;
    RayDiffT r = CameraDifferentials(&cam, pf.x, pf.y, ln.x, ln.y, spp, V3(o.x, o.y, o.z), V3(d.x, d.y, d.z));
    float *q = out + 12 * i;
    q[0] = r.rxO.x; q[1] = r.rxO.y; q[2] = r.rxO.z; q[3] = r.rxD.x; q[4] = r.rxD.y; q[5] = r.rxD.z;
    q[6] = r.ryO.x; q[7] = r.ryO.y; q[8] = r.ryO.z; q[9] = r.ryD.x; q[10] = r.ryD.y; q[11] = r.ryD.z;
}
__global__ void k_stage_export_L(PathState ps, int64_t n, float *L_rgb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 L4 = ps.rec[i].L;
    RGB L(L4.x, L4.y, L4.z);   // same guards as the film path (integrator.cpp:294-315)
    if (L.HasNaNs()) L = RGB(0.f);
    else if ((double)L.y() < -1e-5) L = RGB(0.f);
    else if (__builtin_isinf(L.y())) L = RGB(0.f);
    L_rgb[3 * i] = L.r; L_rgb[3 * i + 1] = L.g; L_rgb[3 * i + 2] = L.b;
}

// ============================================================================ host side
static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return -1; }
#define HIP_TRY(expr)                                                                                    \
    do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

struct DevBuf {   // owns one device allocation: released with the object, so error returns of the entry points do not leak
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <typename T> T *as() const { return (T *)p; }
};

struct mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // second stream for the direct-lighting traversals of bounce b, which overlap the path-extension traversal of bounce b + 1 (PBRT_AMD_OVERLAP, run_pass)
    hipStream_t stream2 = nullptr;
    hipEvent_t evShaded = nullptr, evNeeDone = nullptr;
    bool overlapNee = false;
    int numCUs = 256, gridBlocks = 1024, gridShade = 1024, gridShadeVol = 1024;
    double hotProbeShare = 0;   // share of the probe paths' node visits that fell on the nodes now in nodesq[0 .. n_hot)
    bool hasEnvMap = false, hasSpheres = false;
    bool hasTex = false, hasAlpha = false;   // textured materials / alpha-masked meshes (row f2)
    bool hasInst = false;                    // two-level scene (the host's default): the k_trace / k_shade / k_shade_vol INST instances
    const DevInstance *instPtr = nullptr;
    bool hasNullMat = false;                 // some mesh has no material (medium interfaces): paths may outlive max_depth + 1 wavefront iterations
    int tilesRank = -1, tilesWorld = -1;     // the tile list resident in `tiles` (re-uploaded only when the sharding changes)
    std::vector<uint32_t> tilesHost;         // kept alive: the upload is asynchronous
    size_t tilesCount = 0;
    bool useQ = false;                       // interior steps over the 64-byte quantised BVH4 nodes (pt_bvh4q.h): single-level scenes
    DevTex tex;                              // host copy of c_tex for this scene (device pointers)
    bool volTr = false;                      // ... with BSDF-less interfaces: the shadow / MIS queues are served by k_vol_tr (pt_volpath.h)
    bool volSplit = false;                   // ... with a grid medium (Tr draws sampler dimensions): split form, k_vol_continue samples the continuation after the walks (DevVol::tr_dims)
    uint32_t sssTail = 131072;               // walked BSSRDF probe chains: queue size below which the rest of the walk is one k_sss_probe_tail launch (PBRT_AMD_SSS_TAIL; 0: rounds to the end)
    bool sssWave = false;                    // BSSRDF materials under Integrator "path" in wavefront form: probe chains walked through the queues (k_sss_probe_step / k_sss_entry)
    bool sssLog = true;                      // k_sss_probe_tail lists the counted hits of a first walk instead of walking the chain again (measured in round 5, profiles/r05_a_*: 81.3 -> 89.1 Msamples/s at 16 spp, 133.1 at 64 spp; PBRT_AMD_SSS_LOG=0: walk twice, the A/B partner of the parity tests)
    bool plainTex = false;                   // ... some material WITHOUT a BSSRDF is textured (else the first part takes the untextured k_shade instances)
    bool trLean = true;                      // LAUNCH_TRACE_TR_SHADOW
    bool sssRoute = false;                   // ... under Integrator "path" with plain direct-lighting rays: only the vertices on BSSRDF materials go to k_shade_vol, the others to k_shade (PathState::key_remap; PBRT_AMD_SSS_ROUTE=0: k_shade_vol shades everything)
    const uint32_t *keyRemap = nullptr;
    struct ShadePart { bool vol; uint32_t keyLo, keyHi; };   // vol: k_shade_vol's part; keys [keyLo, keyHi) of the remapped key space, 0xffffffff = the end of the queue
    std::vector<ShadePart> shadeParts;       // the launches that shade the material-sorted queue, in key order
    bool volWave = false;                    // ... and its direct-lighting rays go through the shadow / MIS queues (k_shade_vol<true>; walked: volTr, grid media: volSplit, BSSRDF materials: sssWave)
    bool volKernel = false;                  // Integrator "volpath" or materials with a BSSRDF: k_shade_vol shades (row f4)
    DevVol vol;                              // its extra tables (device pointers)
    const DevScene *scDev = nullptr;         // DevScene in HBM: k_shade_vol's out-of-line routines take it by pointer
    DevScene sc;
    bool haveScene = false;
    std::vector<DevBuf> sceneBufs;
    DevBuf film, counters, tiles;
    float4 *filmPtr = nullptr;   // c->film.p or a caller-owned buffer (mi_film_bind)
    int64_t filmPixels = 0;
    // What the bound film holds (mi_film_gather's sparse form is exact only if it is ONE shard: zero outside that shard's reach): the sharding of the mi_render calls
    // since the last mi_film_clear, and whether a second sharding -- or a freshly bound buffer of unknown contents -- has been mixed in (then the gather adds the whole film).
    int filmShardRank = -1, filmShardWorld = -1;
    bool filmMixed = false;
    // mi_film_gather's per-sender state, kept across frames: the reach list of (rank, world) on this context's device and on the root's, the packed pixels, the root's
    // receive buffer.  Rebuilt only when the sharding, the film size or the root's device changes (round 5 rebuilt, allocated and uploaded all of it every frame).
    struct GatherPart {
        int rank = -1, world = -1, rootDevice = -1;
        int64_t pixels = 0;
        size_t n = 0;
        std::vector<uint32_t> idx;   // kept alive: the uploads are asynchronous
        DevBuf dIdxSrc, dIdxRoot, packed, recv;
    } gather;
    uint64_t gatherBuilds = 0;   // MI_CNT_FILM_GATHER_BUILDS
    // wavefront state
    PathState ps;
    std::vector<DevBuf> stateBufs;
    uint32_t cap = 0;
    uint32_t nkeys = 0;
    uint32_t *cursor2 = nullptr, *spill2 = nullptr;   // fetch cursors / stack spill slices of the kernels on stream2
    // timing
    bool timing = false;
    struct Ev { hipEvent_t a, b; int id; };
    std::vector<Ev> evPool;
    size_t evUsed = 0;
    double msTotal[MI_K_COUNT] = {0};
    uint64_t launches[MI_K_COUNT] = {0};
};

static int upload(mi_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    if (b.alloc(bytes)) return -1;
    if (bytes && src) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// ---- turns on the per-device __constant__ tables.  c_tex / c_instances exist once per DEVICE and every pass of a scene with textures,
// alpha masks or instances rewrites them on its own stream.  Contexts that share a device would overwrite each other's tables while
// kernels still read them, so such passes take turns: the turn's holder first drains the streams of the other table-using contexts of
// its device, and drains its own before it hands the turn on.  A context that is alone on its device only pays an uncontended lock
// and stays asynchronous.
namespace {
std::mutex g_ctxRegistryLock;
std::vector<mi_ctx *> g_ctxRegistry;   // live contexts (mi_ctx_create / mi_ctx_destroy)
std::mutex g_tableTurnLock[64];        // by device ordinal
struct TableTurn {
    mi_ctx *c;
    bool held = false, shared = false;
    explicit TableTurn(mi_ctx *ctx) : c(ctx) {
        if (!(c->hasTex || c->hasAlpha || c->hasInst)) return;
        g_tableTurnLock[c->device & 63].lock();
        held = true;
        std::lock_guard<std::mutex> g(g_ctxRegistryLock);
        for (mi_ctx *o : g_ctxRegistry)
            if (o != c && o->device == c->device && (o->hasTex || o->hasAlpha || o->hasInst)) {
                shared = true;
                (void)hipStreamSynchronize(o->stream);
                if (o->stream2) (void)hipStreamSynchronize(o->stream2);
            }
    }
    ~TableTurn() {
        if (!held) return;
        if (shared) { (void)hipStreamSynchronize(c->stream); if (c->stream2) (void)hipStreamSynchronize(c->stream2); }
        g_tableTurnLock[c->device & 63].unlock();
    }
    TableTurn(const TableTurn &) = delete;
    TableTurn &operator=(const TableTurn &) = delete;
};
}  // namespace

// ---- BVH2 -> BVH4 collapse (host).  Every BVH4 child box is a node box of the binary tree it is given: the reference's, or (single-level scenes, round 5) the library's
// own topology over the reference's leaves (pt_treebuild.h).
#include "pt_treebuild.h"
namespace {
struct B4Builder {
    bool ownTopology = false;            // rebuild the top-level tree's interior over the reference's leaves (treebuild::RebuildOverLeaves)
    bool rebuilt = false;                // ... and it was
    std::vector<mi_bvh2_node> own;
    const mi_bvh2_node *n2;
    uint32_t primBase = 0;   // added to every primitive offset: 0 for the top-level tree, mi_object::first_prim for an instanced object's
    std::vector<BVH4Node> out;
    int maxDepth = 0;
    static float area(const mi_bvh2_node &n) {
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    static void setChild(BVH4Node &nd, int k, const float bmin[3], const float bmax[3], uint32_t ref) {
        nd.lox[k] = bmin[0]; nd.loy[k] = bmin[1]; nd.loz[k] = bmin[2];
        nd.hix[k] = bmax[0]; nd.hiy[k] = bmax[1]; nd.hiz[k] = bmax[2];
        nd.child[k] = ref;
    }
    static void clearNode(BVH4Node &nd) {
        std::memset(&nd, 0, sizeof(nd));
        const float inf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 4; ++k) {   // empty slot = inverted infinite box: fails every interval test
            nd.child[k] = BVH4_EMPTY;
            nd.lox[k] = nd.loy[k] = nd.loz[k] = inf;
            nd.hix[k] = nd.hiy[k] = nd.hiz[k] = -inf;
        }
    }
    // a reference leaf with more than BVH4_LEAF_MAX triangles becomes a small chain of nodes with the leaf's box
    uint32_t leafRef(const mi_bvh2_node &lf, uint32_t first, uint32_t count, int depth) {
        if (count <= BVH4_LEAF_MAX) return BVH4_LEAF | ((count - 1) << 27) | first;
        uint32_t idx = (uint32_t)out.size();
        out.emplace_back();
        clearNode(out[idx]);
        maxDepth = std::max(maxDepth, depth + 1);
        uint32_t per = (count + 3) / 4;
        per = ((per + BVH4_LEAF_MAX - 1) / BVH4_LEAF_MAX) * BVH4_LEAF_MAX;
        for (int k = 0; k < 4 && count > 0; ++k) {
            uint32_t c = std::min(per, count);
            uint32_t ref = leafRef(lf, first, c, depth + 1);
            setChild(out[idx], k, lf.bmin, lf.bmax, ref);
            first += c; count -= c;
        }
        return idx;
    }
    uint32_t build(uint32_t i2, int depth) {   // i2: interior reference node
        uint32_t idx = (uint32_t)out.size();
        out.emplace_back();
        clearNode(out[idx]);
        maxDepth = std::max(maxDepth, depth);
        uint32_t kids[4];
        int nk = 2;
        kids[0] = i2 + 1; kids[1] = (uint32_t)n2[i2].offset;
        while (nk < 4) {   // open the interior child with the largest surface area
            int best = -1;
            float bestA = -1;
            for (int k = 0; k < nk; ++k)
                if (n2[kids[k]].n_prims == 0) { float a = area(n2[kids[k]]); if (a > bestA) { bestA = a; best = k; } }
            if (best < 0) break;
            uint32_t o = kids[best];
            for (int k = nk; k > best + 1; --k) kids[k] = kids[k - 1];
            kids[best] = o + 1; kids[best + 1] = (uint32_t)n2[o].offset;
            ++nk;
        }
        for (int k = 0; k < nk; ++k) {
            const mi_bvh2_node &c = n2[kids[k]];
            uint32_t ref = c.n_prims > 0 ? leafRef(c, primBase + (uint32_t)c.offset, c.n_prims, depth) : build(kids[k], depth + 1);
            setChild(out[idx], k, c.bmin, c.bmax, ref);
        }
        return idx;
    }
    // entries one lane's traversal stack can hold at once; two-level: + the rest of the leaf, the sentinel and the object's own tree
    static int stackNeed(int topDepth, int objDepth, bool twoLevel) { return 3 * (topDepth + 1) + 1 + (twoLevel ? 2 + 3 * (objDepth + 1) + 1 : 0); }
    // tree of one BVH2 (a single-leaf reference tree gets a one-child root node); returns the root's index in `out`
    uint32_t buildTree(const mi_bvh2_node *nodes, uint32_t base) {
        n2 = nodes; primBase = base; maxDepth = 0;
        if (n2[0].n_prims > 0) {
            uint32_t idx = (uint32_t)out.size();
            out.emplace_back();
            clearNode(out[idx]);
            uint32_t ref = leafRef(n2[0], primBase + (uint32_t)n2[0].offset, n2[0].n_prims, 0);
            setChild(out[idx], 0, n2[0].bmin, n2[0].bmax, ref);
            return idx;
        }
        return build(0, 0);
    }
    // what mi_scene_upload decides: single-level scenes traverse the library's own topology unless PBRT_AMD_TREE=reference (the host-side validators build the same tree)
    static bool wantOwnTopology(bool twoLevel) { const char *e = std::getenv("PBRT_AMD_TREE"); return !(e && e[0] == 'r') && !twoLevel; }
    // the top-level tree (root = node 0) and, for two-level scenes, one tree per instanced object behind it
    bool buildScene(const mi_scene_desc *d, std::vector<uint32_t> *objRoot, int *topDepth, int *objDepth, std::string *err) {
        *topDepth = *objDepth = 0;
        objRoot->clear();
        if (d->n_bvh_nodes) {
            rebuilt = ownTopology && !d->n_instances && treebuild::RebuildOverLeaves(d->bvh_nodes, d->n_bvh_nodes, &own);
            buildTree(rebuilt ? own.data() : d->bvh_nodes, 0);
            *topDepth = maxDepth;
        }
        if (!d->n_instances) return true;
        objRoot->resize(d->n_objects);
        for (uint32_t o = 0; o < d->n_objects; ++o) {
            const mi_object &ob = d->objects[o];
            if (!ob.n_nodes || (uint64_t)ob.first_prim + ob.n_prims > d->n_tris || ob.first_node < d->n_bvh_nodes) {
                *err = "bad object record";
                return false;
            }
            (*objRoot)[o] = buildTree(d->bvh_nodes + ob.first_node, ob.first_prim);
            *objDepth = std::max(*objDepth, maxDepth);
        }
        return true;
    }
};
}  // namespace

extern "C" {

const char *mi_last_error(void) { return g_err.c_str(); }
int mi_abi_version(void) { return MI_ABI_VERSION; }

int mi_ctx_create(int device_ordinal, void *stream, mi_ctx **out) {
    if (!out) return fail("mi_ctx_create: null out");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail("mi_ctx_create: no HIP device available (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail("mi_ctx_create: bad device ordinal");
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    mi_ctx *c = new mi_ctx;
    c->device = device_ordinal;
    if (stream) c->stream = (hipStream_t)stream;
    else {
        hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (es != hipSuccess) { delete c; return fail(std::string("mi_ctx_create: hipStreamCreateWithFlags: ") + hipGetErrorString(es)); }
        c->ownStream = true;
    }
    c->numCUs = prop.multiProcessorCount;
    c->gridBlocks = ((c->numCUs * PT_GRID_PER_CU + 7) / 8) * 8;   // multiple of 8 for the XCD mapping
    c->gridShade = ((c->numCUs * PT_SHADE_GRID_PER_CU + 7) / 8) * 8;   // k_shade: a multiple of what is resident at PT_SHADE_WAVES per SIMD
    c->gridShadeVol = ((c->numCUs * 12 + 7) / 8) * 8;   // k_shade_vol and its companions keep rounds 1-4's twelve blocks per CU: one round measured +6 % on the subsurface C3's shading (profiles/r05_v_*)
    std::memset(&c->sc, 0, sizeof(c->sc));
    std::memset(&c->ps, 0, sizeof(c->ps));
    if (c->counters.alloc(PT_CNT_ALLOC * sizeof(uint64_t))) { mi_ctx_destroy(c); return -1; }
    if (hipMemsetAsync(c->counters.p, 0, PT_CNT_ALLOC * sizeof(uint64_t), c->stream) != hipSuccess) { mi_ctx_destroy(c); return fail("mi_ctx_create: hipMemsetAsync failed"); }
    { std::lock_guard<std::mutex> g(g_ctxRegistryLock); g_ctxRegistry.push_back(c); }
    *out = c;
    return 0;
}

void mi_ctx_destroy(mi_ctx *c) {
    if (!c) return;
    { std::lock_guard<std::mutex> g(g_ctxRegistryLock); g_ctxRegistry.erase(std::remove(g_ctxRegistry.begin(), g_ctxRegistry.end(), c), g_ctxRegistry.end()); }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->sceneBufs) b.release();
    for (auto &b : c->stateBufs) b.release();
    c->film.release(); c->counters.release(); c->tiles.release();
    for (auto &e : c->evPool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->evShaded) (void)hipEventDestroy(c->evShaded);
    if (c->evNeeDone) (void)hipEventDestroy(c->evNeeDone);
    if (c->ownStream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mi_scene_upload(mi_ctx *c, const mi_scene_desc *d) {
    if (!c || !d) return fail("mi_scene_upload: null argument");
    if (d->abi_version != MI_ABI_VERSION) return fail("mi_scene_upload: ABI version mismatch");
    if (d->n_tris > BVH4_FIRST_MASK) return fail("mi_scene_upload: more than 2^27 triangles");
    if (d->n_tris && d->n_bvh_nodes == 0) return fail("mi_scene_upload: triangles without a BVH");
    // row f4: Integrator "volpath" and materials with a BSSRDF are shaded by k_shade_vol (pt_volpath.h)
    c->volKernel = d->integrator_type == MI_INTEGRATOR_VOLPATH || d->material_bssrdf != nullptr;
    c->volWave = c->volTr = c->sssWave = c->volSplit = false;
    if (d->integrator_type != MI_INTEGRATOR_PATH && d->integrator_type != MI_INTEGRATOR_VOLPATH) return fail("mi_scene_upload: unknown integrator type");
    if (d->n_media && (!d->media || (d->integrator_type == MI_INTEGRATOR_VOLPATH && d->camera_medium >= (int32_t)d->n_media))) return fail("mi_scene_upload: bad medium table");
    if (d->material_bssrdf && (!d->bssrdf_tables || !d->material_descs || !d->textures)) return fail("mi_scene_upload: BSSRDF materials without tables / material descriptions");
    {   // PBRT_AMD_OVERLAP=1: the shadow / MIS traversals of a bounce on a second stream, overlapping the next bounce's path-extension traversal
        const char *e = std::getenv("PBRT_AMD_OVERLAP");
        c->overlapNee = e && e[0] == '1';
        if (c->overlapNee && !c->stream2) {
            HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreate(&c->evShaded)); HIP_TRY(hipEventCreate(&c->evNeeDone));
        }
    }
    c->hasNullMat = false;
    for (uint32_t m = 0; m < d->n_meshes; ++m) c->hasNullMat |= d->meshes[m].material < 0;
    c->tilesRank = c->tilesWorld = -1;
    c->hasInst = d->n_instances > 0;   // two-level scenes (see TravStateI)
    if (c->hasInst && (!d->instances || !d->objects)) return fail("mi_scene_upload: instances without instance / object tables");
    // textures (row f2): validate the node table before anything is uploaded
    c->hasTex = c->hasAlpha = false;
    if (d->material_descs) for (uint32_t m = 0; m < d->n_materials; ++m) c->hasTex |= d->material_descs[m].textured != 0;
    if (d->mesh_alpha) for (uint32_t m = 0; m < 2 * d->n_meshes; ++m) c->hasAlpha |= d->mesh_alpha[m] >= 0;
    if (c->hasTex || c->hasAlpha) {
        if (!d->textures || !d->n_textures) return fail("mi_scene_upload: textured materials / alpha masks without a texture table");
        for (uint32_t i = 0; i < d->n_textures; ++i) {   // children precede parents (the evaluation programs below rely on it)
            const mi_texture &t = d->textures[i];
            const bool hasChildren = t.type == MI_TEX_SCALE || t.type == MI_TEX_MIX || t.type == MI_TEX_CHECKERBOARD || t.type == MI_TEX_DOTS;
            const int children[3] = {t.tex1, t.tex2, t.amount};
            for (int k = 0; k < (hasChildren ? (t.type == MI_TEX_MIX ? 3 : 2) : 0); ++k)
                if (children[k] < 0 || (uint32_t)children[k] >= i) return fail("mi_scene_upload: texture node refers to a later / missing node");
            if (t.type == MI_TEX_IMAGEMAP && (t.image < 0 || (uint32_t)t.image >= d->n_images || !d->images)) return fail("mi_scene_upload: imagemap without an image");
        }
        for (uint32_t i = 0; i < d->n_images; ++i)
            if (d->images[i].levels > 16 || d->images[i].levels < 1 || (d->images[i].channels != 1 && d->images[i].channels != 3)) return fail("mi_scene_upload: unsupported image pyramid");
        if (c->hasTex) {
            std::vector<int> mdepth(d->n_materials, 1);
            for (uint32_t m = 0; m < d->n_materials; ++m) {   // sub-materials of a mix have smaller indices (host/integrator.cpp materialIndex)
                const mi_material_desc &md = d->material_descs[m];
                if (md.type == MI_MAT_MIX && md.textured) {
                    if (md.m1 < 0 || md.m2 < 0 || (uint32_t)md.m1 >= m || (uint32_t)md.m2 >= m) return fail("mi_scene_upload: mix material refers to a later / missing material");
                    mdepth[m] = 1 + std::max(mdepth[md.m1], mdepth[md.m2]);
                    if (mdepth[m] > PT_MIX_MAX_DEPTH) return fail("mi_scene_upload: mix materials nested deeper than the device evaluates (PT_MIX_MAX_DEPTH)");
                }
            }
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    for (auto &b : c->sceneBufs) b.release();
    c->sceneBufs.clear();
    c->sceneBufs.resize(80 + 6 * (size_t)d->n_envmaps + (size_t)d->n_images + (size_t)d->n_media + 5 * (size_t)d->n_bssrdf_tables);
    int nb = 0;
    auto next = [&]() -> DevBuf & { return c->sceneBufs[nb++]; };
    DevScene &sc = c->sc;
    std::memset(&sc, 0, sizeof(sc));
    // BVH4
    B4Builder bb;
    int topDepth = 0, objDepth = 0;
    std::vector<uint32_t> objRoot;
    std::vector<DevInstance> insts;
    {
        // single-level scenes traverse the library's own topology over the reference's leaves (pt_treebuild.h); PBRT_AMD_TREE=reference keeps the tree as handed over
        // (A/B, and one of the parity suite's traversal modes)
        bb.ownTopology = B4Builder::wantOwnTopology(c->hasInst);
        std::string err;
        const auto t0 = std::chrono::steady_clock::now();
        if (!bb.buildScene(d, &objRoot, &topDepth, &objDepth, &err)) return fail("mi_scene_upload: " + err);
        if (std::getenv("PBRT_AMD_VERBOSE"))
            std::fprintf(stderr, "[pbrt_amd] BVH4 over %s: %zu nodes, depth %d, %.2f s\n", bb.rebuilt ? "the library's own topology (reference leaves)" : "the reference's tree",
                         bb.out.size(), topDepth, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (c->hasInst) {
        insts.resize(d->n_instances);
        for (uint32_t i = 0; i < d->n_instances; ++i) {
            const mi_instance &mi = d->instances[i];
            if (mi.object >= d->n_objects) return fail("mi_scene_upload: instance refers to a missing object");
            std::memset(&insts[i], 0, sizeof(DevInstance));
            std::memcpy(insts[i].w2i, mi.w2i, sizeof(mi.w2i)); std::memcpy(insts[i].i2w, mi.i2w, sizeof(mi.i2w));
            insts[i].root = objRoot[mi.object];
            bool identity = true;
            for (int k = 0; k < 16; ++k) if (mi.i2w[k] != ((k % 5 == 0) ? 1.f : 0.f)) identity = false;
            insts[i].identity = identity ? 1u : 0u;
        }
    }
    { DevBuf &b = next(); if (upload(c, b, bb.out.data(), bb.out.size() * sizeof(BVH4Node))) return -1; sc.nodes = b.as<BVH4Node>(); }
    sc.n_nodes = (uint32_t)bb.out.size();
    sc.stack_need = B4Builder::stackNeed(topDepth, objDepth, c->hasInst);
    c->instPtr = nullptr;
    if (c->hasInst) {
        DevBuf &b = next();
        if (upload(c, b, insts.data(), insts.size() * sizeof(DevInstance))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));   // insts is a local
        c->instPtr = b.as<DevInstance>();
    }
    // triangle records
    c->hasSpheres = false;
    std::vector<float4> tv(3 * (size_t)d->n_tris);
    for (uint32_t t = 0; t < d->n_tris; ++t) {
        const uint32_t *v = d->tri_indices + 3 * (size_t)t;
        if (v[0] == MI_PRIM_INSTANCE) {   // a TransformedPrimitive of a two-level scene: the record carries the instance index and the flag
            if (v[1] >= d->n_instances) return fail("mi_scene_upload: primitive refers to a missing instance");
            uint32_t fl = TRI_FLAG_INSTANCE;
            float flf, idf;
            std::memcpy(&flf, &fl, 4); std::memcpy(&idf, &v[1], 4);
            tv[3 * (size_t)t] = make_float4(idf, 0, 0, flf);
            tv[3 * (size_t)t + 1] = make_float4(0, 0, 0, 0);
            tv[3 * (size_t)t + 2] = make_float4(0, 0, 0, 0);
            continue;
        }
        if (v[0] == MI_PRIM_SPHERE) {   // a Sphere primitive: the record carries its index and the flag
            if (v[1] >= d->n_spheres || !d->spheres) return fail("mi_scene_upload: primitive refers to a missing sphere");
            uint32_t fl = TRI_FLAG_SPHERE;
            float flf, idf;
            std::memcpy(&flf, &fl, 4); std::memcpy(&idf, &v[1], 4);
            tv[3 * (size_t)t] = make_float4(idf, 0, 0, flf);
            tv[3 * (size_t)t + 1] = make_float4(0, 0, 0, 0);
            tv[3 * (size_t)t + 2] = make_float4(0, 0, 0, 0);
            c->hasSpheres = true;
            continue;
        }
        if (v[0] >= d->n_verts || v[1] >= d->n_verts || v[2] >= d->n_verts) return fail("mi_scene_upload: vertex index out of range");
        const float *p0 = d->P + 3 * (size_t)v[0], *p1 = d->P + 3 * (size_t)v[1], *p2 = d->P + 3 * (size_t)v[2];
        // per-triangle rejection of shapes/triangle.cpp:300-315, evaluated with the reference's arithmetic
        uint32_t mflags = d->meshes[d->tri_mesh[t]].flags;
        float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
        if (d->UV && (mflags & MI_MESH_HAS_UV)) for (int k = 0; k < 3; ++k) { uv[k][0] = d->UV[2 * (size_t)v[k]]; uv[k][1] = d->UV[2 * (size_t)v[k] + 1]; }
        float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
        float dp02[3], dp12[3];
        for (int k = 0; k < 3; ++k) { dp02[k] = p0[k] - p2[k]; dp12[k] = p1[k] - p2[k]; }
        float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerateUV = std::abs(determinant) < 1e-8;
        auto crossLen2 = [](const float a[3], const float b[3]) {
            double ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
            float cx = (float)((ay * bz) - (az * by)), cy = (float)((az * bx) - (ax * bz)), cz = (float)((ax * by) - (ay * bx));
            return cx * cx + cy * cy + cz * cz;
        };
        bool reject = false;
        bool needNg = degenerateUV;
        if (!degenerateUV) {
            float invdet = 1 / determinant, dpdu[3], dpdv[3];
            for (int k = 0; k < 3; ++k) {
                dpdu[k] = (duv12[1] * dp02[k] - duv02[1] * dp12[k]) * invdet;
                dpdv[k] = (-duv12[0] * dp02[k] + duv02[0] * dp12[k]) * invdet;
            }
            if (crossLen2(dpdu, dpdv) == 0) needNg = true;
        }
        if (needNg) {
            float a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = p2[k] - p0[k]; b[k] = p1[k] - p0[k]; }
            if (crossLen2(a, b) == 0) reject = true;
        }
        uint32_t fl = reject ? TRI_FLAG_REJECT : 0u;
        if (c->hasAlpha && (d->mesh_alpha[2 * (size_t)d->tri_mesh[t]] >= 0 || d->mesh_alpha[2 * (size_t)d->tri_mesh[t] + 1] >= 0)) fl |= TRI_FLAG_ALPHA;
        float flf;
        std::memcpy(&flf, &fl, 4);
        tv[3 * (size_t)t] = make_float4(p0[0], p0[1], p0[2], flf);
        tv[3 * (size_t)t + 1] = make_float4(p1[0], p1[1], p1[2], 0);
        tv[3 * (size_t)t + 2] = make_float4(p2[0], p2[1], p2[2], 0);
    }
    { DevBuf &b = next(); if (upload(c, b, tv.data(), tv.size() * sizeof(float4))) return -1; sc.tri_verts = b.as<float4>(); }
    {   // Traversal layout: the general steps over the 64-byte quantised BVH4 (pt_bvh4q.h; 4 vector-memory requests per interior step) for
        // every single-level scene -- spheres and alpha masks only touch the leaf step; the full-precision 128-byte nodes (7 requests per
        // step) serve two-level scenes and PBRT_AMD_TRACE=general.  Round 2 measured both and three more layouts on the 10 M-triangle frame
        // (lean BVH4, 80-byte compressed BVH8, 128-byte BVH8: all slower, profiles/r02_a_*, r02_b_*; removed from the library in round 3).
        const char *e = std::getenv("PBRT_AMD_TRACE");
        const bool wantGeneral = e && std::strcmp(e, "general") == 0;
        c->useQ = !wantGeneral && !c->hasInst && d->n_bvh_nodes > 0;
    }
    sc.nodesq = nullptr;
    DevBuf *qnBuf = nullptr;
    if (c->useQ) {
        std::vector<BVH4QNode> qn;
        std::string err;
        // A grid the folded test cannot serve (non-finite / inconsistent bounds, or cells so large that cell * 1e30 -- the slab constant of a
        // zero direction component -- overflows) is not an error: the scene takes the full-precision 128-byte nodes, which are always built.
        bool okq = bvh4q::quantise(bb.out, d->bvh_nodes[0].bmin, d->bvh_nodes[0].bmax, &qn, &sc.qgrid, &err);
        for (int a = 0; okq && a < 3; ++a) okq = std::isfinite(sc.qgrid.cell[a] * 1e30f * 65535.f) && std::isfinite(sc.qgrid.lo[a]);
        // TravNodeStepQ2 decides "child entered" by x >= e alone and relies on the INVERTED box of an empty slot to fail it, which holds for a ray that
        // starts within ~5e5 grid extents of the grid (pt_bvh4q.h, Bvh4qStepEX).  Path vertices lie inside the root box; the camera is checked here (1e5).
        for (int a = 0; okq && a < 3; ++a) {
            const double ext = 65535.0 * (double)sc.qgrid.cell[a], camPos = d->camera.camera_to_world[4 * a + 3];
            okq = std::fabs(camPos - (double)sc.qgrid.lo[a]) <= 1e5 * ext;
        }
        if (!okq) c->useQ = false;
        else {
            DevBuf &b = next();
            if (upload(c, b, qn.data(), qn.size() * sizeof(BVH4QNode))) return -1;
            HIP_TRY(hipStreamSynchronize(c->stream));   // local
            sc.nodesq = b.as<BVH4QNode>();
            qnBuf = &b;
        }
    }
    {   // per-triangle shading records (TriShade): vertex normals + uvs gathered through the index buffer
        std::vector<TriShade> tsd(d->n_tris);
        for (uint32_t t = 0; t < d->n_tris; ++t) {
            const uint32_t *v = d->tri_indices + 3 * (size_t)t;
            const mi_mesh &m = d->meshes[d->tri_mesh[t]];
            TriShade &r = tsd[t];
            std::memset(&r, 0, sizeof(r));
            if (v[0] == MI_PRIM_SPHERE || v[0] == MI_PRIM_INSTANCE) continue;
            if (d->N && (m.flags & MI_MESH_HAS_N))
                for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) r.n[3 * k + a] = d->N[3 * (size_t)v[k] + a];
            if (d->UV && (m.flags & MI_MESH_HAS_UV)) {
                for (int k = 0; k < 3; ++k) { r.uv[2 * k] = d->UV[2 * (size_t)v[k]]; r.uv[2 * k + 1] = d->UV[2 * (size_t)v[k] + 1]; }
            } else { r.uv[0] = 0; r.uv[1] = 0; r.uv[2] = 1; r.uv[3] = 0; r.uv[4] = 1; r.uv[5] = 1; }
        }
        DevBuf &b = next();
        if (upload(c, b, tsd.data(), tsd.size() * sizeof(TriShade))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        sc.tri_shade = b.as<TriShade>();
    }
    {
        std::vector<uint4> ti(d->n_tris);
        for (uint32_t t = 0; t < d->n_tris; ++t) {
            const mi_mesh &m = d->meshes[d->tri_mesh[t]];
            uint32_t mf = m.flags;
            if (!d->N) mf &= ~MI_MESH_HAS_N;
            if (!d->UV) mf &= ~MI_MESH_HAS_UV;
            ti[t] = make_uint4(mf, (uint32_t)m.material, (uint32_t)d->tri_light[t], d->tri_mesh[t]);
        }
        DevBuf &b = next();
        if (upload(c, b, ti.data(), ti.size() * sizeof(uint4))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        sc.tri_info = b.as<uint4>();
    }
    sc.tri_rec = nullptr;
    if (PT_TRI_REC && d->n_tris) {   // 128 bytes per triangle on top of the 128 of the three arrays (10 M triangles: 1.3 GB of 288)
        static_assert(sizeof(TriShade) == 64, "TriShade = four 16-byte words");
        DevBuf &b = next();
        if (b.alloc((size_t)d->n_tris * 128)) return -1;
        hipLaunchKernelGGL(k_build_tri_rec, dim3(c->numCUs * 4), dim3(PT_BLOCK), 0, c->stream, sc.tri_verts, sc.tri_shade, sc.tri_info, b.as<float4>(), d->n_tris);
        HIP_TRY(hipGetLastError());
        sc.tri_rec = b.as<float4>();
    }
    { DevBuf &b = next(); if (upload(c, b, d->materials, (size_t)d->n_materials * sizeof(mi_material))) return -1; sc.materials = b.as<mi_material>(); }
    {   // the BSDF's lobe header per material: count + the lobe types, 4 bits each (one scalar load instead of a chain through the lobe records)
        std::vector<uint2> pack(std::max<uint32_t>(1, d->n_materials));
        for (uint32_t m = 0; m < d->n_materials; ++m) {
            const mi_material &mm = d->materials[m];
            if (mm.n_bxdfs < 0 || mm.n_bxdfs > MI_MAX_BXDFS) return fail("mi_scene_upload: material with more than 8 BxDFs");
            uint32_t t = 0;
            for (int i = 0; i < mm.n_bxdfs; ++i) {
                if (mm.bxdfs[i].type < 0 || mm.bxdfs[i].type > 15) return fail("mi_scene_upload: unknown BxDF type");
                t |= (uint32_t)mm.bxdfs[i].type << (4 * i);
            }
            pack[m] = make_uint2((uint32_t)mm.n_bxdfs, t);
        }
        DevBuf &b = next();
        if (upload(c, b, pack.data(), pack.size() * sizeof(uint2))) return -1;
        sc.mat_pack = b.as<uint2>();
    }
    std::memset(&c->tex, 0, sizeof(c->tex));
    if (c->hasInst && !(c->hasTex || c->hasAlpha)) {   // the INST shading instance is the general (textured) one: give it "no textured material" descriptors
        DevTex &tx = c->tex;
        std::vector<mi_material_desc> descs(std::max<uint32_t>(1, d->n_materials));
        std::memset(descs.data(), 0xff, descs.size() * sizeof(mi_material_desc));
        for (auto &md : descs) { md.type = MI_MAT_MATTE; md.textured = 0; }
        DevBuf &b = next();
        if (upload(c, b, descs.data(), descs.size() * sizeof(mi_material_desc))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        tx.descs = b.as<mi_material_desc>();
        tx.camera = d->camera;
        tx.spp = d->integrator.spp;
    }
    if (c->hasTex || c->hasAlpha) {   // texture node table, image pyramids, material parameter nodes, alpha masks -> c_tex (set per render)
        DevTex &tx = c->tex;
        { DevBuf &b = next(); if (upload(c, b, d->textures, (size_t)d->n_textures * sizeof(mi_texture))) return -1; tx.nodes = b.as<mi_texture>(); }
        tx.n_nodes = d->n_textures; tx.n_images = d->n_images;
        std::vector<DevImage> imgs(d->n_images);
        for (uint32_t i = 0; i < d->n_images; ++i) {
            const mi_image &im = d->images[i];
            DevImage &di = imgs[i];
            std::memset(&di, 0, sizeof(di));
            di.width = im.width; di.height = im.height; di.levels = im.levels; di.channels = im.channels;
            di.trilinear = im.trilinear; di.wrap = im.wrap; di.max_aniso = im.max_aniso;
            size_t off = 0;
            for (int l = 0; l < im.levels; ++l) {
                di.level_off[l] = (uint32_t)off;
                off += (size_t)std::max(1, im.width >> l) * std::max(1, im.height >> l) * im.channels;
            }
            DevBuf &b = next();
            if (upload(c, b, im.texels, off * sizeof(float))) return -1;
            di.texels = b.as<float>();
        }
        { DevBuf &b = next(); if (upload(c, b, imgs.data(), imgs.size() * sizeof(DevImage))) return -1; tx.images = b.as<DevImage>(); }
        {   // flatten the graph below every node into its post-order evaluation program (pt_texture.h TexEval)
            std::vector<int32_t> progOff(d->n_textures + 1, 0);
            std::vector<int4> prog;
            for (uint32_t root = 0; root < d->n_textures; ++root) {
                progOff[root] = (int32_t)prog.size();
                std::vector<std::pair<int, int>> memo;   // node -> step (graphs are tiny)
                const size_t base = prog.size();
                std::function<int(int)> emit = [&](int n) -> int {
                    for (auto &kv : memo) if (kv.first == n) return kv.second;
                    const mi_texture &t = d->textures[n];
                    const bool hasChildren = t.type == MI_TEX_SCALE || t.type == MI_TEX_MIX || t.type == MI_TEX_CHECKERBOARD || t.type == MI_TEX_DOTS;
                    int c1 = hasChildren ? emit(t.tex1) : -1, c2 = hasChildren ? emit(t.tex2) : -1, c3 = t.type == MI_TEX_MIX ? emit(t.amount) : -1;
                    prog.push_back(make_int4(n, c1, c2, c3));
                    memo.push_back({n, (int)(prog.size() - base) - 1});
                    return memo.back().second;
                };
                emit((int)root);
                if (prog.size() - base > PT_TEX_MAX_PROG) return fail("mi_scene_upload: a texture refers to more than PT_TEX_MAX_PROG nodes");
            }
            progOff[d->n_textures] = (int32_t)prog.size();
            { DevBuf &b = next(); if (upload(c, b, progOff.data(), progOff.size() * sizeof(int32_t))) return -1; tx.prog_off = b.as<int32_t>(); }
            { DevBuf &b = next(); if (upload(c, b, prog.data(), prog.size() * sizeof(int4))) return -1; tx.prog = b.as<int4>(); }
            HIP_TRY(hipStreamSynchronize(c->stream));   // locals
        }
        if (!c->hasTex && c->hasInst) {   // masks but no textured material, two-level scene: the general shading instance still wants descriptors
            std::vector<mi_material_desc> descs(std::max<uint32_t>(1, d->n_materials));
            std::memset(descs.data(), 0xff, descs.size() * sizeof(mi_material_desc));
            for (auto &md : descs) { md.type = MI_MAT_MATTE; md.textured = 0; }
            DevBuf &b = next();
            if (upload(c, b, descs.data(), descs.size() * sizeof(mi_material_desc))) return -1;
            HIP_TRY(hipStreamSynchronize(c->stream));
            tx.descs = b.as<mi_material_desc>();
        }
        if (c->hasTex) { DevBuf &b = next(); if (upload(c, b, d->material_descs, (size_t)d->n_materials * sizeof(mi_material_desc))) return -1; tx.descs = b.as<mi_material_desc>(); }
        if (c->hasAlpha) { DevBuf &b = next(); if (upload(c, b, d->mesh_alpha, 2 * (size_t)d->n_meshes * sizeof(int32_t))) return -1; tx.mesh_alpha = b.as<int32_t>(); }
        std::vector<DevMaskFast> maskFast;
        if (c->hasAlpha) {   // the masks pre-resolved per mesh (DevMaskFast, pt_texture.h)
            maskFast.resize(2 * (size_t)d->n_meshes);
            auto isConst = [&](int n) { return n >= 0 && (uint32_t)n < d->n_textures && d->textures[n].type == MI_TEX_CONSTANT; };
            for (size_t k = 0; k < maskFast.size(); ++k) {
                DevMaskFast &mf = maskFast[k];
                std::memset(&mf, 0, sizeof(mf));
                const int n = d->mesh_alpha[k];
                if (n < 0 || (uint32_t)n >= d->n_textures) { mf.kind = n < 0 ? 0 : 1; continue; }   // (an index beyond the table: TexEval answers 0, as before)
                const mi_texture &t = d->textures[n];
                mf.kind = 1; mf.image = -1;
                mf.su = t.su; mf.sv = t.sv; mf.du = t.du; mf.dv = t.dv;
                if (t.type == MI_TEX_CONSTANT) { mf.kind = 2; mf.v_out = t.value[0]; }
                else if (t.type == MI_TEX_DOTS && t.mapping == MI_MAP_UV && isConst(t.tex1) && isConst(t.tex2)) { mf.kind = 3; mf.v_out = d->textures[t.tex1].value[0]; mf.v_in = d->textures[t.tex2].value[0]; }
                else if (t.type == MI_TEX_IMAGEMAP && t.mapping == MI_MAP_UV) { mf.kind = 4; mf.image = t.image; }
            }
            DevBuf &b = next();
            if (upload(c, b, maskFast.data(), maskFast.size() * sizeof(DevMaskFast))) return -1;
            tx.mask_fast = b.as<DevMaskFast>();
        }
        HIP_TRY(hipStreamSynchronize(c->stream));   // `imgs`, `maskFast` are locals
        tx.camera = d->camera;
        tx.spp = d->integrator.spp;
        for (int i = 0; i < 128; ++i) {   // MIPMap::weightLut, mipmap.h:187-195
            float alpha = 2;
            float r2 = float(i) / float(128 - 1);
            tx.ewa_lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
        }
    }
    if (d->n_spheres) { DevBuf &b = next(); if (upload(c, b, d->spheres, (size_t)d->n_spheres * sizeof(mi_sphere))) return -1; sc.spheres = b.as<mi_sphere>(); }
    {   // radiance maps of infinite lights
        std::vector<DevEnvMap> em(d->n_envmaps);
        for (uint32_t i = 0; i < d->n_envmaps; ++i) {
            const mi_envmap &m = d->envmaps[i];
            if (m.width < 1 || m.height < 1 || !m.rgb || !m.cond_func || !m.cond_cdf || !m.cond_func_int || !m.marg_func || !m.marg_cdf)
                return fail("mi_scene_upload: incomplete environment map");
            size_t w = (size_t)m.width, h = (size_t)m.height, nu = 2 * w, nv = 2 * h;
            std::memset(&em[i], 0, sizeof(DevEnvMap));
            em[i].width = m.width; em[i].height = m.height; em[i].marg_func_int = m.marg_func_int;
            if (c->sceneBufs.size() < (size_t)nb + 8) return fail("mi_scene_upload: too many environment maps");
            { DevBuf &b = next(); if (upload(c, b, m.rgb, 3 * w * h * 4)) return -1; em[i].rgb = b.as<float>(); }
            { DevBuf &b = next(); if (upload(c, b, m.cond_func, nu * nv * 4)) return -1; em[i].cond_func = b.as<float>(); }
            { DevBuf &b = next(); if (upload(c, b, m.cond_cdf, (nu + 1) * nv * 4)) return -1; em[i].cond_cdf = b.as<float>(); }
            { DevBuf &b = next(); if (upload(c, b, m.cond_func_int, nv * 4)) return -1; em[i].cond_func_int = b.as<float>(); }
            { DevBuf &b = next(); if (upload(c, b, m.marg_func, nv * 4)) return -1; em[i].marg_func = b.as<float>(); }
            { DevBuf &b = next(); if (upload(c, b, m.marg_cdf, (nv + 1) * 4)) return -1; em[i].marg_cdf = b.as<float>(); }
        }
        DevBuf &b = next();
        if (upload(c, b, em.data(), em.size() * sizeof(DevEnvMap))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        sc.envmaps = b.as<DevEnvMap>();
        c->hasEnvMap = d->n_envmaps > 0;
    }
    const DevEnvMap *envDev = sc.envmaps;
    {
        std::vector<DevLight> dl(d->n_lights);
        for (uint32_t i = 0; i < d->n_lights; ++i) {
            std::memset(&dl[i], 0, sizeof(DevLight));
            const mi_light &ml = d->lights[i];
            dl[i].type = ml.type; dl[i].tri = ml.tri; dl[i].two_sided = ml.two_sided;
            for (int k = 0; k < 3; ++k) { dl[i].L[k] = ml.L[k]; dl[i].pos[k] = ml.pos[k]; }
            dl[i].area = ml.area; dl[i].world_radius = ml.world_radius;
            if (ml.type == MI_LIGHT_INFINITE && ml.env_map) {
                if ((uint32_t)ml.env_map > d->n_envmaps) return fail("mi_scene_upload: light refers to a missing environment map");
                dl[i].ext = envDev + (ml.env_map - 1);
                for (int k = 0; k < 3; ++k) {
                    dl[i].p0[k] = ml.frame[k]; dl[i].p1[k] = ml.frame[3 + k]; dl[i].p2[k] = ml.frame[6 + k];
                    dl[i].l2w0[k] = ml.l2w[k]; dl[i].l2w1[k] = ml.l2w[3 + k]; dl[i].l2w2[k] = ml.l2w[6 + k];
                }
            }
            if (ml.type == MI_LIGHT_AREA_SPHERE) {
                if (ml.sphere < 0 || (uint32_t)ml.sphere >= d->n_spheres) return fail("mi_scene_upload: light refers to a missing sphere");
                dl[i].ext = sc.spheres + ml.sphere;
            }
            if (ml.type == MI_LIGHT_SPOT) {
                dl[i].cos_total = ml.cos_total_width; dl[i].cos_falloff = ml.cos_falloff_start;
                for (int k = 0; k < 3; ++k) { dl[i].p0[k] = ml.frame[k]; dl[i].p1[k] = ml.frame[3 + k]; dl[i].p2[k] = ml.frame[6 + k]; }
            }
            if (d->lights[i].type == MI_LIGHT_AREA_TRI) {
                uint32_t t = (uint32_t)d->lights[i].tri;
                const float4 *v = &tv[3 * (size_t)t];
                dl[i].p0[0] = v[0].x; dl[i].p0[1] = v[0].y; dl[i].p0[2] = v[0].z;
                dl[i].p1[0] = v[1].x; dl[i].p1[1] = v[1].y; dl[i].p1[2] = v[1].z;
                dl[i].p2[0] = v[2].x; dl[i].p2[1] = v[2].y; dl[i].p2[2] = v[2].z;
                uint32_t fl;
                std::memcpy(&fl, &v[0].w, 4);
                uint32_t mf = d->meshes[d->tri_mesh[t]].flags;
                if (!d->N) mf &= ~MI_MESH_HAS_N;
                if (!d->UV) mf &= ~MI_MESH_HAS_UV;
                dl[i].mesh_flags = mf | ((fl & TRI_FLAG_REJECT) ? 0x80000000u : 0u);
            }
        }
        DevBuf &b = next();
        if (upload(c, b, dl.data(), dl.size() * sizeof(DevLight))) return -1;
        HIP_TRY(hipStreamSynchronize(c->stream));
        sc.lights = b.as<DevLight>();
    }
    { DevBuf &b = next(); if (upload(c, b, d->light_func, (size_t)d->n_lights * 4)) return -1; sc.light_func = b.as<float>(); }
    { DevBuf &b = next(); if (upload(c, b, d->light_cdf, ((size_t)d->n_lights + 1) * 4)) return -1; sc.light_cdf = b.as<float>(); }
    // ---- SpatialLightDistribution (lightdistrib.cpp:96-126 sizes the grid; ComputeDistribution for all voxels on the device)
    for (int i = 0; i < 3; ++i) {   // scene bound for the ray bins
        sc.sp_bmin_all[i] = d->n_bvh_nodes ? d->bvh_nodes[0].bmin[i] : 0.f;
        sc.sp_bmax_all[i] = d->n_bvh_nodes ? d->bvh_nodes[0].bmax[i] : 0.f;
    }
    sc.light_strategy = MI_LIGHT_STRATEGY_TABLE;
    if (d->integrator.light_strategy == MI_LIGHT_STRATEGY_SPATIAL && d->n_lights > 1) {
        const int maxVoxels = d->integrator.spatial_max_voxels > 0 ? d->integrator.spatial_max_voxels : 64;
        float bmin[3], bmax[3];
        for (int i = 0; i < 3; ++i) {
            bmin[i] = d->n_bvh_nodes ? d->bvh_nodes[0].bmin[i] : std::numeric_limits<float>::max();      // Bounds3f() geometry.h:650-655
            bmax[i] = d->n_bvh_nodes ? d->bvh_nodes[0].bmax[i] : std::numeric_limits<float>::lowest();
        }
        float diag[3] = {bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2]};
        int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);   // Bounds3::MaximumExtent geometry.h:747-755
        float bm = diag[me];
        uint64_t nvox = 1;
        for (int i = 0; i < 3; ++i) {
            sc.sp_nvox[i] = std::max(1, int(std::round(diag[i] / bm * maxVoxels)));
            sc.sp_bmin[i] = bmin[i]; sc.sp_bmax[i] = bmax[i];
            nvox *= (uint64_t)sc.sp_nvox[i];
        }
        uint64_t nl = d->n_lights, bytes = nvox * (2 * nl + 2) * 4;
        if (nvox >= (1ull << 31) || bytes > (64ull << 30))
            return fail("mi_scene_upload: spatial light distribution table would need " + std::to_string(bytes >> 30) + " GiB");
        // the Halton points of ComputeDistribution: RadicalInverse(0..4, i) (core/lowdiscrepancy.cpp:389-403, 427-445)
        std::vector<float> halton(PT_SPATIAL_SAMPLES * 5);
        {
            static const int primes[5] = {2, 3, 5, 7, 11};
            for (int i = 0; i < PT_SPATIAL_SAMPLES; ++i)
                for (int b = 0; b < 5; ++b) {
                    uint64_t a = (uint64_t)i;
                    float v;
                    if (b == 0) {
                        uint64_t r = 0;
                        for (int k = 0; k < 64; ++k) if (a & (1ull << k)) r |= 1ull << (63 - k);   // ReverseBits64
                        v = (float)((double)r * 0x1p-64);
                    } else {
                        const float invBase = 1.f / (float)primes[b];
                        uint64_t rev = 0;
                        float invBaseN = 1;
                        while (a) {
                            uint64_t nx = a / (uint64_t)primes[b], digit = a - nx * (uint64_t)primes[b];
                            rev = rev * (uint64_t)primes[b] + digit;
                            invBaseN *= invBase;
                            a = nx;
                        }
                        v = std::min((float)rev * invBaseN, 0x1.fffffep-1f);
                    }
                    halton[5 * i + b] = v;
                }
        }
        DevBuf hbuf;
        if (upload(c, hbuf, halton.data(), halton.size() * 4)) return -1;
        DevBuf &bf = next(), &bc = next(), &bi = next();
        if (bf.alloc(nvox * nl * 4) || bc.alloc(nvox * (nl + 1) * 4) || bi.alloc(nvox * 4)) return -1;
        sc.sp_func = bf.as<float>(); sc.sp_cdf = bc.as<float>(); sc.sp_func_int = bi.as<float>();
        sc.n_lights = d->n_lights;   // the kernels below read lights / n_lights / N / tri_indices, all set above
        uint64_t total = nvox * nl;
        unsigned gridc = (unsigned)std::min<uint64_t>((total + PT_BLOCK - 1) / PT_BLOCK, 65536);
        hipLaunchKernelGGL(k_spatial_contrib, dim3(gridc), dim3(PT_BLOCK), 0, c->stream, sc, bf.as<float>(), hbuf.as<float>(), total);
        unsigned gridv = (unsigned)std::min<uint64_t>((nvox + PT_BLOCK - 1) / PT_BLOCK, 65536);
        hipLaunchKernelGGL(k_spatial_cdf, dim3(gridv), dim3(PT_BLOCK), 0, c->stream, (uint32_t)nvox, (uint32_t)nl, bf.as<float>(), bc.as<float>(), bi.as<float>());
        {   // guide table in front of the cdf search (SpatialPick): M = the power of two >= n_lights, at most 256 cells per voxel; PBRT_AMD_LIGHT_GUIDE=0: none
            const char *eg = std::getenv("PBRT_AMD_LIGHT_GUIDE");
            uint32_t M = 1;
            while (M < nl && M < 256) M <<= 1;
            // only where the search needs more than one round of 16 probes: with a handful of lights the guide word is one dependent fetch MORE
            // (measured: k_shade +2..4 % on the one-light / few-light frames C2 and C4, -0.9 % on C3's 130 lights; profiles/r03_d_*)
            if (!(eg && eg[0] == '0') && nl >= 32 && nl + 1 <= 65535 && nvox * M * 4 <= (8ull << 30)) {
                DevBuf &bg = next();
                if (bg.alloc(nvox * M * 4)) return -1;
                hipLaunchKernelGGL(k_spatial_guide, dim3(gridv), dim3(PT_BLOCK), 0, c->stream, (uint32_t)nvox, (uint32_t)nl, M, bc.as<float>(), bg.as<uint32_t>());
                sc.sp_guide = bg.as<uint32_t>(); sc.sp_guide_m = M;
            }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        hbuf.release();
        sc.light_strategy = MI_LIGHT_STRATEGY_SPATIAL;
    }
    { DevBuf &b = next(); if (upload(c, b, d->film.filter_table, sizeof(d->film.filter_table))) return -1; sc.filter_table = b.as<float>(); }
    std::vector<int32_t> inf;
    for (uint32_t i = 0; i < d->n_lights; ++i) if (d->lights[i].type == MI_LIGHT_INFINITE) inf.push_back((int32_t)i);
    { DevBuf &b = next(); if (upload(c, b, inf.data(), inf.size() * 4)) return -1; sc.infinite_lights = b.as<int32_t>(); }
    { DevBuf &b = next(); if (upload(c, b, kSobolMatrices32, sizeof(kSobolMatrices32))) return -1; sc.sobol32 = b.as<uint32_t>(); }
    {
        std::vector<uint32_t> T((size_t)PBRT_AMD_SOBOL_NCOL * PT_SOBOLT_STRIDE, 0u);
        for (int dmn = 0; dmn < PBRT_AMD_SOBOL_NDIM; ++dmn)
            for (int i = 0; i < PBRT_AMD_SOBOL_NCOL; ++i) T[(size_t)i * PT_SOBOLT_STRIDE + dmn] = kSobolMatrices32[dmn * PBRT_AMD_SOBOL_NCOL + i];
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_sobolT), T.data(), T.size() * 4));
    }
    { DevBuf &b = next(); if (upload(c, b, kVdCSobolMatrices, sizeof(kVdCSobolMatrices))) return -1; sc.vdc = b.as<uint64_t>(); }
    { DevBuf &b = next(); if (upload(c, b, kVdCSobolMatricesInv, sizeof(kVdCSobolMatricesInv))) return -1; sc.vdc_inv = b.as<uint64_t>(); }
    sc.n_infinite = (uint32_t)inf.size();
    sc.light_func_int = d->light_func_int;
    sc.n_tris = d->n_tris; sc.n_lights = d->n_lights; sc.n_materials = d->n_materials;
    sc.camera = d->camera;
    for (int i = 0; i < 2; ++i) {
        sc.full_res[i] = d->film.full_res[i]; sc.crop_min[i] = d->film.crop_min[i]; sc.crop_max[i] = d->film.crop_max[i];
        sc.sample_min[i] = d->film.sample_min[i]; sc.sample_max[i] = d->film.sample_max[i];
        sc.pixel_min[i] = d->integrator.pixel_min[i]; sc.pixel_max[i] = d->integrator.pixel_max[i];
        sc.filter_radius[i] = d->film.filter_radius[i];
    }
    sc.max_sample_luminance = d->film.max_sample_luminance;
    sc.max_depth = d->integrator.max_depth; sc.spp = d->integrator.spp;
    sc.sobol_resolution = d->integrator.sobol_resolution; sc.sobol_log2_resolution = d->integrator.sobol_log2_resolution;
    sc.rr_threshold = d->integrator.rr_threshold;
    sc.sampler_type = d->integrator.sampler;
    if (sc.sampler_type < MI_SAMPLER_SOBOL || sc.sampler_type > MI_SAMPLER_MAXMIN) return fail("mi_scene_upload: unknown sampler");
    if (MI_SAMPLER_IS_TILE_SERIAL(sc.sampler_type)) sc.sobol_resolution = sc.sobol_log2_resolution = 0;   // not a GlobalSampler: whatever the caller left in the Sobol' fields is never an index
    sc.pix_maxmin = nullptr;
    sc.pix_rng = nullptr; sc.pix_s1 = sc.pix_s2 = nullptr; sc.pix_nd = 0; sc.strat_nx = sc.strat_ny = 1; sc.strat_jitter = 0;
    if (MI_SAMPLER_IS_TILE_SERIAL(sc.sampler_type)) {   // one PCG32 stream per tile + the current pixel's precomputed dimensions
        const mi_integrator &in = d->integrator;
        sc.pix_nd = sc.sampler_type == MI_SAMPLER_RANDOM ? 0 : in.pixel_sampler_dims;
        sc.strat_nx = in.strat_samples[0]; sc.strat_ny = in.strat_samples[1]; sc.strat_jitter = in.strat_jitter;
        if (sc.pix_nd < 0 || sc.pix_nd > 255 || sc.spp < 1) return fail("mi_scene_upload: pixel sampler dimensions out of range");
        if (sc.sampler_type == MI_SAMPLER_STRATIFIED && (sc.strat_nx < 1 || sc.strat_ny < 1 || (int64_t)sc.strat_nx * sc.strat_ny != sc.spp))
            return fail("mi_scene_upload: stratified sampler: spp != xsamples * ysamples");
        if (sc.sampler_type == MI_SAMPLER_ZEROTWO && (sc.spp & (sc.spp - 1))) return fail("mi_scene_upload: 02sequence sampler: spp is not a power of two");
        if (sc.sampler_type == MI_SAMPLER_MAXMIN) {   // maxmin.h:54-77: a power of two <= 2^16; samples2D[0] is written unconditionally -> at least one sampled dimension
            if ((sc.spp & (sc.spp - 1)) || sc.spp > 65536 || sc.pix_nd < 1) return fail("mi_scene_upload: maxmindist sampler: spp must be a power of two <= 65536 and dimensions >= 1");
            DevBuf &b = next();
            if (upload(c, b, in.maxmin_matrix, sizeof(in.maxmin_matrix))) return -1;
            sc.pix_maxmin = b.as<uint32_t>();
        }
        const size_t nTiles = (size_t)((d->film.sample_max[0] - d->film.sample_min[0] + 15) / 16) * (size_t)((d->film.sample_max[1] - d->film.sample_min[1] + 15) / 16);
        { DevBuf &b = next(); if (b.alloc(std::max<size_t>(1, nTiles) * 2 * sizeof(unsigned long long))) return -1; sc.pix_rng = b.as<unsigned long long>(); }
        const size_t nv = std::max<size_t>(1, nTiles * (size_t)sc.pix_nd * (size_t)sc.spp);
        { DevBuf &b = next(); if (b.alloc(nv * sizeof(float))) return -1; sc.pix_s1 = b.as<float>(); }
        { DevBuf &b = next(); if (b.alloc(2 * nv * sizeof(float))) return -1; sc.pix_s2 = b.as<float>(); }
    }
    if (sc.sampler_type == MI_SAMPLER_HALTON) {
        for (int i = 0; i < 2; ++i) {
            sc.h_base_scales[i] = d->integrator.halton_base_scales[i]; sc.h_base_exps[i] = d->integrator.halton_base_exponents[i];
            sc.h_mult_inv[i] = d->integrator.halton_mult_inverse[i];
        }
        sc.h_stride = d->integrator.halton_sample_stride;
        sc.h_at_center = d->integrator.halton_sample_at_center;
        if (sc.h_stride < 1 || sc.h_base_scales[0] < 1 || sc.h_base_scales[1] < 1 || sc.h_base_scales[0] * sc.h_base_scales[1] != sc.h_stride)
            return fail("mi_scene_upload: inconsistent Halton parameters");
        sc.h_magic_scale1 = sc.h_base_scales[1] > 1 ? ~0ull / (uint64_t)sc.h_base_scales[1] + 1 : 0;   // floor((2^64-1)/d)+1 == floor(2^64/d)+1 unless d | 2^64
        if (sc.h_base_scales[1] == 1) return fail("mi_scene_upload: Halton base scale 1 in y (image of height 1) is not supported");
        // worst-case dimensions: 5 camera + 8 per bounce; HaltonSampler can only sample PrimeTableSize = 1000 (halton.h:72-75)
        if (5 + 8 * (int64_t)(d->integrator.max_depth + 1) > 1000) return fail("mi_scene_upload: maxdepth needs more than 1000 Halton dimensions");
        // Primes / PrimeSums (core/lowdiscrepancy.cpp), digit permutations = ComputeRadicalInversePermutations(RNG()) (:2490-2504,
        // Shuffle core/sampling.h:151-157, PCG32 core/rng.h:60-150)
        const int NP = 1000;
        std::vector<uint32_t> primes;
        for (uint32_t cnd = 2; (int)primes.size() < NP; ++cnd) {
            bool prime = true;
            for (size_t k = 0; k < primes.size() && primes[k] * primes[k] <= cnd; ++k) if (cnd % primes[k] == 0) { prime = false; break; }
            if (prime) primes.push_back(cnd);
        }
        std::vector<uint4> info(NP);
        uint32_t sum = 0;
        for (int i = 0; i < NP; ++i) {
            uint64_t magic = ~0ull / primes[i] + 1;
            if (primes[i] == 2) magic = 0x8000000000000001ull;   // floor(2^64/2) + 1
            info[i] = make_uint4(primes[i], sum, (uint32_t)magic, (uint32_t)(magic >> 32));
            sum += primes[i];
        }
        std::vector<uint16_t> perms(sum);
        {
            uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
            auto next32 = [&]() -> uint32_t {
                uint64_t oldstate = state;
                state = oldstate * 0x5851f42d4c957f2dULL + inc;
                uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u), rot = (uint32_t)(oldstate >> 59u);
                return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
            };
            auto bounded = [&](uint32_t b) -> uint32_t {
                uint32_t threshold = (~b + 1u) % b;
                while (true) { uint32_t r = next32(); if (r >= threshold) return r % b; }
            };
            uint16_t *pp = perms.data();
            for (int i = 0; i < NP; ++i) {
                int n = (int)primes[i];
                for (int j = 0; j < n; ++j) pp[j] = (uint16_t)j;
                for (int j = 0; j < n; ++j) { int other = j + (int)bounded((uint32_t)(n - j)); std::swap(pp[j], pp[other]); }
                pp += n;
            }
        }
        { DevBuf &b = next(); if (upload(c, b, perms.data(), perms.size() * 2)) return -1; sc.h_perms = b.as<uint16_t>(); }
        { DevBuf &b = next(); if (upload(c, b, info.data(), info.size() * sizeof(uint4))) return -1; sc.h_info = b.as<uint4>(); }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    {
        int sppBits = 0;
        while ((1 << sppBits) < sc.spp) ++sppBits;
        sc.sobol_index_bits = std::min(PBRT_AMD_SOBOL_NCOL, 2 * sc.sobol_log2_resolution + sppBits);
    }
    if (sc.sobol_log2_resolution > PBRT_AMD_SOBOL_NRES) return fail("mi_scene_upload: image too large for the Sobol' tables");
    if (sc.sample_max[0] - sc.sample_min[0] > 65535 || sc.sample_max[1] - sc.sample_min[1] > 65535)
        return fail("mi_scene_upload: sample bounds exceed 65535 pixels per axis");
    // worst-case Sobol' dimensions: 5 camera + 8 per bounce (the reference LOG(FATAL)s past 1024, sobol.cpp:48-51)
    if (5 + 8 * (int64_t)(sc.max_depth + 1) > PBRT_AMD_SOBOL_NDIM) return fail("mi_scene_upload: maxdepth needs more than 1024 Sobol' dimensions");
    // hot nodes of the quantised tree (k_hot_probe): measured per scene AND camera, nodes renumbered on the device so that nodesq[0 .. n_hot) are the
    // most visited ones (the root stays node 0: every probe path visits it; ties go to the lower index, so the numbering is a function of the
    // scene alone).  Only for scenes whose closest-hit / any-hit instances keep hot nodes (TraceShape::BIG / MID: all-triangle scenes; the probe itself ignores alpha masks --
    // it only ranks nodes).
    // PBRT_AMD_HOT=0 keeps the reference order and n_hot = 0 (every step through the vector-memory path; A/B and tests).
    sc.n_hot = 0;
    if (c->useQ && qnBuf && PT_HOT_NODES > 0 && (!(c->hasSpheres || c->hasAlpha) || (PT_TRACE_MID && !(c->hasSpheres && c->hasAlpha))) && sc.n_nodes > 1 && sc.spp > 0 && sc.sample_max[0] > sc.sample_min[0] && sc.sample_max[1] > sc.sample_min[1]) {
        const char *e = std::getenv("PBRT_AMD_HOT");
        if (!(e && e[0] == '0')) {
            const uint32_t n = sc.n_nodes, K = std::min<uint32_t>(PT_HOT_NODES, n);
            const uint32_t W = (uint32_t)(sc.sample_max[0] - sc.sample_min[0]), H = (uint32_t)(sc.sample_max[1] - sc.sample_min[1]);
            uint32_t npx = std::min<uint32_t>(W, 256), npy = std::min<uint32_t>(H, 256);   // <= 65536 probe paths on a regular grid of pixels
            uint32_t nProbe = npx * npy;
            if (nProbe < 16384) nProbe = std::min<uint64_t>(16384, (uint64_t)npx * npy * (uint32_t)sc.spp);   // small films: several samples per pixel
            nProbe = (nProbe / (npx * npy)) * (npx * npy);
            const int spillPer = std::max(1, sc.stack_need - PT_LDS_STACK);   // k_hot_probe: TravStack (PT_LDS_STACK entries in LDS)
            DevBuf dv, dsp, dni, dst;
            if (dv.alloc((size_t)n * 4) || dsp.alloc((size_t)((nProbe + PT_BLOCK - 1) / PT_BLOCK) * PT_BLOCK * spillPer * sizeof(StackEntry)) || dni.alloc((size_t)n * 4) || dst.alloc((size_t)n * sizeof(BVH4QNode))) return -1;
            HIP_TRY(hipMemsetAsync(dv.p, 0, (size_t)n * 4, c->stream));
            hipLaunchKernelGGL(k_hot_probe, dim3((nProbe + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, c->stream, sc, nProbe, npx, npy, std::min(4, std::max(0, sc.max_depth)),
                               dv.as<uint32_t>(), dsp.as<StackEntry>(), spillPer);
            HIP_TRY(hipGetLastError());
            std::vector<uint32_t> visits(n), order(n), newIdx(n);
            HIP_TRY(hipMemcpyAsync(visits.data(), dv.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (uint32_t i = 0; i < n; ++i) order[i] = i;
            auto hotter = [&](uint32_t a, uint32_t b) { return visits[a] != visits[b] ? visits[a] > visits[b] : a < b; };
            std::partial_sort(order.begin(), order.begin() + K, order.end(), hotter);
            std::vector<uint8_t> isHot(n, 0);
            for (uint32_t k = 0; k < K; ++k) { newIdx[order[k]] = k; isHot[order[k]] = 1; }
            for (uint32_t i = 0, nextCold = K; i < n; ++i) if (!isHot[i]) newIdx[i] = nextCold++;   // the others keep their (depth-first) order
            if (newIdx[0] != 0) return fail("mi_scene_upload: hot-node probe: the root is not the most visited node");
            HIP_TRY(hipMemcpyAsync(dni.p, newIdx.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_renumber_nodes, dim3(std::min<uint32_t>((n + PT_BLOCK - 1) / PT_BLOCK, 65536u)), dim3(PT_BLOCK), 0, c->stream, sc.nodesq, dst.as<BVH4QNode>(), dni.as<uint32_t>(), n);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            *qnBuf = std::move(dst);
            sc.nodesq = qnBuf->as<BVH4QNode>();
            sc.n_hot = K;
            uint64_t tot = 0, hot = 0;
            for (uint32_t i = 0; i < n; ++i) tot += visits[i];
            for (uint32_t k = 0; k < K; ++k) hot += visits[order[k]];
            c->hotProbeShare = tot ? (double)hot / (double)tot : 0.0;
        }
    }
    // row f4: media, medium interfaces, BSSRDF tables, and the DevScene itself in HBM for k_shade_vol
    std::memset(&c->vol, 0, sizeof(c->vol));
    c->scDev = nullptr;
    c->sssRoute = false; c->keyRemap = nullptr; c->shadeParts.clear();
    std::vector<char> needsVol(d->n_materials, 0);   // (sssRoute) the materials k_shade_vol must see
    if (c->volKernel) {
        DevVol &v = c->vol;
        v.handle_media = d->integrator_type == MI_INTEGRATOR_VOLPATH;
        v.camera_medium = v.handle_media ? d->camera_medium : -1;
        // Which form (pt_volpath.h).  Every combination has a wavefront form; the general form is the A/B partner (environment switches below).
        //  * "volpath", no BSSRDF: direct-lighting rays through the queues; WALKED segment by segment (volTr: k_trace<..., TR> + k_vol_tr_step; the walk's closest-hit kernel
        //    steps through interfaces and evaluates alphaMask, not shadowAlphaMask, exactly where VisibilityTester::Tr's Scene::Intersect does, core/light.cpp:63-82,
        //    shapes/triangle.cpp:333-338) with BSDF-less interfaces, alpha masks or a grid medium, whose Tr draws a
        //    data-dependent number of dimensions between the light sample and the continuation sample -- the vertex is then shaded in two stages around the walk (volSplit)
        //  * BSSRDF materials (sssWave): Sample_S draws its numbers before the probe chain is traced and nothing in between does: the chain is walked through the queues; the
        //    direct-lighting rays of the subsurface vertex and of the entry vertex take the plain traversals or the walk, and with a grid medium both are split like any vertex
        // PBRT_AMD_VOL_INLINE=1 keeps the general form everywhere, PBRT_AMD_VOL_TR_QUEUES=0 for interfaces / masks, PBRT_AMD_VOL_SPLIT=0 for grid media (A/B, parity tests)
        bool allHomogeneous = true;
        for (uint32_t i = 0; i < d->n_media; ++i) allHomogeneous = allHomogeneous && d->media[i].type == MI_MEDIUM_HOMOGENEOUS;
        const bool hasSss = d->material_bssrdf != nullptr;
        bool wave = true;
        const bool split = v.handle_media && !allHomogeneous;
        { const char *e = std::getenv("PBRT_AMD_VOL_INLINE"); if (e && e[0] == '1') wave = false; }
        { const char *e = std::getenv("PBRT_AMD_SSS_TAIL"); if (e && e[0]) c->sssTail = (uint32_t)std::strtoul(e, nullptr, 10); }
        { const char *e = std::getenv("PBRT_AMD_TR_LEAN"); c->trLean = !(e && e[0] == '0'); }
        { const char *e = std::getenv("PBRT_AMD_SSS_LOG"); c->sssLog = !(e && e[0] == '0'); }
        { const char *e = std::getenv("PBRT_AMD_VOL_SPLIT"); if (split && e && e[0] == '0') wave = false; }
        { const char *e = std::getenv("PBRT_AMD_VOL_TR_QUEUES"); if (e && e[0] == '0' && v.handle_media && (c->hasNullMat || c->hasAlpha)) wave = false; }
        c->volWave = wave;
        c->volSplit = wave && split;
        c->volTr = wave && v.handle_media && (c->hasNullMat || c->hasAlpha || c->volSplit);
        c->sssWave = wave && hasSss;
        v.tr_queues = c->volTr ? 1 : 0;
        v.tr_dims = c->volSplit ? 1 : 0;
        v.sss_wave = c->sssWave ? 1 : 0;
        v.textured = c->hasTex ? 1 : 0;   // (alpha masks alone leave c_tex.descs null: the lobe lists stay the constant ones)
        if (v.handle_media && d->n_media) {
            std::vector<mi_medium> med(d->media, d->media + d->n_media);
            for (uint32_t i = 0; i < d->n_media; ++i) {
                mi_medium &m = med[i];
                if (m.type == MI_MEDIUM_GRID) {
                    if (!m.density || m.nx <= 0 || m.ny <= 0 || m.nz <= 0) return fail("mi_scene_upload: grid medium without a density grid");
                    DevBuf &b = next();
                    if (upload(c, b, m.density, (size_t)m.nx * m.ny * m.nz * sizeof(float))) return -1;
                    m.density = b.as<float>();
                } else
                    m.density = nullptr;
            }
            { DevBuf &b = next(); if (upload(c, b, med.data(), med.size() * sizeof(mi_medium))) return -1; v.media = b.as<mi_medium>(); }
            HIP_TRY(hipStreamSynchronize(c->stream));   // local
            if (d->mesh_medium) {
                for (uint32_t m = 0; m < 2 * d->n_meshes; ++m) if (d->mesh_medium[m] >= (int32_t)d->n_media) return fail("mi_scene_upload: mesh_medium refers to a missing medium");
                DevBuf &b = next();
                if (upload(c, b, d->mesh_medium, 2 * (size_t)d->n_meshes * sizeof(int32_t))) return -1;
                v.mesh_medium = b.as<int32_t>();
            }
        }
        if (d->material_bssrdf) {
            std::vector<DevBssrdfTable> tabs(d->n_bssrdf_tables);
            for (uint32_t i = 0; i < d->n_bssrdf_tables; ++i) {
                const mi_bssrdf_table &t = d->bssrdf_tables[i];
                if (t.n_rho < 2 || t.n_radius < 2) return fail("mi_scene_upload: degenerate BSSRDF table");
                tabs[i].n_rho = t.n_rho; tabs[i].n_radius = t.n_radius;
                { DevBuf &b = next(); if (upload(c, b, t.rho_samples, (size_t)t.n_rho * 4)) return -1; tabs[i].rho_samples = b.as<float>(); }
                { DevBuf &b = next(); if (upload(c, b, t.radius_samples, (size_t)t.n_radius * 4)) return -1; tabs[i].radius_samples = b.as<float>(); }
                { DevBuf &b = next(); if (upload(c, b, t.profile, (size_t)t.n_rho * t.n_radius * 4)) return -1; tabs[i].profile = b.as<float>(); }
                { DevBuf &b = next(); if (upload(c, b, t.rho_eff, (size_t)t.n_rho * 4)) return -1; tabs[i].rho_eff = b.as<float>(); }
                { DevBuf &b = next(); if (upload(c, b, t.profile_cdf, (size_t)t.n_rho * t.n_radius * 4)) return -1; tabs[i].profile_cdf = b.as<float>(); }
            }
            for (uint32_t m = 0; m < d->n_materials; ++m) {
                const mi_bssrdf_desc &b = d->material_bssrdf[m];
                if (b.kind == MI_BSSRDF_NONE) continue;
                if (b.table < 0 || (uint32_t)b.table >= d->n_bssrdf_tables) return fail("mi_scene_upload: BSSRDF material refers to a missing table");
                const int nodes[4] = {b.kind == MI_BSSRDF_SUBSURFACE ? b.sigma_a : b.Kd, b.kind == MI_BSSRDF_SUBSURFACE ? b.sigma_s : b.mfp, 0, 0};
                for (int k = 0; k < 2; ++k) if (nodes[k] < 0 || (uint32_t)nodes[k] >= d->n_textures) return fail("mi_scene_upload: BSSRDF material refers to a missing texture node");
            }
            { DevBuf &b = next(); if (upload(c, b, tabs.data(), tabs.size() * sizeof(DevBssrdfTable))) return -1; v.tables = b.as<DevBssrdfTable>(); }
            { DevBuf &b = next(); if (upload(c, b, d->material_bssrdf, (size_t)d->n_materials * sizeof(mi_bssrdf_desc))) return -1; v.bssrdf = b.as<mi_bssrdf_desc>(); }
            // Routing (sssRoute).  Under Integrator "path" a vertex on a material WITHOUT a BSSRDF is PathIntegrator::Li's ordinary loop body: k_shade does it from the same
            // PathRec / NeeRec fields at 3 waves per SIMD with wave-uniform lobe lists, where k_shade_vol (per-lane lobe lists, 2 waves) would treat it as the general case.
            // The material sort orders the keys [materials without BSSRDF, escaped, null-BSDF | materials with BSSRDF]; k_shade takes the first part of the sorted
            // queue and k_shade_vol the second.  Not with a walk or a split vertex (volTr / volSplit: "volpath" scenes): there NeeRec carries the walk's extra words.
            bool route = c->sssWave && !v.handle_media && !c->volTr && !c->volSplit;
            { const char *e = std::getenv("PBRT_AMD_SSS_ROUTE"); if (e && e[0] == '0') route = false; }
            if (route) {
                // which materials k_shade_vol must see: those with a BSSRDF, and mix materials built on one -- MixMaterial evaluates m1 on *si itself (mixmat.cpp:45-64), so
                // the mix's interaction carries m1's BSSRDF (ComputeBSSRDFD follows the m1 links); m2's is included for simplicity.  Sub-materials have smaller indices.
                // (found by tools/fuzz_vs_reference.py --device --sss: a mix of two subsurface materials shaded by k_shade lost its BSSRDF bounce)
                bool any = false;
                for (uint32_t m = 0; m < d->n_materials; ++m) {
                    needsVol[m] = d->material_bssrdf[m].kind != MI_BSSRDF_NONE;
                    if (d->material_descs && d->material_descs[m].type == MI_MAT_MIX) {
                        const int m1 = d->material_descs[m].m1, m2 = d->material_descs[m].m2;
                        if (m1 >= 0 && (uint32_t)m1 < m) needsVol[m] |= needsVol[m1];
                        if (m2 >= 0 && (uint32_t)m2 < m) needsVol[m] |= needsVol[m2];
                    }
                    any = any || needsVol[m];
                }
                c->plainTex = false;
                if (d->material_descs) for (uint32_t m = 0; m < d->n_materials; ++m) c->plainTex |= d->material_descs[m].textured != 0 && !needsVol[m];
                c->sssRoute = any;   // (some material does have a BSSRDF; the key order itself is built below with the shading parts)
            }
            HIP_TRY(hipStreamSynchronize(c->stream));   // local
        }
        // sampler dimensions: ratio / delta tracking in grid media draw a data-dependent number per segment; the reference aborts past its tables
        // (sobol.cpp:48-51, halton.h:72-75) and the device clamps to the last dimension instead -- such paths are outside both samplers' range
        { DevBuf &b = next(); if (upload(c, b, &sc, sizeof(DevScene))) return -1; c->scDev = b.as<DevScene>(); }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // ---- the parts the material-sorted queue is shaded in (PathState::key_remap, ShadeRange): one, or with sssRoute two -- key order [materials k_shade sees, escaped
    // rays, null-BSDF surfaces | the materials k_shade_vol must see]
    {
        const uint32_t nm = d->n_materials, nk = nm + 2;
        if (c->sssRoute) {
            std::vector<uint32_t> remap(nk);
            uint32_t nextKey = 0;
            for (uint32_t m = 0; m < nm; ++m) if (!needsVol[m]) remap[m] = nextKey++;
            remap[nm] = nextKey++; remap[nm + 1] = nextKey++;
            const uint32_t split = nextKey;
            for (uint32_t m = 0; m < nm; ++m) if (needsVol[m]) remap[m] = nextKey++;
            if (nextKey != nk) return fail("mi_scene_upload: internal error (shading parts do not cover the keys)");
            c->shadeParts.push_back({false, 0u, split});
            c->shadeParts.push_back({true, split, 0xffffffffu});
            DevBuf &b = next();
            if (upload(c, b, remap.data(), remap.size() * sizeof(uint32_t))) return -1;
            c->keyRemap = b.as<uint32_t>();
            HIP_TRY(hipStreamSynchronize(c->stream));   // local
        } else
            c->shadeParts.push_back({c->volKernel, 0u, 0xffffffffu});
    }
    c->nkeys = d->n_materials + 2;
    if (c->nkeys > 12288) return fail("mi_scene_upload: more than 12286 distinct materials (LDS histogram of the material sort)");
    // film
    c->filmPixels = (int64_t)std::max(0, sc.crop_max[0] - sc.crop_min[0]) * std::max(0, sc.crop_max[1] - sc.crop_min[1]);
    if (c->film.alloc((size_t)c->filmPixels * sizeof(float4))) return -1;
    HIP_TRY(hipMemsetAsync(c->film.p, 0, c->film.bytes, c->stream));
    c->filmPtr = c->film.as<float4>();
    HIP_TRY(hipStreamSynchronize(c->stream));   // host staging vectors go out of scope
    // state depending on the scene is (re)allocated lazily by ensure_state()
    for (auto &b : c->stateBufs) b.release();
    c->stateBufs.clear();
    c->cap = 0;
    c->haveScene = true;
    return 0;
}

}  // extern "C"

// k_sss_probe_tail's hit lists (PathState::sss_log_*): threads x PT_SSS_LOG_CAP entries of 36 bytes -- 1.2 GB at the default tail threshold; counted against the path-state budget (mi_render)
#define PT_SSS_LOG_CAP 256u
static uint32_t SssLogThreads(const mi_ctx *c, uint32_t cap) { return std::min<uint32_t>(c->sssTail, cap) + 8 * PT_BLOCK; }
static size_t SssLogBytes(const mi_ctx *c, uint32_t cap) {
    return (c->sssWave && c->sssTail && c->sssLog) ? (size_t)SssLogThreads(c, cap) * PT_SSS_LOG_CAP * (2 * sizeof(float4) + sizeof(uint32_t)) : 0;
}
static int ensure_state(mi_ctx *c, uint32_t cap) {
    if (c->cap >= cap && !c->stateBufs.empty()) return 0;
    for (auto &b : c->stateBufs) b.release();
    c->stateBufs.clear();
    c->stateBufs.resize(48);
    int nb = 0;
    PathState &ps = c->ps;
    std::memset(&ps, 0, sizeof(ps));
    auto A = [&](size_t bytes) -> void * { DevBuf &b = c->stateBufs[nb++]; return b.alloc(bytes) ? nullptr : b.p; };
#define ALLOC(field, type, count) do { ps.field = (type *)A(sizeof(type) * (size_t)(count)); if (!ps.field) return -1; } while (0)
    // queue segment capacity: a block class handles one contiguous eighth of the 256-item chunks of whatever it walks (ChunkIter) and
    // appends at most one entry per item to each queue
    // (+ the shading launches' headroom: each of up to PT_SHADE_PARTS_MAX parts of the sorted queue is cut into eighths of its own, ceil(ceil(n_i / 256) / 8) * 256
    // <= n_i / 8 + 256 items per class and part -- ADVICE r4: two parts could overrun a segment sized for one launch by up to 512 entries)
    const uint32_t chunks = (cap + PT_BLOCK - 1) / PT_BLOCK;
    ps.seg_cap = ((chunks + 7) / 8) * PT_BLOCK + PT_SHADE_PARTS_MAX * (PT_DYN_GRAIN > PT_BLOCK ? PT_DYN_GRAIN : PT_BLOCK);   // (DynIter hands a class whole grains: up to one grain beyond n_i / 8 per part)
    const size_t qcap = (size_t)QSEG * ps.seg_cap;
    ALLOC(rec, PathRec, cap); ALLOC(nee, NeeRec, cap); ALLOC(keyrank, uint2, qcap);
    ALLOC(q_ext[0], uint32_t, qcap); ALLOC(q_ext[1], uint32_t, qcap); ALLOC(q_shadow, uint32_t, qcap); ALLOC(q_mis, uint32_t, qcap);
    ALLOC(q_sorted, uint32_t, qcap);
    ps.qrow_shadow = QC_SHADOW; ps.qrow_mis = QC_MIS;
    if (c->volTr || c->sssWave) { ALLOC(trs, TrState, cap); ALLOC(q_tr[0], uint32_t, qcap); ALLOC(q_tr[1], uint32_t, qcap); }
    if (c->sssWave) { ALLOC(sss, SssRec, cap); ALLOC(q_probe[0], uint32_t, qcap); ALLOC(q_probe[1], uint32_t, qcap); }
    if (c->sssWave) ALLOC(q_sss, uint32_t, qcap);
    if (c->sssWave && c->sssTail && c->sssLog) {   // one list per thread the tail launch can occupy when its queue is spread evenly over the eight segments (threads beyond walk twice)
        const uint32_t threads = SssLogThreads(c, cap), lcap = PT_SSS_LOG_CAP;
        ALLOC(sss_log_o, float4, (size_t)threads * lcap); ALLOC(sss_log_d, float4, (size_t)threads * lcap); ALLOC(sss_log_inst, uint32_t, (size_t)threads * lcap);
        ps.sss_log_threads = threads; ps.sss_log_cap = lcap;
    }
    if (c->volSplit) ALLOC(q_cont, uint32_t, qcap);
    ALLOC(qcount, uint32_t, QC_WORDS);
    ALLOC(keycount, uint32_t, c->nkeys); ALLOC(keyoffset, uint32_t, c->nkeys); ALLOC(cursor, uint32_t, QSEG * QC_STRIDE);
    ALLOC(blockhist, uint32_t, (size_t)c->gridBlocks * c->nkeys);
    ps.spill_per_thread = std::max(1, c->sc.stack_need - PT_LDS_STACK_MIN);
    {
        const size_t words4 = (size_t)ps.spill_per_thread * (sizeof(StackEntry) / 4);
        // slices for the largest number of threads any traversal launch has: the 256-thread shape launches gridBlocks x PT_BLOCK, a TraceShape<...>::BIG
        // launch roundup8(numCUs x PER_CU) x BLOCK <= numCUs x PT_GRID_PER_CU x PT_BLOCK + 7 x BLOCK (the static_assert of TraceShape) -- hence the margin
        const size_t spillThreads = (size_t)c->gridBlocks * PT_BLOCK + 8 * 1024;
        ALLOC(spill, uint32_t, spillThreads * words4);
        c->cursor2 = c->spill2 = nullptr;
        if (c->overlapNee) {
            c->cursor2 = (uint32_t *)A(sizeof(uint32_t) * QSEG * QC_STRIDE);
            c->spill2 = (uint32_t *)A(sizeof(uint32_t) * spillThreads * words4);
            if (!c->cursor2 || !c->spill2) return -1;
        }
    }
#undef ALLOC
    ps.counters = c->counters.as<unsigned long long>();
    ps.cap = cap;
    c->cap = cap;
    return 0;
}

// ---- launch bookkeeping (per-kernel device time from HIP events on the ctx stream)
static void tic(mi_ctx *c, int id, hipStream_t onStream = nullptr) {
    if (!c->timing) return;
    if (c->evUsed == c->evPool.size()) {
        mi_ctx::Ev e;
        (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
        c->evPool.push_back(e);
    }
    c->evPool[c->evUsed].id = id;
    (void)hipEventRecord(c->evPool[c->evUsed].a, onStream ? onStream : c->stream);
}
static void toc(mi_ctx *c, hipStream_t onStream = nullptr) {
    if (!c->timing) return;
    (void)hipEventRecord(c->evPool[c->evUsed].b, onStream ? onStream : c->stream);
    ++c->evUsed;
}
static void harvest(mi_ctx *c) {
    for (size_t i = 0; i < c->evUsed; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->evPool[i].a, c->evPool[i].b) == hipSuccess) { c->msTotal[c->evPool[i].id] += ms; c->launches[c->evPool[i].id]++; }
    }
    c->evUsed = 0;
}

// template arguments after COUNT: SPHERES, ALPHA, INST, QN.  Every instance brings its own launch shape (TraceShape: threads per block, blocks per CU)
#define LAUNCH_TRACE_I(MODE, SPH, ALP, INS, QNN)                                                                    \
    do {                                                                                                            \
        typedef TraceShape<MODE, SPH, ALP, QNN> TS_;                                                                     \
        const dim3 g_(((c->numCUs * TS_::PER_CU + 7) / 8) * 8), b_(TS_::BLOCK);   /* multiple of 8 for the XCD mapping */ \
        if (countWork) hipLaunchKernelGGL((k_trace<MODE, true, SPH, ALP, INS, QNN>), g_, b_, 0, st, sc, ps, qin);   \
        else hipLaunchKernelGGL((k_trace<MODE, false, SPH, ALP, INS, QNN>), g_, b_, 0, st, sc, ps, qin);            \
    } while (0)
#define LAUNCH_TRACE(MODE)                                                                                          \
    do {                                                                                                            \
        if (c->hasInst) LAUNCH_TRACE_I(MODE, true, true, true, false);        /* two-level scenes: the general instance (spheres, masks, instances) */ \
        else if (c->useQ && c->hasAlpha && !c->hasSpheres) LAUNCH_TRACE_I(MODE, false, true, false, true);   /* masks, no spheres: TraceShape::MID */          \
        else if (c->useQ && c->hasAlpha) LAUNCH_TRACE_I(MODE, true, true, false, true);     /* quantised nodes; spheres + masks in the leaf step */     \
        else if (c->useQ && c->hasSpheres) LAUNCH_TRACE_I(MODE, true, false, false, true);                             \
        else if (c->useQ) LAUNCH_TRACE_I(MODE, false, false, false, true);    /* the default: all-triangle single-level scenes */                      \
        else if (c->hasAlpha) LAUNCH_TRACE_I(MODE, true, true, false, false); /* PBRT_AMD_TRACE=general: full-precision nodes */                       \
        else if (c->hasSpheres) LAUNCH_TRACE_I(MODE, true, false, false, false);                                       \
        else LAUNCH_TRACE_I(MODE, false, false, false, false);                                                         \
    } while (0)
// the segment traversals of walked shadow / MIS rays (k_trace<..., TR>; volpath scenes with BSDF-less interfaces or alpha masks): five instances per mode
#define LAUNCH_TRACE_TR_I(MODE, SPH, ALP, INS, QNN)                                                                 \
    do {                                                                                                            \
        typedef TraceShape<MODE, SPH, ALP, QNN> TS_;                                                                \
        const dim3 g_(((c->numCUs * TS_::PER_CU + 7) / 8) * 8), b_(TS_::BLOCK);                                     \
        if (countWork) hipLaunchKernelGGL((k_trace<MODE, true, SPH, ALP, INS, QNN, true>), g_, b_, 0, st, sc, ps, qin);   \
        else hipLaunchKernelGGL((k_trace<MODE, false, SPH, ALP, INS, QNN, true>), g_, b_, 0, st, sc, ps, qin);            \
    } while (0)
#define LAUNCH_TRACE_TR(MODE)                                                                                       \
    do {                                                                                                            \
        if (c->hasInst) LAUNCH_TRACE_TR_I(MODE, true, true, true, false);                                           \
        else if (c->useQ && c->hasAlpha) LAUNCH_TRACE_TR_I(MODE, true, true, false, true);                          \
        else if (c->useQ) LAUNCH_TRACE_TR_I(MODE, true, false, false, true);                                        \
        else if (c->hasAlpha) LAUNCH_TRACE_TR_I(MODE, true, true, false, false);                                    \
        else LAUNCH_TRACE_TR_I(MODE, true, false, false, false);                                                    \
    } while (0)
// The shadow-queue walks (MODE 2: the segments of walked shadow rays and of BSSRDF probe chains) of all-triangle single-level scenes without masks take the lean
// instance: the 768-thread shape with hot nodes in LDS (TraceShape::BIG), as the plain traversals of such scenes do.  PBRT_AMD_TR_LEAN=0: the sphere-capable instance (A/B)
#define LAUNCH_TRACE_TR_SHADOW                                                                                      \
    do {                                                                                                            \
        if (c->trLean && c->useQ && !c->hasAlpha && !c->hasInst && !c->hasSpheres) LAUNCH_TRACE_TR_I(2, false, false, false, true); \
        else LAUNCH_TRACE_TR(2);                                                                                    \
    } while (0)
// One pass of the wavefront pipeline over the paths generated by `pass`.
static int run_pass(mi_ctx *c, const PassInfo &pass, bool countWork, bool toFilm) {
    PathState &ps = c->ps;
    const DevScene &sc = c->sc;
    hipStream_t st = c->stream;
    dim3 grid(c->gridBlocks), block(PT_BLOCK);
    TableTurn turn(c);
    ps.key_remap = c->keyRemap;
    ps.shade_key_lo = 0; ps.shade_key_hi = 0xffffffffu;
    HIP_TRY(hipMemsetAsync(ps.qcount, 0, QC_WORDS * sizeof(uint32_t), st));
    if (c->hasTex || c->hasAlpha || c->hasInst)   // the tables of THIS context's scene (stream ordered; contexts sharing a device take turns: TableTurn)
        HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tex), &c->tex, sizeof(DevTex), 0, hipMemcpyHostToDevice, st));
    if (c->hasInst) HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_instances), &c->instPtr, sizeof(c->instPtr), 0, hipMemcpyHostToDevice, st));
    tic(c, MI_K_RAYGEN);
    if (c->hasTex || c->hasInst) hipLaunchKernelGGL(k_raygen<true>, grid, block, 0, st, sc, ps, pass, 0u);
    else hipLaunchKernelGGL(k_raygen<false>, grid, block, 0, st, sc, ps, pass, 0u);
    if (c->volKernel && c->vol.handle_media)
        hipLaunchKernelGGL(k_vol_camera_medium, grid, block, 0, st, ps, pass.list_xy ? pass.npix : pass.npix * pass.ns, c->vol.camera_medium);
    toc(c);
    uint32_t qin = 0;
    int iter = 0;
    while (true) {
        uint32_t qout = qin ^ 1;
        HIP_TRY(hipMemsetAsync(ps.qcount + QCI(qout, 0), 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
        const bool overlap = c->overlapNee && c->stream2 && (!c->volKernel || c->volWave) && !c->volTr && !c->sssWave;
        if (c->sssWave) HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_SSS, 0), 0, 3 * QSEG * QC_STRIDE * sizeof(uint32_t), st));   // QC_SSS + the probe queues (QC_PROBE0, QC_PROBE1)
        if (c->volSplit) HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_CONT, 0), 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));   // the vertices waiting for k_vol_continue
        if (!overlap) HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_SHADOW, 0), 0, 2 * QSEG * QC_STRIDE * sizeof(uint32_t), st));   // shadow + mis (overlap: after the join below)
        HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
        tic(c, MI_K_CLOSEST);
        {
            LAUNCH_TRACE(0);
        }
        toc(c);
        tic(c, MI_K_SORT);
        hipLaunchKernelGGL(k_keycount, grid, block, c->nkeys * sizeof(uint32_t), st, sc, ps, qin, c->nkeys);
        hipLaunchKernelGGL(k_scan_keys, dim3(c->nkeys), block, 0, st, ps, c->nkeys, (uint32_t)c->gridBlocks);
        hipLaunchKernelGGL(k_scan_keys_total, dim3(1), block, 0, st, ps, c->nkeys);
        hipLaunchKernelGGL(k_scatter, grid, block, 0, st, ps, qin, c->nkeys);
        toc(c);
        HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
        if (overlap) {   // join: the previous bounce's shadow / MIS traversals (stream2) add into PathRec::L and read the queues k_shade refills
            if (iter > 0) HIP_TRY(hipStreamWaitEvent(st, c->evNeeDone, 0));
            HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_SHADOW, 0), 0, 2 * QSEG * QC_STRIDE * sizeof(uint32_t), st));
        }
        tic(c, MI_K_SHADE);
        {   // compile-time variants keep the common case (Sobol', no radiance map) free of the other paths' registers
            const bool halton = sc.sampler_type == MI_SAMPLER_HALTON, pixSmp = MI_SAMPLER_IS_TILE_SERIAL(sc.sampler_type);
            auto shade_vol = [&](const PathState &ps) {   // row f4: media / BSSRDF (pt_volpath.h: the general form traces transmittance, MIS and probe rays in the shading lanes, the wavefront forms queue them)
                const dim3 gw(c->gridShadeVol);
#define LAUNCH_VOL(W, I, U, G) hipLaunchKernelGGL((k_shade_vol<W, I, U>), G, block, 0, st, c->scDev, ps, c->vol, qout)
                const bool umat = !c->vol.textured && !c->vol.bssrdf;   // constant lobe lists only: wave-uniform material access (the UMAT instance compiles the BSSRDF branch out)
                if (c->volWave) {
                    if (c->hasInst) { if (umat) LAUNCH_VOL(true, true, true, gw); else LAUNCH_VOL(true, true, false, gw); }
                    else { if (umat) LAUNCH_VOL(true, false, true, gw); else LAUNCH_VOL(true, false, false, gw); }
                } else {
                    if (c->hasInst) { if (umat) LAUNCH_VOL(false, true, true, grid); else LAUNCH_VOL(false, true, false, grid); }
                    else { if (umat) LAUNCH_VOL(false, false, true, grid); else LAUNCH_VOL(false, false, false, grid); }
                }
#undef LAUNCH_VOL
            };
#define LAUNCH_SHADE(ENV, TEX, ...)                                                                                                              \
    do {                                                                                                                                         \
        if (pixSmp) hipLaunchKernelGGL((k_shade<ENV, 2, TEX, ##__VA_ARGS__>), dim3(c->gridShade), block, 0, st, sc, ps, qout);                   \
        else if (halton) hipLaunchKernelGGL((k_shade<ENV, 1, TEX, ##__VA_ARGS__>), dim3(c->gridShade), block, 0, st, sc, ps, qout);               \
        else hipLaunchKernelGGL((k_shade<ENV, 0, TEX, ##__VA_ARGS__>), dim3(c->gridShade), block, 0, st, sc, ps, qout);                           \
    } while (0)
            auto shade_plain = [&](const PathState &ps) {
                // (routed subsurface scenes: the BSSRDF materials are always built per hit, mi_material_desc::textured -- what counts here are the materials k_shade sees)
                const bool tex = c->sssRoute ? c->plainTex : c->hasTex;
                const bool env = c->hasEnvMap || c->hasSpheres;
                if (c->hasInst) LAUNCH_SHADE(true, true, true);   // two-level scenes: the general instance + interactions carried back from the object's space
                else if (tex) LAUNCH_SHADE(true, true);           // textured materials: the general instance (radiance maps, spheres, per-lane lobe lists)
                else if (env) LAUNCH_SHADE(true, false);
                else LAUNCH_SHADE(false, false);
            };
#undef LAUNCH_SHADE
            // one launch per part of the sorted queue (mi_ctx::shadeParts: k_shade's vertices, then those k_shade_vol must see)
            bool firstPart = true;
            for (const mi_ctx::ShadePart &sp : c->shadeParts) {
                if (!firstPart) {
                    toc(c);   // (each part is its own entry of mi_timing_get's launch count)
                    HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                    tic(c, MI_K_SHADE);
                }
                firstPart = false;
                PathState part = ps;
                part.shade_key_lo = sp.keyLo; part.shade_key_hi = sp.keyHi;
                if (sp.vol) shade_vol(part); else shade_plain(part);
            }
        }
        toc(c);
        // the direct-lighting rays of a shading stage through the walk (volTr) or the plain any-hit / closest-hit traversals
        auto nee_walk = [&]() -> int {
            // wavefront form with BSDF-less interfaces between homogeneous media: the shadow and the MIS rays are WALKED through the interfaces, one
            // segment per round -- closest hit of the segment by the persistent-lane kernel (k_trace<..., TR>), then k_vol_tr_step ends the ray or
            // re-aims it behind the interface into the other queue.  Three rounds are queued blindly (a kernel on an empty queue returns at once),
            // then the host looks at the counter (such scenes synchronise once per pass anyway, see below).
            for (int which = 0; which < 2; ++which) {   // 0: shadow rays (MODE 2), 1: MIS rays (MODE 1)
                tic(c, which ? MI_K_MIS_CLOSEST : MI_K_ANYHIT);
                uint32_t rowIn = which ? QC_MIS : QC_SHADOW, rowOut = which ? QC_MIS2 : QC_SHADOW2;
                uint32_t *qIn = which ? ps.q_mis : ps.q_shadow, *qOut = ps.q_tr[which];
                bool drained = false;
                for (int round = 0; round < 4096; ++round) {
                    HIP_TRY(hipMemsetAsync(ps.qcount + QCI(rowOut, 0), 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                    HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                    {
                        PathState psRun = ps;
                        if (which) { psRun.q_mis = qIn; psRun.qrow_mis = rowIn; } else { psRun.q_shadow = qIn; psRun.qrow_shadow = rowIn; }
                        PathState &ps = psRun;
                        if (which) LAUNCH_TRACE_TR(1); else LAUNCH_TRACE_TR_SHADOW;
                    }
#define LAUNCH_TR_STEP(M, I, D) hipLaunchKernelGGL((k_vol_tr_step<M, I, D>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn, qOut, rowOut)
                    if (which) {
                        if (c->hasInst) { if (c->volSplit) LAUNCH_TR_STEP(1, true, true); else LAUNCH_TR_STEP(1, true, false); }
                        else { if (c->volSplit) LAUNCH_TR_STEP(1, false, true); else LAUNCH_TR_STEP(1, false, false); }
                    } else {
                        if (c->hasInst) { if (c->volSplit) LAUNCH_TR_STEP(2, true, true); else LAUNCH_TR_STEP(2, true, false); }
                        else { if (c->volSplit) LAUNCH_TR_STEP(2, false, true); else LAUNCH_TR_STEP(2, false, false); }
                    }
#undef LAUNCH_TR_STEP
                    std::swap(qIn, qOut); std::swap(rowIn, rowOut);
                    if (round >= 2) {
                        uint32_t left = 0, row[QSEG * QC_STRIDE];
                        HIP_TRY(hipMemcpyAsync(row, ps.qcount + QCI(rowIn, 0), sizeof(row), hipMemcpyDeviceToHost, st));
                        HIP_TRY(hipStreamSynchronize(st));
                        for (uint32_t sg = 0; sg < QSEG; ++sg) left += row[sg * QC_STRIDE];
                        if (left == 0) { drained = true; break; }
                    }
                }
                toc(c);
                if (!drained) return fail("mi_render: a walked shadow / MIS ray crossed more than 4096 medium interfaces (direct-lighting terms would be missing)");
            }
            return 0;
        };
        auto nee_plain = [&]() -> int {
            HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
            tic(c, MI_K_ANYHIT);
            LAUNCH_TRACE(2);
            toc(c);
            HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
            tic(c, MI_K_MIS_CLOSEST);
            {
                PathState psRun = ps;
                psRun.vol_tr = c->volWave && c->vol.handle_media ? 1u : 0u;   // (scenes without media: no sigma_t in NeeRec::pad[0] -- k_shade does not write it)
                PathState &ps = psRun;
                LAUNCH_TRACE(1);
            }
            toc(c);
            return 0;
        };
        auto vol_continue = [&]() {
            tic(c, MI_K_SHADE);
            if (c->hasInst) hipLaunchKernelGGL((k_vol_continue<true>), dim3(c->gridShadeVol), block, 0, st, c->scDev, ps, c->vol, qout);
            else hipLaunchKernelGGL((k_vol_continue<false>), dim3(c->gridShadeVol), block, 0, st, c->scDev, ps, c->vol, qout);
            toc(c);
        };
        if (overlap) {
            // The direct-lighting traversals of this bounce run on stream2 while the main stream goes on with the next bounce's path-extension
            // traversal and material sort: they touch disjoint data (NeeRec + PathRec::L vs PathRec::hit / keys / queues), have their own fetch
            // cursors and stack spill slices, and every persistent traversal launch ends in a tail of a few long rays that the other launch fills.
            HIP_TRY(hipEventRecord(c->evShaded, st));
            hipStream_t s2 = c->stream2;
            HIP_TRY(hipStreamWaitEvent(s2, c->evShaded, 0));
            PathState psNee = ps;
            psNee.cursor = c->cursor2; psNee.spill = c->spill2;
            psNee.vol_tr = c->volWave && c->vol.handle_media ? 1u : 0u;
            {
                hipStream_t st = s2;
                PathState &ps = psNee;
                HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                tic(c, MI_K_ANYHIT, st);
                LAUNCH_TRACE(2);
                toc(c, st);
                HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                tic(c, MI_K_MIS_CLOSEST, st);
                LAUNCH_TRACE(1);
                toc(c, st);
            }
            HIP_TRY(hipEventRecord(c->evNeeDone, s2));
        } else if (c->volTr) {
            if (nee_walk()) return -1;
            if (c->volSplit) vol_continue();   // second stage of the vertices whose direct-lighting rays have now consumed their dimensions
        } else if (!c->volKernel || c->volWave) {   // (k_shade_vol<false> traces its own shadow / MIS rays)
            if (nee_plain()) return -1;
        }
        if (c->sssWave) {
            // the subsurface paths k_shade_vol parked in this bounce: walk their probe chains (one segment per round: k_trace<2, ..., TR> finds its closest
            // hit, k_sss_probe_step takes it into the chain), then shade the entry vertices and trace THEIR direct-lighting rays.  The vertex's own
            // shadow / MIS rays are done (above): NeeRec::sh_* now carries the probe segments.
            uint32_t row[QSEG * QC_STRIDE], left = 0;
            HIP_TRY(hipMemcpyAsync(row, ps.qcount + QCI(QC_PROBE0, 0), sizeof(row), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (uint32_t sg = 0; sg < QSEG; ++sg) left += row[sg * QC_STRIDE];
            if (left) {
                uint32_t rowIn = QC_PROBE0, rowOut = QC_PROBE1;
                uint32_t *qIn = ps.q_probe[0], *qOut = ps.q_probe[1];
                tic(c, MI_K_SHADE);
                if (c->hasInst) hipLaunchKernelGGL((k_sss_probe_step<true>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn, qOut, rowOut, 1);
                else hipLaunchKernelGGL((k_sss_probe_step<false>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn, qOut, rowOut, 1);
                toc(c);
                std::swap(qIn, qOut); std::swap(rowIn, rowOut);
                tic(c, MI_K_MIS_CLOSEST);
                for (int round = 0; round < 16384; ++round) {
                    HIP_TRY(hipMemsetAsync(ps.qcount + QCI(rowOut, 0), 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                    HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
                    {
                        PathState psRun = ps;
                        psRun.q_shadow = qIn; psRun.qrow_shadow = rowIn;
                        PathState &ps = psRun;
                        LAUNCH_TRACE_TR_SHADOW;
                    }
                    if (c->hasInst) hipLaunchKernelGGL((k_sss_probe_step<true>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn, qOut, rowOut, 0);
                    else hipLaunchKernelGGL((k_sss_probe_step<false>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn, qOut, rowOut, 0);
                    std::swap(qIn, qOut); std::swap(rowIn, rowOut);
                    if (round >= 3) {   // (a chain has at least two hits per walk on a closed object: the first rounds are queued blindly)
                        left = 0;
                        HIP_TRY(hipMemcpyAsync(row, ps.qcount + QCI(rowIn, 0), sizeof(row), hipMemcpyDeviceToHost, st));
                        HIP_TRY(hipStreamSynchronize(st));
                        for (uint32_t sg = 0; sg < QSEG; ++sg) left += row[sg * QC_STRIDE];
                        if (left == 0) break;
                        if (left <= c->sssTail) {   // the tail of the walk: the few long chains are finished by their own lanes in ONE launch (k_sss_probe_tail)
                            if (c->hasInst) hipLaunchKernelGGL((k_sss_probe_tail<true>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn);
                            else hipLaunchKernelGGL((k_sss_probe_tail<false>), grid, block, 0, st, c->scDev, ps, c->vol, (const uint32_t *)qIn, rowIn);
                            left = 0;   // (a chain that does not end there trips the guard counter: mi_render's callers see MI_CNT_TRACE_GUARD_TRIPS)
                            break;
                        }
                    }
                }
                toc(c);
                if (left) return fail("mi_render: a BSSRDF probe chain did not end within 16384 segments");   // (every segment starts behind the previous hit: no real chain comes near)
                HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_SHADOW, 0), 0, 2 * QSEG * QC_STRIDE * sizeof(uint32_t), st));
                if (c->volSplit) HIP_TRY(hipMemsetAsync(ps.qcount + QCI(QC_CONT, 0), 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));   // (k_vol_continue has served the bounce's first stage)
                tic(c, MI_K_SHADE);
                if (c->hasInst) hipLaunchKernelGGL((k_sss_entry<true>), dim3(c->gridShadeVol), block, 0, st, c->scDev, ps, c->vol, qout);
                else hipLaunchKernelGGL((k_sss_entry<false>), dim3(c->gridShadeVol), block, 0, st, c->scDev, ps, c->vol, qout);
                toc(c);
                if (c->volTr ? nee_walk() : nee_plain()) return -1;   // the entry vertices' direct-lighting rays
                if (c->volSplit) vol_continue();   // ... and, in the split form, their continuation
            }
        }
        qin = qout;
        ++iter;
        if (iter > sc.max_depth) {
            // every ordinary path is done after max_depth+1 segments; only chains of null-BSDF surfaces
            // (which do not count as bounces) can still be alive -- check, and keep going if so.  Scenes without such
            // surfaces (no mesh with a null material) need no check: the pass stays asynchronous on the ctx stream.
            if (!c->hasNullMat) break;
            uint32_t left = 0, row[QSEG * QC_STRIDE];
            HIP_TRY(hipMemcpyAsync(row, ps.qcount + QCI(qin, 0), sizeof(row), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (uint32_t sg = 0; sg < QSEG; ++sg) left += row[sg * QC_STRIDE];
            if (left == 0 || iter > sc.max_depth + 4096) break;
        }
    }
    if (c->overlapNee && c->stream2 && (!c->volKernel || c->volWave) && !c->volTr && !c->sssWave && iter > 0) HIP_TRY(hipStreamWaitEvent(st, c->evNeeDone, 0));   // the last bounce's direct-lighting terms
    if (toFilm) {
        tic(c, MI_K_FILM);
        hipLaunchKernelGGL((k_film<false>), grid, block, 0, st, sc, ps, pass, c->filmPtr);
        hipLaunchKernelGGL((k_film<true>), grid, block, 0, st, sc, ps, pass, c->filmPtr);
        toc(c);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int mi_render(mi_ctx *c, const mi_render_params *rp) {
    if (!c || !rp) return fail("mi_render: null argument");
    if (!c->haveScene) return fail("mi_render: no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    const DevScene &sc = c->sc;
    int world = std::max(1, rp->world), rank = rp->rank;
    if (rank < 0 || rank >= world) return fail("mi_render: rank out of range");
    int s0 = std::max(0, rp->spp_begin), s1 = rp->spp_end < 0 ? sc.spp : std::min(rp->spp_end, sc.spp);
    if (s1 <= s0) return 0;
    // tile grid of SamplerIntegrator::Render (integrator.cpp:233-237); tile (tx, ty) belongs to rank mi_tile_owner(tx, ty, world) (include/pbrt_amd.h)
    const int tileSize = 16;
    int ex = sc.sample_max[0] - sc.sample_min[0], ey = sc.sample_max[1] - sc.sample_min[1];
    int nTx = (ex + tileSize - 1) / tileSize, nTy = (ey + tileSize - 1) / tileSize;
    if (c->filmShardWorld < 0) { c->filmShardRank = rank; c->filmShardWorld = world; }
    else if (c->filmShardRank != rank || c->filmShardWorld != world) c->filmMixed = true;   // a second shard accumulates into the same film
    if (c->tilesRank != rank || c->tilesWorld != world) {   // the owned-tile list changes only with the sharding: no allocation on repeated frames
        HIP_TRY(hipStreamSynchronize(c->stream));           // a pass still reading the old list / the host staging copy
        c->tilesHost.clear();
        c->tilesHost.resize((size_t)mi_owned_tiles(nTx, nTy, rank, world, nullptr));
        mi_owned_tiles(nTx, nTy, rank, world, c->tilesHost.data());
        c->tilesCount = c->tilesHost.size();
        if (c->tilesCount && upload(c, c->tiles, c->tilesHost.data(), c->tilesCount * sizeof(uint32_t))) return -1;
        c->tilesRank = rank; c->tilesWorld = world;
    }
    if (c->tilesCount == 0) return 0;
    if (MI_SAMPLER_IS_TILE_SERIAL(sc.sampler_type)) {
        // Tile-serial rounds.  These samplers give every tile ONE PCG32 stream, and what a sample receives depends on how many numbers all earlier
        // samples of its tile drew (path lengths, Russian roulette, specular flags ...): the reference's image needs each tile's pixels and samples in the
        // reference's order (integrator.cpp:259-325), one path after the other.  So a pass here is ONE sample of ONE pixel position of every owned tile
        // (tiles are independent: that is the parallelism there is) and a frame is 256 x spp passes -- correct and slow; Sobol' / Halton are the
        // samplers to render with.
        if (s0 != 0 || s1 != sc.spp) return fail("mi_render: the random / stratified / 02sequence samplers render whole frames only (spp_begin = 0, spp_end = spp)");
        if (ensure_state(c, (uint32_t)std::max<uint64_t>(c->tilesCount, 256 * 64))) return -1;
        const dim3 grid((unsigned)std::min<size_t>((c->tilesCount + PT_BLOCK - 1) / PT_BLOCK, (size_t)c->gridBlocks)), block(PT_BLOCK);
        hipLaunchKernelGGL(k_pix_seed, grid, block, 0, c->stream, sc, c->tiles.as<uint32_t>(), (uint32_t)c->tilesCount);
        for (uint32_t w = 0; w < 256; ++w) {
            if ((int)(w & 15) >= ex || (int)(w >> 4) >= ey) continue;   // no tile has this pixel
            hipLaunchKernelGGL(k_pix_start_pixel, grid, block, 0, c->stream, sc, c->tiles.as<uint32_t>(), (uint32_t)c->tilesCount, (uint32_t)nTx, w);
            for (int s = 0; s < sc.spp; ++s) {
                PassInfo pass;
                pass.tiles = c->tiles.as<uint32_t>();
                pass.n_tiles_x = (uint32_t)nTx;
                pass.pix0 = 0; pass.npix = (uint32_t)c->tilesCount;
                pass.s0 = (uint32_t)s; pass.ns = 1;
                pass.list_xy = nullptr; pass.list_s = nullptr;
                pass.serial = 1; pass.w = w;
                if (run_pass(c, pass, rp->count_work != 0, true)) return -1;
            }
        }
        return 0;
    }
    uint64_t npixOwned = (uint64_t)c->tilesCount * 256;
    // Paths in flight per pass.  Every launch of the wavefront pipeline ends with a tail (the longest rays) and starts with
    // fixed costs, so the pool is made as large as the frame allows -- up to 2^27 paths (37 GB of path state) and at most
    // 60 % of the free HBM: measured 145 -> 190 Msamples/s on the 1080p/64 spp frame going from 2^23 to a single pass.
    uint32_t cap = rp->max_paths_in_flight > 0 ? (uint32_t)rp->max_paths_in_flight : (1u << 27);
    if (rp->max_paths_in_flight <= 0) {
        size_t freeB = 0, totalB = 0;
        const size_t perPath = sizeof(PathRec) + sizeof(NeeRec) + sizeof(uint32_t) * 6 + sizeof(uint2) + ((c->volTr || c->sssWave) ? sizeof(TrState) + 2 * sizeof(uint32_t) : 0) +
                               (c->sssWave ? sizeof(SssRec) + 3 * sizeof(uint32_t) : 0);
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const size_t logB = SssLogBytes(c, cap);   // (the probe walk's hit lists do not scale with the pool: taken off the top -- ADVICE r4)
            size_t budget = (freeB > logB ? freeB - logB : 0) / 10 * 6 + (size_t)c->cap * perPath;   // what is allocated for path state now would be released
            cap = (uint32_t)std::min<size_t>(cap, std::max<size_t>(budget / perPath, 1u << 20));
        }
    }
    cap = std::max(cap, 256u * 64u);
    uint64_t want = std::min<uint64_t>(cap, npixOwned * (uint64_t)(s1 - s0));
    if (ensure_state(c, (uint32_t)std::max<uint64_t>(want, 256 * 64))) return -1;
    cap = c->cap;
    // chunking: whole pixel range x as many samples as fit, else pixel sub-ranges x 1 sample
    uint32_t pixPerPass = (uint32_t)std::min<uint64_t>(npixOwned, cap & ~255u);
    uint32_t sppPerPass = pixPerPass == npixOwned ? std::max<uint32_t>(1, cap / (uint32_t)npixOwned) : 1;
    for (uint64_t p0 = 0; p0 < npixOwned; p0 += pixPerPass) {
        uint32_t np = (uint32_t)std::min<uint64_t>(pixPerPass, npixOwned - p0);
        for (int s = s0; s < s1; s += (int)sppPerPass) {
            PassInfo pass;
            pass.tiles = c->tiles.as<uint32_t>();
            pass.n_tiles_x = (uint32_t)nTx;
            pass.pix0 = (uint32_t)p0; pass.npix = np;
            pass.s0 = (uint32_t)s; pass.ns = (uint32_t)std::min<int>((int)sppPerPass, s1 - s);
            pass.list_xy = nullptr; pass.list_s = nullptr;
            if (run_pass(c, pass, rp->count_work != 0, true)) return -1;
        }
    }
    return 0;
}

// A traversal wave that hit the non-termination guard dropped its rays: every frame rendered since the last mi_counters_reset is
// invalid.  mi_sync / mi_film_download / mi_film_gather report that as a FAILURE (no caller can hand out such a film by accident).
static int guard_check(mi_ctx *c, const char *who) {
    if (!c->counters.p) return 0;
    uint64_t trips = 0;
    HIP_TRY(hipMemcpyAsync(&trips, (const uint64_t *)c->counters.p + MI_CNT_TRACE_GUARD_TRIPS, sizeof(trips), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (trips) return fail(std::string(who) + ": " + std::to_string((unsigned long long)trips) + " traversal wave(s) hit the non-termination guard -- the frame is invalid");
    return 0;
}
int64_t mi_owned_tiles(int n_tiles_x, int n_tiles_y, int rank, int world, uint32_t *out) {
    if (world < 1) world = 1;
    if (n_tiles_x <= 0 || n_tiles_y <= 0 || rank < 0 || rank >= world) return 0;
    const int skew = mi_tile_skew(world);
    int64_t n = 0;
    for (int ty = 0; ty < n_tiles_y; ++ty)
        for (int tx = 0; tx < n_tiles_x; ++tx)
            if (mi_tile_owner(tx, ty, world, skew) == rank) { if (out) out[n] = (uint32_t)(ty * n_tiles_x + tx); ++n; }
    return n;
}
int mi_sync(mi_ctx *c) {
    if (!c) return fail("mi_sync: null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    harvest(c);
    return guard_check(c, "mi_sync");
}

int mi_film_clear(mi_ctx *c) {
    if (!c || !c->haveScene) return fail("mi_film_clear: no scene");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->filmPtr, 0, (size_t)c->filmPixels * sizeof(float4), c->stream));
    c->filmShardRank = c->filmShardWorld = -1;
    c->filmMixed = false;
    return 0;
}
int mi_film_bind(mi_ctx *c, void *p) {
    if (!c || !c->haveScene) return fail("mi_film_bind: no scene");
    float4 *q = p ? (float4 *)p : c->film.as<float4>();
    if (q != c->filmPtr) { c->filmShardRank = c->filmShardWorld = -1; c->filmMixed = true; }   // contents unknown until the caller clears it
    c->filmPtr = q;
    return 0;
}
int mi_film_download(mi_ctx *c, float *rgbw) {
    if (!c || !c->haveScene || !rgbw) return fail("mi_film_download: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(rgbw, c->filmPtr, (size_t)c->filmPixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    harvest(c);
    return guard_check(c, "mi_film_download");
}
void *mi_film_device_ptr(mi_ctx *c) { return c ? (void *)c->filmPtr : nullptr; }

// ---- mi_film_gather: RCCL reduction of the per-GPU films (see include/pbrt_amd.h).  RCCL is loaded with dlopen so that the
// library has no link-time dependency on it (single-GPU users never touch it); the few entry points used are declared here
// with the signatures of <rccl/rccl.h>.
namespace {
typedef struct ncclComm *ncclComm_t_;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t_ *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void *, void *, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int /*ncclDataType_t*/, int /*peer*/, ncclComm_t_, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::vector<int> devs;
    std::vector<ncclComm_t_> comms;
    bool load(std::string *err) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { *err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !Send || !Recv) { *err = "librccl.so lacks an expected symbol"; dlclose(lib); lib = nullptr; return false; }
        return true;
    }
    bool commsFor(const std::vector<int> &d, std::string *err) {
        if (d == devs && !comms.empty()) return true;
        for (ncclComm_t_ cm : comms) CommDestroy(cm);
        comms.assign(d.size(), nullptr);
        int rc = CommInitAll(comms.data(), (int)d.size(), d.data());
        if (rc != 0) { *err = std::string("ncclCommInitAll: ") + (GetErrorString ? GetErrorString(rc) : "error"); comms.clear(); devs.clear(); return false; }
        devs = d;
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rcclMutex;   // the cached communicators are shared by every caller of this process
// ONE group of point-to-point transfers (ncclGroupStart ... ncclSend / ncclRecv per move ... ncclGroupEnd): move m sends `count` floats from `src` (on the device and
// stream of communicator rank `from`) into `dst` on rank `to`.  No early return inside the group: GroupEnd always runs.  Caller holds the lock on g_rccl.
struct RcclMove { int from, to, fromDevice, toDevice; const void *src; void *dst; size_t count; hipStream_t fromStream, toStream; };
static int rccl_group_moves(const std::vector<RcclMove> &moves, std::string *err) {
    const int ncclFloat32_ = 7;   // ncclDataType_t value of <rccl/rccl.h>
    int rc = g_rccl.GroupStart();
    hipError_t he = hipSuccess;
    for (size_t k = 0; k < moves.size() && rc == 0 && he == hipSuccess; ++k) {
        const RcclMove &m = moves[k];
        he = hipSetDevice(m.fromDevice);
        if (he == hipSuccess) rc = g_rccl.Send(m.src, m.count, ncclFloat32_, m.to, g_rccl.comms[m.from], m.fromStream);
        if (he == hipSuccess && rc == 0) he = hipSetDevice(m.toDevice);
        if (he == hipSuccess && rc == 0) rc = g_rccl.Recv(m.dst, m.count, ncclFloat32_, m.from, g_rccl.comms[m.to], m.toStream);
    }
    int rc2 = g_rccl.GroupEnd();
    if (he != hipSuccess) { *err = std::string("hipSetDevice: ") + hipGetErrorString(he); return -1; }
    if (rc != 0 || rc2 != 0) { *err = std::string("ncclSend / ncclRecv: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "error"); return -1; }
    return 0;
}
// the sparse film exchange (mi_film_gather): pack the FilmTilePixels a context's samples can reach / add a packed list into the root film
__global__ void __launch_bounds__(PT_BLOCK) k_film_pack(const float4 *film, const uint32_t *idx, int64_t n, float4 *out) {
    for (int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT_BLOCK) out[i] = film[idx[i]];
}
__global__ void __launch_bounds__(PT_BLOCK) k_film_add_packed(float4 *film, const uint32_t *idx, int64_t n, const float4 *in) {   // idx ascending and unique: no two threads touch one pixel
    for (int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT_BLOCK) {
        float4 a = film[idx[i]], b = in[i];
        film[idx[i]] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
__global__ void __launch_bounds__(PT_BLOCK) k_film_add(float4 *dst, const float4 *src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PT_BLOCK) {
        float4 a = dst[i], b = src[i];
        dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
}  // namespace

int mi_trace_info(mi_ctx *c, int64_t out[8]) {
    if (!c || !out || !c->haveScene) return fail("mi_trace_info: no scene");
    int mode = c->hasInst ? 4 : (c->useQ ? 5 : 0);
    out[0] = mode;
    out[1] = mode == 5 ? (int64_t)sizeof(BVH4QNode) : 128;
    out[2] = c->sc.n_nodes;
    const bool mid = mode == 5 && (c->hasSpheres != c->hasAlpha) && TraceShape<0, false, true, true>::MID;
    const bool big = mid || (mode == 5 && !c->hasSpheres && !c->hasAlpha && TraceShape<0, false, false, true>::BIG);
    out[3] = big ? TraceShape<0, false, false, true>::NLDS : PT_LDS_STACK;
    out[4] = c->sc.n_hot;
    out[5] = (int64_t)(c->hotProbeShare * 1e6);
    out[6] = mid ? TraceShape<0, false, true, true>::BLOCK : (big ? TraceShape<0, false, false, true>::BLOCK : PT_BLOCK);
    out[7] = mid ? TraceShape<0, false, true, true>::PER_CU : (big ? TraceShape<0, false, false, true>::PER_CU : PT_GRID_PER_CU);
    return 0;
}

// The FilmTilePixels the samples of rank `rank` of `world` can contribute to, as ascending indices into the cropped film: its tiles (mi_tile_owner) grown by
// floor(radius + 1/2) pixels per axis and clipped to the film -- what Film::GetFilmTile computes for one tile (core/film.cpp:95-106); parallel.reach_pixels is the same rule.
static void reach_indices(const DevScene &sc, int rank, int world, std::vector<uint32_t> *out) {
    const int sx0 = sc.sample_min[0], sy0 = sc.sample_min[1], sx1 = sc.sample_max[0], sy1 = sc.sample_max[1];
    const int ntx = (sx1 - sx0 + 15) / 16, nty = (sy1 - sy0 + 15) / 16;
    const int W = sc.crop_max[0] - sc.crop_min[0], H = sc.crop_max[1] - sc.crop_min[1];
    const int hx = (int)std::floor(sc.filter_radius[0] + 0.5f), hy = (int)std::floor(sc.filter_radius[1] + 0.5f);
    std::vector<uint8_t> mask((size_t)std::max(0, W) * (size_t)std::max(0, H), 0);
    const int skew = mi_tile_skew(world);
    for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
            if (mi_tile_owner(tx, ty, world, skew) != rank) continue;
            const int x0 = std::max(0, sx0 + 16 * tx - hx - sc.crop_min[0]), y0 = std::max(0, sy0 + 16 * ty - hy - sc.crop_min[1]);
            const int x1 = std::min(W, std::min(sx0 + 16 * tx + 16, sx1) + hx - sc.crop_min[0]), y1 = std::min(H, std::min(sy0 + 16 * ty + 16, sy1) + hy - sc.crop_min[1]);
            for (int y = y0; y < y1; ++y) for (int x = x0; x < x1; ++x) mask[(size_t)y * W + x] = 1;
        }
    out->clear();
    for (size_t i = 0; i < mask.size(); ++i) if (mask[i]) out->push_back((uint32_t)i);
}

// Round 5: the exchange is SPARSE, like parallel.FilmExchange -- context i packs the pixels its samples can reach (its tiles of the last mi_render + the filter's ring: 1.27 / N
// of the film under the box filter) and the root adds them into its film in context order, exact for every filter because a context's film is zero everywhere else.  Contexts
// on distinct devices move the packed lists with ONE group of ncclSend / ncclRecv (N - 1 point-to-point transfers over N - 1 xGMI links into the root's GPU; rounds 1-4:
// ncclReduce of N full films); contexts sharing a device (one-GPU boxes) hand the root their packed list directly, contexts on another device without a full set of distinct
// devices through hipMemcpyPeer.  A context that has not rendered a shard (world 1) takes the dense add.
int mi_film_gather(mi_ctx **ctxs, int n, int root) {
    if (!ctxs || n < 1 || root < 0 || root >= n) return fail("mi_film_gather: bad argument");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->haveScene) return fail("mi_film_gather: context without a scene");
        if (ctxs[i]->filmPixels != ctxs[root]->filmPixels) return fail("mi_film_gather: films of different size");
    }
    for (int i = 0; i < n; ++i) if (mi_sync(ctxs[i])) return -1;   // the renders (each on its own ctx stream)
    if (n == 1) return 0;
    mi_ctx *r = ctxs[root];
    std::vector<int> devs(n);
    bool distinct = true;
    for (int i = 0; i < n; ++i) { devs[i] = ctxs[i]->device; for (int j = 0; j < i; ++j) distinct &= devs[j] != devs[i]; }
    // per sender: the reach list on its own device and on the root's, the packed pixels, the root's receive buffer -- cached in the sender's context (mi_ctx::gather)
    typedef mi_ctx::GatherPart Part;
    std::vector<Part *> parts(n, nullptr);
    std::vector<char> dense(n, 0);
    bool built = false;
    for (int i = 0; i < n; ++i) {
        if (i == root) continue;
        mi_ctx *c = ctxs[i];
        // the whole film may carry samples: no shard rendered (world 1), or more than one sharding since the last clear (the sparse form would drop them)
        dense[i] = c->filmMixed || c->filmShardWorld <= 1 || c->filmShardRank < 0;
        if (dense[i]) continue;
        Part &p = c->gather;
        parts[i] = &p;
        if (p.rank != c->filmShardRank || p.world != c->filmShardWorld || p.pixels != c->filmPixels || p.rootDevice != r->device) {
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));   // nobody still reads the buffers about to be replaced
            p.rank = -1;
            reach_indices(c->sc, c->filmShardRank, c->filmShardWorld, &p.idx);
            p.n = p.idx.size();
            if (upload(c, p.dIdxSrc, p.idx.data(), p.n * sizeof(uint32_t)) || p.packed.alloc(p.n * sizeof(float4))) return -1;
            p.dIdxRoot.release(); p.recv.release();
            if (c->device != r->device) {
                HIP_TRY(hipSetDevice(r->device));
                if (upload(r, p.dIdxRoot, p.idx.data(), p.n * sizeof(uint32_t)) || p.recv.alloc(p.n * sizeof(float4))) return -1;
            }
            p.rank = c->filmShardRank; p.world = c->filmShardWorld; p.pixels = c->filmPixels; p.rootDevice = r->device;
            ++c->gatherBuilds;
            built = true;
        }
        if (!p.n) continue;
        HIP_TRY(hipSetDevice(c->device));
        hipLaunchKernelGGL(k_film_pack, dim3(c->gridBlocks), dim3(PT_BLOCK), 0, c->stream, (const float4 *)c->filmPtr, p.dIdxSrc.as<uint32_t>(), (int64_t)p.n, p.packed.as<float4>());
        HIP_TRY(hipGetLastError());
    }
    // the packed lists (and, after a rebuild, the index uploads) are in place before the root's stream reads them
    for (int i = 0; i < n; ++i) if (i != root || built) { HIP_TRY(hipSetDevice(ctxs[i]->device)); HIP_TRY(hipStreamSynchronize(ctxs[i]->stream)); }
    bool viaRccl = false;
    if (distinct) {
        bool anySparse = false;
        for (int i = 0; i < n; ++i) anySparse = anySparse || (parts[i] && parts[i]->n);
        if (anySparse) {
            std::string err;
            std::lock_guard<std::mutex> lock(g_rcclMutex);
            if (!g_rccl.load(&err) || !g_rccl.commsFor(devs, &err)) return fail("mi_film_gather: " + err);
            std::vector<RcclMove> moves;
            for (int i = 0; i < n; ++i)
                if (parts[i] && parts[i]->n) moves.push_back({i, root, ctxs[i]->device, r->device, parts[i]->packed.p, parts[i]->recv.p, parts[i]->n * 4, ctxs[i]->stream, r->stream});
            if (rccl_group_moves(moves, &err)) return fail("mi_film_gather: " + err);
            viaRccl = true;
        }
    }
    HIP_TRY(hipSetDevice(r->device));
    DevBuf stage;
    const size_t count = (size_t)r->filmPixels * 4;   // floats of a whole film (dense parts)
    for (int i = 0; i < n; ++i) {   // context order: the sum is deterministic
        if (i == root) continue;
        mi_ctx *c = ctxs[i];
        if (dense[i]) {
            const float4 *src = c->filmPtr;
            if (c->device != r->device) {
                if (!stage.p && stage.alloc(count * sizeof(float))) return -1;
                HIP_TRY(hipMemcpyPeerAsync(stage.p, r->device, c->filmPtr, c->device, count * sizeof(float), r->stream));
                src = stage.as<float4>();
            }
            hipLaunchKernelGGL(k_film_add, dim3(r->gridBlocks), dim3(PT_BLOCK), 0, r->stream, r->filmPtr, src, (int64_t)r->filmPixels);
            HIP_TRY(hipGetLastError());
            if (c->device != r->device) HIP_TRY(hipStreamSynchronize(r->stream));   // the staging buffer is reused
            continue;
        }
        Part &p = *parts[i];
        if (!p.n) continue;
        const uint32_t *idx = c->device == r->device ? p.dIdxSrc.as<uint32_t>() : p.dIdxRoot.as<uint32_t>();
        const float4 *in = c->device == r->device ? p.packed.as<float4>() : p.recv.as<float4>();
        if (c->device != r->device && !viaRccl) HIP_TRY(hipMemcpyPeerAsync(p.recv.p, r->device, p.packed.p, c->device, p.n * sizeof(float4), r->stream));
        hipLaunchKernelGGL(k_film_add_packed, dim3(r->gridBlocks), dim3(PT_BLOCK), 0, r->stream, r->filmPtr, idx, (int64_t)p.n, in);
        HIP_TRY(hipGetLastError());
    }
    // the root's film now holds every sharding that was added: a later gather FROM it must take the dense form
    r->filmMixed = true;
    for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(ctxs[i]->device)); HIP_TRY(hipStreamSynchronize(ctxs[i]->stream)); }
    HIP_TRY(hipSetDevice(r->device));
    stage.release();
    return 0;
}
int64_t mi_film_pixel_count(mi_ctx *c) { return c ? c->filmPixels : 0; }

// mi_film_gather's RCCL step on one GPU (include/pbrt_amd.h): communicator of one device, one grouped send / recv of packed pixels to itself, the gather's add kernel
int mi_rccl_probe(mi_ctx *c, int64_t nPix) {
    if (!c || nPix < 1 || nPix > (int64_t)1 << 28) return fail("mi_rccl_probe: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float> host((size_t)nPix * 4), back((size_t)nPix * 4);
    uint32_t st = 12345u;
    for (float &v : host) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) * (1.0f / 16777216.0f) + 0.25f; }
    std::vector<uint32_t> idx((size_t)nPix);
    for (int64_t i = 0; i < nPix; ++i) idx[(size_t)i] = (uint32_t)(nPix - 1 - i);   // a permutation: the add kernel scatters
    DevBuf packed, recv, film, dIdx;
    if (upload(c, packed, host.data(), host.size() * sizeof(float)) || recv.alloc(host.size() * sizeof(float)) || film.alloc(host.size() * sizeof(float)) ||
        upload(c, dIdx, idx.data(), idx.size() * sizeof(uint32_t))) return -1;
    HIP_TRY(hipMemsetAsync(recv.p, 0, recv.bytes, c->stream));
    HIP_TRY(hipMemsetAsync(film.p, 0, film.bytes, c->stream));
    {
        std::string err;
        std::lock_guard<std::mutex> lock(g_rcclMutex);
        if (!g_rccl.load(&err) || !g_rccl.commsFor(std::vector<int>{c->device}, &err)) return fail("mi_rccl_probe: " + err);
        if (rccl_group_moves({{0, 0, c->device, c->device, packed.p, recv.p, (size_t)nPix * 4, c->stream, c->stream}}, &err)) return fail("mi_rccl_probe: " + err);
    }
    hipLaunchKernelGGL(k_film_add_packed, dim3(c->gridBlocks), dim3(PT_BLOCK), 0, c->stream, film.as<float4>(), dIdx.as<uint32_t>(), nPix, (const float4 *)recv.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(back.data(), film.p, back.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < nPix; ++i)
        if (std::memcmp(&back[(size_t)idx[(size_t)i] * 4], &host[(size_t)i * 4], 16) != 0) return fail("mi_rccl_probe: pixel " + std::to_string(i) + " did not arrive intact");
    return 0;
}

int mi_counters(mi_ctx *c, uint64_t out[MI_CNT_COUNT]) {
    if (!c || !out) return fail("mi_counters: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(out, c->counters.p, MI_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
#if PT_SHADE_PROF
    {   // developer build only: per-phase wave cycles of k_shade
        uint64_t all[PT_CNT_ALLOC];
        HIP_TRY(hipMemcpyAsync(all, c->counters.p, sizeof(all), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 24; ++k)
            if (all[40 + k]) fprintf(stderr, "[shade-prof] phase %2d: visits %12llu  cycles/visit %10.1f  lanes %5.1f  total Gcycles %8.3f\n", k,
                                     (unsigned long long)all[40 + k], (double)all[16 + k] / (double)all[40 + k], (double)all[68 + k] / (double)all[40 + k], (double)all[16 + k] * 1e-9);
    }
#endif
    HIP_TRY(hipStreamSynchronize(c->stream));
    out[MI_CNT_FILM_GATHER_BUILDS] = c->gatherBuilds;   // host-side: not a device counter
    return 0;
}
int mi_trace_clock(mi_ctx *c, double out[4]) {
    if (!c || !out) return fail("mi_trace_clock: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    uint64_t t[4];
    HIP_TRY(hipMemcpyAsync(t, (const uint64_t *)c->counters.p + PT_CNT_CLK, sizeof(t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int wallKHz = 0;
    if (hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || wallKHz <= 0) wallKHz = 100000;
    out[0] = t[1] ? (double)t[0] / (double)t[1] * wallKHz * 1e-6 : 0.0;   // GHz, closest hit
    out[1] = t[3] ? (double)t[2] / (double)t[3] * wallKHz * 1e-6 : 0.0;   // GHz, any hit
    out[2] = (double)t[0];
    out[3] = (double)t[2];
    return 0;
}
int mi_bvh4_validate(const mi_scene_desc *d, int64_t stats[8]) {
    if (!d || !stats) return fail("mi_bvh4_validate: null argument");
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    if (!d->n_bvh_nodes) return 0;
    B4Builder bb;
    bb.ownTopology = B4Builder::wantOwnTopology(d->n_instances > 0);   // the tree mi_scene_upload would build
    int topDepth = 0, objDepth = 0;
    std::vector<uint32_t> objRoot;
    {
        std::string err;
        if (!bb.buildScene(d, &objRoot, &topDepth, &objDepth, &err)) return fail("mi_bvh4_validate: " + err);
    }
    std::vector<uint8_t> covered(d->n_tris, 0);
    int64_t leaves = 0, maxDepth = 0;
    struct Item { uint32_t node; int depth; };
    std::vector<Item> st{{0u, 0}};
    for (uint32_t r : objRoot) st.push_back({r, 0});   // two-level scenes: the instanced objects' trees, each from depth 0
    auto primBox = [&](uint32_t t, float lo[3], float hi[3]) -> bool {   // false for spheres and instances (their box is the reference node's)
        const uint32_t *v = d->tri_indices + 3 * (size_t)t;
        if (v[0] == MI_PRIM_SPHERE || v[0] == MI_PRIM_INSTANCE) return false;
        for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
        for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) { float x = d->P[3 * (size_t)v[k] + a]; lo[a] = std::min(lo[a], x); hi[a] = std::max(hi[a], x); }
        return true;
    };
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node >= bb.out.size()) return fail("mi_bvh4_validate: child index out of range");
        maxDepth = std::max<int64_t>(maxDepth, it.depth);
        const BVH4Node &n = bb.out[it.node];
        for (int k = 0; k < 4; ++k) {
            uint32_t c = n.child[k];
            if (c == BVH4_EMPTY) { if (!(n.lox[k] > n.hix[k])) return fail("mi_bvh4_validate: empty slot without an inverted box"); continue; }
            if (c & BVH4_LEAF) {
                uint32_t first = c & BVH4_FIRST_MASK, count = ((c >> 27) & 0xfu) + 1;
                ++leaves;
                if (count > BVH4_LEAF_MAX || first + count > d->n_tris) return fail("mi_bvh4_validate: bad leaf reference");
                for (uint32_t t = first; t < first + count; ++t) {
                    if (covered[t]++) return fail("mi_bvh4_validate: primitive referenced twice");
                    float lo[3], hi[3];
                    if (primBox(t, lo, hi))
                        if (lo[0] < n.lox[k] || lo[1] < n.loy[k] || lo[2] < n.loz[k] || hi[0] > n.hix[k] || hi[1] > n.hiy[k] || hi[2] > n.hiz[k])
                            return fail("mi_bvh4_validate: primitive outside its leaf box");
                }
            } else {
                const BVH4Node &ch = bb.out[c];   // the child's own children must lie inside the box the parent holds for it
                for (int j = 0; j < 4; ++j) {
                    if (ch.child[j] == BVH4_EMPTY) continue;
                    if (ch.lox[j] < n.lox[k] || ch.loy[j] < n.loy[k] || ch.loz[j] < n.loz[k] || ch.hix[j] > n.hix[k] || ch.hiy[j] > n.hiy[k] || ch.hiz[j] > n.hiz[k])
                        return fail("mi_bvh4_validate: child box not contained in its parent's");
                }
                st.push_back({c, it.depth + 1});
            }
        }
    }
    int64_t ncov = 0;
    for (uint8_t cflag : covered) ncov += cflag;
    if (ncov != (int64_t)d->n_tris) return fail("mi_bvh4_validate: " + std::to_string((int64_t)d->n_tris - ncov) + " primitives not covered by any leaf");
    if (maxDepth != std::max(topDepth, objDepth)) return fail("mi_bvh4_validate: depth bookkeeping differs from the tree");
    stats[0] = (int64_t)bb.out.size(); stats[1] = leaves; stats[2] = maxDepth; stats[3] = B4Builder::stackNeed(topDepth, objDepth, d->n_instances > 0); stats[4] = ncov;
    stats[5] = (int64_t)objRoot.size();
    stats[6] = bb.rebuilt ? 1 : 0;   // the library's own topology over the reference's leaves (pt_treebuild.h)
    return 0;
}
__global__ void __launch_bounds__(PT_BLOCK) k_stage_spheres(const mi_sphere *spheres, const mi_ray *rays, int64_t n, mi_sphere_hit *hits) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    mi_ray r = rays[i];
    SphereIsectOut o;
    SphereIsect(spheres + i, V3(r.o[0], r.o[1], r.o[2]), V3(r.d[0], r.d[1], r.d[2]), r.tmax, &o);
    mi_sphere_hit h;
    std::memset(&h, 0, sizeof(h));
    if (o.hit) {
        h.hit = 1;
        h.t = SphereIntersectT(spheres + i, V3(r.o[0], r.o[1], r.o[2]), V3(r.d[0], r.d[1], r.d[2]), r.tmax);
        h.p[0] = o.p.x; h.p[1] = o.p.y; h.p[2] = o.p.z;
        h.p_error[0] = o.pError.x; h.p_error[1] = o.pError.y; h.p_error[2] = o.pError.z;
        h.n[0] = o.n.x; h.n[1] = o.n.y; h.n[2] = o.n.z;
    }
    hits[i] = h;
}
int mi_sphere_intersect(int device, const mi_sphere *spheres, const mi_ray *rays, int64_t n, mi_sphere_hit *hits) {
    if (!spheres || !rays || !hits || n < 0) return fail("mi_sphere_intersect: bad argument");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("mi_sphere_intersect: no HIP device (this library has no CPU path)");
    HIP_TRY(hipSetDevice(device));
    DevBuf ds, dr, dh;
    if (ds.alloc((size_t)n * sizeof(mi_sphere)) || dr.alloc((size_t)n * sizeof(mi_ray)) || dh.alloc((size_t)n * sizeof(mi_sphere_hit))) return -1;
    HIP_TRY(hipMemcpy(ds.p, spheres, (size_t)n * sizeof(mi_sphere), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, rays, (size_t)n * sizeof(mi_ray), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_spheres, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, 0, ds.as<mi_sphere>(), dr.as<mi_ray>(), n, dh.as<mi_sphere_hit>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(hits, dh.p, (size_t)n * sizeof(mi_sphere_hit), hipMemcpyDeviceToHost));
    return 0;
}
// stage-level BxDF evaluation: f / Pdf / Sample_f of one lobe per record, per-lane lobe records (the instantiation the textured shading
// kernel uses; the constant-material kernel runs the same code on wave-uniform records)
__global__ void __launch_bounds__(PT_BLOCK) k_stage_bxdf(const mi_bxdf *b, const float *wo, const float *wi, const float *u, int64_t n, float *f, float *pdf, float *wi_s,
                                                         float *pdf_s, float *f_s, int32_t *type_s) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const mi_bxdf *bp = b + i;
    V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
    RGB fv = BxdfF<false>(bp, o, w);
    f[3 * i] = fv.r; f[3 * i + 1] = fv.g; f[3 * i + 2] = fv.b;
    pdf[i] = BxdfPdf<false>(bp, o, w);
    BxdfSample s = BxdfSample_f<false>(bp, o, u[2 * i], u[2 * i + 1], BxdfFlags(bp->type));
    wi_s[3 * i] = s.wi.x; wi_s[3 * i + 1] = s.wi.y; wi_s[3 * i + 2] = s.wi.z;
    pdf_s[i] = s.pdf;
    f_s[3 * i] = s.f.r; f_s[3 * i + 1] = s.f.g; f_s[3 * i + 2] = s.f.b;
    type_s[i] = s.sampledType;
}
int mi_bxdf_eval(int device_ordinal, const mi_bxdf *bxdfs, const float *wo, const float *wi, const float *u, int64_t n, float *f, float *pdf, float *wi_s, float *pdf_s,
                 float *f_s, int32_t *type_s) {
    if (!bxdfs || !wo || !wi || !u || !f || !pdf || !wi_s || !pdf_s || !f_s || !type_s || n < 0) return fail("mi_bxdf_eval: bad argument");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return fail("mi_bxdf_eval: no such HIP device (this library has no CPU fallback)");
    HIP_TRY(hipSetDevice(device_ordinal));
    DevBuf db, dwo, dwi, du, df, dp, dws, dps, dfs, dts;
    size_t N = (size_t)n;
    if (db.alloc(N * sizeof(mi_bxdf)) || dwo.alloc(N * 12) || dwi.alloc(N * 12) || du.alloc(N * 8) || df.alloc(N * 12) || dp.alloc(N * 4) || dws.alloc(N * 12) ||
        dps.alloc(N * 4) || dfs.alloc(N * 12) || dts.alloc(N * 4)) return -1;
    HIP_TRY(hipMemcpy(db.p, bxdfs, N * sizeof(mi_bxdf), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dwo.p, wo, N * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dwi.p, wi, N * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(du.p, u, N * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_bxdf, dim3((unsigned)((N + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, 0, db.as<mi_bxdf>(), dwo.as<float>(), dwi.as<float>(), du.as<float>(), n,
                       df.as<float>(), dp.as<float>(), dws.as<float>(), dps.as<float>(), dfs.as<float>(), dts.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(f, df.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf, dp.p, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(wi_s, dws.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pdf_s, dps.p, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(f_s, dfs.p, N * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(type_s, dts.p, N * 4, hipMemcpyDeviceToHost));
    for (DevBuf *x : {&db, &dwo, &dwi, &du, &df, &dp, &dws, &dps, &dfs, &dts}) x->release();
    return 0;
}
// stage-level light sampling: Sample_Li / Pdf_Li of one light of the uploaded scene per record, the routines k_shade calls
__global__ void __launch_bounds__(PT_BLOCK) k_stage_light(DevScene sc, const mi_light_query *q, int64_t n, mi_light_result *out) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const mi_light_query &qi = q[i];
    const DevLight *dl = sc.lights + qi.light;
    V3 p(qi.p[0], qi.p[1], qi.p[2]), nr(qi.n[0], qi.n[1], qi.n[2]), zero(0.f, 0.f, 0.f), w(qi.wi[0], qi.wi[1], qi.wi[2]);
    LightSample ls = SampleLiAny(GeomTables(sc), dl, p, zero, nr, qi.u[0], qi.u[1]);
    mi_light_result r;
    r.wi[0] = ls.wi.x; r.wi[1] = ls.wi.y; r.wi[2] = ls.wi.z;
    r.pdf = ls.pdf;
    r.Li[0] = ls.Li.r; r.Li[1] = ls.Li.g; r.Li[2] = ls.Li.b;
    r.ray_o[0] = ls.shadow.o.x; r.ray_o[1] = ls.shadow.o.y; r.ray_o[2] = ls.shadow.o.z;
    r.ray_d[0] = ls.shadow.d.x; r.ray_d[1] = ls.shadow.d.y; r.ray_d[2] = ls.shadow.d.z;
    r.ray_tmax = ls.shadow.tMax;
    r.delta = ls.delta ? 1 : 0;
    r.pdf_wi = PdfLiAny(GeomTables(sc), dl, p, zero, nr, w);
    RGB le = dl->type == MI_LIGHT_INFINITE ? InfiniteLe(dl, w) : RGB(0.f);
    r.le_wi[0] = le.r; r.le_wi[1] = le.g; r.le_wi[2] = le.b;
    out[i] = r;
}
int mi_light_sample(mi_ctx *c, const mi_light_query *queries, int64_t n, mi_light_result *out) {
    if (!c || !queries || !out || n < 0) return fail("mi_light_sample: bad argument");
    if (!c->haveScene) return fail("mi_light_sample: no scene uploaded");
    for (int64_t i = 0; i < n; ++i)
        if (queries[i].light < 0 || (uint32_t)queries[i].light >= c->sc.n_lights) return fail("mi_light_sample: light index out of range");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dq, dr;
    if (dq.alloc((size_t)n * sizeof(mi_light_query)) || dr.alloc((size_t)n * sizeof(mi_light_result))) return -1;
    HIP_TRY(hipMemcpyAsync(dq.p, queries, (size_t)n * sizeof(mi_light_query), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_stage_light, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, c->stream, c->sc, dq.as<mi_light_query>(), n, dr.as<mi_light_result>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, dr.p, (size_t)n * sizeof(mi_light_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
// stage-level row f4: the radial profile of the tabulated BSSRDF and the phase function, the routines k_shade_vol calls
__global__ void __launch_bounds__(PT_BLOCK) k_stage_bssrdf(const DevBssrdfTable *tb, float eta, const mi_bssrdf_query *q, int64_t n, mi_bssrdf_result *out) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const mi_bssrdf_query &x = q[i];
    DevBSSRDF bs;
    bs.table = tb; bs.eta = eta; bs.material = 0;
    bs.poP = V3(0.f, 0.f, 0.f); bs.ns = V3(0.f, 0.f, 1.f); bs.ss = V3(1.f, 0.f, 0.f); bs.ts = V3(0.f, 1.f, 0.f);
    RGB sa(x.sigma_a[0], x.sigma_a[1], x.sigma_a[2]), ss(x.sigma_s[0], x.sigma_s[1], x.sigma_s[2]);
    bs.sigma_t = sa + ss;
    bs.rho = RGB(bs.sigma_t.r != 0 ? ss.r / bs.sigma_t.r : 0.f, bs.sigma_t.g != 0 ? ss.g / bs.sigma_t.g : 0.f, bs.sigma_t.b != 0 ? ss.b / bs.sigma_t.b : 0.f);
    mi_bssrdf_result r;
    RGB sr = BssrdfSr(&bs, x.r);
    r.sr[0] = sr.r; r.sr[1] = sr.g; r.sr[2] = sr.b;
    r.sample_sr = BssrdfSample_Sr(&bs, x.ch, x.u);
    r.pdf_sr = BssrdfPdf_Sr(&bs, x.ch, x.r);
    for (int c = 0; c < 3; ++c) {   // SubsurfaceFromDiffuse
        Float rho = InvertCatmullRom(tb->n_rho, tb->rho_samples, tb->rho_eff, x.kd[c]);
        r.sigma_s[c] = rho / x.mfp[c];
        r.sigma_a[c] = (1 - rho) / x.mfp[c];
    }
    out[i] = r;
}
__global__ void __launch_bounds__(PT_BLOCK) k_stage_hg(const mi_hg_query *q, int64_t n, mi_hg_result *out) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const mi_hg_query &x = q[i];
    V3 wo(x.wo[0], x.wo[1], x.wo[2]), wi(x.wi[0], x.wi[1], x.wi[2]), ws;
    mi_hg_result r;
    r.p = HGp(x.g, wo, wi);
    r.p_s = HGSample_p(x.g, wo, &ws, x.u[0], x.u[1]);
    r.wi_s[0] = ws.x; r.wi_s[1] = ws.y; r.wi_s[2] = ws.z;
    out[i] = r;
}
// Stage entry for the libm restatement (csrc/pt_libm.h): routine `fn` over n floats -- the -m gpu test compares the device's results with the GPU box's own glibc.
__global__ void __launch_bounds__(PT_BLOCK) k_stage_libm(int fn, const float *a, const float *b, int64_t n, float *out, float *out2) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    float r = 0, r2 = 0;
    switch (fn) {
    case MI_LIBM_SINF: r = pt_sinf(x); break;
    case MI_LIBM_COSF: r = pt_cosf(x); break;
    case MI_LIBM_SINCOSF: pt_sincosf(x, &r, &r2); break;
    case MI_LIBM_EXPF: r = pt_expf(x); break;
    case MI_LIBM_LOGF: r = pt_logf(x); break;
    case MI_LIBM_ACOSF: r = pt_acosf(x); break;
    case MI_LIBM_ATANF: r = pt_atanf(x); break;
    case MI_LIBM_ATAN2F: r = pt_atan2f(x, b[i]); break;
    }
    out[i] = r;
    if (out2) out2[i] = r2;
}
static int stage_device(int device_ordinal, const char *who) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return fail(std::string(who) + ": no such HIP device (this library has no CPU fallback)");
    HIP_TRY(hipSetDevice(device_ordinal));
    return 0;
}
int mi_bssrdf_eval(int device_ordinal, const mi_bssrdf_table *t, float eta, const mi_bssrdf_query *queries, int64_t n, mi_bssrdf_result *out) {
    if (!t || !queries || !out || n < 0 || t->n_rho < 4 || t->n_radius < 4) return fail("mi_bssrdf_eval: bad argument");
    if (n == 0) return 0;
    if (stage_device(device_ordinal, "mi_bssrdf_eval")) return -1;
    DevBuf a[5], dt, dq, dr;
    const float *src[5] = {t->rho_samples, t->radius_samples, t->profile, t->rho_eff, t->profile_cdf};
    const size_t cnt[5] = {(size_t)t->n_rho, (size_t)t->n_radius, (size_t)t->n_rho * t->n_radius, (size_t)t->n_rho, (size_t)t->n_rho * t->n_radius};
    for (int k = 0; k < 5; ++k) {
        if (a[k].alloc(cnt[k] * 4)) return -1;
        HIP_TRY(hipMemcpy(a[k].p, src[k], cnt[k] * 4, hipMemcpyHostToDevice));
    }
    DevBssrdfTable h;
    h.n_rho = t->n_rho; h.n_radius = t->n_radius;
    h.rho_samples = a[0].as<float>(); h.radius_samples = a[1].as<float>(); h.profile = a[2].as<float>(); h.rho_eff = a[3].as<float>(); h.profile_cdf = a[4].as<float>();
    if (dt.alloc(sizeof(h)) || dq.alloc((size_t)n * sizeof(mi_bssrdf_query)) || dr.alloc((size_t)n * sizeof(mi_bssrdf_result))) return -1;
    HIP_TRY(hipMemcpy(dt.p, &h, sizeof(h), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dq.p, queries, (size_t)n * sizeof(mi_bssrdf_query), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_bssrdf, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, 0, dt.as<DevBssrdfTable>(), eta, dq.as<mi_bssrdf_query>(), n, dr.as<mi_bssrdf_result>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dr.p, (size_t)n * sizeof(mi_bssrdf_result), hipMemcpyDeviceToHost));
    return 0;
}
int mi_phase_hg(int device_ordinal, const mi_hg_query *queries, int64_t n, mi_hg_result *out) {
    if (!queries || !out || n < 0) return fail("mi_phase_hg: bad argument");
    if (n == 0) return 0;
    if (stage_device(device_ordinal, "mi_phase_hg")) return -1;
    DevBuf dq, dr;
    if (dq.alloc((size_t)n * sizeof(mi_hg_query)) || dr.alloc((size_t)n * sizeof(mi_hg_result))) return -1;
    HIP_TRY(hipMemcpy(dq.p, queries, (size_t)n * sizeof(mi_hg_query), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_hg, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, 0, dq.as<mi_hg_query>(), n, dr.as<mi_hg_result>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dr.p, (size_t)n * sizeof(mi_hg_result), hipMemcpyDeviceToHost));
    return 0;
}
int mi_libm_eval(int device_ordinal, int fn, const float *a, const float *b, int64_t n, float *out, float *out2) {
    if (!a || !out || n < 0 || fn < MI_LIBM_SINF || fn > MI_LIBM_ATAN2F || (fn == MI_LIBM_ATAN2F && !b) || (fn == MI_LIBM_SINCOSF && !out2)) return fail("mi_libm_eval: bad argument");
    if (n == 0) return 0;
    if (stage_device(device_ordinal, "mi_libm_eval")) return -1;
    DevBuf da, db, dr, dr2;
    const size_t bytes = (size_t)n * sizeof(float);
    if (da.alloc(bytes) || dr.alloc(bytes) || (b && db.alloc(bytes)) || (out2 && dr2.alloc(bytes))) return -1;
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_libm, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, 0, fn, da.as<float>(), b ? db.as<float>() : nullptr, n, dr.as<float>(), out2 ? dr2.as<float>() : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dr.p, bytes, hipMemcpyDeviceToHost));
    if (out2) HIP_TRY(hipMemcpy(out2, dr2.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}
// host check of the 64-byte quantised BVH4 (pt_bvh4q.h): quantisation in exact arithmetic + the kernel's per-ray state machine on the host
int mi_bvh4q_validate(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int any_hit, mi_hit *hits, int64_t stats[8]) {
    if (!d || !stats || (n > 0 && !rays)) return fail("mi_bvh4q_validate: null argument");
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    if (d->n_instances) return fail("mi_bvh4q_validate: two-level scenes are not handled");
    if (!d->n_bvh_nodes) {   // no geometry: every ray misses
        if (hits) for (int64_t i = 0; i < n; ++i) { std::memset(&hits[i], 0, sizeof(mi_hit)); hits[i].prim = -1; }
        return 0;
    }
    B4Builder bb;
    bb.ownTopology = B4Builder::wantOwnTopology(false);   // the tree mi_scene_upload would build
    int topDepth = 0, objDepth = 0;
    std::vector<uint32_t> objRoot;
    { std::string err; if (!bb.buildScene(d, &objRoot, &topDepth, &objDepth, &err)) return fail("mi_bvh4q_validate: " + err); }
    std::vector<BVH4QNode> qn;
    Bvh4qGrid g;
    std::string err;
    if (!bvh4q::quantise(bb.out, d->bvh_nodes[0].bmin, d->bvh_nodes[0].bmax, &qn, &g, &err)) return fail("mi_bvh4q_validate: " + err);
    bvh4q::Stats st;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t prim; float t, b[3];
        bool hit = bvh4q::traverse(d, qn, g, rays[i], any_hit != 0, &prim, &t, b, &st);
        if (hits) {
            std::memset(&hits[i], 0, sizeof(mi_hit));
            hits[i].prim = hit ? (int32_t)prim : -1;
            hits[i].t = hit ? t : 0; hits[i].b0 = b[0]; hits[i].b1 = b[1]; hits[i].b2 = b[2];
        }
    }
    stats[0] = (int64_t)qn.size(); stats[2] = topDepth; stats[3] = (int64_t)st.maxStack; stats[4] = d->n_tris;
    stats[5] = (int64_t)st.nodes; stats[6] = (int64_t)st.tris; stats[7] = (int64_t)st.hits;
    return 0;
}
// stage-level texture evaluation: Texture<T>::Evaluate of node `node` at n recorded interactions
__global__ void __launch_bounds__(PT_BLOCK) k_stage_texture(int node, const mi_tex_query *q, int64_t n, float *rgb) {
    NoiseLdsInit();
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    TexCtx tc;
    tc.p = v3(q[i].p); tc.u = q[i].uv[0]; tc.v = q[i].uv[1];
    tc.dpdx = v3(q[i].dpdx); tc.dpdy = v3(q[i].dpdy);
    tc.dudx = q[i].dudx; tc.dvdx = q[i].dvdx; tc.dudy = q[i].dudy; tc.dvdy = q[i].dvdy;
    RGB v = TexEval(node, tc);
    rgb[3 * i] = v.r; rgb[3 * i + 1] = v.g; rgb[3 * i + 2] = v.b;
}
int mi_texture_eval(mi_ctx *c, int32_t node, const mi_tex_query *queries, int64_t n, float *rgb_out) {
    if (!c || !queries || !rgb_out || n < 0) return fail("mi_texture_eval: bad argument");
    if (!c->haveScene || !(c->hasTex || c->hasAlpha)) return fail("mi_texture_eval: the uploaded scene has no textures");
    if (node < 0 || (uint32_t)node >= c->tex.n_nodes) return fail("mi_texture_eval: node out of range");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dq, dr;
    if (dq.alloc((size_t)n * sizeof(mi_tex_query)) || dr.alloc((size_t)n * 3 * sizeof(float))) return -1;
    TableTurn turn(c);
    HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tex), &c->tex, sizeof(DevTex), 0, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dq.p, queries, (size_t)n * sizeof(mi_tex_query), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_stage_texture, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, c->stream, (int)node, dq.as<mi_tex_query>(), n, dr.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_out, dr.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
__global__ void __launch_bounds__(256) k_stream_read(const float4 *p, uint64_t n, float4 *out) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) out[0] = acc;   // keeps the loads alive
}
int mi_stream_read_gbps(mi_ctx *c, uint64_t bytes, double *gbps) {
    if (!c || !gbps || bytes < (1u << 20)) return fail("mi_stream_read_gbps: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf buf;
    if (buf.alloc(bytes)) return -1;
    HIP_TRY(hipMemsetAsync(buf.p, 0, bytes, c->stream));
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    uint64_t n = bytes / 16;
    double best = 0;
    for (int it = 0; it < 4; ++it) {   // first launch = warm-up
        HIP_TRY(hipEventRecord(a, c->stream));
        hipLaunchKernelGGL(k_stream_read, dim3(c->numCUs * 16), dim3(256), 0, c->stream, buf.as<float4>(), n, buf.as<float4>());
        HIP_TRY(hipEventRecord(b, c->stream));
        HIP_TRY(hipEventSynchronize(b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms > 0) best = std::max(best, (double)(n * 16) / (ms * 1e-3) * 1e-9);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    buf.release();
    *gbps = best;
    return 0;
}
// ---- the request-rate ceiling the traversal kernels are measured against: every lane walks a chain of DEPENDENT fetches of
// 64-byte records at random places of a buffer -- LOADS 16-byte loads per record, issued together, the next record's index taken from the data
// just loaded -- with the traversal kernels' launch shape (256-thread blocks, PT_GRID_PER_CU blocks per CU) and no arithmetic beyond the index.
// That is the memory-side work of one BVH4Q interior step (4 x 16 B of one node, next node from its child words) stripped of the box tests.
}   // extern "C"
template <int LOADS>
__global__ void __launch_bounds__(PT_BLOCK) k_gather_probe(const uint4 *buf, uint32_t nrec, int iters, uint4 *out) {
    uint32_t s = (blockIdx.x * PT_BLOCK + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t rec = (uint32_t)(((uint64_t)s * nrec) >> 32);
        const uint4 *p = buf + (size_t)rec * 4;
        uint4 a = p[0], b = LOADS > 1 ? p[1] : a, c = LOADS > 2 ? p[2] : a, d = LOADS > 3 ? p[3] : a;
        Pin(a); if (LOADS > 1) Pin(b); if (LOADS > 2) Pin(c); if (LOADS > 3) Pin(d);
        acc += a.y ^ b.z ^ c.w ^ d.x;
        s = s * 1664525u + a.x;   // the next index depends on the record
    }
    if (acc == 0x12345678u) out[0] = make_uint4(acc, s, 0, 0);   // keeps the loads alive
}
extern "C" {
int mi_gather_rate(mi_ctx *c, uint64_t bytes, int loads_per_record, double *grequests_per_s) {
    if (!c || !grequests_per_s || bytes < 4096 || loads_per_record < 1 || loads_per_record > 4) return fail("mi_gather_rate: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nrec = (uint32_t)std::min<uint64_t>(bytes / 64, 0xffffffffull);
    DevBuf buf;
    if (buf.alloc((size_t)nrec * 64)) return -1;
    {   // random words (the chain must not fall into a short cycle): a host-side LCG fill
        std::vector<uint32_t> h((size_t)nrec * 16);
        uint32_t x = 2463534242u;
        for (auto &w : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w = x; }
        HIP_TRY(hipMemcpyAsync(buf.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
    const int iters = 2048;
    const dim3 grid(c->gridBlocks), block(PT_BLOCK);
    double best = 0;
    for (int it = 0; it < 4; ++it) {   // first launch = warm-up
        HIP_TRY(hipEventRecord(a, c->stream));
        switch (loads_per_record) {
        case 1: hipLaunchKernelGGL(k_gather_probe<1>, grid, block, 0, c->stream, buf.as<uint4>(), nrec, iters, buf.as<uint4>()); break;
        case 2: hipLaunchKernelGGL(k_gather_probe<2>, grid, block, 0, c->stream, buf.as<uint4>(), nrec, iters, buf.as<uint4>()); break;
        case 3: hipLaunchKernelGGL(k_gather_probe<3>, grid, block, 0, c->stream, buf.as<uint4>(), nrec, iters, buf.as<uint4>()); break;
        default: hipLaunchKernelGGL(k_gather_probe<4>, grid, block, 0, c->stream, buf.as<uint4>(), nrec, iters, buf.as<uint4>()); break;
        }
        HIP_TRY(hipEventRecord(b, c->stream));
        HIP_TRY(hipEventSynchronize(b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms > 0) best = std::max(best, (double)c->gridBlocks * PT_BLOCK * iters * loads_per_record / (ms * 1e-3) * 1e-9);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    buf.release();
    *grequests_per_s = best;
    return 0;
}
int mi_counters_reset(mi_ctx *c) {
    if (!c) return fail("mi_counters_reset: null ctx");
    HIP_TRY(hipMemsetAsync(c->counters.p, 0, PT_CNT_ALLOC * sizeof(uint64_t), c->stream));
    return 0;
}
int mi_timing_enable(mi_ctx *c, int on) {
    if (!c) return fail("mi_timing_enable: null ctx");
    c->timing = on != 0;
    for (int i = 0; i < MI_K_COUNT; ++i) { c->msTotal[i] = 0; c->launches[i] = 0; }
    return 0;
}
int mi_timing_get(mi_ctx *c, double ms_total[MI_K_COUNT], uint64_t launches[MI_K_COUNT]) {
    if (!c) return fail("mi_timing_get: null ctx");
    HIP_TRY(hipStreamSynchronize(c->stream));
    harvest(c);
    for (int i = 0; i < MI_K_COUNT; ++i) { ms_total[i] = c->msTotal[i]; launches[i] = c->launches[i]; }
    return 0;
}

// ---- stage-level entry points
static int stage_common(mi_ctx *c) {
    if (!c || !c->haveScene) return fail("no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    return ensure_state(c, std::max<uint32_t>(c->cap, 256 * 64));
}

// rays -> queue -> the scene's own k_trace<0> (closest hit) or k_trace<2> (any hit) instance -> hits; in chunks of the path-state capacity
static int stage_trace(mi_ctx *c, const mi_ray *rays, int64_t n, mi_hit *hits, uint8_t *occluded) {
    if (!c || !c->haveScene) return fail("no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    if (n <= 0) return 0;
    if (c->useQ) {   //  the quantised interior step serves rays that start within 1e5 grid extents of the scene (mi_scene_upload checks the camera the same way): caller's rays are checked here
        for (int a = 0; a < 3; ++a) {
            const double ext = 65535.0 * (double)c->sc.qgrid.cell[a], lo = (double)c->sc.qgrid.lo[a];
            for (int64_t i = 0; i < n; ++i)
                if (!(std::fabs((double)rays[i].o[a] - lo) <= 1e5 * ext))
                    return fail("mi_intersect / mi_intersect_p: a ray starts farther than 1e5 scene extents from the scene (or at a non-finite point): outside the range of the quantised BVH4 traversal -- upload the scene with PBRT_AMD_TRACE=general for such rays");
        }
    }
    if (ensure_state(c, (uint32_t)std::max<int64_t>(c->cap, std::min<int64_t>(std::max<int64_t>(n, 256 * 64), 1 << 22)))) return -1;
    PathState &ps = c->ps;
    const DevScene &sc = c->sc;
    hipStream_t st = c->stream;
    dim3 grid(c->gridBlocks), block(PT_BLOCK);
    const bool countWork = false;
    const uint32_t qin = 0;
    TableTurn turn(c);
    if (c->hasTex || c->hasAlpha || c->hasInst) HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tex), &c->tex, sizeof(DevTex), 0, hipMemcpyHostToDevice, st));
    if (c->hasInst) HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_instances), &c->instPtr, sizeof(c->instPtr), 0, hipMemcpyHostToDevice, st));
    DevBuf dr, dh;
    const int64_t chunk = (int64_t)c->cap;
    if (dr.alloc((size_t)std::min(n, chunk) * sizeof(mi_ray)) || dh.alloc((size_t)std::min(n, chunk) * (occluded ? 1 : sizeof(mi_hit)))) return -1;
    for (int64_t i0 = 0; i0 < n; i0 += chunk) {
        const int64_t m = std::min(chunk, n - i0);
        HIP_TRY(hipMemcpyAsync(dr.p, rays + i0, (size_t)m * sizeof(mi_ray), hipMemcpyHostToDevice, st));
        uint32_t row[QSEG * QC_STRIDE] = {0};
        for (uint32_t sg = 0; sg < QSEG; ++sg) row[sg * QC_STRIDE] = (uint32_t)((m - sg + 7) / 8);   // items sg, sg + 8, ...
        HIP_TRY(hipMemcpyAsync(ps.qcount + QCI(occluded ? QC_SHADOW : qin, 0), row, sizeof(row), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(ps.cursor, 0, QSEG * QC_STRIDE * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_stage_fill_rays, grid, block, 0, st, ps, dr.as<mi_ray>(), m, occluded ? 1 : 0);
        if (occluded) { LAUNCH_TRACE(2); } else { LAUNCH_TRACE(0); }
        hipLaunchKernelGGL(k_stage_collect_hits, grid, block, 0, st, sc, ps, dr.as<mi_ray>(), m, occluded ? (mi_hit *)nullptr : dh.as<mi_hit>(), occluded ? dh.as<uint8_t>() : (uint8_t *)nullptr);
        HIP_TRY(hipGetLastError());
        if (occluded) HIP_TRY(hipMemcpyAsync(occluded + i0, dh.p, (size_t)m, hipMemcpyDeviceToHost, st));
        else HIP_TRY(hipMemcpyAsync(hits + i0, dh.p, (size_t)m * sizeof(mi_hit), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));   // `row` and the staging buffers are reused by the next chunk
    }
    dr.release(); dh.release();
    return 0;
}
int mi_intersect(mi_ctx *c, const mi_ray *rays, int64_t n, mi_hit *hits) {
    if (n > 0 && (!rays || !hits)) return fail("mi_intersect: null argument");
    return stage_trace(c, rays, n, hits, nullptr);
}
int mi_triangle_intersect(int device_ordinal, const float *tri9, const mi_ray *rays, int64_t n, mi_hit *hits) {
    if (n <= 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("mi_triangle_intersect: no HIP device available");
    HIP_TRY(hipSetDevice(device_ordinal));
    DevBuf dt, dr, dh;
    if (dt.alloc((size_t)n * 36) || dr.alloc((size_t)n * sizeof(mi_ray)) || dh.alloc((size_t)n * sizeof(mi_hit))) return -1;
    HIP_TRY(hipMemcpy(dt.p, tri9, (size_t)n * 36, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, rays, (size_t)n * sizeof(mi_ray), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage_triangles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dt.as<float>(), dr.as<mi_ray>(), n, dh.as<mi_hit>());
    HIP_TRY(hipMemcpy(hits, dh.p, (size_t)n * sizeof(mi_hit), hipMemcpyDeviceToHost));
    dt.release(); dr.release(); dh.release();
    return 0;
}
int mi_intersect_p(mi_ctx *c, const mi_ray *rays, int64_t n, uint8_t *occluded) {
    if (n > 0 && (!rays || !occluded)) return fail("mi_intersect_p: null argument");
    return stage_trace(c, rays, n, nullptr, occluded);
}
int mi_sobol(mi_ctx *c, int px, int py, int n_samples, int n_dims, float *out, uint64_t *index_out) {
    if (stage_common(c)) return -1;
    int64_t n = (int64_t)n_samples * n_dims;
    if (n <= 0) return 0;
    DevBuf d_out, d_idx;
    if (d_out.alloc((size_t)n * 4) || d_idx.alloc((size_t)n_samples * 8)) return -1;
    hipLaunchKernelGGL(k_stage_sobol, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->sc, px, py, n_samples, n_dims, d_out.as<float>(), d_idx.as<unsigned long long>());
    HIP_TRY(hipMemcpyAsync(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (index_out) HIP_TRY(hipMemcpyAsync(index_out, d_idx.p, (size_t)n_samples * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    d_out.release(); d_idx.release();
    return 0;
}
static int list_pass(mi_ctx *c, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, int64_t off, uint32_t cnt, bool trace,
                     DevBuf &dxy, DevBuf &ds, bool keepLens = false) {
    if (MI_SAMPLER_IS_TILE_SERIAL(c->sc.sampler_type)) return fail("per-sample queries (mi_li / mi_camera_rays / mi_camera_differentials) need a GlobalSampler: with random / stratified / 02sequence a sample's values depend on its tile's earlier samples");
    (void)n;
    if (upload(c, dxy, pixels_xy + 2 * off, (size_t)cnt * 8) || upload(c, ds, sample_num + off, (size_t)cnt * 4)) return -1;
    PassInfo pass;
    std::memset(&pass, 0, sizeof(pass));
    pass.npix = cnt; pass.ns = 1;
    pass.list_xy = dxy.as<int32_t>(); pass.list_s = ds.as<int32_t>();
    if (trace) return run_pass(c, pass, false, false);
    HIP_TRY(hipMemsetAsync(c->ps.qcount, 0, QC_WORDS * sizeof(uint32_t), c->stream));
    if (keepLens) hipLaunchKernelGGL(k_raygen<true>, dim3(c->gridBlocks), dim3(PT_BLOCK), 0, c->stream, c->sc, c->ps, pass, 0u);
    else hipLaunchKernelGGL(k_raygen<false>, dim3(c->gridBlocks), dim3(PT_BLOCK), 0, c->stream, c->sc, c->ps, pass, 0u);
    return 0;
}
int mi_camera_rays(mi_ctx *c, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, mi_ray *rays, float *p_film) {
    if (stage_common(c)) return -1;
    DevBuf dxy, ds, dr, dp;
    for (int64_t off = 0; off < n; off += c->cap) {
        uint32_t cnt = (uint32_t)std::min<int64_t>(c->cap, n - off);
        if (list_pass(c, pixels_xy, sample_num, n, off, cnt, false, dxy, ds)) return -1;
        if (dr.alloc((size_t)cnt * sizeof(mi_ray)) || dp.alloc((size_t)cnt * 8)) return -1;
        hipLaunchKernelGGL(k_stage_export_rays, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->ps, (int64_t)cnt, dr.as<mi_ray>(), dp.as<float>());
        HIP_TRY(hipMemcpyAsync(rays + off, dr.p, (size_t)cnt * sizeof(mi_ray), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(p_film + 2 * off, dp.p, (size_t)cnt * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    dxy.release(); ds.release(); dr.release(); dp.release();
    return 0;
}
int mi_camera_differentials(mi_ctx *c, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, float *diffs) {
    if (stage_common(c)) return -1;
    if (!pixels_xy || !sample_num || !diffs || n < 0) return fail("mi_camera_differentials: bad argument");
    DevBuf dxy, ds, dd;
    for (int64_t off = 0; off < n; off += c->cap) {
        uint32_t cnt = (uint32_t)std::min<int64_t>(c->cap, n - off);
        if (list_pass(c, pixels_xy, sample_num, n, off, cnt, false, dxy, ds, true)) return -1;
        if (dd.alloc((size_t)cnt * 48)) return -1;
        hipLaunchKernelGGL(k_stage_export_diffs, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->ps, c->sc.camera, c->sc.spp, (int64_t)cnt, dd.as<float>());
        HIP_TRY(hipMemcpyAsync(diffs + 12 * off, dd.p, (size_t)cnt * 48, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    dxy.release(); ds.release(); dd.release();
    return 0;
}
int mi_li(mi_ctx *c, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, float *L_rgb) {
    if (stage_common(c)) return -1;
    DevBuf dxy, ds, dl;
    for (int64_t off = 0; off < n; off += c->cap) {
        uint32_t cnt = (uint32_t)std::min<int64_t>(c->cap, n - off);
        if (list_pass(c, pixels_xy, sample_num, n, off, cnt, true, dxy, ds)) return -1;
        if (dl.alloc((size_t)cnt * 12)) return -1;
        hipLaunchKernelGGL(k_stage_export_L, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, c->ps, (int64_t)cnt, dl.as<float>());
        HIP_TRY(hipMemcpyAsync(L_rgb + 3 * off, dl.p, (size_t)cnt * 12, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    dxy.release(); ds.release(); dl.release();
    return 0;
}

}  // extern "C"
