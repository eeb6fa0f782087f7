// Device scene layout (HBM) + Sobol' sampler + watertight triangle test + BVH4 traversal.
#pragma once
#include "pbrt_amd.h"
#include "pt_math.h"
#include "pt_sphere.h"
#include "pt_bvh4q.h"

// ------------------------------------------------------------------ HBM layout
// BVH4 node: exactly one 128-byte cache line, fetched by a lane as 8 x global_load_dwordx4.
// Child bounds are SoA so a lane tests 4 boxes from 6 float4 registers.  Built by collapsing
// the reference's own BVH2 (LinearBVHNode[], accelerators/bvh.cpp:95-104): every BVH4 child box is
// a BVH2 node box, so the set of triangles whose ancestors' boxes a ray passes is a superset of
// what BVHAccel::Intersect visits and the closest hit is identical (SURVEY.md s.7 "BVH topology").
struct __attribute__((aligned(128))) BVH4Node {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];   // interior: node index | leaf: BVH4_LEAF | (count-1) << 27 | first triangle | BVH4_EMPTY
    uint32_t pad[4];
};
#define BVH4_LEAF 0x80000000u
#define BVH4_EMPTY 0xFFFFFFFFu
#define BVH4_LEAF_MAX 16          /* triangles per leaf reference; bigger reference leaves are chained */
#define BVH4_FIRST_MASK 0x07ffffffu

// Triangle record: 3 x float4 = 48 B, pre-gathered world-space vertices in BVH primitive order
// (removes the reference's Primitive -> Shape -> mesh -> index -> vertex pointer chain).
// v0.w carries flag bits (as uint): bit0 = "always rejected": the triangle is degenerate in the sense of
// shapes/triangle.cpp:308-315 (decided per triangle, independent of the ray).
#define TRI_FLAG_REJECT 1u
// the primitive is a Sphere (shapes/sphere.cpp): record.v0.x holds the index into DevScene::spheres
#define TRI_FLAG_SPHERE 2u
#define TRI_FLAG_ALPHA 4u
/* the triangle's mesh has an alpha or shadow-alpha mask (triangle.cpp:333-338,532-570) */
#define TRI_FLAG_INSTANCE 8u /* two-level scenes: the primitive is a TransformedPrimitive; record.v0.x holds the index into the instance table */

struct DevEnvMap;
struct DevLight {   // 10 x 16 bytes; the first 7, fetched with independent 16-byte loads (LoadLight)
    int32_t type, tri, two_sided; uint32_t mesh_flags;   // mesh flags of the emissive triangle (bit 31: TRI_FLAG_REJECT)
    float L[3], area;
    float pos[3], world_radius;
    float cos_total, cos_falloff; const void *ext;   // spot cone; infinite: its radiance map (DevEnvMap*, null: constant L); sphere light: its mi_sphere*
    float p0[3], padA;                                   // area light: triangle vertices; spot: rows of WorldToLight
    float p1[3], padB;
    float p2[3], padC;
    float l2w0[3], padD;                                 // infinite light with a map: rows of LightToWorld (p0..p2 hold WorldToLight)
    float l2w1[3], padE;
    float l2w2[3], padF;
};
// radiance map of an InfiniteAreaLight (mi_envmap on the device)
struct DevEnvMap {
    int32_t width, height;
    const float *rgb, *cond_func, *cond_cdf, *cond_func_int, *marg_func, *marg_cdf;
    float marg_func_int, pad;
};

// Per-triangle shading record: the vertex normals and uvs the interaction needs, gathered per triangle at upload time
// (one 64-byte fetch instead of the reference's indices -> per-vertex arrays chain).  Meshes without normals / uvs
// hold zeros / the default (0,0),(1,0),(1,1) of Triangle::GetUVs (shapes/triangle.h:98-108).
struct __attribute__((aligned(64))) TriShade {
    float n[9];    // n0, n1, n2
    float uv[6];   // uv0, uv1, uv2
    uint32_t pad;
};

struct DevScene {
    const BVH4Node *nodes;
    const float4 *tri_verts;        // 3 per triangle
    const BVH4QNode *nodesq;        // the same tree as `nodes` with 16-bit planes on one grid (pt_bvh4q.h); null: not built
    Bvh4qGrid qgrid;
    const TriShade *tri_shade;      // per triangle: vertex normals + uvs
    const uint4 *tri_info;          // per triangle, ONE 16-byte load: x = mesh flags (MI_MESH_*), y = material (int), z = light (int), w = mesh
    const float4 *tri_rec;          // (round 6) per triangle ONE 128-byte line for the shading kernels: [0..2] the tri_verts record, [3..6] the TriShade record, [7] tri_info --
                                    // a vertex reloads vertices + normals / uvs + flags from one line instead of three or four (the 48-byte vertex records straddle lines)
    const mi_material *materials;
    const uint2 *mat_pack;          // per material: {n_bxdfs, the lobe types 4 bits each} -- the BSDF's lobe header (pt_shade.h, PT_LOBE_HEADER)
    const DevEnvMap *envmaps;
    const mi_sphere *spheres;
    const DevLight *lights;         // mi_light + (area lights) the triangle's vertices and mesh flags: no extra hops while sampling
    const float *light_func, *light_cdf;
    const float *filter_table;
    const int32_t *infinite_lights; // indices of MI_LIGHT_INFINITE lights
    const uint32_t *sobol32;        // [1024][52] generator matrices (sobol_tables.inc), L2 resident
    int32_t sobol_index_bits;       // upper bound on the bits of a Sobol' index: 2*log2(resolution) + log2(spp)
    const uint64_t *vdc, *vdc_inv;  // [26][52] pixel <-> index maps
    float light_func_int;
    // SpatialLightDistribution (core/lightdistrib.cpp:96-300) as a dense table: one Distribution1D per voxel
    const float *sp_func, *sp_cdf, *sp_func_int;   // [nvox][n_lights], [nvox][n_lights + 1], [nvox]
    const uint32_t *sp_guide;                      // [nvox][sp_guide_m]: cut points of the voxel's cdf at u = j / sp_guide_m (SpatialPick, pt_shade.h); null: none
    uint32_t sp_guide_m;                           // a power of two
    int32_t light_strategy, sp_nvox[3];
    float sp_bmin[3], sp_bmax[3];                  // scene.WorldBound()
    float sp_bmin_all[3], sp_bmax_all[3];          // the same for every scene (sp_bmin / sp_bmax are set for the spatial light strategy only): ray binning
    uint32_t n_tris, n_nodes, n_lights, n_materials, n_infinite;
    uint32_t n_hot;                 // nodesq[0 .. n_hot) are the scene's most visited nodes (hot-node probe at upload): the traversal blocks keep them in LDS
    int32_t stack_need;             // 3 * BVH4 depth + 1
    mi_camera camera;
    // film / sampler / integrator scalars
    int32_t full_res[2], crop_min[2], crop_max[2], sample_min[2], sample_max[2], pixel_min[2], pixel_max[2];
    float filter_radius[2], max_sample_luminance;
    int32_t max_depth, spp, sobol_resolution, sobol_log2_resolution;
    // HaltonSampler (samplers/halton.cpp): constructor results + tables
    int32_t sampler_type;                 // MI_SAMPLER_*
    int32_t h_base_scales[2], h_base_exps[2], h_stride, h_mult_inv[2], h_at_center;
    uint64_t h_magic_scale1;              // floor(2^64 / baseScales[1]) + 1
    const uint16_t *h_perms;              // digit permutations of all 1000 prime bases, concatenated
    const uint4 *h_info;                  // per dimension: {prime, offset into h_perms, magic lo, magic hi}
    float rr_threshold;
    // the tile-serial samplers (ABI v11: MI_SAMPLER_RANDOM / STRATIFIED / ZEROTWO): per 16x16 tile (global tile index) its PCG32 stream and the
    // current pixel's precomputed dimensions (PixelSampler::samples1D / samples2D, core/sampler.h:118-121)
    unsigned long long *pix_rng;          // [2 * tile]: state, inc
    const uint32_t *pix_maxmin;           // MI_SAMPLER_MAXMIN: the 32 columns of the generator matrix of the first 2D dimension (ABI v12)
    float *pix_s1;                        // [(tile * pix_nd + d) * spp + s]
    float *pix_s2;                        // [2 * ((tile * pix_nd + d) * spp + s)]
    int32_t pix_nd, strat_nx, strat_ny, strat_jitter;
};

// ------------------------------------------------------------------ PCG32 (RNG core/rng.h:64-144)
PT_DEV uint32_t Pcg32Next(unsigned long long &state, unsigned long long inc) {
    unsigned long long oldstate = state;
    state = oldstate * 0x5851f42d4c957f2dULL + inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
PT_DEV uint32_t Pcg32Bounded(unsigned long long &state, unsigned long long inc, uint32_t b) {   // UniformUInt32(b) rng.h:72-78
    uint32_t threshold = (~b + 1u) % b;
    while (true) {
        uint32_t r = Pcg32Next(state, inc);
        if (r >= threshold) return r % b;
    }
}
PT_DEV Float Pcg32Float(unsigned long long &state, unsigned long long inc) {   // UniformFloat rng.h:83-90
    Float v = (Float)Pcg32Next(state, inc) * 0x1p-32f;
    return v < PT_ONE_MINUS_EPS ? v : PT_ONE_MINUS_EPS;
}

// ------------------------------------------------------------------ Sobol' (integer, bit exact)
#ifndef PBRT_AMD_SOBOL_NDIM
#define PBRT_AMD_SOBOL_NDIM 1024
#define PBRT_AMD_SOBOL_NCOL 52
#define PBRT_AMD_SOBOL_NRES 26
#endif

PT_DEV uint64_t SobolIntervalToIndex(const DevScene &sc, uint32_t m, uint64_t frame, int px, int py) {   // core/lowdiscrepancy.h:229-249
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = uint64_t(frame) << m2;
    // The reference loops over the set bits of `frame` and of the pixel word; column c of the two tables is the same for
    // every lane, so the loops run over a wave-uniform column range with the columns coming through scalar loads (constant
    // address space) and a per-lane select -- same XOR set, no per-lane dependent loads.
    typedef const __attribute__((address_space(4))) uint64_t *ConstU64;
    ConstU64 vdc = (ConstU64)(unsigned long long)(sc.vdc + (m - 1) * PBRT_AMD_SOBOL_NCOL);
    ConstU64 vdcInv = (ConstU64)(unsigned long long)(sc.vdc_inv + (m - 1) * PBRT_AMD_SOBOL_NCOL);
    uint64_t delta = 0;
    const int frameBits = sc.sobol_index_bits - (int)m2 > 0 ? sc.sobol_index_bits - (int)m2 : 0;   // log2(spp): frame < spp
    for (int c = 0; c < frameBits; ++c)
        if ((frame >> c) & 1) delta ^= vdc[c];
    if (frame >> frameBits) {   // sample numbers beyond the scene's spp (stage-level calls): the reference's loop as it is
        uint64_t fr = frame >> frameBits;
        for (int c = frameBits; fr; fr >>= 1, ++c)
            if (fr & 1) delta ^= sc.vdc[(m - 1) * PBRT_AMD_SOBOL_NCOL + c];
    }
    uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;
    for (uint32_t c = 0; c < m2; ++c)   // px, py < 2^m and delta < 2^2m: b has at most 2m bits
        if ((b >> c) & 1) index ^= vdcInv[c];
    if (b >> m2) {
        uint64_t br = b >> m2;
        for (uint32_t c = m2; br; br >>= 1, ++c)
            if (br & 1) index ^= sc.vdc_inv[(m - 1) * PBRT_AMD_SOBOL_NCOL + c];
    }
    return index;
}
PT_DEV Float SobolSampleFloat(const DevScene &sc, uint64_t a, int dimension) {   // core/lowdiscrepancy.h:259-274 (scramble 0)
    uint32_t v = 0;
    for (int i = dimension * PBRT_AMD_SOBOL_NCOL; a != 0; a >>= 1, i++)
        if (a & 1) v ^= sc.sobol32[i];
    return mn(v * 0x1p-32f, PT_ONE_MINUS_EPS);
}

// N consecutive dimensions dim0..dim0+N-1 of one Sobol' index in a single sweep over the index bits.
// Inside a wave all paths normally sit at the same dimension (same bounce history), so the generator words
// are wave-uniform: one broadcast load per (bit, dimension) instead of 64 divergent dependent ones, and the
// per-lane work is an AND/XOR.  Values are bit-identical to SobolSampleFloat (same XOR set).
#define PT_SOBOLT_STRIDE (PBRT_AMD_SOBOL_NDIM + 16)
// transposed generator matrices [bit][dimension] in the constant address space: with a wave-uniform
// dimension the words come through the scalar cache (s_load), off the vector memory pipe
__constant__ uint32_t c_sobolT[PBRT_AMD_SOBOL_NCOL * PT_SOBOLT_STRIDE];
template <int N>
PT_DEV void SobolBatch(const DevScene &sc, uint64_t index, int dim0, Float *out) {
    uint32_t v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = 0;
    // Lanes of a wave may sit at different dimensions (paths with specular bounces consume fewer): loop over the distinct
    // values so that every group still reads its matrix words through the scalar cache (a "waterfall" loop).
    bool todo = true;
    while (todo) {   // divergent loop: the exec mask holds the lanes still waiting, readfirstlane picks one of them
        int d0 = UniformInt(dim0);
        if (SameAs(dim0, d0)) {
            const uint32_t *row = c_sobolT + d0;   // scalar address: the words come through s_load
            for (int i = 0; i < sc.sobol_index_bits; ++i, row += PT_SOBOLT_STRIDE) {
                uint32_t mask = 0u - ((uint32_t)(index >> i) & 1u);
#pragma unroll
                for (int k = 0; k < N; ++k) v[k] ^= row[k] & mask;
            }
            todo = false;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = mn(v[k] * 0x1p-32f, PT_ONE_MINUS_EPS);
}

// ---- HaltonSampler (samplers/halton.cpp; radical inverses core/lowdiscrepancy.cpp:389-424, 427-445, 2506-2520).
// Integer parts are exact; a / base uses a multiply-high with magic = floor(2^64 / base) + 1, exact for a < 2^64 / base
// (indices stay below 2^46: stride <= 128 * 243, sample numbers < 2^31).
PT_DEV uint64_t MulHi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }
PT_DEV Float HaltonRadicalInverse3(uint64_t a) {   // RadicalInverseSpecialized<3>
    const Float invBase = (Float)1 / (Float)3;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = MulHi64(a, 0x5555555555555556ull);   // a / 3
        uint64_t digit = a - next * 3;
        reversedDigits = reversedDigits * 3 + digit;
        invBaseN *= invBase;
        a = next;
    }
    return mn((Float)reversedDigits * invBaseN, PT_ONE_MINUS_EPS);
}
// ScrambledRadicalInverseSpecialized<base>(perm, a) for dimension dim >= 2 (out of line: k_shade draws 8 of them per vertex)
__device__ __noinline__ Float HaltonScrambled(const uint16_t *perms, const uint4 *info, uint64_t a, int dim) {
    uint4 in = info[dim];
    const uint32_t base = in.x;
    const uint16_t *perm = perms + in.y;
    const uint64_t magic = (uint64_t)in.z | ((uint64_t)in.w << 32);
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = MulHi64(a, magic);
        uint32_t digit = (uint32_t)(a - next * base);
        reversedDigits = reversedDigits * base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    return mn(invBaseN * ((Float)reversedDigits + invBase * (Float)(int)perm[0] / (1 - invBase)), PT_ONE_MINUS_EPS);
}
PT_DEV Float HaltonSampleDimension(const DevScene &sc, uint64_t index, int dim) {   // halton.cpp:123-132
    if (sc.h_at_center && (dim == 0 || dim == 1)) return 0.5f;
    if (dim == 0) {   // RadicalInverse(0, a) = ReverseBits64(a) * 2^-64, evaluated in double, rounded to Float on return
        uint64_t a = index >> sc.h_base_exps[0];
        uint64_t r = ((uint64_t)__brev((uint32_t)a) << 32) | (uint64_t)__brev((uint32_t)(a >> 32));
        return (Float)((double)r * 0x1p-64);
    }
    if (dim == 1) return HaltonRadicalInverse3(MulHi64(index, sc.h_magic_scale1));   // index / baseScales[1]
    if (dim > 999) dim = 999;   // the reference LOG(FATAL)s (halton.h:72-75); mi_scene_upload rejects such depths
    return HaltonScrambled(sc.h_perms, sc.h_info, index, dim);
}
PT_DEV uint64_t HaltonIndexForSample(const DevScene &sc, int x, int y, uint64_t sampleNum) {   // halton.cpp:100-121
    uint64_t offset = 0;
    if (sc.h_stride > 1) {
        int pm0 = x - (x / 128) * 128, pm1 = y - (y / 128) * 128;   // Mod(currentPixel, kMaxResolution) core/pbrt.h:310-313
        if (pm0 < 0) pm0 += 128;
        if (pm1 < 0) pm1 += 128;
        uint64_t d0 = 0, d1 = 0;   // InverseRadicalInverse<2>, <3> core/lowdiscrepancy.h:82-91
        { uint32_t inv = (uint32_t)pm0; for (int i = 0; i < sc.h_base_exps[0]; ++i) { d0 = d0 * 2 + (inv & 1u); inv >>= 1; } }
        { uint32_t inv = (uint32_t)pm1; for (int i = 0; i < sc.h_base_exps[1]; ++i) { d1 = d1 * 3 + (inv % 3u); inv /= 3u; } }
        offset = d0 * (uint64_t)(sc.h_stride / sc.h_base_scales[0]) * (uint64_t)sc.h_mult_inv[0] +
                 d1 * (uint64_t)(sc.h_stride / sc.h_base_scales[1]) * (uint64_t)sc.h_mult_inv[1];
        offset %= (uint64_t)sc.h_stride;
    }
    return offset + sampleNum * (uint64_t)sc.h_stride;
}

// N consecutive dimensions of the scene's sampler for one path
template <int N>
PT_DEV void SamplerBatch(const DevScene &sc, uint64_t index, int dim0, Float *out) {
    if (sc.sampler_type == MI_SAMPLER_HALTON) {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = HaltonSampleDimension(sc, index, dim0 + k);
    } else
        SobolBatch<N>(sc, index, dim0, out);
}

struct Sampler {   // GlobalSampler state per path (core/sampler.cpp:136-195) over SobolSampler / HaltonSampler
    uint64_t index;
    int dimension;
    int px, py;
    PT_DEV void Start(const DevScene &sc, int x, int y, uint64_t sampleNum) {
        px = x; py = y; dimension = 0;
        if (sc.sampler_type == MI_SAMPLER_HALTON) index = HaltonIndexForSample(sc, x, y, sampleNum);
        else index = SobolIntervalToIndex(sc, sc.sobol_log2_resolution, sampleNum, x - sc.sample_min[0], y - sc.sample_min[1]);
    }
    PT_DEV Float SampleDimension(const DevScene &sc, int dim) const {
        if (sc.sampler_type == MI_SAMPLER_HALTON) return HaltonSampleDimension(sc, index, dim);
        // sobol.cpp:47-59
        if (dim >= PBRT_AMD_SOBOL_NDIM) dim = PBRT_AMD_SOBOL_NDIM - 1;   // the reference LOG(FATAL)s here; host rejects such depths
        Float s = SobolSampleFloat(sc, index, dim);
        if (dim == 0 || dim == 1) {
            s = s * sc.sobol_resolution + sc.sample_min[dim];
            s = clampf(s - (dim == 0 ? px : py), (Float)0, PT_ONE_MINUS_EPS);
        }
        return s;
    }
    // The tile-serial samplers (sc.sampler_type >= MI_SAMPLER_RANDOM): index = tile | sample << 32, dimension = current1DDimension | current2DDimension << 8.
    // ONE path per tile is in flight (mi_render's tile-serial rounds), so the path's own lane is the only one touching the tile's stream.
    PT_DEV static bool Pix(const DevScene &sc) { return sc.sampler_type >= MI_SAMPLER_RANDOM; }
    PT_DEV void PixStart(uint32_t tile, uint32_t sampleNum, int x, int y) { index = (uint64_t)tile | ((uint64_t)sampleNum << 32); dimension = 0; px = x; py = y; }
    PT_DEV Float PixGet1D(const DevScene &sc) {   // PixelSampler::Get1D core/sampler.cpp:118-125, RandomSampler::Get1D random.cpp:42-46
        const uint32_t tile = (uint32_t)index, s = (uint32_t)(index >> 32);
        const int c1 = dimension & 0xff;
        if (c1 < sc.pix_nd) { dimension += 1; return sc.pix_s1[((size_t)tile * sc.pix_nd + c1) * sc.spp + s]; }
        unsigned long long st = sc.pix_rng[2 * (size_t)tile];
        Float v = Pcg32Float(st, sc.pix_rng[2 * (size_t)tile + 1]);
        sc.pix_rng[2 * (size_t)tile] = st;
        return v;
    }
    PT_DEV void PixGet2D(const DevScene &sc, Float *u0, Float *u1) {   // PixelSampler::Get2D core/sampler.cpp:127-134, RandomSampler::Get2D random.cpp:48-52
        const uint32_t tile = (uint32_t)index, s = (uint32_t)(index >> 32);
        const int c2 = (dimension >> 8) & 0xff;
        if (c2 < sc.pix_nd) {
            dimension += 0x100;
            const float *p = sc.pix_s2 + 2 * (((size_t)tile * sc.pix_nd + c2) * sc.spp + s);
            *u0 = p[0]; *u1 = p[1];
            return;
        }
        unsigned long long st = sc.pix_rng[2 * (size_t)tile];
        const unsigned long long inc = sc.pix_rng[2 * (size_t)tile + 1];
        // RandomSampler: `return {a(), b()};` draws x first; PixelSampler: `return Point2f(a(), b());` -- the reference as built here (g++) evaluates the
        // second argument first (oracle/pt_oracle.cpp Sampler::Get2D, pinned by tests/golden/edge_sampler_*.pfm)
        if (sc.sampler_type == MI_SAMPLER_RANDOM) { *u0 = Pcg32Float(st, inc); *u1 = Pcg32Float(st, inc); }
        else { *u1 = Pcg32Float(st, inc); *u0 = Pcg32Float(st, inc); }
        sc.pix_rng[2 * (size_t)tile] = st;
    }
    PT_DEV Float Get1D(const DevScene &sc) {
        if (Pix(sc)) return PixGet1D(sc);
        return SampleDimension(sc, dimension++);
    }
    PT_DEV void Get2D(const DevScene &sc, Float *u0, Float *u1) {
        if (Pix(sc)) { PixGet2D(sc, u0, u1); return; }
        *u0 = SampleDimension(sc, dimension);
        *u1 = SampleDimension(sc, dimension + 1);
        dimension += 2;
    }
};

// ------------------------------------------------------------------ triangle
struct TriHit { Float t, b0, b1, b2; };

// Per-ray constants of the watertight test (shapes/triangle.cpp:203-216): the permutation (kz = dimension of
// largest |d|) and the shear Sx,Sy,Sz depend on the ray only, so a ray computes them once, not per triangle.
// The compiler sinks a load into the branch that consumes it; in a traversal step that turns "fetch the node, test,
// maybe use the child references" into two dependent memory round trips.  Pin() makes the loaded registers live at
// the point where all loads of a step have been issued, so they are fetched together and waited for once.
PT_DEV void Pin(float4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
PT_DEV void Pin(uint4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
PT_DEV void Pin(float4 &a, float4 &b, float4 &c) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
}

// gfx950-exact forms of a few small operations (device build: one instruction each, opaque to the optimiser; host build of this header: plain C++)
PT_DEV float PtMinRaw(float a, float b) {   // v_min_f32 / v_max_f32 straight: the distances are never NaN, the builtins would first canonicalise both operands
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return b < a ? b : a;
#endif
}
// max(|a|, |b|, |c|) of three non-NaN floats: MaxComponent(Abs(V3(a, b, c))) as ONE v_max3_f32 with abs source modifiers (the (a < b) ? b : a form
// compiles to two compares, two selects and an AND per call; equal for every non-NaN input, signed zeros included since all operands are magnitudes)
PT_DEV float PtMaxAbs3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    const float x = __builtin_fabsf(a), y = __builtin_fabsf(b), z = __builtin_fabsf(c), m = y < z ? z : y;
    return x < m ? m : x;
#endif
}
PT_DEV float PtMin3Raw(float a, float b, float c) {   // min / max of three non-NaN floats as one v_min3_f32 / v_max3_f32
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    const float m = b < a ? b : a; return c < m ? c : m;
#endif
}
PT_DEV float PtMax3Raw(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
    const float m = b < a ? a : b; return c < m ? m : c;
#endif
}
PT_DEV float PtRcpBox(float x) {   // 1 / x for the quantised box test's per-ray constants only (see TravStateQ::init)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1 / x;
#endif
}
PT_DEV float PtMaxRaw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
    return b < a ? a : b;
#endif
}
// 0 / 0xffffffff from the sign bit, as ONE v_ashrrev_i32 the optimiser cannot turn back into a compare + select
PT_DEV uint32_t PtSignMask(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r; asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(v)); return r;
#else
    return (uint32_t)((int32_t)v >> 31);
#endif
}
#ifndef PT_TRI_SELECT
#define PT_TRI_SELECT 1
#endif
PT_DEV void PinV3(V3 &a, V3 &b, V3 &c) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(c.x), "+v"(c.y), "+v"(c.z)); }

struct RayShear {
    int kz;
    Float Sx, Sy, Sz;
    PT_DEV void init(const V3 &dir) {
        Float ax = absf(dir.x), ay = absf(dir.y), az = absf(dir.z);
        kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);   // MaxDimension(Abs(d))
        Float dx, dy, dz;                                               // Permute(d, kx, ky, kz), kx = kz+1, ky = kx+1 (mod 3)
        if (kz == 0) { dx = dir.y; dy = dir.z; dz = dir.x; }
        else if (kz == 1) { dx = dir.z; dy = dir.x; dz = dir.y; }
        else { dx = dir.x; dy = dir.y; dz = dir.z; }
        Sx = -dx / dz; Sy = -dy / dz; Sz = 1.f / dz;
    }
    PT_DEV V3 permute(const V3 &v) const {
        return kz == 0 ? V3(v.y, v.z, v.x) : (kz == 1 ? V3(v.z, v.x, v.y) : v);
    }
};

// Watertight ray-triangle test: Triangle::Intersect shapes/triangle.cpp:188-291 (through the
// conservative t > delta_t test).  The per-triangle degeneracy rejection of :308-315 is the
// TRI_FLAG_REJECT bit.  tMax is the ray's current tMax (accept t == tMax: :258-261).
PT_DEV bool TriangleTest(const V3 &p0, const V3 &p1, const V3 &p2, const V3 &o, const RayShear &rs, Float tMax, TriHit *h) {
    // the three differences are formed once and pinned, then the axes are SELECTED (6 v_cndmask per vertex): written as rs.permute(p - o) the
    // compiler sank the subtractions into three EXEC-masked branches per vertex, ~90 of the leaf step's ~300 instructions (round 3: VALU-bound)
#if PT_TRI_SELECT
    V3 d0 = p0 - o, d1 = p1 - o, d2 = p2 - o;
    PinV3(d0, d1, d2);
    const bool k0 = rs.kz == 0, k1 = rs.kz == 1;
#define PT_PERM(d) V3(k0 ? d.y : (k1 ? d.z : d.x), k0 ? d.z : (k1 ? d.x : d.y), k0 ? d.x : (k1 ? d.y : d.z))
    V3 p0t = PT_PERM(d0), p1t = PT_PERM(d1), p2t = PT_PERM(d2);
#undef PT_PERM
#else
    V3 p0t = rs.permute(p0 - o), p1t = rs.permute(p1 - o), p2t = rs.permute(p2 - o);
#endif
    const Float Sx = rs.Sx, Sy = rs.Sy, Sz = rs.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {   // :234-245 fp64 re-evaluation at triangle edges
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    // The four rejections of :247-261 as ONE predicate and one branch (round 4: the traversal kernels are bound by the NUMBER of VALU instructions, and the
    // nested early returns cost a compare pair, an EXEC save / restore and three register copies of the lane's running result per level, while a wave
    // of 64 rays practically never leaves early as a whole).  Same decisions for every non-NaN input:
    //   edge signs   (e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)   <=>   min3 < 0 && max3 > 0
    //   range        det < 0: reject tScaled >= 0 || tScaled < tMax det;  det > 0: reject tScaled <= 0 || tScaled > tMax det
    //                <=>  with both sides multiplied by sign(det) (a sign-bit flip, exact):  accept iff  0 < a && a <= b
    const Float eMin = PtMin3Raw(e0, e1, e2), eMax = PtMax3Raw(e0, e1, e2);
    Float det = e0 + e1 + e2;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const uint32_t detSign = __float_as_uint(det) & 0x80000000u;
    const Float a = __uint_as_float(__float_as_uint(tScaled) ^ detSign), b = __uint_as_float(__float_as_uint(tMax * det) ^ detSign);
    if (!(!(eMin < 0 && eMax > 0) && det != 0 && a > 0 && a <= b)) return false;
    Float invDet = 1 / det;
    Float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    Float t = tScaled * invDet;
    Float maxZt = PtMaxAbs3(p0t.z, p1t.z, p2t.z);   // MaxComponent(Abs(Vector3f(p0t.z, p1t.z, p2t.z)))
    Float deltaZ = gamma_n(3) * maxZt;
    Float maxXt = PtMaxAbs3(p0t.x, p1t.x, p2t.x);
    Float maxYt = PtMaxAbs3(p0t.y, p1t.y, p2t.y);
    Float deltaX = gamma_n(5) * (maxXt + maxZt);
    Float deltaY = gamma_n(5) * (maxYt + maxZt);
    Float deltaE = 2 * (gamma_n(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    Float maxE = PtMaxAbs3(e0, e1, e2);
    Float deltaT = 3 * (gamma_n(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absf(invDet);
    if (t <= deltaT) return false;
    h->t = t; h->b0 = b0; h->b1 = b1; h->b2 = b2;
    return true;
}
PT_DEV bool TriangleTest(const V3 &p0, const V3 &p1, const V3 &p2, const V3 &o, const V3 &dir, Float tMax, TriHit *h) {
    RayShear rs;
    rs.init(dir);
    return TriangleTest(p0, p1, p2, o, rs, tMax, h);
}

PT_DEV void LoadTri(const DevScene &sc, uint32_t prim, V3 *p0, V3 *p1, V3 *p2, uint32_t *flags) {
    const float4 *tv = sc.tri_verts + 3 * (size_t)prim;
    float4 a = tv[0], b = tv[1], c = tv[2];
    Pin(a, b, c);
    *p0 = V3(a.x, a.y, a.z); *p1 = V3(b.x, b.y, b.z); *p2 = V3(c.x, c.y, c.z);
    *flags = __float_as_uint(a.w);
}

// ------------------------------------------------------------------ BVH4 traversal
struct TraceCounters { uint32_t nodes, tris, hot; };

// Ray-box test.  Reference: Bounds3::IntersectP(ray, invDir, dirIsNeg) core/geometry.h:1412-1438 --
//   tNear_a = (near_a - o_a) * invDir_a,  tFar_a = ((far_a - o_a) * invDir_a) * (1 + 2 gamma(3)),  accept iff the
//   three [tNear, tFar] intervals overlap, tEnter < ray.tMax and tExit > 0.
// The box test only has to be CONSERVATIVE with respect to the reference's (every box the reference enters is
// entered here; extra boxes cost time, never change the closest hit), so it is evaluated in the cheaper form
//   tEnter = max3(tNear_x, tNear_y, tNear_z),  tExit = min3(tFar_x, tFar_y, tFar_z),  tFar_a = (far_a - o_a) * invFar_a
// with invFar_a = invDir_a * (1 + 4 gamma(3)) precomputed per ray: the wider factor covers the one rounding that
// differs from the reference's two-step product (relative 2^-24 << 2 gamma(3)); max3/min3 ignore a NaN operand
// (0 * inf on a slab plane) where the reference's comparisons reject, again the conservative side.
struct RayBox {
    V3 o, invNear, invFar;
    uint32_t offNearX, offNearY, offNearZ;   // byte offsets of the near planes inside a BVH4Node (lo or hi array by ray sign)
    PT_DEV void init(const V3 &o_, const V3 &invDir) {
        o = o_;
        invNear = invDir;
        const Float w = 1 + 4 * gamma_n(3);
        invFar = V3(invDir.x * w, invDir.y * w, invDir.z * w);
        offNearX = invDir.x < 0 ? 48u : 0u;
        offNearY = invDir.y < 0 ? 64u : 16u;
        offNearZ = invDir.z < 0 ? 80u : 32u;
    }
};
// Per-lane traversal stack: the first PT_LDS_STACK entries live in LDS ([entry][lane] layout: a
// lane's entries sit in one bank column, so pushes/pops of a whole wave are conflict free whatever
// the per-lane depth), deeper entries spill to a per-thread slice of an HBM buffer (rare).
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 24
#endif
#define PT_BLOCK 256
typedef uint32_t StackEntry;
typedef __attribute__((address_space(3))) StackEntry LdsStackEntry;
// STRIDE = threads per block of the kernel that owns the stack (the [entry][lane] rows are one block wide); NLDS = entries held in LDS.  The spill
// slices are sized for the shallowest LDS part any kernel uses (PT_LDS_STACK_MIN)
#ifndef PT_LDS_STACK_MIN
#define PT_LDS_STACK_MIN 15   /* TravStackB<.., 16>: entry 0 is its sentinel */
#endif
template <int STRIDE, int NLDS = PT_LDS_STACK>
struct TravStackT {
    enum { LDS_ENTRIES = NLDS, STRIDE_ = STRIDE };
    typedef LdsStackEntry *LdsPtr;
    LdsStackEntry *lds;   // &stack[0][threadIdx.x]; typed as LDS so that pushes / pops are ds_write / ds_read, never flat
    StackEntry *spill;    // per-thread spill slice
    int sp;
    PT_DEV void reset() { sp = 0; }
    PT_DEV void push(uint32_t v, Float t) {
        StackEntry e = v;
        if (sp < NLDS) lds[sp * STRIDE] = e; else spill[sp - NLDS] = e;
        ++sp;
    }
    // next node / leaf to look at, or TRAV_DONE
    PT_DEV uint32_t pop(Float tMax) {
        while (sp) {
            --sp;
            StackEntry e = (sp < NLDS) ? lds[sp * STRIDE] : spill[sp - NLDS];
            return e;
        }
        return 0xFFFFFFFFu;
    }
};
typedef TravStackT<PT_BLOCK> TravStack;

// Traversal as a per-lane state machine, so that a wave can keep its lanes busy with DIFFERENT rays at
// different stages (persistent lanes with dynamic ray fetch, see k_trace): `cur` is the next thing to
// look at -- an interior BVH4 node, a leaf reference, or TRAV_DONE.
// Closest hit (ANY == false) follows BVHAccel::Intersect (accelerators/bvh.cpp:662-700), any hit
// (ANY == true) BVHAccel::IntersectP (:702-738); `prim` != MISS marks a hit / an occlusion.
#define TRAV_DONE 0xFFFFFFFFu
#define TRAV_MISS 0xFFFFFFFFu
struct TravState {
    V3 o, d;
    RayBox box;
    RayShear shear;
    Float tMax, tHit;
    uint32_t prim, cur;
    template <class ST> PT_DEV void init(const DevScene &sc, const V3 &o_, const V3 &d_, Float tMax_, ST &st) {
        o = o_; d = d_; tMax = tMax_; tHit = 0; prim = TRAV_MISS;
        box.init(o, V3(1 / d.x, 1 / d.y, 1 / d.z));
        shear.init(d);
        st.reset();
        cur = sc.n_nodes ? 0u : TRAV_DONE;   // the root is always an interior BVH4 node
    }
    PT_DEV bool done() const { return cur == TRAV_DONE; }
    PT_DEV bool atLeaf() const { return cur != TRAV_DONE && (cur & BVH4_LEAF); }
    PT_DEV bool atNode() const { return !(cur & BVH4_LEAF); }
};

// one interior-node step: fetch the 128-byte node (near / far planes picked by address, per ray sign), test its
// four boxes, go to the nearest hit child and push the others far-to-near.  Empty child slots hold an inverted
// infinite box, so they fail the interval test without a separate check.
#ifndef PT_ANY_NOSORT
#define PT_ANY_NOSORT 0   /* experiment: shadow rays take the hit children in slot order instead of near-to-far */
#endif
template <bool COUNT, bool ORDERED = true>
PT_DEV void TravNodeStep(const DevScene &sc, TravState &ts, TravStack &st, TraceCounters *cnt) {
    const char *node = reinterpret_cast<const char *>(sc.nodes + ts.cur);
    const RayBox &rb = ts.box;
    float4 nx = *reinterpret_cast<const float4 *>(node + rb.offNearX), fx = *reinterpret_cast<const float4 *>(node + (48u - rb.offNearX));
    float4 ny = *reinterpret_cast<const float4 *>(node + rb.offNearY), fy = *reinterpret_cast<const float4 *>(node + (80u - rb.offNearY));
    float4 nz = *reinterpret_cast<const float4 *>(node + rb.offNearZ), fz = *reinterpret_cast<const float4 *>(node + (112u - rb.offNearZ));
    uint4 ch = *reinterpret_cast<const uint4 *>(node + 96);
    Pin(nx, fx, ny); Pin(fy, nz, fz); Pin(ch);
    if (COUNT) ++cnt->nodes;
    Float t0, t1, t2, t3;
    bool h0, h1, h2, h3;
#define PT_BOX(k, c, tk, hk)                                                                                       \
    {                                                                                                              \
        Float e = __builtin_fmaxf(__builtin_fmaxf((nx.c - rb.o.x) * rb.invNear.x, (ny.c - rb.o.y) * rb.invNear.y), \
                                  (nz.c - rb.o.z) * rb.invNear.z);                                                 \
        Float x = __builtin_fminf(__builtin_fminf((fx.c - rb.o.x) * rb.invFar.x, (fy.c - rb.o.y) * rb.invFar.y),   \
                                  (fz.c - rb.o.z) * rb.invFar.z);                                                  \
        hk = (e <= x) && (e < ts.tMax) && (x > 0);                                                                  \
        tk = hk ? e : PT_INFINITY;                                                                                 \
    }
    PT_BOX(0, x, t0, h0) PT_BOX(1, y, t1, h1) PT_BOX(2, z, t2, h2) PT_BOX(3, w, t3, h3)
#undef PT_BOX
    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
    if (!ORDERED) {   // any-hit: the visiting order cannot change the answer
        uint32_t nxt = TRAV_DONE;
        Float tn = 0;
        if (h3) { nxt = c3; tn = t3; }
        if (h2) { if (nxt != TRAV_DONE) st.push(nxt, tn); nxt = c2; tn = t2; }
        if (h1) { if (nxt != TRAV_DONE) st.push(nxt, tn); nxt = c1; tn = t1; }
        if (h0) { if (nxt != TRAV_DONE) st.push(nxt, tn); nxt = c0; tn = t0; }
        ts.cur = nxt != TRAV_DONE ? nxt : st.pop(ts.tMax);
        return;
    }
#define PT_CSWAP(ta, ca, tb, cb) if (tb < ta) { Float tt = ta; ta = tb; tb = tt; uint32_t cc = ca; ca = cb; cb = cc; }
    PT_CSWAP(t0, c0, t1, c1) PT_CSWAP(t2, c2, t3, c3) PT_CSWAP(t0, c0, t2, c2) PT_CSWAP(t1, c1, t3, c3) PT_CSWAP(t1, c1, t2, c2)
#undef PT_CSWAP
    int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (nh == 0) { ts.cur = st.pop(ts.tMax); return; }
    if (nh > 3) st.push(c3, t3);
    if (nh > 2) st.push(c2, t2);
    if (nh > 1) st.push(c1, t1);
    ts.cur = c0;
}

// ------------------------------------------------------------------ quantised nodes (pt_bvh4q.h): four requests per step instead of seven
// Same per-lane state machine and the same stack / leaf step as above; only the interior step differs: 4 x 16-byte loads of the 64-byte
// node, planes converted from 16-bit grid indices (v_cvt_f32_u32 with a word select), folded slab distances t = q A + B with the per-RAY
// constants of Bvh4qRayInit, hit children to the stack far to near.  The step itself is Bvh4qStepWords, the function the host emulation
// (mi_bvh4q_validate) checks against the oracle's BVH2 traversal.
struct TravStateQ : TravState {
    Bvh4qRay q;
    uint32_t pend;   // PT_PEND_LEAF: the leaf the lane still has triangles to test in (a leaf reference), TRAV_DONE: none
    template <class ST> PT_DEV void init(const DevScene &sc, const V3 &o_, const V3 &d_, Float tMax_, ST &st) {
        o = o_; d = d_; tMax = tMax_; tHit = 0; prim = TRAV_MISS;
        const float oo[3] = {o.x, o.y, o.z};
        // SlabRayInit's convention for zero direction components.  The reciprocal may be v_rcp_f32's (1 ulp) instead of the IEEE division's twelve
        // instructions: the folded box test is only required to be CONSERVATIVE, and its slack of 16 eps (|B| + 65535 |A|) covers the 3 eps a one-ulp
        // reciprocal adds to the evaluation error of pt_bvh4q.h's header (~10 of 16 eps in total); the triangle test's shear constants stay exact
        // (components below 1e-30 in magnitude are treated like zeros: their crossing times are beyond any tMax unless the origin is within 1e-27 of the
        // plane, and v_rcp_f32 must not see a denormal)
        const float inv[3] = {absf(d.x) < 1e-30f ? __builtin_copysignf(1e30f, d.x) : PtRcpBox(d.x), absf(d.y) < 1e-30f ? __builtin_copysignf(1e30f, d.y) : PtRcpBox(d.y),
                              absf(d.z) < 1e-30f ? __builtin_copysignf(1e30f, d.z) : PtRcpBox(d.z)};
        Bvh4qRayInit(q, sc.qgrid, oo, inv);
        shear.init(d);
        st.reset();
        cur = sc.n_nodes ? 0u : TRAV_DONE;
        pend = TRAV_DONE;
    }
};
// HOT > 0: the scene's sc.n_hot most visited nodes (indices 0 .. n_hot - 1 after mi_scene_upload's renumbering; pbrt_amd.hip: hot-node probe) sit in
// the block's LDS as four word planes hot[word][node] (stride HOT; a wave's lanes read the SAME word of different nodes, so consecutive nodes lie in
// consecutive 16-byte bank groups and lanes at the same node -- the root, its children -- are one broadcast): a step at such a node is
// 4 x ds_read_b128 and leaves the vector-memory path alone.  The words are the node's own, so the step is the same either way.
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const U32x4 LdsNodeWord;
template <bool COUNT, class ST> PT_DEV void TravNodeStepQWords(uint4 w0, uint4 w1, uint4 w2, uint4 ch, TravStateQ &ts, ST &st, TraceCounters *cnt);
template <bool COUNT, int HOT = 0, class ST = TravStack>
PT_DEV void TravNodeStepQ(const DevScene &sc, TravStateQ &ts, ST &st, TraceCounters *cnt, LdsNodeWord *hot = nullptr) {
    uint4 w0, w1, w2, ch;
    const bool isHot = HOT > 0 && ts.cur < sc.n_hot;
    if (isHot) {   // issued first: the LDS reads of the hot lanes are in flight while the others' global loads are
        LdsNodeWord *h = hot + ts.cur;
        const U32x4 a = h[0], b = h[HOT], c = h[2 * HOT], e = h[3 * HOT];   // 4 x ds_read_b128
        w0 = make_uint4(a.x, a.y, a.z, a.w); w1 = make_uint4(b.x, b.y, b.z, b.w); w2 = make_uint4(c.x, c.y, c.z, c.w); ch = make_uint4(e.x, e.y, e.z, e.w);
        if (COUNT) ++cnt->hot;
    }
    if (!isHot) {
        const uint4 *w = reinterpret_cast<const uint4 *>(sc.nodesq + ts.cur);
        w0 = w[0]; w1 = w[1]; w2 = w[2]; ch = w[3];
        if (HOT == 0) { Pin(w0); Pin(w1); Pin(w2); Pin(ch); }   // with hot nodes the wait belongs after the join: LDS reads and global loads of a mixed wave overlap
    }
    TravNodeStepQWords<COUNT>(w0, w1, w2, ch, ts, st, cnt);
}
// the step on the node's four 16-byte words, however they were fetched
// (round 3: the traversal kernels are bound by VALU issue as much as by memory, profiles/r03_c_*) the tail of the step with few
// instructions -- compare-exchanges as v_min / v_max on the distances + two selects on the references (the swap condition
// is still tb < ta), and the up-to-three pushes as one address + three predicated LDS stores while the lane's stack stays inside its LDS part.
template <bool COUNT, class ST>
PT_DEV void TravNodeStepQWords(uint4 w0, uint4 w1, uint4 w2, uint4 ch, TravStateQ &ts, ST &st, TraceCounters *cnt) {
    if (COUNT) ++cnt->nodes;
    const uint32_t wd[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, ch.x, ch.y, ch.z, ch.w};
    Float t[4];
    const uint32_t mask = Bvh4qStepWords(wd, ts.q, ts.tMax, t);
    Float t0 = (mask & 1u) ? t[0] : PT_INFINITY, t1 = (mask & 2u) ? t[1] : PT_INFINITY, t2 = (mask & 4u) ? t[2] : PT_INFINITY, t3 = (mask & 8u) ? t[3] : PT_INFINITY;
    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
    const int nh = __builtin_popcount(mask);
    // v_min_f32 / v_max_f32 straight (the distances are never NaN; the builtins would first canonicalise both operands: two more instructions each)
#define PT_MINMAX(lo, hi, a, b) lo = PtMinRaw(a, b); hi = PtMaxRaw(a, b);
#define PT_CSWAP(ta, ca, tb, cb) { const bool sw_ = tb < ta; Float lo_, hi_; PT_MINMAX(lo_, hi_, ta, tb) const uint32_t cl_ = sw_ ? cb : ca, ch_ = sw_ ? ca : cb; ta = lo_; tb = hi_; ca = cl_; cb = ch_; }
    PT_CSWAP(t0, c0, t1, c1) PT_CSWAP(t2, c2, t3, c3) PT_CSWAP(t0, c0, t2, c2) PT_CSWAP(t1, c1, t3, c3) PT_CSWAP(t1, c1, t2, c2)
#undef PT_CSWAP
#undef PT_MINMAX
    if (nh == 0) { ts.cur = st.pop(ts.tMax); return; }
    ts.cur = c0;
    if (nh > 1) {
        if (st.sp + 3 <= ST::LDS_ENTRIES) {   // entries sp, sp + 1, sp + 2 <- the hit children far to near: c[nh - 1], c[nh - 2], c[nh - 3]
            const uint32_t e0 = nh == 4 ? c3 : (nh == 3 ? c2 : c1), e1 = nh == 4 ? c2 : c1;
            typename ST::LdsPtr base = st.lds + st.sp * ST::STRIDE_;
            base[0] = e0;
            if (nh > 2) base[ST::STRIDE_] = e1;
            if (nh > 3) base[2 * ST::STRIDE_] = c1;
            st.sp += nh - 1;
        } else {
            if (nh > 3) st.push(c3, t3);
            if (nh > 2) st.push(c2, t2);
            st.push(c1, t1);
        }
    }
}

// ------------------------------------------------------------------ round 4: the interior step with a shorter tail (TraceShape::BIG instances of k_trace)
// The traversal kernels are bound by VALU issue (profiles/r04_*: ~3.6 cycles of SIMD time per VALU instruction of the real mix, VALU busy ~95 % at 24 waves
// per CU), and what counts is the NUMBER of VALU instructions: an A/B that moved a quarter of the step's instructions from the 4.1-cycle class of
// tools/valu_probe/issue_probe (compares, selects, conversions) to the 2.3-cycle class (adds, logic) at equal count changed nothing
// (profiles/r04_g_*).  The box arithmetic (12 plane selects, 24 conversions, 24 FMAs, 8 min3 / max3) and the 5-exchange sorting network are what the
// node format prescribes; this form shortens everything around them:
//   * entered <=> max(e, +0) <= min(x, pred(tMax)): ONE compare per child (was three compares + an empty-slot compare + two scalar ANDs), the sort key by one
//     select, the hit count by one carry-add per child (was four selects, two ORs and a population count);
//   * the stack pointer is the LDS byte ADDRESS of the lane's top entry and entry 0 of every lane holds TRAV_DONE, so a pop is `read, subtract` without an
//     emptiness test and the new top is one expression, top + (hits - 1) entries, "nothing hit" included;
//   * the three candidate pushes are UNCONDITIONAL ds_write_b32 at max(target, first free entry): a missed child lands in the first free entry and is
//     overwritten by the valid entry that belongs there (LDS operations of one wave complete in order), or stays as garbage above the top;
//   * the entry a pop would return is read speculatively with the node words, so "nothing hit" costs a select instead of an LDS round trip;
//   * deep stacks (a lane within 4 entries of its LDS part: 2-3 % of the steps) take the branching generic tail -- decided per WAVE.
// Same children entered as before up to one relaxed boundary case (x == +0 is accepted: a superset), same visiting order, same hits.
#if defined(__HIP_DEVICE_COMPILE__)
typedef int32_t PtLdsInt;     // an LDS address is 32 bits wide on the device (SIGNED: an address a few entries below a lane's stack may be negative, never huge)
#else
typedef intptr_t PtLdsInt;    // host build of this header: the "LDS" arrays are ordinary memory
#endif
#ifndef PT_FIFTH_LOAD
#define PT_FIFTH_LOAD 0   /* measurement switch, see TravNodeStepQ2 */
#endif
template <int STRIDE, int NLDS>
struct TravStackB {
    enum { LDS_ENTRIES = NLDS, STRIDE_ = STRIDE, SBYTES = STRIDE * 4 };
    typedef LdsStackEntry *LdsPtr;
    LdsStackEntry *tp;     // the lane's TOP entry ([entry][lane] rows, SBYTES apart); tp == base: only the sentinel is left
    LdsStackEntry *base;   // entry 0 = TRAV_DONE, written once by the kernel
    StackEntry *spill;     // entries beyond the LDS part, oldest first (non-empty only while the LDS part is full)
    int nspill;
    PT_DEV void reset() { tp = base; nspill = 0; }
    PT_DEV bool deep() const { return tp > base + (NLDS - 4) * STRIDE; }   // fewer than three free LDS entries (or spilled ones): the generic tail
    PT_DEV void push(uint32_t v, Float) {
        if (tp < base + (NLDS - 1) * STRIDE) { tp += STRIDE; *tp = v; } else spill[nspill++] = v;
    }
    PT_DEV uint32_t pop(Float) {
        if (nspill) return spill[--nspill];
        const uint32_t v = *tp;
        tp -= STRIDE;   // (below `base` after the sentinel went out: the lane's ray is finished, reset() comes before the next access)
        return v;
    }
};
template <bool COUNT, int HOT, class ST>
PT_DEV void TravNodeStepQ2(const DevScene &sc, TravStateQ &ts, ST &st, TraceCounters *cnt, LdsNodeWord *hot) {
    const bool deepWave = __builtin_amdgcn_ballot_w64(st.deep()) != 0ull;   // over the lanes that take this step (EXEC): some lane is within three entries of its LDS part
    uint4 w0, w1, w2, ch;
    const bool isHot = HOT > 0 && ts.cur < sc.n_hot;
    if (isHot) {
        LdsNodeWord *h = hot + ts.cur;
        const U32x4 a = h[0], b = h[HOT], c = h[2 * HOT], e = h[3 * HOT];   // 4 x ds_read_b128
        w0 = make_uint4(a.x, a.y, a.z, a.w); w1 = make_uint4(b.x, b.y, b.z, b.w); w2 = make_uint4(c.x, c.y, c.z, c.w); ch = make_uint4(e.x, e.y, e.z, e.w);
        if (COUNT) ++cnt->hot;
    }
    if (!isHot) {   // (a second `if`, not an `else`: the LDS reads are issued first and are in flight while the others' global loads are)
        const uint4 *w = reinterpret_cast<const uint4 *>(sc.nodesq + ts.cur);
        w0 = w[0]; w1 = w[1]; w2 = w[2]; ch = w[3];
    }
    const uint32_t spec = *st.tp;   // what a pop would return (valid whenever the fast tail runs: nothing spilled); in flight with the node words
#if PT_FIFTH_LOAD
    // MEASUREMENT ONLY (never shipped; profiles/r06_w_f16_node_probe.txt): what would the fifth 16-byte request of an 80-byte node cost this step?  The first word of
    // the NEXT node stands in for it -- the next 128-byte line for every second node, as with 80-byte nodes; one more ds_read_b128 for a hot node -- and is waited for
    // with the node words, unused.  (Reads 16 bytes of the node after the last one: inside the allocation's granule on the scenes this is run on.)
    {
        uint4 w4;
        w4 = make_uint4(0, 0, 0, 0);
        if (isHot && PT_FIFTH_LOAD != 3) { const U32x4 f = (hot + ts.cur + 1u)[0]; w4 = make_uint4(f.x, f.y, f.z, f.w); }   // (2: the LDS request only, 3: the global one only)
        if (!isHot && PT_FIFTH_LOAD != 2 && PT_FIFTH_LOAD != 4) w4 = reinterpret_cast<const uint4 *>(sc.nodesq + ts.cur + 1u)[0];
#if PT_FIFTH_LOAD == 4   // 4: the cold lanes load their OWN node's last word a second time (same 128-byte line, no new miss): is it the request or the line that costs?
        if (!isHot) {
            const uint4 *again = reinterpret_cast<const uint4 *>(sc.nodesq + ts.cur) + 3;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w4) : "v"(again) : "memory");   // (the wait inside: the compiler does not count this load)
        }
#endif
        Pin(w4);
    }
#endif
    if (COUNT) ++cnt->nodes;
    const uint32_t wd[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, ch.x, ch.y, ch.z, ch.w};
    Float e[4], x[4];
    Bvh4qStepEX(wd, ts.q, e, x);
    // entered <=> max(e, +0) <= min(x, pred(tMax)), i.e. e < tMax as the reference has it (a hit ON a box plane -- axis-aligned walls -- ends the
    // search there): two min / max, ONE compare, one select for the key (+inf for a miss) and one carry-add for the count
    const Float tMaxP = __uint_as_float(__float_as_uint(ts.tMax) - 1u);   // the float below tMax (tMax > 0; +inf -> FLT_MAX)
    Float t0, t1, t2, t3;
    uint32_t nh = 0;
#define PT_HITKEY(i, tk)                                                        \
    {                                                                           \
        const Float ec = PtMaxRaw(e[i], 0.0f), xc = PtMinRaw(x[i], tMaxP);      \
        const bool h = ec <= xc;                                                \
        tk = h ? e[i] : PT_INFINITY;   /* (the unclamped distance orders boxes the ray starts in, as before) */ \
        nh += h ? 1u : 0u;                                                      \
    }
    PT_HITKEY(0, t0) PT_HITKEY(1, t1) PT_HITKEY(2, t2) PT_HITKEY(3, t3)
#undef PT_HITKEY
    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
#define PT_CSWAP(ta, ca, tb, cb) { const bool sw_ = tb < ta; const Float lo_ = PtMinRaw(ta, tb), hi_ = PtMaxRaw(ta, tb); const uint32_t cl_ = sw_ ? cb : ca, ch_ = sw_ ? ca : cb; ta = lo_; tb = hi_; ca = cl_; cb = ch_; }
    PT_CSWAP(t0, c0, t1, c1) PT_CSWAP(t2, c2, t3, c3) PT_CSWAP(t0, c0, t2, c2) PT_CSWAP(t1, c1, t3, c3) PT_CSWAP(t1, c1, t2, c2)
#undef PT_CSWAP
    if (!deepWave) {
        // sorted position i (1..3) is a hit iff nh > i and then belongs at entry top + (nh - i); a missed position is sent to the first free entry top + 1,
        // written BEFORE the valid entry that belongs there (c3, c2, c1 in this order; LDS operations of one wave complete in order) or, when
        // nothing is pushed, left as garbage above the top
        constexpr uint32_t S = ST::SBYTES;
        const PtLdsInt tp0 = (PtLdsInt)(uintptr_t)st.tp, free0 = tp0 + (PtLdsInt)S, T = tp0 + (PtLdsInt)(nh * S);
        const PtLdsInt a3 = T - 3 * (PtLdsInt)S, a2 = T - 2 * (PtLdsInt)S, a1 = T - (PtLdsInt)S;
        *(typename ST::LdsPtr)(uintptr_t)(a3 > free0 ? a3 : free0) = c3;
        *(typename ST::LdsPtr)(uintptr_t)(a2 > free0 ? a2 : free0) = c2;
        *(typename ST::LdsPtr)(uintptr_t)(a1 > free0 ? a1 : free0) = c1;
        ts.cur = nh ? c0 : spec;                               // nothing hit: the speculative pop
        st.tp = (typename ST::LdsPtr)(uintptr_t)a1;            // top + (hits - 1): one down when nothing was hit
        return;
    }
    // generic tail (some lane of the wave is deep): the same decisions with per-lane branches
    if (nh == 0) { ts.cur = st.pop(ts.tMax); return; }
    ts.cur = c0;
    if (nh > 3) st.push(c3, 0);
    if (nh > 2) st.push(c2, 0);
    if (nh > 1) st.push(c1, 0);
}

// ------------------------------------------------------------------ two-level instancing (the host's default since round 2; PBRT_AMD_INSTANCING=0 flattens)
// TransformedPrimitive::Intersect / IntersectP (core/primitive.cpp:76-111) inside the per-lane state machine: meeting an instance
// primitive in a leaf, the lane pushes the rest of that leaf and a SENTINEL, takes its ray into the object's space
// (Transform::operator()(Ray), transform.h:252-264: error-bounded origin, tMax shortened accordingly) and continues at the object's
// own BVH; when the object's sub-tree is used up the sentinel comes off the stack and the world ray is restored with
// r.tMax = ray.tMax if something was hit in there.  Compiled in round 1, not yet run on a GPU (the default host mode flattens).
#define TRAV_SENTINEL 0xFFFFFFFEu   /* carries the leaf bit: arrives in the leaf step */
#define TRAV_NO_INSTANCE 0xFFFFFFFFu
struct DevInstance {
    float w2i[16], i2w[16];   // Inverse(PrimitiveToWorld), PrimitiveToWorld (row major)
    uint32_t root;            // BVH4 node of the object's tree
    uint32_t identity;        // PrimitiveToWorld.IsIdentity(): the interaction is not transformed back (primitive.cpp:92)
    uint32_t pad[2];
};
extern __constant__ const DevInstance *c_instances;
struct TravStateI : TravState {
    V3 wo, wd;            // the world-space ray while the lane is inside an instance
    Float wtMax;
    uint32_t inst, hitInst;
    bool ihit;
    template <class ST> PT_DEV void init(const DevScene &sc, const V3 &o_, const V3 &d_, Float tMax_, ST &st) {
        TravState::init(sc, o_, d_, tMax_, st);
        inst = hitInst = TRAV_NO_INSTANCE; ihit = false; wtMax = tMax_;
    }
};
PT_DEV void EnterInstance(TravStateI &ts, uint32_t idx) {
    const DevInstance &in = c_instances[idx];
    ts.wo = ts.o; ts.wd = ts.d; ts.wtMax = ts.tMax; ts.inst = idx; ts.ihit = false;
    V3 oErr;
    V3 o = SXfPointErr(in.w2i, ts.o, &oErr);
    V3 d = SXfVector(in.w2i, ts.d);
    Float lengthSquared = d.LengthSquared(), tm = ts.tMax;
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), oErr) / lengthSquared;
        o = o + d * dt;
        tm -= dt;
    }
    ts.o = o; ts.d = d; ts.tMax = tm;
    ts.box.init(o, V3(1 / d.x, 1 / d.y, 1 / d.z));
    ts.shear.init(d);
    ts.cur = in.root;
}
PT_DEV void LeaveInstance(TravStateI &ts) {
    Float t = ts.ihit ? ts.tMax : ts.wtMax;   // r.tMax = ray.tMax only after a hit (primitive.cpp:88-90)
    ts.o = ts.wo; ts.d = ts.wd; ts.tMax = t; ts.inst = TRAV_NO_INSTANCE;
    ts.box.init(ts.o, V3(1 / ts.d.x, 1 / ts.d.y, 1 / ts.d.z));
    ts.shear.init(ts.d);
}

// one leaf step = ONE triangle of the leaf (in primitive order; ties at equal t: the later one wins, as in the
// reference's loop, bvh.cpp:677-681 with triangle.cpp:258-261).  A lane stays at the leaf until its triangles are
// used up, so a leaf phase of the wave costs one watertight test whatever the leaf sizes of its lanes are.
// alpha masks (scenes with masked meshes only): true if the mask(s) of `prim`'s mesh evaluate to 0 at the hit (pt_material.h)
__device__ bool TriAlphaRejects(const uint4 *tri_info, const TriShade *tri_shade, uint32_t prim, const V3 p0, const V3 p1, const V3 p2, Float b0, Float b1,
                                Float b2, bool anyHit);
template <bool ANY, bool COUNT, bool SPHERES = false, bool ALPHA = false, class TS = TravState, class ST = TravStack, bool INST = false>
PT_DEV void TravLeafStep(const DevScene &sc, TS &ts, ST &st, TraceCounters *cnt) {
    if constexpr (INST) {
        if (ts.cur == TRAV_SENTINEL) { LeaveInstance(ts); ts.cur = st.pop(ts.tMax); return; }
    }
    uint32_t first = ts.cur & BVH4_FIRST_MASK, left = (ts.cur >> 27) & 0xfu;   // left = triangles after this one
    V3 p0, p1, p2;
    uint32_t flags;
    LoadTri(sc, first, &p0, &p1, &p2, &flags);
    if (COUNT) ++cnt->tris;
    if constexpr (INST) {
        if (flags & TRI_FLAG_INSTANCE) {   // the rest of this leaf, the sentinel, then into the object
            if (left) st.push(BVH4_LEAF | ((left - 1) << 27) | (first + 1), -PT_INFINITY);
            st.push(TRAV_SENTINEL, -PT_INFINITY);
            EnterInstance(ts, __float_as_uint(p0.x));
            return;
        }
    }
    TriHit th;
    bool hitPrim;
    if (SPHERES && (flags & TRI_FLAG_SPHERE)) {   // Sphere::Intersect / IntersectP (scenes with spheres only: separate kernel instance)
        th.t = SphereIntersectT(sc.spheres + __float_as_uint(p0.x), ts.o, ts.d, ts.tMax);
        th.b0 = th.b1 = th.b2 = 0;
        hitPrim = th.t >= 0;
    } else
        hitPrim = !(flags & TRI_FLAG_REJECT) && TriangleTest(p0, p1, p2, ts.o, ts.shear, ts.tMax, &th);
    if (ALPHA && hitPrim && (flags & TRI_FLAG_ALPHA)) hitPrim = !TriAlphaRejects(sc.tri_info, sc.tri_shade, first, p0, p1, p2, th.b0, th.b1, th.b2, ANY);
    if (hitPrim) {
        ts.prim = first;
        ts.tHit = th.t;
        if constexpr (INST) { ts.hitInst = ts.inst; if (ts.inst != TRAV_NO_INSTANCE) ts.ihit = true; }
        if (ANY) { ts.cur = TRAV_DONE; return; }
        ts.tMax = th.t;   // GeometricPrimitive::Intersect shrinks ray.tMax (core/primitive.cpp:120)
    }
    if (left) ts.cur = BVH4_LEAF | ((left - 1) << 27) | (first + 1);
    else ts.cur = st.pop(ts.tMax);
}

// PT_PEND_LEAF (round 3).  The traversal kernels are bound by VALU issue and a partially filled wave pays the full price per instruction
// (tools/valu_probe: ~3 cycles per wave instruction whatever the EXEC mask; profiles/r03_c_*: 47 % of the lanes active per VALU instruction), and the
// biggest idle group are lanes that reached a leaf and wait for the wave's next leaf phase.  With this option such a lane PARKS the leaf
// (TravStateQ::pend) and goes on with node steps from its stack; leaf phases test one triangle of every parked leaf.  A second leaf waits until
// the parked one is used up, so a ray's triangles are tested in the same order as before; node steps in between use the tMax of the moment
// (a few more nodes are visited: +1.5 % in the wave simulator of tools/bvh_study, for -12 % wave instructions).  Quantised single-level scenes only.
#ifndef PT_PEND_LEAF
#define PT_PEND_LEAF 1
#endif
// PT_ALPHA_DEFER (round 3, masked scenes with parked leaves): alpha masks are evaluated in wave-wide alpha phases, run when PT_ALPHA_MIN lanes wait
// for one or when the waiting lanes outnumber those that can go on without (a lane whose parked triangle waits for its mask still takes node steps)
#ifndef PT_ALPHA_DEFER
#define PT_ALPHA_DEFER 1
#endif
#ifndef PT_ALPHA_MIN
#define PT_ALPHA_MIN 32
#endif
#ifndef PT_ALPHA_GO_MUL   // ... outnumber PT_ALPHA_GO_MUL x the lanes that can go on
#define PT_ALPHA_GO_MUL 1
#endif
// if the lane stands at a leaf and has none parked: park it and take the next stack entry
template <class ST> PT_DEV void TravParkLeaf(TravStateQ &ts, ST &st) {
    if (ts.pend == TRAV_DONE && ts.cur != TRAV_DONE && (ts.cur & BVH4_LEAF)) { ts.pend = ts.cur; ts.cur = st.pop(ts.tMax); }
}
// one triangle of the parked leaf (TravLeafStep on ts.pend; the leaf's end frees the slot instead of popping).
// DEFER (PT_ALPHA_DEFER, masked scenes): a candidate hit on a masked triangle is not decided here -- *cand is set, the triangle stays parked, and the
// wave's next ALPHA PHASE repeats this step with DEFER = false for all such lanes together (k_trace): the mask texture is an interpreter call of
// hundreds of instructions, and in a kernel bound by VALU issue it costs the same for one lane as for sixty-four
template <bool ANY, bool COUNT, bool SPHERES, bool ALPHA, class ST, bool DEFER = false>
PT_DEV void TravPendStep(const DevScene &sc, TravStateQ &ts, ST &st, TraceCounters *cnt, bool *cand = nullptr) {
    const uint32_t first = ts.pend & BVH4_FIRST_MASK, left = (ts.pend >> 27) & 0xfu;
    V3 p0, p1, p2;
    uint32_t flags;
    LoadTri(sc, first, &p0, &p1, &p2, &flags);
    if (COUNT) ++cnt->tris;
    TriHit th;
    bool hitPrim;
    if (SPHERES && (flags & TRI_FLAG_SPHERE)) {
        th.t = SphereIntersectT(sc.spheres + __float_as_uint(p0.x), ts.o, ts.d, ts.tMax);
        th.b0 = th.b1 = th.b2 = 0;
        hitPrim = th.t >= 0;
    } else
        hitPrim = !(flags & TRI_FLAG_REJECT) && TriangleTest(p0, p1, p2, ts.o, ts.shear, ts.tMax, &th);
    if (ALPHA && hitPrim && (flags & TRI_FLAG_ALPHA)) {
        if constexpr (DEFER) { *cand = true; return; }   // (tMax cannot change before the alpha phase: only this lane's leaf steps shrink it)
        hitPrim = !TriAlphaRejects(sc.tri_info, sc.tri_shade, first, p0, p1, p2, th.b0, th.b1, th.b2, ANY);
    }
    if (hitPrim) {
        ts.prim = first;
        ts.tHit = th.t;
        if (ANY) { ts.cur = TRAV_DONE; ts.pend = TRAV_DONE; return; }
        ts.tMax = th.t;
    }
    if (left) ts.pend = BVH4_LEAF | ((left - 1) << 27) | (first + 1);
    else { ts.pend = TRAV_DONE; TravParkLeaf(ts, st); }
}
