// BVH8 with quantised child boxes: the next traversal layout, CPU-validated groundwork (no kernel launches it yet).
//
// Why: k_trace is bounded by the dependent fetch of a wave's slowest lane per step.  Collapsing the
// reference's BVH2 to 8-wide nodes cuts the interior steps per ray by ~38 % (36.5 -> 22.6 on the C3 stand-in) and an
// 8-bit quantisation of the child boxes on a per-node power-of-two grid keeps a node in ONE 128-byte cache line.  The box
// tests are evaluated in the folded form  t = q * (s / d) + (p - o) / d  (three operations per plane instead of five; static
// count of the step on gfx950: 215 VALU instructions for 8 boxes against 93 for the 4 full-precision boxes of the round-1
// node, tools/isa_probe), with explicit slack so that the test stays CONSERVATIVE with respect to the reference's
// Bounds3::IntersectP on the original boxes: every box the reference enters is entered, the closest hit is the same.
//
// This header holds (a) the node layout, (b) the per-node step `Bvh8Step`, written once for host and device, and, host only,
// (c) the builder (collapse + quantisation, checked in exact arithmetic) and (d) an emulation of the per-ray traversal state
// machine the kernel will run (nearest child first, the others pushed with their entry distance and culled at pop time, one
// triangle per leaf step) with a host copy of the watertight triangle test.  mi_bvh8_validate runs (c) + (d); the CPU test
// suite compares its hits with the oracle's BVH2 traversal bit for bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

struct __attribute__((aligned(128))) BVH8Node {
    float p[3];          // origin of the quantisation grid (= the node's lower corner)
    float s[3];          // cell size per axis, a power of two
    uint32_t child[8];   // as in BVH4Node: interior index | BVH4_LEAF | (count-1) << 27 | first | BVH4_EMPTY
    uint8_t qlo[3][8];   // child k spans [p + qlo * s, p + qhi * s] per axis (a superset of its reference box)
    uint8_t qhi[3][8];
    uint32_t pad[6];
};
static_assert(sizeof(BVH8Node) == 128, "one cache line per node");

struct Ray8 {   // per-ray constants of the node step
    float o[3], inv[3];   // inv: 1/d, with +-1e30 standing in for the infinities of axis-parallel rays (see Ray8Init)
    int neg[3];
};
PT_HD void Ray8Init(Ray8 &r, const float o[3], const float d[3]) {
    for (int a = 0; a < 3; ++a) {
        r.o[a] = o[a];
        // d == 0: the reference works with inv = +-inf and lets NaN comparisons fall through (conservative accept on that axis iff the
        // origin lies inside the slab); a huge finite inverse gives exactly "inside the (wider) quantised slab" without NaNs
        r.inv[a] = d[a] == 0 ? __builtin_copysignf(1e30f, d[a]) : 1 / d[a];
        r.neg[a] = r.inv[a] < 0;
    }
}
// One node step: entry distances of the (up to) 8 children whose quantised boxes the ray may enter before tMax; returns the hit mask.
//   A = s * inv, B = (p - o) * inv  per axis;  plane at q:  t(q) = q * A + B  up to rounding.  |error| <= 4 eps (|B| + 255 |A|) for the
//   evaluation itself, and the reference's own far distance is inflated by (1 + 2 gamma(3)) ~ 6 eps: delta = 16 eps (|B| + 255 |A|)
//   moved onto the near / far offsets covers both (eps = 2^-24).
PT_HD uint32_t Bvh8Step(const BVH8Node &n, const Ray8 &r, float tMax, float tNear[8]) {
    const float K = 16 * 5.9604644775390625e-08f;
    float A[3], Bn[3], Bf[3];
    for (int a = 0; a < 3; ++a) {
        A[a] = n.s[a] * r.inv[a];
        float B = (n.p[a] - r.o[a]) * r.inv[a];
        float delta = K * (__builtin_fabsf(B) + 255 * __builtin_fabsf(A[a]));
        Bn[a] = B - delta; Bf[a] = B + delta;
    }
    uint32_t mask = 0;
    for (int k = 0; k < 8; ++k) {
        float qn[3], qf[3];
        for (int a = 0; a < 3; ++a) {
            qn[a] = (float)(r.neg[a] ? n.qhi[a][k] : n.qlo[a][k]);
            qf[a] = (float)(r.neg[a] ? n.qlo[a][k] : n.qhi[a][k]);
        }
        float e = __builtin_fmaxf(__builtin_fmaxf(qn[0] * A[0] + Bn[0], qn[1] * A[1] + Bn[1]), qn[2] * A[2] + Bn[2]);
        float x = __builtin_fminf(__builtin_fminf(qf[0] * A[0] + Bf[0], qf[1] * A[1] + Bf[1]), qf[2] * A[2] + Bf[2]);
        bool hit = (e <= x) && (e < tMax) && (x > 0) && n.child[k] != 0xFFFFFFFFu;
        tNear[k] = e;
        if (hit) mask |= 1u << k;
    }
    return mask;
}

// The same step on the node's 28 payload words as the kernel holds them in registers (7 x 16-byte loads): word / byte selection by
// ray sign included.  The kernel (TravNodeStep8, pt_scene.h) and the host emulation below both call THIS function, so the packed
// path is what the CPU tests validate.  Word layout = BVH8Node: [0..2] p, [3..5] s, [6..13] child, [14..19] qlo[x|y|z][0..7], [20..25] qhi.
PT_HD float Bvh8BitsToFloat(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_HD uint32_t Bvh8StepWords(const uint32_t w[28], float ox, float oy, float oz, float ix, float iy, float iz, float tMax, float t[8]) {
    const float px = Bvh8BitsToFloat(w[0]), py = Bvh8BitsToFloat(w[1]), pz = Bvh8BitsToFloat(w[2]);
    const float sx = Bvh8BitsToFloat(w[3]), sy = Bvh8BitsToFloat(w[4]), sz = Bvh8BitsToFloat(w[5]);
    const bool nx = ix < 0, ny = iy < 0, nz = iz < 0;
    const uint32_t nxa = nx ? w[20] : w[14], nxb = nx ? w[21] : w[15], fxa = nx ? w[14] : w[20], fxb = nx ? w[15] : w[21];
    const uint32_t nya = ny ? w[22] : w[16], nyb = ny ? w[23] : w[17], fya = ny ? w[16] : w[22], fyb = ny ? w[17] : w[23];
    const uint32_t nza = nz ? w[24] : w[18], nzb = nz ? w[25] : w[19], fza = nz ? w[18] : w[24], fzb = nz ? w[19] : w[25];
    const float K = 16 * 5.9604644775390625e-08f;
    const float Ax = sx * ix, Bx = (px - ox) * ix, dx = K * (__builtin_fabsf(Bx) + 255 * __builtin_fabsf(Ax));
    const float Ay = sy * iy, By = (py - oy) * iy, dy = K * (__builtin_fabsf(By) + 255 * __builtin_fabsf(Ay));
    const float Az = sz * iz, Bz = (pz - oz) * iz, dz = K * (__builtin_fabsf(Bz) + 255 * __builtin_fabsf(Az));
    const float Bnx = Bx - dx, Bfx = Bx + dx, Bny = By - dy, Bfy = By + dy, Bnz = Bz - dz, Bfz = Bz + dz;
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sh = 8 * (k & 3);
        const uint32_t wnx = k < 4 ? nxa : nxb, wny = k < 4 ? nya : nyb, wnz = k < 4 ? nza : nzb;
        const uint32_t wfx = k < 4 ? fxa : fxb, wfy = k < 4 ? fya : fyb, wfz = k < 4 ? fza : fzb;
        float e = __builtin_fmaxf(__builtin_fmaxf((float)((wnx >> sh) & 255u) * Ax + Bnx, (float)((wny >> sh) & 255u) * Ay + Bny), (float)((wnz >> sh) & 255u) * Az + Bnz);
        float x = __builtin_fminf(__builtin_fminf((float)((wfx >> sh) & 255u) * Ax + Bfx, (float)((wfy >> sh) & 255u) * Ay + Bfy), (float)((wfz >> sh) & 255u) * Az + Bfz);
        t[k] = e;
        if ((e <= x) && (e < tMax) && (x > 0) && w[6 + k] != 0xFFFFFFFFu) mask |= 1u << k;
    }
    return mask;
}

// ---- host side (plain host functions: parsed in both compilation passes, emitted in the host pass only)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace bvh8 {
const uint32_t LEAF = 0x80000000u, EMPTY = 0xFFFFFFFFu, FIRST_MASK = 0x07ffffffu;
const uint32_t LEAF_MAX = 16;

struct Builder {
    const mi_bvh2_node *n2 = nullptr;
    std::vector<BVH8Node> out;
    int maxDepth = 0;
    std::string error;
    static float area(const mi_bvh2_node &n) {
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    static void clearNode(BVH8Node &nd) {
        std::memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 8; ++k) nd.child[k] = EMPTY;
        for (int a = 0; a < 3; ++a) nd.s[a] = 1;
    }
    // grid of a node with box [lo, hi]: p = lo, s = the smallest power of two with (hi - lo) / s <= 255 (exact arithmetic)
    static void setGrid(BVH8Node &nd, const float lo[3], const float hi[3]) {
        for (int a = 0; a < 3; ++a) {
            nd.p[a] = lo[a];
            double ext = (double)hi[a] - (double)lo[a];
            int e = -60;   // degenerate extent: any cell size works, a tiny one keeps the slack of the step tight
            if (ext > 0) {
                e = (int)std::ceil(std::log2(ext / 255.0));
                while (std::ldexp(1.0, e) * 255.0 < ext) ++e;
                while (e > -120 && std::ldexp(1.0, e - 1) * 255.0 >= ext) --e;
            }
            nd.s[a] = (float)std::ldexp(1.0, e);
        }
    }
    bool setChild(BVH8Node &nd, int k, const float lo[3], const float hi[3], uint32_t ref) {
        for (int a = 0; a < 3; ++a) {
            double s = nd.s[a], p = nd.p[a];
            double ql = std::floor(((double)lo[a] - p) / s), qh = std::ceil(((double)hi[a] - p) / s);
            ql = std::min(255.0, std::max(0.0, ql)); qh = std::min(255.0, std::max(0.0, qh));
            // exact check (p, s, q are short binary numbers: p + q * s is exact in double)
            if (p + ql * s > (double)lo[a] || p + qh * s < (double)hi[a]) { error = "quantised box does not contain the reference box"; return false; }
            nd.qlo[a][k] = (uint8_t)ql; nd.qhi[a][k] = (uint8_t)qh;
        }
        nd.child[k] = ref;
        return true;
    }
    // a reference leaf with more than LEAF_MAX primitives becomes a small chain of nodes with the leaf's box
    uint32_t leafRef(const mi_bvh2_node &lf, uint32_t first, uint32_t count, int depth) {
        if (count <= LEAF_MAX) return LEAF | ((count - 1) << 27) | first;
        uint32_t idx = (uint32_t)out.size();
        out.emplace_back();
        clearNode(out[idx]);
        setGrid(out[idx], lf.bmin, lf.bmax);
        maxDepth = std::max(maxDepth, depth + 1);
        uint32_t per = (count + 7) / 8;
        per = ((per + LEAF_MAX - 1) / LEAF_MAX) * LEAF_MAX;
        for (int k = 0; k < 8 && count > 0; ++k) {
            uint32_t c = std::min(per, count);
            uint32_t ref = leafRef(lf, first, c, depth + 1);
            if (!setChild(out[idx], k, lf.bmin, lf.bmax, ref)) return EMPTY;
            first += c; count -= c;
        }
        return idx;
    }
    uint32_t build(uint32_t i2, int depth) {   // i2: interior reference node
        uint32_t idx = (uint32_t)out.size();
        out.emplace_back();
        clearNode(out[idx]);
        setGrid(out[idx], n2[i2].bmin, n2[i2].bmax);
        maxDepth = std::max(maxDepth, depth);
        uint32_t kids[8];
        int nk = 2;
        kids[0] = i2 + 1; kids[1] = (uint32_t)n2[i2].offset;
        while (nk < 8) {   // open the interior child with the largest surface area (keeps the reference's left-to-right order)
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
            uint32_t ref = c.n_prims > 0 ? leafRef(c, (uint32_t)c.offset, c.n_prims, depth) : build(kids[k], depth + 1);
            if (!setChild(out[idx], k, c.bmin, c.bmax, ref)) return EMPTY;
        }
        return idx;
    }
    bool run(const mi_scene_desc *d) {
        n2 = d->bvh_nodes;
        if (!d->n_bvh_nodes) return true;
        if (n2[0].n_prims > 0) {   // single-leaf tree: wrap it in one node
            out.emplace_back();
            clearNode(out[0]);
            setGrid(out[0], n2[0].bmin, n2[0].bmax);
            uint32_t ref = leafRef(n2[0], (uint32_t)n2[0].offset, n2[0].n_prims, 0);
            return setChild(out[0], 0, n2[0].bmin, n2[0].bmax, ref);
        }
        build(0, 0);
        return error.empty();
    }
};

// ---- host copy of the watertight test (Triangle::Intersect shapes/triangle.cpp:188-291, as TriangleTest in pt_scene.h)
struct Shear { int kz; float Sx, Sy, Sz; };
inline void shearInit(Shear &rs, const float d[3]) {
    float ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
    rs.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    float dx, dy, dz;
    if (rs.kz == 0) { dx = d[1]; dy = d[2]; dz = d[0]; }
    else if (rs.kz == 1) { dx = d[2]; dy = d[0]; dz = d[1]; }
    else { dx = d[0]; dy = d[1]; dz = d[2]; }
    rs.Sx = -dx / dz; rs.Sy = -dy / dz; rs.Sz = 1.f / dz;
}
inline void permute(const Shear &rs, const float v[3], float o[3]) {
    if (rs.kz == 0) { o[0] = v[1]; o[1] = v[2]; o[2] = v[0]; }
    else if (rs.kz == 1) { o[0] = v[2]; o[1] = v[0]; o[2] = v[1]; }
    else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
}
inline float gammaN(int n) { const float e = 5.9604644775390625e-08f; return (n * e) / (1 - n * e); }
inline float max3(float a, float b, float c) { return std::max(a, std::max(b, c)); }
inline bool triangleTest(const float *P0, const float *P1, const float *P2, const float o[3], const Shear &rs, float tMax, float *tOut, float bOut[3]) {
    float a[3], b[3], c[3], p0t[3], p1t[3], p2t[3];
    for (int k = 0; k < 3; ++k) { a[k] = P0[k] - o[k]; b[k] = P1[k] - o[k]; c[k] = P2[k] - o[k]; }
    permute(rs, a, p0t); permute(rs, b, p1t); permute(rs, c, p2t);
    p0t[0] += rs.Sx * p0t[2]; p0t[1] += rs.Sy * p0t[2];
    p1t[0] += rs.Sx * p1t[2]; p1t[1] += rs.Sy * p1t[2];
    p2t[0] += rs.Sx * p2t[2]; p2t[1] += rs.Sy * p2t[2];
    float e0 = p1t[0] * p2t[1] - p1t[1] * p2t[0];
    float e1 = p2t[0] * p0t[1] - p2t[1] * p0t[0];
    float e2 = p0t[0] * p1t[1] - p0t[1] * p1t[0];
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t[0] * (double)p1t[1], p2typ1tx = (double)p2t[1] * (double)p1t[0];
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t[0] * (double)p2t[1], p0typ2tx = (double)p0t[1] * (double)p2t[0];
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t[0] * (double)p0t[1], p1typ0tx = (double)p1t[1] * (double)p0t[0];
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t[2] *= rs.Sz; p1t[2] *= rs.Sz; p2t[2] *= rs.Sz;
    float tScaled = e0 * p0t[2] + e1 * p1t[2] + e2 * p2t[2];
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = max3(std::fabs(p0t[2]), std::fabs(p1t[2]), std::fabs(p2t[2]));
    float deltaZ = gammaN(3) * maxZt;
    float maxXt = max3(std::fabs(p0t[0]), std::fabs(p1t[0]), std::fabs(p2t[0]));
    float maxYt = max3(std::fabs(p0t[1]), std::fabs(p1t[1]), std::fabs(p2t[1]));
    float deltaX = gammaN(5) * (maxXt + maxZt);
    float deltaY = gammaN(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammaN(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max3(std::fabs(e0), std::fabs(e1), std::fabs(e2));
    float deltaT = 3 * (gammaN(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * std::fabs(invDet);
    if (t <= deltaT) return false;
    *tOut = t; bOut[0] = b0; bOut[1] = b1; bOut[2] = b2;
    return true;
}
// per-triangle rejection of shapes/triangle.cpp:300-315 (what TRI_FLAG_REJECT records at upload time)
inline bool triangleRejected(const mi_scene_desc *d, uint32_t t) {
    const uint32_t *v = d->tri_indices + 3 * (size_t)t;
    const float *p0 = d->P + 3 * (size_t)v[0], *p1 = d->P + 3 * (size_t)v[1], *p2 = d->P + 3 * (size_t)v[2];
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
        if (crossLen2(a, b) == 0) return true;
    }
    return false;
}

struct Stats { uint64_t nodes = 0, tris = 0, maxStack = 0, rays = 0, hits = 0, mismatch = 0; };
// the per-ray state machine of the future kernel: closest hit (anyHit = false) or first hit found (anyHit = true)
inline bool traverse(const mi_scene_desc *d, const std::vector<BVH8Node> &nodes, const mi_ray &ray, bool anyHit, uint32_t *primOut, float *tOut, float bOut[3],
                     Stats *st) {
    struct Entry { uint32_t ref; float t; };
    Entry stack[8 * 64];
    int sp = 0;
    Ray8 r8;
    Ray8Init(r8, ray.o, ray.d);
    Shear sh;
    shearInit(sh, ray.d);
    float tMax = ray.tmax;
    uint32_t prim = EMPTY;
    float bary[3] = {0, 0, 0}, tHit = 0;
    uint32_t cur = nodes.empty() ? EMPTY : 0u;
    auto pop = [&]() -> uint32_t {
        while (sp) { --sp; if (stack[sp].t < tMax) return stack[sp].ref; }   // a box beyond the hit found meanwhile is dropped unfetched
        return EMPTY;
    };
    while (cur != EMPTY) {
        if (!(cur & LEAF)) {
            float tn[8], tn2[8];
            uint32_t words[32];
            std::memcpy(words, &nodes[cur], 128);
            uint32_t mask = Bvh8StepWords(words, r8.o[0], r8.o[1], r8.o[2], r8.inv[0], r8.inv[1], r8.inv[2], tMax, tn);   // what the kernel executes
            uint32_t mask2 = Bvh8Step(nodes[cur], r8, tMax, tn2);                                                       // the readable form
            if (mask != mask2 || std::memcmp(tn, tn2, sizeof(tn)) != 0) ++st->mismatch;
            ++st->nodes;
            const BVH8Node &n = nodes[cur];
            int best = -1;
            for (int k = 0; k < 8; ++k) if ((mask >> k) & 1u) if (best < 0 || tn[k] < tn[best]) best = k;
            if (best < 0) { cur = pop(); continue; }
            for (int k = 7; k >= 0; --k)   // the other hit children, with their entry distances (slot order, unsorted)
                if (((mask >> k) & 1u) && k != best) { stack[sp].ref = n.child[k]; stack[sp].t = tn[k]; ++sp; }
            st->maxStack = std::max<uint64_t>(st->maxStack, (uint64_t)sp);
            cur = n.child[best];
        } else {
            uint32_t first = cur & FIRST_MASK, count = ((cur >> 27) & 0xfu) + 1;
            for (uint32_t t = first; t < first + count; ++t) {   // primitive order; at equal t the later one wins (triangle.cpp:258-261)
                ++st->tris;
                const uint32_t *v = d->tri_indices + 3 * (size_t)t;
                if (v[0] == MI_PRIM_SPHERE || triangleRejected(d, t)) continue;   // spheres: not part of this study
                float th, b[3];
                if (triangleTest(d->P + 3 * (size_t)v[0], d->P + 3 * (size_t)v[1], d->P + 3 * (size_t)v[2], ray.o, sh, tMax, &th, b)) {
                    prim = t; tHit = th; bary[0] = b[0]; bary[1] = b[1]; bary[2] = b[2];
                    tMax = th;
                    if (anyHit) { sp = 0; break; }
                }
            }
            cur = (anyHit && prim != EMPTY) ? EMPTY : pop();
        }
    }
    ++st->rays;
    if (prim != EMPTY) ++st->hits;
    *primOut = prim; *tOut = tHit; bOut[0] = bary[0]; bOut[1] = bary[1]; bOut[2] = bary[2];
    return prim != EMPTY;
}
}  // namespace bvh8
