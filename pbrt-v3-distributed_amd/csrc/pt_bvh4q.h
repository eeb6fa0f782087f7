// BVH4Q: the BVH4 of pt_scene.h with its child boxes on a 16-bit grid over the scene bound -- 64-byte nodes, FOUR 16-byte loads per step.
//
// Why (round 2, profiles/r02_c_*).  What bounds the traversal kernels is the number of per-lane vector-memory requests the CU's
// texture-address / L1 pipe has to serve, not bytes, HBM traffic, arithmetic or occupancy: halving the instructions of a step, cutting
// the HBM traffic by 63 % (ray binning) and going from 24 to 16 waves per CU all left the closest-hit kernel where it was, whereas three
// more 16-byte loads per node step (7 -> 10, same cache line, distinct addresses) made it 31 % slower.  A step of the 128-byte node
// issues seven such requests per lane (6 x 4 planes + the child references); this layout issues four:
//     lo[3][4], hi[3][4]   uint16 plane indices: child k spans [g.lo + lo * g.cell, g.lo + hi * g.cell] per axis, a superset of its
//                          reference box (floor / ceil, checked in exact arithmetic);  48 B = three loads
//     child[4]             as BVH4Node::child;                                           16 B = one load
// with ONE grid for the whole tree (g.lo = the root box's lower corner, g.cell = a power of two with extent / cell <= 65535 per axis: 1 mm
// cells on a 40 m scene).  No per-node header: the folded slab constants A = cell / d, B = (g.lo - o) / d are per RAY.
// The box test is t = q A + B with explicit slack: delta = 16 eps (|B| + 65535 |A|) moved onto the
// near / far offsets keeps it CONSERVATIVE with respect to Bounds3::IntersectP on the reference's boxes (evaluation error <= 2 eps q |A| +
// 3 eps |B| + the reference's own (1 + 2 gamma(3)) on the far side): every box the reference enters is entered, the closest hit is the same.
#pragma once
#include <stdint.h>

#include "pt_hostcheck.h"   // PT_HD, SlabRay / SlabRayInit (+-1e30 for zero direction components), host copy of the watertight test

struct __attribute__((aligned(64))) BVH4QNode {
    uint16_t lo[3][4], hi[3][4];
    uint32_t child[4];
};
static_assert(sizeof(BVH4QNode) == 64, "four 16-byte loads per node");
struct Bvh4qGrid { float lo[3], cell[3]; };

// per-ray constants of the folded test
struct Bvh4qRay { float A[3], Bn[3], Bf[3]; int neg[3]; };
PT_HD void Bvh4qRayInit(Bvh4qRay &r, const Bvh4qGrid &g, const float o[3], const float inv[3]) {
    const float K = 16 * 5.9604644775390625e-08f;
    for (int a = 0; a < 3; ++a) {
        r.A[a] = g.cell[a] * inv[a];
        // zero direction components come in as inv = +-1e30: B and the slack may overflow on huge scenes / far origins; clamped to the
        // largest finite float they stay on the conservative side and inf - inf (NaN: every child culled) cannot occur.  A itself is
        // finite for every grid mi_scene_upload accepts (cell * 1e30 finite), else the scene takes the full-precision nodes.
        const float FM = 3.4028234663852886e+38f;
        float B = __builtin_fminf(__builtin_fmaxf((g.lo[a] - o[a]) * inv[a], -FM), FM);
        float delta = __builtin_fminf(K * (__builtin_fabsf(B) + 65535 * __builtin_fabsf(r.A[a])), FM);
        r.Bn[a] = B - delta; r.Bf[a] = B + delta;
        r.neg[a] = inv[a] < 0;
    }
}
// One node step on the node's 16 words (4 x 16-byte loads): mask of the children the ray may enter before tMax, t[k] = entry distance.
// words: [0,1] lo.x[0..3]  [2,3] lo.y  [4,5] lo.z  [6,7] hi.x  [8,9] hi.y  [10,11] hi.z  [12..15] child
PT_HD uint32_t Bvh4qStepWords(const uint32_t w[16], const Bvh4qRay &r, float tMax, float t[4]) {
    const uint32_t nx0 = r.neg[0] ? w[6] : w[0], nx1 = r.neg[0] ? w[7] : w[1], fx0 = r.neg[0] ? w[0] : w[6], fx1 = r.neg[0] ? w[1] : w[7];
    const uint32_t ny0 = r.neg[1] ? w[8] : w[2], ny1 = r.neg[1] ? w[9] : w[3], fy0 = r.neg[1] ? w[2] : w[8], fy1 = r.neg[1] ? w[3] : w[9];
    const uint32_t nz0 = r.neg[2] ? w[10] : w[4], nz1 = r.neg[2] ? w[11] : w[5], fz0 = r.neg[2] ? w[4] : w[10], fz1 = r.neg[2] ? w[5] : w[11];
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sh = 16 * (k & 1);
        const uint32_t wnx = k < 2 ? nx0 : nx1, wny = k < 2 ? ny0 : ny1, wnz = k < 2 ? nz0 : nz1;
        const uint32_t wfx = k < 2 ? fx0 : fx1, wfy = k < 2 ? fy0 : fy1, wfz = k < 2 ? fz0 : fz1;
        float e = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((wnx >> sh) & 65535u), r.A[0], r.Bn[0]), __builtin_fmaf((float)((wny >> sh) & 65535u), r.A[1], r.Bn[1])),
                                  __builtin_fmaf((float)((wnz >> sh) & 65535u), r.A[2], r.Bn[2]));
        float x = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((wfx >> sh) & 65535u), r.A[0], r.Bf[0]), __builtin_fmaf((float)((wfy >> sh) & 65535u), r.A[1], r.Bf[1])),
                                  __builtin_fmaf((float)((wfz >> sh) & 65535u), r.A[2], r.Bf[2]));
        t[k] = e;
        if ((e <= x) && (e < tMax) && (x > 0) && w[12 + k] != 0xFFFFFFFFu) mask |= 1u << k;
    }
    return mask;
}

// The same node step in the form the round-4 traversal tail wants it (TravNodeStepQ2, pt_scene.h): per child the entry distance e = max3(near
// planes) and the exit distance x = min3(far planes), nothing else.  The caller enters a child iff max(e, +0) <= min(x, pred(tMax)): Bvh4qStepWords' rule
// with x > 0 relaxed to x >= +0 (a superset, still conservative with respect to Bounds3::IntersectP) and
// WITHOUT the explicit empty-slot test: an empty slot (inverted box q_lo = 65535, q_hi = 0, reference 0xFFFFFFFF) fails e <= x by itself as long as the
// ray starts within ~5e5 grid extents of the grid on some axis -- x - e <= -(65535 |A_a| - 2 delta_a) with delta_a = 16 eps (|B_a| + 65535 |A_a|) for every
// axis a.  Path vertices lie inside the root box; mi_scene_upload checks the camera (else the scene takes the full-precision nodes).
PT_HD void Bvh4qStepEX(const uint32_t w[16], const Bvh4qRay &r, float e[4], float x[4]) {
    const uint32_t nx0 = r.neg[0] ? w[6] : w[0], nx1 = r.neg[0] ? w[7] : w[1], fx0 = r.neg[0] ? w[0] : w[6], fx1 = r.neg[0] ? w[1] : w[7];
    const uint32_t ny0 = r.neg[1] ? w[8] : w[2], ny1 = r.neg[1] ? w[9] : w[3], fy0 = r.neg[1] ? w[2] : w[8], fy1 = r.neg[1] ? w[3] : w[9];
    const uint32_t nz0 = r.neg[2] ? w[10] : w[4], nz1 = r.neg[2] ? w[11] : w[5], fz0 = r.neg[2] ? w[4] : w[10], fz1 = r.neg[2] ? w[5] : w[11];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sh = 16 * (k & 1);
        const uint32_t wnx = k < 2 ? nx0 : nx1, wny = k < 2 ? ny0 : ny1, wnz = k < 2 ? nz0 : nz1;
        const uint32_t wfx = k < 2 ? fx0 : fx1, wfy = k < 2 ? fy0 : fy1, wfz = k < 2 ? fz0 : fz1;
        e[k] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((wnx >> sh) & 65535u), r.A[0], r.Bn[0]), __builtin_fmaf((float)((wny >> sh) & 65535u), r.A[1], r.Bn[1])),
                               __builtin_fmaf((float)((wnz >> sh) & 65535u), r.A[2], r.Bn[2]));
        x[k] = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((wfx >> sh) & 65535u), r.A[0], r.Bf[0]), __builtin_fmaf((float)((wfy >> sh) & 65535u), r.A[1], r.Bf[1])),
                               __builtin_fmaf((float)((wfz >> sh) & 65535u), r.A[2], r.Bf[2]));
    }
}

// ---- host side: quantiser over an existing BVH4 (the collapse of the reference's BVH2), checks, emulation of the traversal
#include <cmath>
#include <string>
#include <vector>

namespace bvh4q {
const uint32_t LEAF = 0x80000000u, EMPTY = 0xFFFFFFFFu, FIRST_MASK = 0x07ffffffu;

// BVH4NodeF: any struct with lox/loy/loz/hix/hiy/hiz[4] + child[4] (BVH4Node of pt_scene.h)
template <class BVH4NodeF>
inline bool quantise(const std::vector<BVH4NodeF> &in, const float rootLo[3], const float rootHi[3], std::vector<BVH4QNode> *out, Bvh4qGrid *g, std::string *err) {
    for (int a = 0; a < 3; ++a) {
        g->lo[a] = rootLo[a];
        double ext = (double)rootHi[a] - (double)rootLo[a];
        int e = -20;
        if (ext > 0) {
            e = (int)std::ceil(std::log2(ext / 65535.0));
            while (std::ldexp(1.0, e) * 65535.0 < ext) ++e;
            while (e > -120 && std::ldexp(1.0, e - 1) * 65535.0 >= ext) --e;
        }
        g->cell[a] = (float)std::ldexp(1.0, e);
    }
    out->resize(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
        const BVH4NodeF &n = in[i];
        BVH4QNode &q = (*out)[i];
        for (int k = 0; k < 4; ++k) {
            q.child[k] = n.child[k];
            const float lo[3] = {n.lox[k], n.loy[k], n.loz[k]}, hi[3] = {n.hix[k], n.hiy[k], n.hiz[k]};
            for (int a = 0; a < 3; ++a) {
                if (n.child[k] == EMPTY) { q.lo[a][k] = 65535; q.hi[a][k] = 0; continue; }
                double s = g->cell[a], p = g->lo[a];
                double ql = std::floor(((double)lo[a] - p) / s), qh = std::ceil(((double)hi[a] - p) / s);
                if (ql < 0 || qh > 65535 || ql > qh) { *err = "a child box lies outside the root box (the grid)"; return false; }
                if (p + ql * s > (double)lo[a] || p + qh * s < (double)hi[a]) { *err = "quantised box does not contain the reference box"; return false; }
                q.lo[a][k] = (uint16_t)ql; q.hi[a][k] = (uint16_t)qh;
            }
        }
    }
    return true;
}

struct Stats { uint64_t nodes = 0, tris = 0, maxStack = 0, rays = 0, hits = 0; };
// the per-ray state machine of the kernel (nearest hit child first, the others pushed far to near; one triangle per leaf step) on the host
inline bool traverse(const mi_scene_desc *d, const std::vector<BVH4QNode> &nodes, const Bvh4qGrid &g, const mi_ray &ray, bool anyHit, uint32_t *primOut, float *tOut, float bOut[3],
                     Stats *st) {
    std::vector<uint32_t> stack(4 * 96);
    int sp = 0;
    SlabRay r8;
    SlabRayInit(r8, ray.o, ray.d);
    Bvh4qRay qr;
    Bvh4qRayInit(qr, g, r8.o, r8.inv);
    hostcheck::Shear sh;
    hostcheck::shearInit(sh, ray.d);
    float tMax = ray.tmax;
    uint32_t prim = EMPTY;
    float bary[3] = {0, 0, 0}, tHit = 0;
    uint32_t cur = nodes.empty() ? EMPTY : 0u;
    while (cur != EMPTY) {
        if (!(cur & LEAF)) {
            float t[4];
            uint32_t words[16];
            std::memcpy(words, &nodes[cur], 64);
            uint32_t mask = Bvh4qStepWords(words, qr, tMax, t);
            ++st->nodes;
            int order[4], nh = 0;
            for (int k = 0; k < 4; ++k) if ((mask >> k) & 1u) order[nh++] = k;
            for (int i = 1; i < nh; ++i) for (int j = i; j > 0 && t[order[j]] < t[order[j - 1]]; --j) std::swap(order[j], order[j - 1]);   // near to far, stable
            if (nh == 0) { cur = sp ? stack[--sp] : EMPTY; continue; }
            for (int i = nh - 1; i >= 1; --i) stack[sp++] = words[12 + order[i]];
            st->maxStack = std::max<uint64_t>(st->maxStack, (uint64_t)sp);
            cur = words[12 + order[0]];
        } else {
            uint32_t first = cur & FIRST_MASK, count = ((cur >> 27) & 0xfu) + 1;
            for (uint32_t t = first; t < first + count; ++t) {
                ++st->tris;
                const uint32_t *v = d->tri_indices + 3 * (size_t)t;
                if (v[0] == MI_PRIM_SPHERE || v[0] == MI_PRIM_INSTANCE || hostcheck::triangleRejected(d, t)) continue;
                float th, b[3];
                if (hostcheck::triangleTest(d->P + 3 * (size_t)v[0], d->P + 3 * (size_t)v[1], d->P + 3 * (size_t)v[2], ray.o, sh, tMax, &th, b)) {
                    prim = t; tHit = th; bary[0] = b[0]; bary[1] = b[1]; bary[2] = b[2];
                    tMax = th;
                    if (anyHit) { sp = 0; break; }
                }
            }
            cur = (anyHit && prim != EMPTY) ? EMPTY : (sp ? stack[--sp] : EMPTY);
        }
    }
    ++st->rays;
    if (prim != EMPTY) ++st->hits;
    *primOut = prim; *tOut = tHit; bOut[0] = bary[0]; bOut[1] = bary[1]; bOut[2] = bary[2];
    return prim != EMPTY;
}
}  // namespace bvh4q
