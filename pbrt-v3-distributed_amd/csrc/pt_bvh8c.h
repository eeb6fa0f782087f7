// Compressed 8-wide BVH ("BVH8C"): 80-byte nodes, the traversal layout of round 2.
//
// Why.  Round 2's measurements on the 10 M-triangle frame (profiles/r02_*): halving the instructions of the BVH4 steps
// (pt_trace_fast.h) did not move the closest-hit kernel, adding three more 16-byte loads per step from the same cache line
// cost 7 %, the 128-byte BVH8 of round 1 (fewer steps, twice the arithmetic) was slower -- what the kernel pays for is the
// MEMORY SIDE of a step: 770 MB of 128-byte BVH4 nodes visited at random (36.5 per ray) against 4 MiB of L2 per XCD and a
// 256 MiB Infinity Cache.  This layout cuts both factors: 8-wide nodes (22-24 steps per ray) of 80 bytes (5 x 16-byte
// loads instead of 7; the whole tree of the 10 M-triangle scene is ~190 MB and fits the Infinity Cache):
//     p[3]        origin of the node's quantisation grid (its lower corner)                      12 B
//     e[3], imask cell size 2^(e-127) per axis (the IEEE exponent byte); bit k: child k is interior  4 B
//     child_base  index of the first interior child (interior children are stored consecutively)   4 B
//     tri_base    first triangle of the leaf children in the TRAVERSAL-ORDER triangle array          4 B
//     meta[8]     leaf child k: 0x80 | (count-1) << 5 | offset from tri_base (count <= 4); else 0     8 B
//     qlo / qhi   [axis][child] 8-bit planes: child k spans [p + qlo s, p + qhi s], a superset of its reference box  48 B
// The triangles of a node's leaf children sit back to back in a traversal-private copy of the records (tri order table
// trav -> reference primitive index kept for the hit record), so a child reference never needs more than the two bases.
// Box tests run in the folded form t = q (s / d) + (p - o) / d with explicit slack (as the round-1 BVH8, pt_bvh8.h): CONSERVATIVE
// with respect to the reference's Bounds3::IntersectP on the ORIGINAL boxes, hence the same closest hit; the watertight
// triangle test is the reference's, op for op.  Host side: builder (collapse of the reference's BVH2, checked in exact
// arithmetic) and an emulation of the per-ray state machine; mi_bvh8c_validate compares its hits with the oracle's bit for bit.
#pragma once
#include <stdint.h>

#include "pt_bvh8.h"   // PT_HD, the host copy of the watertight test

struct __attribute__((aligned(16))) BVH8CNode {
    float p[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t child_base, tri_base;
    uint8_t meta[8];
    uint8_t qlo[3][8], qhi[3][8];
};
static_assert(sizeof(BVH8CNode) == 80, "five 16-byte loads per node");
#define BVH8C_LEAF_MAX 4u

// One node step on the node's 20 words as the kernel holds them (5 x 16-byte loads); shared by the kernel and the host emulation.
// o / inv: ray origin and reciprocal direction (+-1e30 standing in for the infinities of zero components, Ray8Init in pt_bvh8.h).
// Returns the mask of children whose (quantised, slack-widened) boxes the ray may enter before tMax; t[k] = entry distance.
//   A = s inv, B = (p - o) inv per axis; plane q: t(q) = q A + B.  |error| of the evaluation <= 4 eps (|B| + 255 |A|); the reference's
//   far distance carries (1 + 2 gamma(3)) ~ 6 eps: delta = 16 eps (|B| + 255 |A|) on the near / far offsets covers both.
PT_HD uint32_t Bvh8cStepWords(const uint32_t w[20], float ox, float oy, float oz, float ix, float iy, float iz, float tMax, float t[8]) {
    const float px = Bvh8BitsToFloat(w[0]), py = Bvh8BitsToFloat(w[1]), pz = Bvh8BitsToFloat(w[2]);
    const uint32_t eb = w[3];
    const float sx = Bvh8BitsToFloat((eb & 255u) << 23), sy = Bvh8BitsToFloat(((eb >> 8) & 255u) << 23), sz = Bvh8BitsToFloat(((eb >> 16) & 255u) << 23);
    const uint32_t imask = eb >> 24;
    // valid children: interior or leaf (bit 7 of the meta byte)
    const uint32_t m0 = w[6], m1 = w[7];
    uint32_t leafmask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { leafmask |= ((m0 >> (8 * k + 7)) & 1u) << k; leafmask |= ((m1 >> (8 * k + 7)) & 1u) << (k + 4); }
    const uint32_t valid = imask | leafmask;
    const bool nx = ix < 0, ny = iy < 0, nz = iz < 0;
    // words: [8,9] qlo.x  [10,11] qlo.y  [12,13] qlo.z  [14,15] qhi.x  [16,17] qhi.y  [18,19] qhi.z
    const uint32_t nxa = nx ? w[14] : w[8], nxb = nx ? w[15] : w[9], fxa = nx ? w[8] : w[14], fxb = nx ? w[9] : w[15];
    const uint32_t nya = ny ? w[16] : w[10], nyb = ny ? w[17] : w[11], fya = ny ? w[10] : w[16], fyb = ny ? w[11] : w[17];
    const uint32_t nza = nz ? w[18] : w[12], nzb = nz ? w[19] : w[13], fza = nz ? w[12] : w[18], fzb = nz ? w[13] : w[19];
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
        if ((e <= x) && (e < tMax) && (x > 0) && ((valid >> k) & 1u)) mask |= 1u << k;
    }
    return mask;
}
// reference of child k: interior -> node index; leaf -> 0x80000000 | (count-1) << 27 | first triangle (traversal order), as BVH4 leaf references
PT_HD uint32_t Bvh8cChildRef(const uint32_t w[20], int k) {
    const uint32_t imask = w[3] >> 24;
    const uint32_t meta = ((k < 4 ? w[6] : w[7]) >> (8 * (k & 3))) & 255u;
    const uint32_t interior = w[4] + (uint32_t)__builtin_popcount(imask & ((1u << k) - 1u));
    const uint32_t leaf = 0x80000000u | (((meta >> 5) & 3u) << 27) | (w[5] + (meta & 31u));
    return ((imask >> k) & 1u) ? interior : leaf;
}

// ---- host side
#include <deque>

namespace bvh8c {
const uint32_t LEAF = 0x80000000u, EMPTY = 0xFFFFFFFFu, FIRST_MASK = 0x07ffffffu;

struct Builder {
    const mi_bvh2_node *n2 = nullptr;
    std::vector<BVH8CNode> out;
    std::vector<uint32_t> triOrder;   // traversal order -> reference primitive index
    int maxDepth = 0;
    std::string error;
    struct Src { bool big; uint32_t i2; uint32_t first, count; float lo[3], hi[3]; int depth; };   // a reference interior node, or a (piece of a) leaf with more than BVH8C_LEAF_MAX primitives
    struct Kid { bool interior; Src src; uint32_t first, count; float lo[3], hi[3]; };

    static float area(const mi_bvh2_node &n) {
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    void setGrid(BVH8CNode &nd, const float lo[3], const float hi[3]) {
        for (int a = 0; a < 3; ++a) {
            nd.p[a] = lo[a];
            double ext = (double)hi[a] - (double)lo[a];
            int e = -60;
            if (ext > 0) {
                e = (int)std::ceil(std::log2(ext / 255.0));
                while (std::ldexp(1.0, e) * 255.0 < ext) ++e;
                while (e > -120 && std::ldexp(1.0, e - 1) * 255.0 >= ext) --e;
            }
            e = std::max(-126, std::min(127, e));
            nd.e[a] = (uint8_t)(e + 127);
        }
    }
    bool setBox(BVH8CNode &nd, int k, const float lo[3], const float hi[3]) {
        for (int a = 0; a < 3; ++a) {
            double s = std::ldexp(1.0, (int)nd.e[a] - 127), p = nd.p[a];
            double ql = std::floor(((double)lo[a] - p) / s), qh = std::ceil(((double)hi[a] - p) / s);
            ql = std::min(255.0, std::max(0.0, ql)); qh = std::min(255.0, std::max(0.0, qh));
            if (p + ql * s > (double)lo[a] || p + qh * s < (double)hi[a]) { error = "quantised box does not contain the reference box"; return false; }
            nd.qlo[a][k] = (uint8_t)ql; nd.qhi[a][k] = (uint8_t)qh;
        }
        return true;
    }
    static Kid leafKid(uint32_t first, uint32_t count, const float lo[3], const float hi[3], int depth) {
        Kid k;
        std::memset(&k, 0, sizeof(k));
        for (int a = 0; a < 3; ++a) { k.lo[a] = lo[a]; k.hi[a] = hi[a]; }
        if (count <= BVH8C_LEAF_MAX) { k.interior = false; k.first = first; k.count = count; }
        else {   // too many primitives for one leaf reference: a node of its own whose children split the range (same box)
            k.interior = true;
            k.src.big = true; k.src.first = first; k.src.count = count; k.src.depth = depth + 1;
            for (int a = 0; a < 3; ++a) { k.src.lo[a] = lo[a]; k.src.hi[a] = hi[a]; }
        }
        return k;
    }
    bool process(uint32_t idx, const Src &s) {
        std::vector<Kid> kids;
        if (s.big) {
            uint32_t per = (s.count + 7) / 8;
            if (per > BVH8C_LEAF_MAX) per = ((per + BVH8C_LEAF_MAX - 1) / BVH8C_LEAF_MAX) * BVH8C_LEAF_MAX;
            uint32_t first = s.first, left = s.count;
            while (left > 0) { uint32_t c = std::min(per, left); kids.push_back(leafKid(first, c, s.lo, s.hi, s.depth)); first += c; left -= c; }
        } else {
            uint32_t ids[8];
            int nk = 2;
            ids[0] = s.i2 + 1; ids[1] = (uint32_t)n2[s.i2].offset;
            while (nk < 8) {   // open the interior child with the largest surface area, keeping the reference's left-to-right order
                int best = -1;
                float bestA = -1;
                for (int k = 0; k < nk; ++k)
                    if (n2[ids[k]].n_prims == 0) { float a = area(n2[ids[k]]); if (a > bestA) { bestA = a; best = k; } }
                if (best < 0) break;
                uint32_t o = ids[best];
                for (int k = nk; k > best + 1; --k) ids[k] = ids[k - 1];
                ids[best] = o + 1; ids[best + 1] = (uint32_t)n2[o].offset;
                ++nk;
            }
            for (int k = 0; k < nk; ++k) {
                const mi_bvh2_node &c = n2[ids[k]];
                if (c.n_prims > 0) kids.push_back(leafKid((uint32_t)c.offset, c.n_prims, c.bmin, c.bmax, s.depth));
                else {
                    Kid kd;
                    std::memset(&kd, 0, sizeof(kd));
                    kd.interior = true; kd.src.big = false; kd.src.i2 = ids[k]; kd.src.depth = s.depth + 1;
                    for (int a = 0; a < 3; ++a) { kd.lo[a] = c.bmin[a]; kd.hi[a] = c.bmax[a]; }
                    kids.push_back(kd);
                }
            }
        }
        if (kids.size() > 8) { error = "more than 8 children"; return false; }
        maxDepth = std::max(maxDepth, s.depth);
        BVH8CNode nd;
        std::memset(&nd, 0, sizeof(nd));
        float lo[3], hi[3];
        if (s.big) for (int a = 0; a < 3; ++a) { lo[a] = s.lo[a]; hi[a] = s.hi[a]; }
        else for (int a = 0; a < 3; ++a) { lo[a] = n2[s.i2].bmin[a]; hi[a] = n2[s.i2].bmax[a]; }
        setGrid(nd, lo, hi);
        for (int k = 0; k < 8; ++k) for (int a = 0; a < 3; ++a) { nd.qlo[a][k] = 255; nd.qhi[a][k] = 0; }   // empty slots: inverted (and not valid)
        uint32_t nInterior = 0;
        for (auto &k : kids) nInterior += k.interior;
        nd.child_base = (uint32_t)out.size();
        nd.tri_base = (uint32_t)triOrder.size();
        uint32_t nextChild = nd.child_base;
        out.resize(out.size() + nInterior);
        for (size_t k = 0; k < kids.size(); ++k) {
            if (!setBox(nd, (int)k, kids[k].lo, kids[k].hi)) return false;
            if (kids[k].interior) { nd.imask |= (uint8_t)(1u << k); pending.push_back({nextChild++, kids[k].src}); }
            else {
                uint32_t off = (uint32_t)triOrder.size() - nd.tri_base;
                if (off > 31 || kids[k].count < 1 || kids[k].count > BVH8C_LEAF_MAX) { error = "leaf offset / count out of range"; return false; }
                nd.meta[k] = (uint8_t)(0x80u | ((kids[k].count - 1) << 5) | off);
                for (uint32_t t = 0; t < kids[k].count; ++t) triOrder.push_back(kids[k].first + t);
            }
        }
        out[idx] = nd;
        return true;
    }
    std::deque<std::pair<uint32_t, Src>> pending;
    bool run(const mi_bvh2_node *nodes, uint32_t nNodes) {
        n2 = nodes;
        out.clear(); triOrder.clear(); pending.clear(); maxDepth = 0; error.clear();
        if (!nNodes) return true;
        out.resize(1);
        Src root;
        std::memset(&root, 0, sizeof(root));
        if (n2[0].n_prims > 0) {   // single-leaf tree: one node holding it
            root.big = true; root.first = (uint32_t)n2[0].offset; root.count = n2[0].n_prims;
            for (int a = 0; a < 3; ++a) { root.lo[a] = n2[0].bmin[a]; root.hi[a] = n2[0].bmax[a]; }
        } else { root.big = false; root.i2 = 0; }
        pending.push_back({0u, root});
        while (!pending.empty()) {
            auto it = pending.front();
            pending.pop_front();
            if (!process(it.first, it.second)) return false;
        }
        if (triOrder.size() > FIRST_MASK) { error = "more than 2^27 triangles"; return false; }
        return true;
    }
};

struct Stats { uint64_t nodes = 0, tris = 0, maxStack = 0, rays = 0, hits = 0; };
// the per-ray state machine of the kernel on the host: closest hit (anyHit = false) or first hit found (anyHit = true); primOut = REFERENCE index
inline bool traverse(const mi_scene_desc *d, const std::vector<BVH8CNode> &nodes, const std::vector<uint32_t> &triOrder, const mi_ray &ray, bool anyHit,
                     uint32_t *primOut, float *tOut, float bOut[3], Stats *st) {
    struct Entry { uint32_t ref; float t; };
    std::vector<Entry> stack(8 * 96);
    int sp = 0;
    Ray8 r8;
    Ray8Init(r8, ray.o, ray.d);
    bvh8::Shear sh;
    bvh8::shearInit(sh, ray.d);
    float tMax = ray.tmax;
    uint32_t prim = EMPTY;
    float bary[3] = {0, 0, 0}, tHit = 0;
    uint32_t cur = nodes.empty() ? EMPTY : 0u;
    auto pop = [&]() -> uint32_t {
        while (sp) { --sp; if (stack[sp].t < tMax) return stack[sp].ref; }
        return EMPTY;
    };
    while (cur != EMPTY) {
        if (!(cur & LEAF)) {
            float tn[8];
            uint32_t words[20];
            std::memcpy(words, &nodes[cur], 80);
            uint32_t mask = Bvh8cStepWords(words, r8.o[0], r8.o[1], r8.o[2], r8.inv[0], r8.inv[1], r8.inv[2], tMax, tn);
            ++st->nodes;
            int best = -1;
            for (int k = 0; k < 8; ++k) if ((mask >> k) & 1u) if (best < 0 || tn[k] < tn[best]) best = k;
            if (best < 0) { cur = pop(); continue; }
            for (int k = 7; k >= 0; --k)
                if (((mask >> k) & 1u) && k != best) { stack[sp].ref = Bvh8cChildRef(words, k); stack[sp].t = tn[k]; ++sp; }
            st->maxStack = std::max<uint64_t>(st->maxStack, (uint64_t)sp);
            cur = Bvh8cChildRef(words, best);
        } else {
            uint32_t first = cur & FIRST_MASK, count = ((cur >> 27) & 0xfu) + 1;
            for (uint32_t tt = first; tt < first + count; ++tt) {
                ++st->tris;
                uint32_t t = triOrder[tt];
                const uint32_t *v = d->tri_indices + 3 * (size_t)t;
                if (v[0] == MI_PRIM_SPHERE || v[0] == MI_PRIM_INSTANCE || bvh8::triangleRejected(d, t)) continue;
                float th, b[3];
                if (bvh8::triangleTest(d->P + 3 * (size_t)v[0], d->P + 3 * (size_t)v[1], d->P + 3 * (size_t)v[2], ray.o, sh, tMax, &th, b)) {
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
}  // namespace bvh8c
