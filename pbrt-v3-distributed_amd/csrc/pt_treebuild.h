// The library's own TOPOLOGY over the reference's leaves (round 5; host side, included by pbrt_amd.hip).
//
// The boundary hands over the reference's BVHAccel (LinearBVHNode[], bvh.cpp:185-234): its leaves fix the primitive order every
// other table is indexed by, and that stays.  Its interior nodes are what the reference's build made of them -- 12 SAH buckets on
// the axis of the widest centroid extent (bvh.cpp:270-360) -- and nothing downstream depends on them: the device collapses
// whatever binary tree it is given into BVH4 nodes whose child boxes are that tree's node boxes.  RebuildOverLeaves builds a new
// binary tree over the SAME leaves (each leaf a unit with its box and its triangle count): all three axes, 32 bins per axis, an
// exact sweep below 2048 leaves.  Same leaves, same triangles per leaf, same order inside a leaf -- the set of triangles a ray
// tests stays a superset of what it must test, so the closest hit is the one the reference finds (ties at exactly equal t across
// leaves were never pinned: the BVH4 traversal visits children by distance, not in the reference's order).  tools/bvh_study.py
// --rebuild measured such a tree at -8..9 % node visits on the San-Miguel-class stand-in (profiles/r03_c_tree_study_1M.txt).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <system_error>
#include <thread>
#include <vector>

namespace treebuild {

struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); } }
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void growP(const float p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double area() const {
        double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0;
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
};
struct Leaf { Box b; float c[3]; int32_t offset; uint16_t nPrims; };
// joins its threads when the scope is left, also by an exception (bad_alloc in a split, a rethrown system_error): a joinable std::thread that is destroyed ends the process
struct JoinAll {
    std::vector<std::thread> th;
    ~JoinAll() { for (auto &x : th) if (x.joinable()) x.join(); }
};

struct Builder {
    std::vector<Leaf> leaves;   // permuted in place by the build
    std::vector<mi_bvh2_node> *out;
    unsigned nThreads = 1;
    static constexpr int NBINS = 32;
    static constexpr uint32_t SWEEP_BELOW = 2048;
    static constexpr uint32_t PAR_ABOVE = 1u << 19;   // ranges this long are scanned by all threads (the top levels of the tree, where the recursion itself has no parallelism yet)

    // f(chunkBegin, chunkEnd, chunkIndex) over [a, b) cut into nThreads chunks, concurrently
    template <class F> void parallelChunks(uint32_t a, uint32_t b, F f) const {
        const unsigned T = std::max(1u, nThreads);
        const uint32_t per = (b - a + T - 1) / T;
        JoinAll j;
        j.th.reserve(T);
        for (unsigned t = 1; t < T; ++t) {
            const uint32_t c0 = std::min<uint64_t>(b, (uint64_t)a + (uint64_t)t * per), c1 = std::min<uint64_t>(b, (uint64_t)c0 + per);
            if (c0 >= c1) continue;
            try { j.th.emplace_back([=]() { f(c0, c1, t); }); }
            catch (const std::system_error &) { f(c0, c1, t); }   // no thread to be had: the chunk on this one
        }
        f(a, std::min<uint64_t>(b, (uint64_t)a + per), 0u);
    }

    // nodes of the subtree over leaves [a, b) in DFS order (first child = this + 1): 2 (b - a) - 1 of them, starting at out[at]
    void build(uint32_t a, uint32_t b, uint32_t at, int depthLeft) {
        mi_bvh2_node *nd = out->data() + at;
        Box nb, cb;
        nb.clear(); cb.clear();
        if (b - a >= PAR_ABOVE && nThreads > 1) {
            std::vector<Box> pn(nThreads), pc(nThreads);
            for (unsigned t = 0; t < nThreads; ++t) { pn[t].clear(); pc[t].clear(); }
            parallelChunks(a, b, [&](uint32_t c0, uint32_t c1, unsigned t) { for (uint32_t i = c0; i < c1; ++i) { pn[t].grow(leaves[i].b); pc[t].growP(leaves[i].c); } });
            for (unsigned t = 0; t < nThreads; ++t) { nb.grow(pn[t]); cb.grow(pc[t]); }
        } else
            for (uint32_t i = a; i < b; ++i) { nb.grow(leaves[i].b); cb.growP(leaves[i].c); }
        std::memcpy(nd->bmin, nb.lo, sizeof(nb.lo)); std::memcpy(nd->bmax, nb.hi, sizeof(nb.hi));
        nd->pad = 0;
        if (b - a == 1) { nd->offset = leaves[a].offset; nd->n_prims = leaves[a].nPrims; nd->axis = 0; return; }
        uint32_t mid = 0;
        int axis = 0;
        split(a, b, cb, &mid, &axis);
        nd->n_prims = 0; nd->axis = (uint8_t)axis;
        const uint32_t left = at + 1, right = at + 1 + (2 * (mid - a) - 1);
        nd->offset = (int32_t)right;
        bool forked = false;
        if (depthLeft > 0 && b - a > 65536) {   // the two halves in parallel near the top of the tree
            try {
                JoinAll j;   //  joined also when the other half throws
                j.th.emplace_back([=]() { build(a, mid, left, depthLeft - 1); });
                forked = true;
                build(mid, b, right, depthLeft - 1);
            } catch (const std::system_error &) {   // no more threads to be had: this subtree serially
                if (forked) throw;                  // (only the constructor throws before `forked`; anything later is not ours to swallow)
            }
        }
        if (!forked) {
            build(a, mid, left, 0);
            build(mid, b, right, 0);
        }
    }
    // SAH over the leaves as units weighted by their triangle counts: cost(split) = A_l N_l + A_r N_r
    void split(uint32_t a, uint32_t b, const Box &cb, uint32_t *midOut, int *axisOut) {
        const uint32_t n = b - a;
        double best = std::numeric_limits<double>::infinity();
        int bestAxis = -1;
        if (n <= SWEEP_BELOW) {   // exact sweep: every position along every axis
            uint32_t bestPos = 0;
            int sortedBy = -1;
            std::vector<double> rightA(n);
            std::vector<uint32_t> rightN(n);
            for (int ax = 0; ax < 3; ++ax) {
                if (!(cb.hi[ax] > cb.lo[ax])) continue;
                std::sort(leaves.begin() + a, leaves.begin() + b, [ax](const Leaf &x, const Leaf &y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.offset < y.offset); });
                sortedBy = ax;
                Box rb; rb.clear();
                uint32_t rn = 0;
                for (uint32_t i = n; i-- > 1;) { rb.grow(leaves[a + i].b); rn += leaves[a + i].nPrims; rightA[i] = rb.area(); rightN[i] = rn; }
                Box lb; lb.clear();
                uint32_t ln = 0;
                for (uint32_t i = 1; i < n; ++i) {
                    lb.grow(leaves[a + i - 1].b); ln += leaves[a + i - 1].nPrims;
                    const double cst = lb.area() * ln + rightA[i] * rightN[i];
                    if (cst < best) { best = cst; bestAxis = ax; bestPos = i; }
                }
            }
            if (bestAxis >= 0) {
                const int ax = bestAxis;
                if (ax != sortedBy) std::sort(leaves.begin() + a, leaves.begin() + b, [ax](const Leaf &x, const Leaf &y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.offset < y.offset); });
                *midOut = a + bestPos; *axisOut = ax;
                return;
            }
        } else {
            int bestBin = -1;
            for (int ax = 0; ax < 3; ++ax) {
                if (!(cb.hi[ax] > cb.lo[ax])) continue;
                Box bb[NBINS];
                uint32_t cnt[NBINS];
                for (int k = 0; k < NBINS; ++k) { bb[k].clear(); cnt[k] = 0; }
                const double scale = NBINS / ((double)cb.hi[ax] - cb.lo[ax]);
                auto accumulate = [&](uint32_t c0, uint32_t c1, Box *pb, uint32_t *pcnt) {
                    for (uint32_t i = c0; i < c1; ++i) {
                        int k = (int)(((double)leaves[i].c[ax] - cb.lo[ax]) * scale);
                        k = k < 0 ? 0 : (k > NBINS - 1 ? NBINS - 1 : k);
                        pb[k].grow(leaves[i].b); pcnt[k] += leaves[i].nPrims;
                    }
                };
                if (n >= PAR_ABOVE && nThreads > 1) {   // (bin boxes are min / max and counts are integers: the merged result does not depend on the chunking)
                    std::vector<Box> pb((size_t)nThreads * NBINS);
                    std::vector<uint32_t> pcnt((size_t)nThreads * NBINS, 0u);
                    for (auto &x : pb) x.clear();
                    parallelChunks(a, b, [&](uint32_t c0, uint32_t c1, unsigned t) { accumulate(c0, c1, pb.data() + (size_t)t * NBINS, pcnt.data() + (size_t)t * NBINS); });
                    for (unsigned t = 0; t < nThreads; ++t) for (int k = 0; k < NBINS; ++k) { bb[k].grow(pb[(size_t)t * NBINS + k]); cnt[k] += pcnt[(size_t)t * NBINS + k]; }
                } else
                    accumulate(a, b, bb, cnt);
                double rA[NBINS];
                uint32_t rN[NBINS];
                Box rb; rb.clear();
                uint32_t rn = 0;
                for (int k = NBINS - 1; k >= 1; --k) { rb.grow(bb[k]); rn += cnt[k]; rA[k] = rb.area(); rN[k] = rn; }
                Box lb; lb.clear();
                uint32_t ln = 0;
                for (int k = 1; k < NBINS; ++k) {
                    lb.grow(bb[k - 1]); ln += cnt[k - 1];
                    if (ln == 0 || rN[k] == 0) continue;
                    const double cst = lb.area() * ln + rA[k] * rN[k];
                    if (cst < best) { best = cst; bestAxis = ax; bestBin = k; }
                }
            }
            if (bestAxis >= 0) {
                const int ax = bestAxis;
                const double scale = NBINS / ((double)cb.hi[ax] - cb.lo[ax]), lo = cb.lo[ax];
                const int bin = bestBin;
                auto m = std::partition(leaves.begin() + a, leaves.begin() + b, [=](const Leaf &x) {
                    int k = (int)(((double)x.c[ax] - lo) * scale);
                    k = k < 0 ? 0 : (k > NBINS - 1 ? NBINS - 1 : k);
                    return k < bin;
                });
                const uint32_t mid = (uint32_t)(m - leaves.begin());
                if (mid > a && mid < b) { *midOut = mid; *axisOut = ax; return; }
            }
        }
        // all centroids coincide (or no split separates anything): halve in the reference's order
        std::sort(leaves.begin() + a, leaves.begin() + b, [](const Leaf &x, const Leaf &y) { return x.offset < y.offset; });
        *midOut = a + n / 2; *axisOut = 0;
    }
};

// false (and *out untouched) when the tree has no interior node or is malformed
inline bool RebuildOverLeaves(const mi_bvh2_node *nodes, uint32_t nNodes, std::vector<mi_bvh2_node> *out) {
    if (!nodes || nNodes < 3 || nodes[0].n_prims > 0) return false;
    Builder bd;
    bd.leaves.reserve((nNodes + 1) / 2);
    for (uint32_t i = 0; i < nNodes; ++i) {
        const mi_bvh2_node &n = nodes[i];
        if (n.n_prims == 0) { if (n.offset <= (int32_t)i || (uint32_t)n.offset >= nNodes) return false; continue; }
        Leaf lf;
        std::memcpy(lf.b.lo, n.bmin, sizeof(n.bmin)); std::memcpy(lf.b.hi, n.bmax, sizeof(n.bmax));
        for (int a = 0; a < 3; ++a) {
            lf.c[a] = 0.5f * n.bmin[a] + 0.5f * n.bmax[a];
            if (!std::isfinite(lf.c[a])) return false;   // (boxes of spheres at infinity etc.: keep the reference's tree)
        }
        lf.offset = n.offset; lf.nPrims = n.n_prims;
        bd.leaves.push_back(lf);
    }
    if (2 * bd.leaves.size() - 1 != nNodes) return false;   // not a full binary tree
    std::vector<mi_bvh2_node> res(nNodes);
    bd.out = &res;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("PBRT_AMD_NTHREADS")) { int v = std::atoi(e); if (v > 0) hw = (unsigned)v; }
    bd.nThreads = std::min(64u, std::max(1u, hw));
    int depth = 0;
    while ((1u << depth) < std::max(1u, hw) && depth < 6) ++depth;
    bd.build(0, (uint32_t)bd.leaves.size(), 0, depth);
    out->swap(res);
    return true;
}

}  // namespace treebuild
