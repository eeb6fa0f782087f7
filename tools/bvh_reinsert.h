// Study code (NOT product): insertion-based optimisation of the topology the product builds over the reference's leaves (tools/bvh_study.py --reinsert).
// Measured in round 6's last session (profiles/r06_w_reinsertion_study_1M.txt) and not built into the library: see the figures there.
#pragma once
#include <utility>
#include "../pbrt-v3-distributed_amd/csrc/pt_treebuild.h"

namespace treebuild {

// ------------------------------------------------------------------ insertion-based optimisation of the topology (round 6, last session; study only)
// A top-down build decides every split once, from above; what it gets wrong stays.  This pass (after Bittner, Hapala, Havran, "Fast Insertion-Based
// Optimization of Bounding Volume Hierarchies", 2013, in the subtree form of Meister & Bittner 2018) takes a subtree X out of the finished tree -- its
// parent P goes with it, its sibling moves up -- and puts it back where the surface-area cost of the whole tree grows least: a branch-and-bound search
// from the root for the node Y that minimises  area(Y u X) + sum over Y's ancestors A of (area(A u X) - area(A)),  P becoming the parent of (Y, X).
// Subtrees are taken largest first (the upper levels carry most of the visits: 61 % of C3's land on 512 nodes), `frac` of the nodes per pass.
// Only links between nodes change: the leaves -- which triangles sit together, in which order -- are the reference's, as before.
struct Reinserter {
    struct N { Box b; double area; int32_t parent, c0, c1; };   // c0 < 0: leaf
    std::vector<N> nd;
    std::vector<int32_t> leafOffset;
    std::vector<uint16_t> leafPrims;
    int32_t root = 0;
    uint64_t moved = 0, searched = 0;

    void load(const std::vector<mi_bvh2_node> &in) {
        const size_t n = in.size();
        nd.resize(n); leafOffset.assign(n, 0); leafPrims.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(nd[i].b.lo, in[i].bmin, sizeof(in[i].bmin)); std::memcpy(nd[i].b.hi, in[i].bmax, sizeof(in[i].bmax));
            nd[i].area = nd[i].b.area();
            nd[i].parent = -1;
            if (in[i].n_prims) { nd[i].c0 = nd[i].c1 = -1; leafOffset[i] = in[i].offset; leafPrims[i] = in[i].n_prims; }
            else { nd[i].c0 = (int32_t)i + 1; nd[i].c1 = in[i].offset; }
        }
        for (size_t i = 0; i < n; ++i) if (nd[i].c0 >= 0) { nd[nd[i].c0].parent = (int32_t)i; nd[nd[i].c1].parent = (int32_t)i; }
        root = 0;
    }
    static double unionArea(const Box &a, const Box &b) { Box u = a; u.grow(b); return u.area(); }
    void refitFrom(int32_t i) {   // boxes of i and its ancestors from their children, until one does not change
        while (i >= 0) {
            Box u = nd[nd[i].c0].b; u.grow(nd[nd[i].c1].b);
            if (std::memcmp(&u, &nd[i].b, sizeof(Box)) == 0) break;
            nd[i].b = u; nd[i].area = u.area();
            i = nd[i].parent;
        }
    }
    void replaceChild(int32_t parent, int32_t was, int32_t now) {
        if (parent < 0) { root = now; nd[now].parent = -1; return; }
        if (nd[parent].c0 == was) nd[parent].c0 = now; else nd[parent].c1 = now;
        nd[now].parent = parent;
    }
    struct QE { double ci; int32_t node; bool operator<(const QE &o) const { return ci > o.ci; } };
    std::vector<QE> heap;
    // one reinsertion of the subtree at x; true if it ended somewhere else
    bool reinsert(int32_t x) {
        const int32_t p = nd[x].parent;
        if (p < 0 || nd[p].parent < 0) return false;   // the root and its children stay
        const int32_t s = nd[p].c0 == x ? nd[p].c1 : nd[p].c0, g = nd[p].parent;
        replaceChild(g, p, s);
        refitFrom(g);
        const Box xb = nd[x].b;
        const double xa = nd[x].area;
        double best = std::numeric_limits<double>::infinity();
        int32_t bestY = s;
        heap.clear();
        heap.push_back(QE{0.0, root});
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const QE e = heap.back(); heap.pop_back();
            if (e.ci + xa >= best) break;   // every remaining entry costs at least its induced cost + area(X)
            ++searched;
            const N &y = nd[e.node];
            const double cd = unionArea(y.b, xb), total = e.ci + cd;
            if (total < best) { best = total; bestY = e.node; }
            const double ci = e.ci + cd - y.area;   // what Y's own box grows by when X goes below it
            if (y.c0 >= 0 && ci + xa < best) {
                heap.push_back(QE{ci, y.c0}); std::push_heap(heap.begin(), heap.end());
                heap.push_back(QE{ci, y.c1}); std::push_heap(heap.begin(), heap.end());
            }
        }
        const int32_t y = bestY, gy = nd[y].parent;
        replaceChild(gy, y, p);
        nd[p].c0 = y; nd[p].c1 = x; nd[y].parent = p; nd[x].parent = p;
        nd[p].b = nd[y].b; nd[p].b.grow(xb); nd[p].area = nd[p].b.area();
        refitFrom(gy);
        if (y != s) ++moved;
        return y != s;
    }
    double sah() const {   // sum of interior areas + leaf areas x triangles, over the root's area
        double c = 0;
        for (size_t i = 0; i < nd.size(); ++i) c += nd[i].c0 >= 0 ? nd[i].area : nd[i].area * leafPrims[i];
        return c / std::max(nd[root].area, 1e-300);
    }
    void pass(double frac) {
        std::vector<std::pair<double, int32_t>> order;
        order.reserve(nd.size());
        for (size_t i = 0; i < nd.size(); ++i) if (nd[i].parent >= 0 && nd[nd[i].parent].parent >= 0) order.emplace_back(-nd[i].area, (int32_t)i);
        const size_t k = std::min(order.size(), (size_t)std::max(0.0, frac * (double)order.size()));
        std::partial_sort(order.begin(), order.begin() + k, order.end());
        for (size_t j = 0; j < k; ++j) reinsert(order[j].second);
    }
    // DFS order again (first child = this + 1): the form every consumer of mi_bvh2_node[] expects; children ordered along the axis their centroids differ most on
    void store(std::vector<mi_bvh2_node> *out) const {
        std::vector<mi_bvh2_node> res(nd.size());
        std::vector<std::pair<int32_t, uint32_t>> st;   // (node, index of the interior node whose second child it is, or ~0)
        uint32_t next = 0;
        st.emplace_back(root, ~0u);
        while (!st.empty()) {
            const auto e = st.back(); st.pop_back();
            const uint32_t at = next++;
            if (e.second != ~0u) res[e.second].offset = (int32_t)at;
            const N &n = nd[e.first];
            mi_bvh2_node &o = res[at];
            std::memcpy(o.bmin, n.b.lo, sizeof(o.bmin)); std::memcpy(o.bmax, n.b.hi, sizeof(o.bmax));
            o.pad = 0;
            if (n.c0 < 0) { o.offset = leafOffset[e.first]; o.n_prims = leafPrims[e.first]; o.axis = 0; continue; }
            int32_t a = n.c0, b = n.c1;
            int axis = 0;
            double sep = -1;
            for (int k = 0; k < 3; ++k) {
                const double ca = (double)nd[a].b.lo[k] + nd[a].b.hi[k], cb = (double)nd[b].b.lo[k] + nd[b].b.hi[k];
                if (std::fabs(ca - cb) > sep) { sep = std::fabs(ca - cb); axis = k; }
            }
            if ((double)nd[a].b.lo[axis] + nd[a].b.hi[axis] > (double)nd[b].b.lo[axis] + nd[b].b.hi[axis]) std::swap(a, b);
            o.n_prims = 0; o.axis = (uint8_t)axis; o.offset = 0;
            st.emplace_back(b, at);     // second child: after the whole first subtree
            st.emplace_back(a, ~0u);    // first child: next
        }
        out->swap(res);
    }
};
// `passes` passes over the `frac` largest subtrees of a tree in DFS order; returns the tree's SAH cost relative to before
inline double OptimizeByReinsertion(std::vector<mi_bvh2_node> *tree, double frac, int passes, uint64_t *movedOut = nullptr) {
    if (!tree || tree->size() < 7 || passes <= 0 || !(frac > 0)) return 1.0;
    Reinserter r;
    r.load(*tree);
    const double before = r.sah();
    for (int i = 0; i < passes; ++i) r.pass(frac);
    const double after = r.sah();
    r.store(tree);
    if (movedOut) *movedOut = r.moved;
    return after / std::max(before, 1e-300);
}

}  // namespace treebuild
