// Offline study tool (NOT product code): node / triangle visit statistics of alternative BVH layouts for the traversal kernel,
// on the host, over rays supplied by tools/bvh_study.py.  Variants: the reference's BVH2 order, the device's BVH4 collapse
// (ordered, with and without dropping popped nodes beyond the current hit), and a BVH8 collapse of the same tree.
//   g++ -O2 -std=c++14 -shared -fPIC tools/bvh_study.cpp -Iinclude -o /tmp/libbvhstudy.so
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pbrt_amd.h"
#include <cstdlib>
#include "bvh_reinsert.h"   // the product's own-topology builder (pt_treebuild.h) + the reinsertion pass studied on top of it (--reinsert)

namespace {
struct WNode { int n; float lo[8][3], hi[8][3]; uint32_t child[8]; };   // child: leaf bit 31 | first prim, count in leafCount
struct Wide {
    std::vector<WNode> nodes;
    std::vector<uint8_t> leafCount;   // per leaf ref -> stored separately keyed by (node, slot)
    std::vector<std::vector<uint8_t>> cnt;
};
const uint32_t LEAF = 0x80000000u;

float area(const mi_bvh2_node &n) {
    float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
uint32_t buildWide(const mi_bvh2_node *n2, uint32_t i2, int width, std::vector<WNode> &out, std::vector<std::vector<uint8_t>> &cnt) {
    uint32_t idx = (uint32_t)out.size();
    out.emplace_back();
    cnt.emplace_back(8, 0);
    uint32_t kids[8];
    int nk = 2;
    kids[0] = i2 + 1; kids[1] = (uint32_t)n2[i2].offset;
    while (nk < width) {
        int best = -1; float bestA = -1;
        for (int k = 0; k < nk; ++k) if (n2[kids[k]].n_prims == 0) { float a = area(n2[kids[k]]); if (a > bestA) { bestA = a; best = k; } }
        if (best < 0) break;
        uint32_t o = kids[best];
        for (int k = nk; k > best + 1; --k) kids[k] = kids[k - 1];
        kids[best] = o + 1; kids[best + 1] = (uint32_t)n2[o].offset;
        ++nk;
    }
    out[idx].n = nk;
    for (int k = 0; k < nk; ++k) {
        const mi_bvh2_node &c = n2[kids[k]];
        for (int a = 0; a < 3; ++a) { out[idx].lo[k][a] = c.bmin[a]; out[idx].hi[k][a] = c.bmax[a]; }
        if (c.n_prims > 0) { out[idx].child[k] = LEAF | (uint32_t)c.offset; cnt[idx][k] = (uint8_t)std::min<int>(255, c.n_prims); }
        else { uint32_t ch = buildWide(n2, kids[k], width, out, cnt); out[idx].child[k] = ch; }
    }
    return idx;
}

// ---- SAH-optimal collapse (dynamic programme over the reference tree, after Ylitie et al. 2017 s.3.2): which descendants of a
// BVH2 node become the <= W children of its wide node so that  sum over wide nodes A*cNode + sum over leaves A*n*cTri  is least.
// mergeMax > 0 also lets a whole subtree of <= mergeMax primitives (contiguous in the reference's ordering) become one leaf.
struct Collapse {
    const mi_bvh2_node *n2; int W; float cNode, cTri; int mergeMax;
    std::vector<float> A; std::vector<uint32_t> np, first;
    std::vector<float> F;          // F[n*W + k-1]: least cost of the subtree of n under at most k slots of its parent's node (k = 1..W)
    std::vector<uint8_t> how, howR; // how[n*W + k-1]: 0 = one slot (leaf or own wide node, see asLeaf), s>0 = split: left gets s slots, right howR
    std::vector<uint8_t> asLeaf;   // the one-slot form of n is a leaf
    void run(const mi_bvh2_node *nodes, size_t N, int width, float cn, float ct, int mm) {
        n2 = nodes; W = width; cNode = cn; cTri = ct; mergeMax = mm;
        A.resize(N); np.resize(N); first.resize(N); F.assign(N * W, 0); how.assign(N * W, 0); howR.assign(N * W, 0); asLeaf.assign(N, 0);
        for (size_t i = N; i-- > 0;) {   // children have larger indices than their parent (depth-first layout)
            const mi_bvh2_node &b = n2[i];
            A[i] = area(b);
            if (b.n_prims > 0) {
                np[i] = b.n_prims; first[i] = (uint32_t)b.offset; asLeaf[i] = 1;
                for (int k = 1; k <= W; ++k) F[i * W + k - 1] = A[i] * b.n_prims * cTri;
                continue;
            }
            size_t l = i + 1, r = (size_t)b.offset;
            np[i] = np[l] + np[r]; first[i] = std::min(first[l], first[r]);
            bool contiguous = first[l] + np[l] == first[r] || first[r] + np[r] == first[l];
            // D(i, k): the subtree as a forest of <= k slots = left under s slots + right under k - s
            float D[9]; uint8_t Ds[9];
            for (int k = 2; k <= W; ++k) {
                D[k] = 1e38f; Ds[k] = 1;
                for (int s = 1; s < k; ++s) { float c = F[l * W + s - 1] + F[r * W + k - s - 1]; if (c < D[k]) { D[k] = c; Ds[k] = (uint8_t)s; } }
            }
            float cInt = A[i] * cNode + D[W];
            float cLeaf = (mergeMax > 0 && (int)np[i] <= mergeMax && contiguous) ? A[i] * np[i] * cTri : 1e38f;
            asLeaf[i] = cLeaf < cInt;
            F[i * W] = std::min(cLeaf, cInt); how[i * W] = 0;
            for (int k = 2; k <= W; ++k) {
                if (D[k] < F[i * W + k - 2]) { F[i * W + k - 1] = D[k]; how[i * W + k - 1] = Ds[k]; howR[i * W + k - 1] = (uint8_t)(k - Ds[k]); }
                else { F[i * W + k - 1] = F[i * W + k - 2]; how[i * W + k - 1] = how[i * W + k - 2]; howR[i * W + k - 1] = howR[i * W + k - 2]; }
            }
        }
    }
    // the slots node n occupies when given at most k of them, in the reference's left-to-right order
    void slots(uint32_t n, int k, std::vector<uint32_t> &outv) const {
        uint8_t s = how[(size_t)n * W + k - 1];
        if (s == 0) { outv.push_back(n); return; }
        slots(n + 1, s, outv);
        slots((uint32_t)n2[n].offset, howR[(size_t)n * W + k - 1], outv);
    }
};
uint32_t buildWideDP(const Collapse &c, uint32_t i2, std::vector<WNode> &out, std::vector<std::vector<uint8_t>> &cnt) {
    uint32_t idx = (uint32_t)out.size();
    out.emplace_back();
    cnt.emplace_back(8, 0);
    std::vector<uint32_t> kids;
    // the node's own children: the W-slot forest of its two reference children
    {
        uint8_t s = 0; float best = 1e38f;
        size_t l = i2 + 1, r = (size_t)c.n2[i2].offset;
        for (int t = 1; t < c.W; ++t) { float v = c.F[l * c.W + t - 1] + c.F[r * c.W + c.W - t - 1]; if (v < best) { best = v; s = (uint8_t)t; } }
        c.slots((uint32_t)l, s, kids);
        c.slots((uint32_t)r, c.W - s, kids);
    }
    out[idx].n = (int)kids.size();
    for (int k = 0; k < (int)kids.size(); ++k) {
        const mi_bvh2_node &b = c.n2[kids[k]];
        for (int a = 0; a < 3; ++a) { out[idx].lo[k][a] = b.bmin[a]; out[idx].hi[k][a] = b.bmax[a]; }
        if (c.asLeaf[kids[k]]) { out[idx].child[k] = LEAF | c.first[kids[k]]; cnt[idx][k] = (uint8_t)std::min<uint32_t>(255, c.np[kids[k]]); }
        else { uint32_t ch = buildWideDP(c, kids[k], out, cnt); out[idx].child[k] = ch; }
    }
    return idx;
}
// Moller-Trumbore in double: the study only needs hit distances, not the watertight test
bool triHit(const mi_scene_desc *d, uint32_t prim, const double o[3], const double dir[3], double tMax, double *t) {
    const uint32_t *v = d->tri_indices + 3 * (size_t)prim;
    if (v[0] == MI_PRIM_SPHERE) return false;
    const float *p0 = d->P + 3 * (size_t)v[0], *p1 = d->P + 3 * (size_t)v[1], *p2 = d->P + 3 * (size_t)v[2];
    double e1[3], e2[3], pv[3], tv[3], qv[3];
    for (int a = 0; a < 3; ++a) { e1[a] = p1[a] - p0[a]; e2[a] = p2[a] - p0[a]; }
    pv[0] = dir[1] * e2[2] - dir[2] * e2[1]; pv[1] = dir[2] * e2[0] - dir[0] * e2[2]; pv[2] = dir[0] * e2[1] - dir[1] * e2[0];
    double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (det == 0) return false;
    double inv = 1 / det;
    for (int a = 0; a < 3; ++a) tv[a] = o[a] - p0[a];
    double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0 || u > 1) return false;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    double vv = (dir[0] * qv[0] + dir[1] * qv[1] + dir[2] * qv[2]) * inv;
    if (vv < 0 || u + vv > 1) return false;
    double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    if (tt <= 1e-9 || tt > tMax) return false;
    *t = tt;
    return true;
}
}  // namespace

int g_dp = 0, g_merge = 0; float g_cNode = 1, g_cTri = 1;
extern "C" {
void bvh_study_collapse(int dp, float cNode, float cTri, int mergeMax) { g_dp = dp; g_cNode = cNode; g_cTri = cTri; g_merge = mergeMax; }
// out[0] = nodes visited, out[1] = triangles tested, out[2] = hits, out[3] = number of wide nodes; width 2 = the reference's BVH2 traversal
static int g_anyOrder = 0;   // any-hit rays: 0 near-to-far (as shipped), 1 longest overlap first, 2 largest box first, 3 long-and-near first (study only)
void bvh_study_any_order(int o) { g_anyOrder = o; }
void bvh_study(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int width, int cull_on_pop, int any_hit, double *out) {
    double nodes = 0, tris = 0, hits = 0;
    if (width == 2) {
        const mi_bvh2_node *nd = d->bvh_nodes;
        for (int64_t r = 0; r < n; ++r) {
            double o[3] = {rays[r].o[0], rays[r].o[1], rays[r].o[2]}, dir[3] = {rays[r].d[0], rays[r].d[1], rays[r].d[2]}, tMax = rays[r].tmax;
            double inv[3] = {1 / dir[0], 1 / dir[1], 1 / dir[2]};
            int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
            int stack[64], sp = 0, cur = 0;
            bool hit = false;
            while (true) {
                const mi_bvh2_node &b = nd[cur];
                ++nodes;
                double t0 = 0, t1 = tMax;
                bool ok = true;
                for (int a = 0; a < 3 && ok; ++a) {
                    double tn = ((neg[a] ? b.bmax[a] : b.bmin[a]) - o[a]) * inv[a], tf = ((neg[a] ? b.bmin[a] : b.bmax[a]) - o[a]) * inv[a];
                    if (tn > t0) t0 = tn;
                    if (tf < t1) t1 = tf;
                    if (t0 > t1) ok = false;
                }
                if (ok) {
                    if (b.n_prims > 0) {
                        for (int i = 0; i < b.n_prims; ++i) { ++tris; double t; if (triHit(d, b.offset + i, o, dir, tMax, &t)) { tMax = t; hit = true; if (any_hit) goto done2; } }
                        if (!sp) break;
                        cur = stack[--sp];
                    } else {
                        if (neg[b.axis]) { stack[sp++] = cur + 1; cur = b.offset; } else { stack[sp++] = b.offset; cur = cur + 1; }
                    }
                } else { if (!sp) break; cur = stack[--sp]; }
            }
        done2:
            hits += hit;
        }
        out[0] = nodes; out[1] = tris; out[2] = hits; out[3] = d->n_bvh_nodes;
        return;
    }
    std::vector<WNode> wn;
    std::vector<std::vector<uint8_t>> cnt;
    if (d->n_bvh_nodes && d->bvh_nodes[0].n_prims == 0) {
        if (g_dp) { Collapse c; c.run(d->bvh_nodes, d->n_bvh_nodes, width, g_cNode, g_cTri, g_merge); buildWideDP(c, 0, wn, cnt); }
        else buildWide(d->bvh_nodes, 0, width, wn, cnt);
    }
    struct Ent { uint32_t ref; uint8_t count; double t; };
    for (int64_t r = 0; r < n && !wn.empty(); ++r) {
        double o[3] = {rays[r].o[0], rays[r].o[1], rays[r].o[2]}, dir[3] = {rays[r].d[0], rays[r].d[1], rays[r].d[2]}, tMax = rays[r].tmax;
        double inv[3] = {1 / dir[0], 1 / dir[1], 1 / dir[2]};
        std::vector<Ent> st;
        Ent cur{0, 0, 0};
        bool hit = false;
        while (true) {
            if (cur.ref & LEAF) {
                uint32_t first = cur.ref & ~LEAF;
                for (int i = 0; i < cur.count; ++i) { ++tris; double t; if (triHit(d, first + i, o, dir, tMax, &t)) { tMax = t; hit = true; if (any_hit) goto done; } }
            } else {
                const WNode &w = wn[cur.ref];
                ++nodes;
                Ent h[8];
                int nh = 0;
                for (int k = 0; k < w.n; ++k) {
                    double t0 = 0, t1 = tMax;
                    bool ok = true;
                    for (int a = 0; a < 3 && ok; ++a) {
                        double tn = ((inv[a] < 0 ? w.hi[k][a] : w.lo[k][a]) - o[a]) * inv[a], tf = ((inv[a] < 0 ? w.lo[k][a] : w.hi[k][a]) - o[a]) * inv[a];
                        if (tn > t0) t0 = tn;
                        if (tf < t1) t1 = tf;
                        if (t0 > t1) ok = false;
                    }
                    if (ok) {
                        double key = t0;
                        if (any_hit && g_anyOrder == 1) key = -(t1 - t0);                       // longest overlap with the box first
                        else if (any_hit && g_anyOrder == 2) { double a = 0; for (int q = 0; q < 3; ++q) { int r2 = (q + 1) % 3; a += (w.hi[k][q] - w.lo[k][q]) * (double)(w.hi[k][r2] - w.lo[k][r2]); } key = -a; }   // largest box first (SATO)
                        else if (any_hit && g_anyOrder == 3) key = -(t1 - t0) / (1e-9 + t0 + 0.5 * (t1 - t0));   // long and near first
                        h[nh++] = Ent{w.child[k], cnt[cur.ref][k], key};
                    }
                }
                if (!(any_hit && std::getenv("BVH_STUDY_ANY_NOSORT"))) std::sort(h, h + nh, [](const Ent &a, const Ent &b) { return a.t > b.t; });   // far first: nearest ends on top (BVH_STUDY_ANY_NOSORT: any-hit rays take the hit children in slot order)
                for (int k = 0; k < nh; ++k) st.push_back(h[k]);
            }
            bool got = false;
            while (!st.empty()) {
                cur = st.back(); st.pop_back();
                if (!cull_on_pop || any_hit || cur.t < tMax) { got = true; break; }
            }
            if (!got) break;
        }
    done:
        hits += hit;
    }
    out[0] = nodes; out[1] = tris; out[2] = hits; out[3] = (double)wn.size();
}
// ---- hot-node study (round 3): which share of the interior-node visits of the device's BVH4 lands on the K nodes a block could keep in LDS?
// Two choices of the K nodes: the K most visited ones (an oracle: needs the rays) and the first K nodes of a largest-surface-area-first expansion from the
// root (static: what mi_scene_upload can compute).  out[2 * i] / out[2 * i + 1] = share of the visits under choice one / two for K = Ks[i]; out[2 nK] = visits per ray.
void bvh_study_hot(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int any_hit, const int *Ks, int nK, double *out) {
    std::vector<WNode> wn;
    std::vector<std::vector<uint8_t>> cnt;
    if (!(d->n_bvh_nodes && d->bvh_nodes[0].n_prims == 0)) return;
    buildWide(d->bvh_nodes, 0, 4, wn, cnt);
    std::vector<uint64_t> visits(wn.size(), 0);
    struct Ent { uint32_t ref; uint8_t count; double t; };
    double total = 0;
    for (int64_t r = 0; r < n; ++r) {
        double o[3] = {rays[r].o[0], rays[r].o[1], rays[r].o[2]}, dir[3] = {rays[r].d[0], rays[r].d[1], rays[r].d[2]}, tMax = rays[r].tmax;
        double inv[3] = {1 / dir[0], 1 / dir[1], 1 / dir[2]};
        std::vector<Ent> st;
        Ent cur{0, 0, 0};
        while (true) {
            if (cur.ref & LEAF) {
                uint32_t first = cur.ref & ~LEAF;
                bool stop = false;
                for (int i = 0; i < cur.count; ++i) { double t; if (triHit(d, first + i, o, dir, tMax, &t)) { tMax = t; if (any_hit) { stop = true; break; } } }
                if (stop) break;
            } else {
                const WNode &w = wn[cur.ref];
                ++visits[cur.ref]; ++total;
                Ent h[8];
                int nh = 0;
                for (int k = 0; k < w.n; ++k) {
                    double t0 = 0, t1 = tMax;
                    bool ok = true;
                    for (int a = 0; a < 3 && ok; ++a) {
                        double tn = ((inv[a] < 0 ? w.hi[k][a] : w.lo[k][a]) - o[a]) * inv[a], tf = ((inv[a] < 0 ? w.lo[k][a] : w.hi[k][a]) - o[a]) * inv[a];
                        if (tn > t0) t0 = tn;
                        if (tf < t1) t1 = tf;
                        if (t0 > t1) ok = false;
                    }
                    if (ok) {
                        double key = t0;
                        if (any_hit && g_anyOrder == 1) key = -(t1 - t0);                       // longest overlap with the box first
                        else if (any_hit && g_anyOrder == 2) { double a = 0; for (int q = 0; q < 3; ++q) { int r2 = (q + 1) % 3; a += (w.hi[k][q] - w.lo[k][q]) * (double)(w.hi[k][r2] - w.lo[k][r2]); } key = -a; }   // largest box first (SATO)
                        else if (any_hit && g_anyOrder == 3) key = -(t1 - t0) / (1e-9 + t0 + 0.5 * (t1 - t0));   // long and near first
                        h[nh++] = Ent{w.child[k], cnt[cur.ref][k], key};
                    }
                }
                std::sort(h, h + nh, [](const Ent &a, const Ent &b) { return a.t > b.t; });
                for (int k = 0; k < nh; ++k) st.push_back(h[k]);
            }
            bool got = false;
            while (!st.empty()) { cur = st.back(); st.pop_back(); if (cur.t < tMax) { got = true; break; } }
            if (!got) break;
        }
    }
    std::vector<uint64_t> sorted(visits);
    std::sort(sorted.begin(), sorted.end(), [](uint64_t a, uint64_t b) { return a > b; });
    // static choice: expand from the root, largest child box first
    int Kmax = 0;
    for (int i = 0; i < nK; ++i) Kmax = std::max(Kmax, Ks[i]);
    std::vector<uint32_t> order;
    {
        struct Q { float a; uint32_t node; bool operator<(const Q &o) const { return a < o.a; } };
        std::vector<Q> heap;
        heap.push_back(Q{1e38f, 0});
        while (!heap.empty() && (int)order.size() < Kmax) {
            std::pop_heap(heap.begin(), heap.end());
            Q q = heap.back(); heap.pop_back();
            order.push_back(q.node);
            const WNode &w = wn[q.node];
            for (int k = 0; k < w.n; ++k) if (!(w.child[k] & LEAF)) {
                float dx = w.hi[k][0] - w.lo[k][0], dy = w.hi[k][1] - w.lo[k][1], dz = w.hi[k][2] - w.lo[k][2];
                heap.push_back(Q{2 * (dx * dy + dx * dz + dy * dz), w.child[k]});
                std::push_heap(heap.begin(), heap.end());
            }
        }
    }
    for (int i = 0; i < nK; ++i) {
        double a = 0, b = 0;
        for (int k = 0; k < Ks[i] && k < (int)sorted.size(); ++k) a += (double)sorted[k];
        for (int k = 0; k < Ks[i] && k < (int)order.size(); ++k) b += (double)visits[order[k]];
        out[2 * i] = total > 0 ? a / total : 0; out[2 * i + 1] = total > 0 ? b / total : 0;
    }
    out[2 * nK] = n > 0 ? total / (double)n : 0;
}
}

// ---- tree study (round 3): the SAME triangles under a tree of our own instead of the reference's (12 buckets on the axis of the
// largest centroid extent, bvh.cpp:236-420).  Builder: top-down SAH over all three axes -- `bins` buckets per axis for nodes above `sweepBelow`
// primitives, the exact sweep (every split position of the centroid order) below --, leaves of at most `leafMax` primitives where the SAH says a
// leaf is cheaper (cost model: cNode per interior visit, cTri per triangle).  The result replaces bvh_nodes / the triangle order of a COPY of the
// scene description, so every counter of this file runs on it unchanged.  Closest hits do not depend on the tree (ties at equal t excepted).
namespace {
struct RB {
    const mi_scene_desc *d;
    std::vector<float> lo, hi, ce;       // per triangle: box, centroid (3 floats each)
    std::vector<uint32_t> idx;           // permutation being sorted in place
    std::vector<mi_bvh2_node> nodes;
    int bins, sweepBelow, leafMax; float cNode, cTri;
    static float areaOf(const float *a, const float *b) { float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2]; return dx < 0 ? 0.f : 2 * (dx * dy + dx * dz + dy * dz); }
    uint32_t build(uint32_t begin, uint32_t end) {
        uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f}, cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t i = begin; i < end; ++i) {
            uint32_t t = idx[i];
            for (int a = 0; a < 3; ++a) {
                bl[a] = std::min(bl[a], lo[3 * t + a]); bh[a] = std::max(bh[a], hi[3 * t + a]);
                cl[a] = std::min(cl[a], ce[3 * t + a]); ch[a] = std::max(ch[a], ce[3 * t + a]);
            }
        }
        for (int a = 0; a < 3; ++a) { nodes[me].bmin[a] = bl[a]; nodes[me].bmax[a] = bh[a]; }
        const uint32_t n = end - begin;
        const float A = areaOf(bl, bh);
        auto makeLeaf = [&]() { nodes[me].offset = (int32_t)begin; nodes[me].n_prims = (uint16_t)n; nodes[me].axis = 0; return me; };
        if (n == 1) return makeLeaf();
        float bestCost = 1e38f; int bestAxis = -1; uint32_t bestMid = 0; float bestPos = 0; bool bestBinned = false;
        if ((int)n > sweepBelow) {
            for (int a = 0; a < 3; ++a) {
                if (!(ch[a] > cl[a])) continue;
                std::vector<uint32_t> cnt(bins, 0);
                std::vector<float> bbl(3 * bins, 1e30f), bbh(3 * bins, -1e30f);
                const float sc = bins / (ch[a] - cl[a]);
                for (uint32_t i = begin; i < end; ++i) {
                    uint32_t t = idx[i];
                    int b = std::min(bins - 1, (int)((ce[3 * t + a] - cl[a]) * sc));
                    ++cnt[b];
                    for (int k = 0; k < 3; ++k) { bbl[3 * b + k] = std::min(bbl[3 * b + k], lo[3 * t + k]); bbh[3 * b + k] = std::max(bbh[3 * b + k], hi[3 * t + k]); }
                }
                std::vector<float> rA(bins, 0.f); std::vector<uint32_t> rN(bins, 0);
                float rl[3] = {1e30f, 1e30f, 1e30f}, rh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t rn = 0;
                for (int b = bins - 1; b > 0; --b) {
                    for (int k = 0; k < 3; ++k) { rl[k] = std::min(rl[k], bbl[3 * b + k]); rh[k] = std::max(rh[k], bbh[3 * b + k]); }
                    rn += cnt[b]; rA[b] = areaOf(rl, rh); rN[b] = rn;
                }
                float ll[3] = {1e30f, 1e30f, 1e30f}, lh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t ln = 0;
                for (int b = 0; b < bins - 1; ++b) {
                    for (int k = 0; k < 3; ++k) { ll[k] = std::min(ll[k], bbl[3 * b + k]); lh[k] = std::max(lh[k], bbh[3 * b + k]); }
                    ln += cnt[b];
                    if (ln == 0 || rN[b + 1] == 0) continue;
                    float c = cNode + cTri * (areaOf(ll, lh) * ln + rA[b + 1] * rN[b + 1]) / A;
                    if (c < bestCost) { bestCost = c; bestAxis = a; bestPos = cl[a] + (b + 1) / sc; bestBinned = true; }
                }
            }
        } else {
            std::vector<float> rA(n);
            for (int a = 0; a < 3; ++a) {
                std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t x, uint32_t y) { return ce[3 * x + a] < ce[3 * y + a] || (ce[3 * x + a] == ce[3 * y + a] && x < y); });
                float rl[3] = {1e30f, 1e30f, 1e30f}, rh[3] = {-1e30f, -1e30f, -1e30f};
                for (uint32_t i = n - 1; i > 0; --i) {
                    uint32_t t = idx[begin + i];
                    for (int k = 0; k < 3; ++k) { rl[k] = std::min(rl[k], lo[3 * t + k]); rh[k] = std::max(rh[k], hi[3 * t + k]); }
                    rA[i] = areaOf(rl, rh);
                }
                float ll[3] = {1e30f, 1e30f, 1e30f}, lh[3] = {-1e30f, -1e30f, -1e30f};
                for (uint32_t i = 0; i + 1 < n; ++i) {
                    uint32_t t = idx[begin + i];
                    for (int k = 0; k < 3; ++k) { ll[k] = std::min(ll[k], lo[3 * t + k]); lh[k] = std::max(lh[k], hi[3 * t + k]); }
                    float c = cNode + cTri * (areaOf(ll, lh) * (i + 1) + rA[i + 1] * (n - i - 1)) / A;
                    if (c < bestCost) { bestCost = c; bestAxis = a; bestMid = begin + i + 1; bestBinned = false; }
                }
            }
        }
        const float leafCost = cTri * n;
        if ((int)n <= leafMax && (bestAxis < 0 || leafCost <= bestCost)) return makeLeaf();
        uint32_t mid;
        if (bestAxis < 0) { mid = begin + n / 2; }   // all centroids equal
        else if (bestBinned) {
            const int a = bestAxis;
            mid = (uint32_t)(std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t x) { return ce[3 * x + a] < bestPos; }) - idx.begin());
            if (mid == begin || mid == end) { mid = begin + n / 2; std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end, [&](uint32_t x, uint32_t y) { return ce[3 * x + a] < ce[3 * y + a]; }); }
        } else {
            const int a = bestAxis;
            if (a != 2) std::sort(idx.begin() + begin, idx.begin() + end, [&](uint32_t x, uint32_t y) { return ce[3 * x + a] < ce[3 * y + a] || (ce[3 * x + a] == ce[3 * y + a] && x < y); });
            mid = bestMid;
        }
        nodes[me].n_prims = 0; nodes[me].axis = (uint8_t)std::max(0, bestAxis);
        build(begin, mid);
        uint32_t second = build(mid, end);
        nodes[me].offset = (int32_t)second;
        return me;
    }
};
}  // namespace

extern "C" {
// counters of bvh_study on a rebuilt tree; out[4] = SAH cost of the new BVH2 relative to the reference's (same cost model), out[5] = leaves, out[6] = mean leaf size
void bvh_study_rebuilt(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int width, int cull_on_pop, int any_hit, int bins, int sweepBelow, int leafMax, float cNode, float cTri, double *out) {
    RB rb; rb.d = d; rb.bins = bins; rb.sweepBelow = sweepBelow; rb.leafMax = std::min(leafMax, 255); rb.cNode = cNode; rb.cTri = cTri;
    const uint32_t nt = d->n_tris;
    rb.lo.resize(3 * (size_t)nt); rb.hi.resize(3 * (size_t)nt); rb.ce.resize(3 * (size_t)nt); rb.idx.resize(nt);
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t *v = d->tri_indices + 3 * (size_t)t;
        rb.idx[t] = t;
        for (int a = 0; a < 3; ++a) {
            float x0 = d->P[3 * (size_t)v[0] + a], x1 = d->P[3 * (size_t)v[1] + a], x2 = d->P[3 * (size_t)v[2] + a];
            rb.lo[3 * (size_t)t + a] = std::min(x0, std::min(x1, x2)); rb.hi[3 * (size_t)t + a] = std::max(x0, std::max(x1, x2));
            rb.ce[3 * (size_t)t + a] = 0.5f * (rb.lo[3 * (size_t)t + a] + rb.hi[3 * (size_t)t + a]);   // the reference's centroid: of the box (bvh.cpp:59-62)
        }
    }
    rb.nodes.reserve(2 * (size_t)nt);
    rb.build(0, nt);
    std::vector<uint32_t> tri(3 * (size_t)nt);
    for (uint32_t i = 0; i < nt; ++i) for (int k = 0; k < 3; ++k) tri[3 * (size_t)i + k] = d->tri_indices[3 * (size_t)rb.idx[i] + k];
    mi_scene_desc d2 = *d;
    d2.bvh_nodes = rb.nodes.data(); d2.n_bvh_nodes = (uint32_t)rb.nodes.size(); d2.tri_indices = tri.data();
    bvh_study(&d2, rays, n, width, cull_on_pop, any_hit, out);
    auto sah = [&](const mi_bvh2_node *nd, size_t N) { double c = 0; const double A0 = area(nd[0]); for (size_t i = 0; i < N; ++i) c += area(nd[i]) / A0 * (nd[i].n_prims ? cTri * nd[i].n_prims : cNode); return c; };
    out[4] = sah(rb.nodes.data(), rb.nodes.size()) / sah(d->bvh_nodes, d->n_bvh_nodes);
    double leaves = 0; for (auto &x : rb.nodes) leaves += x.n_prims > 0;
    out[5] = leaves; out[6] = nt / std::max(1.0, leaves);
}
}

// ---- spatial-split study (round 6): SBVH (Stich, Friedrich, Dietrich 2009) over the same triangles.  At every node the best OBJECT split
// (binned SAH on the reference centroids, three axes) competes with the best SPATIAL split (chopped binning: every reference is clipped, as the triangle it is, to the
// bins it straddles); a spatial split duplicates the references that cross its plane.  Spatial splits are only tried where the children of the object split overlap
// by more than `alpha` of the root's area.  The tree replaces bvh_nodes / the primitive list of a COPY of the description (a triangle may appear in several leaves),
// so the counters of this file run on it unchanged.  out[4] = references / triangles, out[5] = leaves, out[6] = mean leaf size.
namespace {
struct SRef { float lo[3], hi[3]; uint32_t tri; };
struct SB {
    const mi_scene_desc *d;
    std::vector<mi_bvh2_node> nodes;
    std::vector<uint32_t> order;   // triangle of every leaf reference, in leaf order
    int bins, leafMax; float cNode, cTri, alpha; double rootArea = 0;
    static float areaOf(const float *a, const float *b) { float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2]; return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2 * (dx * dy + dx * dz + dy * dz); }
    // box of the part of triangle t between the planes x_a = p0 and x_a = p1, intersected with the reference's own box
    bool clipBox(const SRef &r, int a, float p0, float p1, float *lo, float *hi) const {
        const uint32_t *v = d->tri_indices + 3 * (size_t)r.tri;
        double poly[2][10][3]; int n = 3, cur = 0;
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) poly[0][k][c] = d->P[3 * (size_t)v[k] + c];
        for (int side = 0; side < 2; ++side) {   // keep x_a >= p0, then x_a <= p1
            const double pl = side ? p1 : p0, sg = side ? -1 : 1;
            int m = 0;
            for (int k = 0; k < n; ++k) {
                const double *A = poly[cur][k], *B = poly[cur][(k + 1) % n];
                const double da = sg * (A[a] - pl), db = sg * (B[a] - pl);
                if (da >= 0) { for (int c = 0; c < 3; ++c) poly[cur ^ 1][m][c] = A[c]; ++m; }
                if ((da > 0 && db < 0) || (da < 0 && db > 0)) { const double t = da / (da - db); for (int c = 0; c < 3; ++c) poly[cur ^ 1][m][c] = A[c] + t * (B[c] - A[c]); poly[cur ^ 1][m][a] = pl; ++m; }
            }
            n = m; cur ^= 1;
            if (n == 0) return false;
        }
        for (int c = 0; c < 3; ++c) { lo[c] = 1e30f; hi[c] = -1e30f; }
        for (int k = 0; k < n; ++k) for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], (float)poly[cur][k][c]); hi[c] = std::max(hi[c], (float)poly[cur][k][c]); }
        for (int c = 0; c < 3; ++c) { lo[c] = std::max(lo[c], r.lo[c]); hi[c] = std::min(hi[c], r.hi[c]); if (lo[c] > hi[c]) { if (lo[c] - hi[c] < 1e-4f * (1 + std::fabs(lo[c]))) hi[c] = lo[c]; else return false; } }
        return true;
    }
    uint32_t build(std::vector<SRef> &refs, int depth) {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f}, cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
        for (const SRef &r : refs) for (int a = 0; a < 3; ++a) {
            bl[a] = std::min(bl[a], r.lo[a]); bh[a] = std::max(bh[a], r.hi[a]);
            const float c = 0.5f * (r.lo[a] + r.hi[a]); cl[a] = std::min(cl[a], c); ch[a] = std::max(ch[a], c);
        }
        for (int a = 0; a < 3; ++a) { nodes[me].bmin[a] = bl[a]; nodes[me].bmax[a] = bh[a]; }
        const uint32_t n = (uint32_t)refs.size();
        const float A = areaOf(bl, bh);
        if (me == 0) rootArea = A;
        auto makeLeaf = [&]() { nodes[me].offset = (int32_t)order.size(); nodes[me].n_prims = (uint16_t)n; nodes[me].axis = 0; for (const SRef &r : refs) order.push_back(r.tri); return me; };
        if (n == 1 || depth > 60) return makeLeaf();
        // object split
        float objCost = 1e38f; int objAxis = -1; float objPos = 0; float ovl = 0;
        for (int a = 0; a < 3; ++a) {
            if (!(ch[a] > cl[a])) continue;
            std::vector<uint32_t> cnt(bins, 0);
            std::vector<float> bbl(3 * bins, 1e30f), bbh(3 * bins, -1e30f);
            const float sc = bins / (ch[a] - cl[a]);
            for (const SRef &r : refs) {
                int b = std::min(bins - 1, (int)((0.5f * (r.lo[a] + r.hi[a]) - cl[a]) * sc));
                ++cnt[b];
                for (int k = 0; k < 3; ++k) { bbl[3 * b + k] = std::min(bbl[3 * b + k], r.lo[k]); bbh[3 * b + k] = std::max(bbh[3 * b + k], r.hi[k]); }
            }
            std::vector<float> rA(bins, 0.f), rL(3 * bins), rH(3 * bins); std::vector<uint32_t> rN(bins, 0);
            float rl[3] = {1e30f, 1e30f, 1e30f}, rh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t rn = 0;
            for (int b = bins - 1; b > 0; --b) {
                for (int k = 0; k < 3; ++k) { rl[k] = std::min(rl[k], bbl[3 * b + k]); rh[k] = std::max(rh[k], bbh[3 * b + k]); rL[3 * b + k] = rl[k]; rH[3 * b + k] = rh[k]; }
                rn += cnt[b]; rA[b] = areaOf(rl, rh); rN[b] = rn;
            }
            float ll[3] = {1e30f, 1e30f, 1e30f}, lh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t ln = 0;
            for (int b = 0; b < bins - 1; ++b) {
                for (int k = 0; k < 3; ++k) { ll[k] = std::min(ll[k], bbl[3 * b + k]); lh[k] = std::max(lh[k], bbh[3 * b + k]); }
                ln += cnt[b];
                if (ln == 0 || rN[b + 1] == 0) continue;
                float c = cNode + cTri * (areaOf(ll, lh) * ln + rA[b + 1] * rN[b + 1]) / A;
                if (c < objCost) {
                    objCost = c; objAxis = a; objPos = cl[a] + (b + 1) / sc;
                    float ol[3], oh[3]; for (int k = 0; k < 3; ++k) { ol[k] = std::max(ll[k], rL[3 * (b + 1) + k]); oh[k] = std::min(lh[k], rH[3 * (b + 1) + k]); }
                    ovl = areaOf(ol, oh);
                }
            }
        }
        // spatial split (chopped binning), only where the object split's children overlap enough
        float spCost = 1e38f; int spAxis = -1; float spPos = 0;
        if (alpha >= 0 && objAxis >= 0 && ovl > alpha * rootArea) {
            for (int a = 0; a < 3; ++a) {
                if (!(bh[a] > bl[a])) continue;
                const float sc = bins / (bh[a] - bl[a]);
                std::vector<uint32_t> ent(bins, 0), ext(bins, 0);
                std::vector<float> bbl(3 * bins, 1e30f), bbh(3 * bins, -1e30f);
                for (const SRef &r : refs) {
                    int b0 = std::max(0, std::min(bins - 1, (int)((r.lo[a] - bl[a]) * sc))), b1 = std::max(0, std::min(bins - 1, (int)((r.hi[a] - bl[a]) * sc)));
                    ++ent[b0]; ++ext[b1];
                    for (int b = b0; b <= b1; ++b) {
                        float lo[3], hi[3];
                        if (b0 == b1) { for (int k = 0; k < 3; ++k) { lo[k] = r.lo[k]; hi[k] = r.hi[k]; } }
                        else if (!clipBox(r, a, bl[a] + b / sc, bl[a] + (b + 1) / sc, lo, hi)) continue;
                        for (int k = 0; k < 3; ++k) { bbl[3 * b + k] = std::min(bbl[3 * b + k], lo[k]); bbh[3 * b + k] = std::max(bbh[3 * b + k], hi[k]); }
                    }
                }
                std::vector<float> rA(bins, 0.f); std::vector<uint32_t> rN(bins, 0);
                float rl[3] = {1e30f, 1e30f, 1e30f}, rh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t rn = 0;
                for (int b = bins - 1; b > 0; --b) {
                    for (int k = 0; k < 3; ++k) { rl[k] = std::min(rl[k], bbl[3 * b + k]); rh[k] = std::max(rh[k], bbh[3 * b + k]); }
                    rn += ext[b]; rA[b] = areaOf(rl, rh); rN[b] = rn;
                }
                float ll[3] = {1e30f, 1e30f, 1e30f}, lh[3] = {-1e30f, -1e30f, -1e30f}; uint32_t ln = 0;
                for (int b = 0; b < bins - 1; ++b) {
                    for (int k = 0; k < 3; ++k) { ll[k] = std::min(ll[k], bbl[3 * b + k]); lh[k] = std::max(lh[k], bbh[3 * b + k]); }
                    ln += ent[b];
                    if (ln == 0 || rN[b + 1] == 0 || ln == n || rN[b + 1] == n) continue;
                    float c = cNode + cTri * (areaOf(ll, lh) * ln + rA[b + 1] * rN[b + 1]) / A;
                    if (c < spCost) { spCost = c; spAxis = a; spPos = bl[a] + (b + 1) / sc; }
                }
            }
        }
        const float best = std::min(objCost, spCost), leafCost = cTri * n;
        if ((int)n <= leafMax && (best >= 1e38f || leafCost <= best)) return makeLeaf();
        std::vector<SRef> L, R;
        int axis = 0;
        if (spCost < objCost) {
            axis = spAxis;
            for (const SRef &r : refs) {
                if (r.hi[axis] <= spPos) L.push_back(r);
                else if (r.lo[axis] >= spPos) R.push_back(r);
                else {
                    SRef a = r, b = r;
                    float lo[3], hi[3];
                    bool okL = clipBox(r, axis, -1e30f, spPos, lo, hi);
                    if (okL) { for (int k = 0; k < 3; ++k) { a.lo[k] = lo[k]; a.hi[k] = hi[k]; } L.push_back(a); }
                    bool okR = clipBox(r, axis, spPos, 1e30f, lo, hi);
                    if (okR) { for (int k = 0; k < 3; ++k) { b.lo[k] = lo[k]; b.hi[k] = hi[k]; } R.push_back(b); }
                    if (!okL && !okR) L.push_back(r);
                }
            }
            if (L.empty() || R.empty() || L.size() == n && R.size() == n) { L.clear(); R.clear(); spCost = 1e38f; }
        }
        if (L.empty()) {
            if (objAxis >= 0) {
                axis = objAxis;
                for (const SRef &r : refs) (0.5f * (r.lo[axis] + r.hi[axis]) < objPos ? L : R).push_back(r);
            }
            if (L.empty() || R.empty()) { L.assign(refs.begin(), refs.begin() + n / 2); R.assign(refs.begin() + n / 2, refs.end()); }
        }
        std::vector<SRef>().swap(refs);
        nodes[me].n_prims = 0; nodes[me].axis = (uint8_t)axis;
        build(L, depth + 1);
        const uint32_t second = build(R, depth + 1);
        nodes[me].offset = (int32_t)second;
        return me;
    }
};
}  // namespace

extern "C" void bvh_study_sbvh(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int width, int cull_on_pop, int any_hit, int bins, int leafMax, float cNode, float cTri, float alpha, double *out) {
    SB sb; sb.d = d; sb.bins = bins; sb.leafMax = std::min(leafMax, 255); sb.cNode = cNode; sb.cTri = cTri; sb.alpha = alpha;
    const uint32_t nt = d->n_tris;
    std::vector<SRef> refs(nt);
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t *v = d->tri_indices + 3 * (size_t)t;
        refs[t].tri = t;
        for (int a = 0; a < 3; ++a) {
            float x0 = d->P[3 * (size_t)v[0] + a], x1 = d->P[3 * (size_t)v[1] + a], x2 = d->P[3 * (size_t)v[2] + a];
            refs[t].lo[a] = std::min(x0, std::min(x1, x2)); refs[t].hi[a] = std::max(x0, std::max(x1, x2));
        }
    }
    sb.nodes.reserve(3 * (size_t)nt);
    sb.build(refs, 0);
    std::vector<uint32_t> tri(3 * sb.order.size());
    for (size_t i = 0; i < sb.order.size(); ++i) for (int k = 0; k < 3; ++k) tri[3 * i + k] = d->tri_indices[3 * (size_t)sb.order[i] + k];
    mi_scene_desc d2 = *d;
    d2.bvh_nodes = sb.nodes.data(); d2.n_bvh_nodes = (uint32_t)sb.nodes.size(); d2.tri_indices = tri.data(); d2.n_tris = (uint32_t)sb.order.size();
    bvh_study(&d2, rays, n, width, cull_on_pop, any_hit, out);
    out[4] = (double)sb.order.size() / nt;
    double leaves = 0; for (auto &x : sb.nodes) leaves += x.n_prims > 0;
    out[5] = leaves; out[6] = sb.order.size() / std::max(1.0, leaves);
}

// ---- wave-scheduling simulator (round 3): the per-lane state machine of k_trace run for 64-lane waves on the host, to count how many wave-level
// node phases / leaf phases a policy needs and how many lanes take part in each (SIMT efficiency) -- the traversal kernels turned out to be bound by
// VALU issue as much as by memory (profiles/r03_c_*: 47 % of the lanes active per VALU instruction), and policies can be compared here without a GPU.
//   policy 0: k_trace today -- up to `nodeSteps` node phases while any lane wants one (stop early once `leafMin` lanes wait at a leaf), then ONE
//             triangle per lane at a leaf; refill when `refill` lanes are idle
//   policy 1: as 0, but a leaf phase walks ALL triangles of each lane's leaf (phases = the longest leaf among the lanes)
//   policy 2: postponed leaves -- a lane reaching a leaf parks it (one slot) and keeps traversing; parked leaves are tested when `leafMin` lanes have
//             one (or a lane needs its slot again / has nothing else to do).  Speculative: nodes are visited with the old tMax
// out: [0] rays, [1] node phases, [2] lane node steps, [3] leaf phases, [4] lane triangle tests, [5] outer iterations
extern "C" void bvh_study_wavesim(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int policy, int nodeSteps, int leafMin, int refill, double *out) {
    std::vector<WNode> wn;
    std::vector<std::vector<uint8_t>> cnt;
    if (!(d->n_bvh_nodes && d->bvh_nodes[0].n_prims == 0)) return;
    buildWide(d->bvh_nodes, 0, 4, wn, cnt);
    const uint32_t DONE = 0xFFFFFFFFu;
    struct Lane {
        bool active = false; double o[3], dir[3], inv[3], tMax; uint32_t cur = 0xFFFFFFFFu; int left = 0;   // cur: node index, LEAF | first prim (left = triangles after this one), DONE
        std::vector<std::pair<uint32_t, int>> st;   // (ref, count)
        std::vector<double> stT;                    // entry distance of each stack entry (cull-on-pop variants)
        uint32_t parked = 0xFFFFFFFFu; int parkedCount = 0;
    };
    double phN = 0, laN = 0, phL = 0, laL = 0, iters = 0, nrays = 0, deep13 = 0, deep21 = 0, deepLanes = 0;
    for (int64_t base = 0; base < n;) {   // one wave at a time over a contiguous slice of the rays (a wave's batches come from one queue segment)
        int64_t sliceEnd = std::min<int64_t>(n, base + 4096);
        int64_t next = base;
        const int S = policy == 6 ? 2 : 1;   // ray slots per lane
        std::vector<Lane> L(64 * S);
        int pref[64] = {0};
        const bool cullPop = nodeSteps >= 1000;   // nodeSteps + 1000: drop popped entries whose entry distance is not below the current tMax (PT_STACK_T)
        if (cullPop) nodeSteps -= 1000;
        auto pop = [&](Lane &l) {
            while (true) {
                if (l.st.empty()) { l.cur = DONE; return; }
                l.cur = l.st.back().first; l.left = l.st.back().second - 1;
                double t = l.stT.back();
                l.st.pop_back(); l.stT.pop_back();
                if (!cullPop || t < l.tMax) return;
            }
        };
        auto nodeStep = [&](Lane &l) {
            const WNode &w = wn[l.cur];
            struct H { uint32_t ref; int c; double t; } h[4]; int nh = 0;
            for (int k = 0; k < w.n; ++k) {
                double t0 = 0, t1 = l.tMax; bool ok = true;
                for (int a = 0; a < 3 && ok; ++a) {
                    double tn = ((l.inv[a] < 0 ? w.hi[k][a] : w.lo[k][a]) - l.o[a]) * l.inv[a], tf = ((l.inv[a] < 0 ? w.lo[k][a] : w.hi[k][a]) - l.o[a]) * l.inv[a];
                    if (tn > t0) t0 = tn; if (tf < t1) t1 = tf; if (t0 > t1) ok = false;
                }
                if (ok) h[nh++] = H{w.child[k], cnt[l.cur][k], t0};
            }
            std::sort(h, h + nh, [](const H &a, const H &b) { return a.t < b.t; });
            if (nh == 0) { pop(l); return; }
            for (int k = nh - 1; k >= 1; --k) { l.st.push_back({h[k].ref, h[k].c}); l.stT.push_back(h[k].t); }
            l.cur = h[0].ref; l.left = h[0].c - 1;
        };
        auto triStep = [&](Lane &l, uint32_t prim) { double t; if (triHit(d, prim, l.o, l.dir, l.tMax, &t)) l.tMax = t; };
        auto atNode = [&](const Lane &l) { return l.active && !(l.cur & LEAF); };
        auto atLeaf = [&](const Lane &l) { return l.active && l.cur != DONE && (l.cur & LEAF); };
        while (true) {
            int nIdle = 0; for (auto &l : L) nIdle += !l.active;
            if (nIdle >= refill * S && next < sliceEnd) {
                for (auto &l : L) if (!l.active && next < sliceEnd) {
                    const mi_ray &r = rays[next++]; ++nrays;
                    for (int a = 0; a < 3; ++a) { l.o[a] = r.o[a]; l.dir[a] = r.d[a]; l.inv[a] = 1.0 / r.d[a]; }
                    l.tMax = r.tmax; l.cur = 0; l.left = 0; l.st.clear(); l.stT.clear(); l.active = true; l.parked = DONE;
                }
            }
            int nAct = 0; for (auto &l : L) nAct += l.active;
            if (!nAct) break;
            const bool mayRefill = next < sliceEnd;
            while (true) {
                ++iters;
                if (policy == 2) {
                    // node phases; a lane that reaches a leaf parks it and pops on (if its slot is free)
                    for (int g = 0; g < nodeSteps; ++g) {
                        int nWant = 0; for (auto &l : L) nWant += atNode(l);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        for (auto &l : L) if (atNode(l)) {
                            nodeStep(l);
                            if (atLeaf(l) && l.parked == DONE) { l.parked = l.cur; l.parkedCount = l.left + 1; pop(l); if (l.cur == DONE && l.parked != DONE) l.cur = LEAF | 0x7ffffffe; }   // sentinel: only the parked leaf is left
                        }
                        int nPark = 0, nBlocked = 0; for (auto &l : L) { nPark += l.active && l.parked != DONE; nBlocked += l.active && l.parked != DONE && (atLeaf(l)); }
                        if (nPark >= leafMin || nBlocked >= leafMin / 2) break;
                    }
                    // leaf phases over the parked leaves (all triangles)
                    int maxC = 0, nP = 0; for (auto &l : L) if (l.active && l.parked != DONE) { maxC = std::max(maxC, l.parkedCount); ++nP; }
                    bool need = false; for (auto &l : L) if (l.active && l.parked != DONE && (atLeaf(l) || l.cur == DONE)) need = true;
                    int nNode = 0; for (auto &l : L) nNode += atNode(l);
                    if (nP && (nP >= leafMin || need || nNode == 0)) {
                        for (int k = 0; k < maxC; ++k) { ++phL; for (auto &l : L) if (l.active && l.parked != DONE && k < l.parkedCount) { ++laL; triStep(l, (l.parked & ~LEAF) + k); } }
                        for (auto &l : L) if (l.active && l.parked != DONE) {
                            l.parked = DONE;
                            if (l.cur == (LEAF | 0x7ffffffe)) l.cur = DONE;
                            else if (atLeaf(l)) { l.parked = l.cur; l.parkedCount = l.left + 1; pop(l); if (l.cur == DONE) l.cur = LEAF | 0x7ffffffe; }
                        }
                    }
                } else if (policy == 4 || policy == 5) {
                    // as 3 with a FIFO of 2 (policy 4) or 3 (policy 5) parked leaves per lane
                    const size_t slots = policy == 4 ? 2 : 3;
                    static thread_local std::vector<std::pair<uint32_t, int>> q[64];
                    auto parkQ = [&](Lane &l, int li) { q[li].push_back({l.cur & ~LEAF, l.left + 1}); pop(l); };
                    for (int g = 0; g < nodeSteps; ++g) {
                        int nWant = 0; for (auto &l : L) nWant += atNode(l);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        for (int li = 0; li < 64; ++li) { Lane &l = L[li]; if (atNode(l)) { nodeStep(l); while (atLeaf(l) && q[li].size() < slots) parkQ(l, li); } }
                        int nPend = 0; for (int li = 0; li < 64; ++li) nPend += L[li].active && !q[li].empty();
                        if (nPend >= leafMin) break;
                    }
                    int nPend = 0; for (int li = 0; li < 64; ++li) nPend += L[li].active && !q[li].empty();
                    if (nPend) {
                        ++phL; laL += nPend;
                        for (int li = 0; li < 64; ++li) { Lane &l = L[li]; if (l.active && !q[li].empty()) {
                            triStep(l, q[li].front().first);
                            if (--q[li].front().second > 0) ++q[li].front().first;
                            else { q[li].erase(q[li].begin()); while (atLeaf(l) && q[li].size() < slots) parkQ(l, li); }
                        } }
                    }
                    for (int li = 0; li < 64; ++li) { Lane &l = L[li]; l.parked = q[li].empty() ? DONE : 0u; }
                } else if (policy == 6) {
                    // TWO rays per lane (slots 2 li, 2 li + 1), each with policy 3's state machine (one parked leaf per ray): in a node phase a lane steps ONE of its
                    // rays that wants a node step, in a leaf phase it tests ONE parked triangle of one of its rays -- a lane idles only when NEITHER ray wants the
                    // phase.  What it would cost on the device: the second ray's ~40 registers (or its state in LDS) and the selects between the two states.
                    auto park = [&](Lane &l) { l.parked = l.cur & ~LEAF; l.parkedCount = l.left + 1; pop(l); };
                    for (int g = 0; g < nodeSteps; ++g) {
                        int nWant = 0; for (int li = 0; li < 64; ++li) nWant += atNode(L[2 * li]) || atNode(L[2 * li + 1]);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        for (int li = 0; li < 64; ++li) {
                            int s = 2 * li + pref[li];
                            if (!atNode(L[s])) s ^= 1;
                            if (!atNode(L[s])) continue;
                            nodeStep(L[s]); if (atLeaf(L[s]) && L[s].parked == DONE) park(L[s]);
                            pref[li] ^= 1;
                        }
                        int nPend = 0; for (int li = 0; li < 64; ++li) nPend += (L[2 * li].active && L[2 * li].parked != DONE) || (L[2 * li + 1].active && L[2 * li + 1].parked != DONE);
                        if (nPend >= leafMin) break;
                    }
                    int nPend = 0; for (int li = 0; li < 64; ++li) nPend += (L[2 * li].active && L[2 * li].parked != DONE) || (L[2 * li + 1].active && L[2 * li + 1].parked != DONE);
                    if (nPend) {
                        ++phL; laL += nPend;
                        for (int li = 0; li < 64; ++li) {
                            int s = 2 * li + pref[li];
                            if (!(L[s].active && L[s].parked != DONE)) s ^= 1;
                            Lane &l = L[s];
                            if (!(l.active && l.parked != DONE)) continue;
                            triStep(l, l.parked);
                            if (--l.parkedCount > 0) ++l.parked;
                            else { l.parked = DONE; if (atLeaf(l)) park(l); }
                        }
                    }
                } else if (policy == 7) {
                    // as policy 3, but a leaf phase spreads the parked leaves' triangles over ALL 64 lanes: first every parked leaf's next triangle, then the second ones,
                    // ... up to 4 per leaf and 64 tests per phase (idle lanes test triangles of other lanes' rays; what it would cost on the device: the pair
                    // assignment, the ray's constants and the result through LDS -- priced by the caller as a dearer triangle phase)
                    auto park = [&](Lane &l) { l.parked = l.cur & ~LEAF; l.parkedCount = l.left + 1; pop(l); };
                    for (int g = 0; g < nodeSteps; ++g) {
                        int nWant = 0; for (auto &l : L) nWant += atNode(l);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        for (auto &l : L) if (atNode(l)) { nodeStep(l); if (atLeaf(l) && l.parked == DONE) park(l); }
                        int nPend = 0; for (auto &l : L) nPend += l.active && l.parked != DONE;
                        if (nPend >= leafMin) break;
                    }
                    int nPend = 0; for (auto &l : L) nPend += l.active && l.parked != DONE;
                    if (nPend) {
                        int take[64] = {0}, total = 0;
                        for (int k = 0; k < 4 && total < 64; ++k)
                            for (int li = 0; li < 64 && total < 64; ++li) { Lane &l = L[li]; if (l.active && l.parked != DONE && l.parkedCount > k) { ++take[li]; ++total; } }
                        ++phL; laL += total;
                        for (int li = 0; li < 64; ++li) { Lane &l = L[li];
                            for (int k = 0; k < take[li]; ++k) { triStep(l, l.parked); ++l.parked; --l.parkedCount; }
                            if (take[li] && l.parkedCount == 0) { l.parked = DONE; if (atLeaf(l)) park(l); }
                        }
                    }
                } else if (policy == 3) {
                    // one pending leaf per lane, tested ONE triangle per leaf phase while the lane goes on with node steps (speculative: stale tMax)
                    auto park = [&](Lane &l) { l.parked = l.cur & ~LEAF; l.parkedCount = l.left + 1; pop(l); };
                    for (int g = 0; g < nodeSteps; ++g) {
                        int nWant = 0; for (auto &l : L) nWant += atNode(l);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        { size_t mx = 0; for (auto &l : L) if (atNode(l)) mx = std::max(mx, l.st.size()); deep13 += mx > 13; deep21 += mx > 21; deepLanes += 0; for (auto &l : L) if (atNode(l)) deepLanes += l.st.size() > 13; }
                        for (auto &l : L) if (atNode(l)) { nodeStep(l); if (atLeaf(l) && l.parked == DONE) park(l); }
                        int nPend = 0; for (auto &l : L) nPend += l.active && l.parked != DONE;
                        if (nPend >= leafMin) break;
                    }
                    int nPend = 0, nNode = 0; for (auto &l : L) { nPend += l.active && l.parked != DONE; nNode += atNode(l); }
                    if (nPend && (nPend >= leafMin || nNode == 0 || true)) {
                        ++phL; laL += nPend;
                        for (auto &l : L) if (l.active && l.parked != DONE) {
                            triStep(l, l.parked);
                            if (--l.parkedCount > 0) ++l.parked;
                            else { l.parked = DONE; if (atLeaf(l)) park(l); }
                        }
                    }
                } else {
                    int guard = 0;
                    while (true) {
                        int nWant = 0; for (auto &l : L) nWant += atNode(l);
                        if (!nWant) break;
                        ++phN; laN += nWant;
                        for (auto &l : L) if (atNode(l)) nodeStep(l);
                        int nLeaf = 0; for (auto &l : L) nLeaf += atLeaf(l);
                        if (nLeaf >= leafMin || ++guard >= nodeSteps) break;
                    }
                    int nLeaf = 0; for (auto &l : L) nLeaf += atLeaf(l);
                    if (nLeaf) {
                        if (policy == 0) {
                            ++phL; laL += nLeaf;
                            for (auto &l : L) if (atLeaf(l)) { uint32_t first = l.cur & ~LEAF; triStep(l, first); if (l.left) { l.cur = LEAF | (first + 1); --l.left; } else pop(l); }
                        } else {
                            int maxC = 0; for (auto &l : L) if (atLeaf(l)) maxC = std::max(maxC, l.left + 1);
                            for (int k = 0; k < maxC; ++k) { ++phL; for (auto &l : L) if (atLeaf(l) && k <= l.left) { ++laL; triStep(l, (l.cur & ~LEAF) + k); } }
                            for (auto &l : L) if (atLeaf(l)) pop(l);
                        }
                    }
                }
                for (auto &l : L) if (l.active && l.cur == DONE && l.parked == DONE) l.active = false;
                nAct = 0; for (auto &l : L) nAct += l.active;
                if (nAct == 0 || (mayRefill && nAct <= (64 - refill) * S)) break;
            }
        }
        base = sliceEnd;
    }
    out[0] = nrays; out[1] = phN; out[2] = laN; out[3] = phL; out[4] = laL; out[5] = iters;
    out[6] = deep13; out[7] = deep21;   // policy 3: node phases in which some stepping lane's stack holds more than 13 / 21 entries
    static bool said = false;
    if (policy == 3 && !said && (said = true)) std::fprintf(stderr, "[wavesim] policy 3: node phases with a lane deeper than 13 entries: %.1f %%, deeper than 21: %.1f %%; lane steps deeper than 13: %.2f %%\n", 100 * deep13 / std::max(1.0, phN), 100 * deep21 / std::max(1.0, phN), 100 * deepLanes / std::max(1.0, laN));
}


// ---- round 6, last session: the PRODUCT's topology over the reference's leaves (treebuild::RebuildOverLeaves) with and without the insertion-based
// optimisation pass (treebuild::OptimizeByReinsertion); out[0..3] as bvh_study, out[4] = SAH relative to the tree before the pass, out[5] = subtrees moved, out[6] = seconds
#include <chrono>
extern "C" void bvh_study_reinsert(const mi_scene_desc *d, const mi_ray *rays, int64_t n, int cull_on_pop, int any_hit, double frac, int passes, double *out) {
    std::vector<mi_bvh2_node> own;
    if (!treebuild::RebuildOverLeaves(d->bvh_nodes, d->n_bvh_nodes, &own)) { out[0] = -1; return; }
    uint64_t moved = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const double rel = treebuild::OptimizeByReinsertion(&own, frac, passes, &moved);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    mi_scene_desc d2 = *d;
    d2.bvh_nodes = own.data();
    d2.n_bvh_nodes = (uint32_t)own.size();
    bvh_study(&d2, rays, n, 4, cull_on_pop, any_hit, out);
    out[4] = rel; out[5] = (double)moved; out[6] = secs;
}
