// BVHAccel construction (host).  Same tree as the reference builds -- SAH with 12 buckets,
// identical bounds arithmetic, std::partition / std::nth_element with the same predicates
// (accelerators/bvh.cpp:236-402) and the same DFS flattening (:640-658) -- so node order,
// primitive order (tie-breaks at equal t) and LinearBVHNode contents are
// bit-identical.  Differences are structural only: triangles are (primitive, tri) index pairs
// instead of shared_ptr<Primitive>, and big sub-trees build on worker threads (the leaf
// range of a sub-tree is its [start,end) slice of the partitioned array, so the ordered
// primitive list does not depend on scheduling).
#include <atomic>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scene.h"

namespace pbrt_amd {

namespace {

struct PrimInfo {   // BVHPrimitiveInfo, bvh.cpp:50-59
    size_t primitiveNumber;
    Bounds3 bounds;
    Vec3 centroid;
};

struct BuildNode {   // BVHBuildNode, bvh.cpp:61-83
    Bounds3 bounds;
    BuildNode *children[2] = {nullptr, nullptr};
    int splitAxis = 0, firstPrimOffset = 0, nPrimitives = 0;
};

}  // namespace
// Threads for host-side scene construction: --nthreads / PBRT_AMD_NTHREADS, else the CPUs this process may really use
// (hardware threads, capped by the cgroup CPU quota of a container).
int NumHostThreads() {
    extern int g_optionNThreads;
    if (g_optionNThreads > 0) return g_optionNThreads;
    if (const char *e = std::getenv("PBRT_AMD_NTHREADS")) { int v = std::atoi(e); if (v > 0) return v; }
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long per = 0;
        if (std::fscanf(f, "%63s %ld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) n = std::max(1, std::min(n, (int)((std::atol(q) + per / 2) / per)));
        std::fclose(f);
    }
    return n;
}
namespace {
struct Builder {
    std::vector<PrimInfo> &info;
    int maxPrimsInNode;
    BVHAccel::SplitMethod method;
    std::mutex poolMutex;
    std::vector<std::unique_ptr<std::deque<BuildNode>>> pools;
    std::atomic<int> liveTasks{0};
    int maxTasks;

    Builder(std::vector<PrimInfo> &info, int maxPrims, BVHAccel::SplitMethod m)
        : info(info), maxPrimsInNode(maxPrims), method(m) {
        maxTasks = NumHostThreads();
    }
    std::deque<BuildNode> *newPool() {
        std::lock_guard<std::mutex> g(poolMutex);
        pools.emplace_back(new std::deque<BuildNode>());
        return pools.back().get();
    }
    static BuildNode *leaf(BuildNode *n, int start, int end, const Bounds3 &b) {
        n->firstPrimOffset = start;   // == orderedPrims.size() at this point of the reference's DFS
        n->nPrimitives = end - start;
        n->bounds = b;
        return n;
    }
    BuildNode *build(int start, int end, std::deque<BuildNode> *pool);
};

BuildNode *Builder::build(int start, int end, std::deque<BuildNode> *pool) {
    pool->emplace_back();
    BuildNode *node = &pool->back();
    Bounds3 bounds;
    for (int i = start; i < end; ++i) bounds = Union(bounds, info[i].bounds);
    int nPrimitives = end - start;
    if (nPrimitives == 1) return leaf(node, start, end, bounds);

    Bounds3 centroidBounds;
    for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, info[i].centroid);
    int dim = centroidBounds.MaximumExtent();
    int mid = (start + end) / 2;
    if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) return leaf(node, start, end, bounds);

    auto equalCounts = [&]() {
        mid = (start + end) / 2;
        std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                         [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
    };
    bool done = false;
    if (method == BVHAccel::SplitMethod::Middle) {
        Float pmid = (centroidBounds.pMin[dim] + centroidBounds.pMax[dim]) / 2;
        PrimInfo *midPtr = std::partition(&info[start], &info[end - 1] + 1,
                                          [dim, pmid](const PrimInfo &pi) { return pi.centroid[dim] < pmid; });
        mid = midPtr - &info[0];
        if (mid != start && mid != end) done = true;
    }
    if (!done && (method == BVHAccel::SplitMethod::Middle || method == BVHAccel::SplitMethod::EqualCounts)) {
        equalCounts();
    } else if (!done) {   // SAH (bvh.cpp:315-393)
        if (nPrimitives <= 2) {
            equalCounts();
        } else {
            const int nBuckets = 12;
            struct Bucket { int count = 0; Bounds3 bounds; } buckets[nBuckets];
            auto bucketOf = [&](const PrimInfo &pi) {
                int b = nBuckets * centroidBounds.Offset(pi.centroid)[dim];
                if (b == nBuckets) b = nBuckets - 1;
                return b;
            };
            for (int i = start; i < end; ++i) {
                int b = bucketOf(info[i]);
                buckets[b].count++;
                buckets[b].bounds = Union(buckets[b].bounds, info[i].bounds);
            }
            Float cost[nBuckets - 1];
            for (int i = 0; i < nBuckets - 1; ++i) {
                Bounds3 b0, b1;
                int count0 = 0, count1 = 0;
                for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
                for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
                cost[i] = 1 + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
            }
            Float minCost = cost[0];
            int minCostSplitBucket = 0;
            for (int i = 1; i < nBuckets - 1; ++i)
                if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
            Float leafCost = nPrimitives;
            if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
                PrimInfo *pmid = std::partition(&info[start], &info[end - 1] + 1, [&](const PrimInfo &pi) {
                    return bucketOf(pi) <= minCostSplitBucket;
                });
                mid = pmid - &info[0];
            } else {
                return leaf(node, start, end, bounds);
            }
        }
    }
    node->splitAxis = dim;
    node->nPrimitives = 0;
    // children: fork the left half onto a worker when the sub-tree is big enough
    const int forkThreshold = 1 << 16;
    if (mid - start > forkThreshold && end - mid > forkThreshold && liveTasks.load() < maxTasks) {
        ++liveTasks;
        std::deque<BuildNode> *p2 = newPool();
        auto fut = std::async(std::launch::async, [this, start, mid, p2]() { return build(start, mid, p2); });
        node->children[1] = build(mid, end, pool);
        node->children[0] = fut.get();
        --liveTasks;
    } else {
        node->children[0] = build(start, mid, pool);
        node->children[1] = build(mid, end, pool);
    }
    node->bounds = Union(node->children[0]->bounds, node->children[1]->bounds);   // InitInterior, bvh.cpp:75
    return node;
}

// ---- SplitMethod::HLBVH (accelerators/bvh.cpp:404-638): Morton-ordered treelets under a SAH top.  Restated with the reference's arithmetic so that the
// tree -- node bounds, split axes, leaf contents and their order -- is the one the reference builds: 10-bit Morton codes of the centroids' offsets in the centroid
// bounds, a stable LSD radix sort (6 bits x 5), one treelet per run of equal top 12 bits, treelets split at the highest differing remaining bit (leaves below
// maxPrimsInNode primitives or out of bits), the treelet roots joined by 12-bucket SAH with the 0.125 traversal cost and std::partition.  The reference hands out
// a leaf's firstPrimOffset from an atomic counter inside a ParallelFor, i.e. in scheduling order; this build takes them in treelet order (the reference with one
// thread) -- which primitives a leaf holds, and their order inside it, do not depend on that.
struct MortonPrim { int primitiveIndex; uint32_t mortonCode; };

inline uint32_t LeftShift3(uint32_t x) {   // bvh.cpp:107-131: bit i of the 10-bit value moves to bit 3 i
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

struct HLBVH {
    const std::vector<PrimInfo> &info;
    int maxPrimsInNode;
    std::deque<BuildNode> pool;
    std::vector<size_t> ordered;   // orderedPrims, as primitive numbers
    size_t orderedOffset = 0;
    bool degenerate = false;

    HLBVH(const std::vector<PrimInfo> &info, int maxPrims) : info(info), maxPrimsInNode(maxPrims) {}

    BuildNode *emit(const MortonPrim *mp, int n, int bitIndex) {   // emitLBVH, bvh.cpp:472-538
        if (bitIndex == -1 || n < maxPrimsInNode) {
            pool.emplace_back();
            BuildNode *node = &pool.back();
            Bounds3 b;
            node->firstPrimOffset = (int)orderedOffset;
            for (int i = 0; i < n; ++i) {
                ordered[orderedOffset + i] = (size_t)mp[i].primitiveIndex;
                b = Union(b, info[mp[i].primitiveIndex].bounds);
            }
            orderedOffset += n;
            node->nPrimitives = n;
            node->bounds = b;
            return node;
        }
        const uint32_t mask = 1u << bitIndex;
        if ((mp[0].mortonCode & mask) == (mp[n - 1].mortonCode & mask)) return emit(mp, n, bitIndex - 1);
        int lo = 0, hi = n - 1;   // the first primitive with the bit set (the codes are sorted)
        while (lo + 1 != hi) {
            int mid = (lo + hi) / 2;
            if ((mp[lo].mortonCode & mask) == (mp[mid].mortonCode & mask)) lo = mid; else hi = mid;
        }
        pool.emplace_back();
        BuildNode *node = &pool.back();   // (created before its children, like the reference: the DFS flattening does not depend on it)
        node->children[0] = emit(mp, hi, bitIndex - 1);
        node->children[1] = emit(mp + hi, n - hi, bitIndex - 1);
        node->splitAxis = bitIndex % 3;
        node->nPrimitives = 0;
        node->bounds = Union(node->children[0]->bounds, node->children[1]->bounds);
        return node;
    }

    BuildNode *upper(std::vector<BuildNode *> &roots, int start, int end) {   // buildUpperSAH, bvh.cpp:540-638
        if (end - start == 1) return roots[start];
        pool.emplace_back();
        BuildNode *node = &pool.back();
        Bounds3 bounds, centroidBounds;
        for (int i = start; i < end; ++i) bounds = Union(bounds, roots[i]->bounds);
        for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, (roots[i]->bounds.pMin + roots[i]->bounds.pMax) * 0.5f);
        const int dim = centroidBounds.MaximumExtent();
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) { degenerate = true; return roots[start]; }   // the reference CHECK-fails here (bvh.cpp:563)
        const int nBuckets = 12;
        struct Bucket { int count = 0; Bounds3 bounds; } buckets[nBuckets];
        const Float cmin = centroidBounds.pMin[dim], cmax = centroidBounds.pMax[dim];
        auto bucketOf = [=](const BuildNode *r) {
            Float centroid = (r->bounds.pMin[dim] + r->bounds.pMax[dim]) * 0.5f;
            int b = nBuckets * ((centroid - cmin) / (cmax - cmin));
            if (b == nBuckets) b = nBuckets - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            int b = bucketOf(roots[i]);
            buckets[b].count++;
            buckets[b].bounds = Union(buckets[b].bounds, roots[i]->bounds);
        }
        Float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            Bounds3 b0, b1;
            int count0 = 0, count1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
            for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
            cost[i] = .125f + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
        }
        Float minCost = cost[0];
        int minCostSplitBucket = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
        BuildNode **pmid = std::partition(&roots[start], &roots[end - 1] + 1, [&](const BuildNode *r) { return bucketOf(r) <= minCostSplitBucket; });
        const int mid = (int)(pmid - &roots[0]);
        if (mid <= start || mid >= end) { degenerate = true; return roots[start]; }   // (CHECK_GT / CHECK_LT, bvh.cpp:630-631)
        node->splitAxis = dim;
        node->nPrimitives = 0;
        node->children[0] = upper(roots, start, mid);
        node->children[1] = upper(roots, mid, end);
        node->bounds = Union(node->children[0]->bounds, node->children[1]->bounds);
        return node;
    }

    BuildNode *build() {   // HLBVHBuild, bvh.cpp:404-470
        Bounds3 bounds;
        for (const PrimInfo &pi : info) bounds = Union(bounds, pi.centroid);
        std::vector<MortonPrim> mp(info.size()), tmp(info.size());
        for (size_t i = 0; i < info.size(); ++i) {
            const Float mortonScale = 1 << 10;
            Vec3 o = bounds.Offset(info[i].centroid) * mortonScale;
            mp[i].primitiveIndex = (int)info[i].primitiveNumber;
            mp[i].mortonCode = (LeftShift3((uint32_t)o.z) << 2) | (LeftShift3((uint32_t)o.y) << 1) | LeftShift3((uint32_t)o.x);
        }
        for (int pass = 0; pass < 5; ++pass) {   // RadixSort, bvh.cpp:140-179 (stable, 6 bits per pass)
            const int lowBit = pass * 6;
            std::vector<MortonPrim> &in = (pass & 1) ? tmp : mp, &out = (pass & 1) ? mp : tmp;
            int count[64] = {0}, outIndex[64];
            for (const MortonPrim &m : in) ++count[(m.mortonCode >> lowBit) & 63];
            outIndex[0] = 0;
            for (int i = 1; i < 64; ++i) outIndex[i] = outIndex[i - 1] + count[i - 1];
            for (const MortonPrim &m : in) out[outIndex[(m.mortonCode >> lowBit) & 63]++] = m;
        }
        std::swap(mp, tmp);   // five passes: the result is in the temporary
        ordered.resize(info.size());
        std::vector<BuildNode *> roots;
        const uint32_t mask = 0x3ffc0000u;   // the top 12 of the 30 bits
        for (int start = 0, end = 1; end <= (int)mp.size(); ++end)
            if (end == (int)mp.size() || (mp[start].mortonCode & mask) != (mp[end].mortonCode & mask)) {
                roots.push_back(emit(&mp[start], end - start, 29 - 12));
                start = end;
            }
        return upper(roots, 0, (int)roots.size());
    }
};

size_t countNodes(const BuildNode *n) {   // iterative: trees can be deep
    size_t c = 0;
    std::vector<const BuildNode *> st{n};
    while (!st.empty()) {
        const BuildNode *x = st.back();
        st.pop_back();
        ++c;
        if (x->nPrimitives == 0) { st.push_back(x->children[0]); st.push_back(x->children[1]); }
    }
    return c;
}

}  // namespace

BVHAccel::BVHAccel(const std::vector<GeometricPrimitive> &prims, int maxPrims, SplitMethod m)
    : maxPrimsInNode(std::min(255, maxPrims)), splitMethod(m) {
    // one entry per triangle, in scene order (api.cpp:1365: one GeometricPrimitive per Triangle)
    std::vector<PrimRef> refs;
    size_t total = 0;
    for (auto &gp : prims) total += gp.shape->nTriangles() + (gp.sphere ? 1 : 0) + (gp.instance ? 1 : 0);
    if (total == 0) return;
    refs.reserve(total);
    std::vector<PrimInfo> info(total);
    size_t k = 0;
    for (uint32_t pi = 0; pi < prims.size(); ++pi) {
        const TriangleMesh &mesh = *prims[pi].shape;
        for (int t = 0; t < mesh.nTriangles(); ++t, ++k) {
            const Vec3 &p0 = mesh.p[mesh.indices[3 * t]], &p1 = mesh.p[mesh.indices[3 * t + 1]],
                       &p2 = mesh.p[mesh.indices[3 * t + 2]];
            Bounds3 b = Union(Bounds3(p0, p1), p2);   // Triangle::WorldBound, triangle.cpp:180-186
            info[k].primitiveNumber = k;
            info[k].bounds = b;
            info[k].centroid = .5f * b.pMin + .5f * b.pMax;
            refs.push_back({pi, (uint32_t)t});
        }
        if (prims[pi].instance) {   // a TransformedPrimitive: one primitive with the bounds the reference gives it
            Bounds3 b = prims[pi].instance->worldBound;
            info[k].primitiveNumber = k;
            info[k].bounds = b;
            info[k].centroid = .5f * b.pMin + .5f * b.pMax;
            refs.push_back({pi, 0u});
            ++k;
        }
        if (prims[pi].sphere) {   // one primitive for the whole sphere
            Bounds3 b = prims[pi].sphere->WorldBound();
            info[k].primitiveNumber = k;
            info[k].bounds = b;
            info[k].centroid = .5f * b.pMin + .5f * b.pMax;
            refs.push_back({pi, 0u});
            ++k;
        }
    }
    Builder builder(info, maxPrimsInNode, splitMethod);
    HLBVH hl(info, maxPrimsInNode);
    BuildNode *root;
    primitives.resize(total);
    if (splitMethod == SplitMethod::HLBVH) {
        root = hl.build();
        if (hl.degenerate) {   // the reference aborts on such input (CHECK_NE / CHECK_GT in buildUpperSAH): no tree to reproduce
            Unsupported("Accelerator \"bvh\" \"string splitmethod\" \"hlbvh\": the treelet roots cannot be split (coinciding centroids); the reference's build stops with a CHECK failure on this scene");
            splitMethod = SplitMethod::SAH;
            root = builder.build(0, (int)total, builder.newPool());
            for (size_t i = 0; i < total; ++i) primitives[i] = refs[info[i].primitiveNumber];
        } else {
            for (size_t i = 0; i < total; ++i) primitives[i] = refs[hl.ordered[i]];
        }
    } else {
        root = builder.build(0, (int)total, builder.newPool());
        for (size_t i = 0; i < total; ++i) primitives[i] = refs[info[i].primitiveNumber];
    }

    // flattenBVHTree (bvh.cpp:640-658), iteratively
    nodes.resize(countNodes(root));
    struct Frame { const BuildNode *n; int parent; };   // parent >= 0: this node is parent's second child
    std::vector<Frame> st{{root, -1}};
    int offset = 0;
    while (!st.empty()) {
        Frame f = st.back();
        st.pop_back();
        int my = offset++;
        if (f.parent >= 0) nodes[f.parent].offset = my;
        mi_bvh2_node &ln = nodes[my];
        std::memset(&ln, 0, sizeof(ln));
        for (int a = 0; a < 3; ++a) { ln.bmin[a] = f.n->bounds.pMin[a]; ln.bmax[a] = f.n->bounds.pMax[a]; }
        if (f.n->nPrimitives > 0) {
            ln.offset = f.n->firstPrimOffset;
            ln.n_prims = (uint16_t)f.n->nPrimitives;
        } else {
            ln.axis = (uint8_t)f.n->splitAxis;
            ln.n_prims = 0;
            st.push_back({f.n->children[1], my});   // visited after the whole first sub-tree
            st.push_back({f.n->children[0], -1});
        }
    }
}

Bounds3 BVHAccel::WorldBound() const {
    if (nodes.empty()) return Bounds3();
    Bounds3 b;
    b.pMin = Vec3(nodes[0].bmin[0], nodes[0].bmin[1], nodes[0].bmin[2]);
    b.pMax = Vec3(nodes[0].bmax[0], nodes[0].bmax[1], nodes[0].bmax[2]);
    return b;
}

std::shared_ptr<BVHAccel> CreateBVHAccelerator(const std::vector<GeometricPrimitive> &prims, const ParamSet &ps) {
    std::string name = ps.FindOneString("splitmethod", "sah");   // bvh.cpp:740-760
    BVHAccel::SplitMethod m;
    if (name == "sah") m = BVHAccel::SplitMethod::SAH;
    else if (name == "hlbvh") m = BVHAccel::SplitMethod::HLBVH;
    else if (name == "middle") m = BVHAccel::SplitMethod::Middle;
    else if (name == "equal") m = BVHAccel::SplitMethod::EqualCounts;
    else {
        Warning("BVH split method \"%s\" unknown.  Using \"sah\".", name.c_str());
        m = BVHAccel::SplitMethod::SAH;
    }
    int maxPrimsInNode = ps.FindOneInt("maxnodeprims", 4);
    return std::make_shared<BVHAccel>(prims, maxPrimsInNode, m);
}

}  // namespace pbrt_amd
