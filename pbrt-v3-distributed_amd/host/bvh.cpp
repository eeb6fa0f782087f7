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
    if (m == SplitMethod::HLBVH) {
        Warning("BVH split method \"hlbvh\" builds with \"sah\" in this implementation (same hits; different tree).");
        splitMethod = SplitMethod::SAH;
    }
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
    BuildNode *root = builder.build(0, (int)total, builder.newPool());

    primitives.resize(total);
    for (size_t i = 0; i < total; ++i) primitives[i] = refs[info[i].primitiveNumber];

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
