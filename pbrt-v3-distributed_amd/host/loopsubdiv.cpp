// Loop subdivision surfaces -> TriangleMesh (host, scene-load time).
// Behaviour follows the reference's shapes/loopsubdiv.cpp:149-470 step by step (same refinement
// rules, same one-ring traversal order and therefore the same float summation order, limit-surface
// push and tangent-derived normals), written over index arrays instead of pointer-linked
// SDVertex/SDFace objects.  killeroo-simple.pbrt (the one scene the reference ships) needs it.
#include <map>
#include <set>

#include "scene.h"

namespace pbrt_amd {

namespace {

inline int NEXT(int i) { return (i + 1) % 3; }
inline int PREV(int i) { return (i + 2) % 3; }

struct SDVertex {
    Vec3 p;
    int startFace = -1, child = -1;
    bool regular = false, boundary = false;
};
struct SDFace {
    int v[3] = {-1, -1, -1};
    int f[3] = {-1, -1, -1};
    int children[4] = {-1, -1, -1, -1};
};

struct Mesh {
    std::vector<SDVertex> V;
    std::vector<SDFace> F;
    int vnum(int face, int vert) const {
        for (int i = 0; i < 3; ++i) if (F[face].v[i] == vert) return i;
        Error("Basic logic error in SDFace::vnum()");
        return 0;
    }
    int nextFace(int face, int vert) const { return F[face].f[vnum(face, vert)]; }
    int prevFace(int face, int vert) const { return F[face].f[PREV(vnum(face, vert))]; }
    int nextVert(int face, int vert) const { return F[face].v[NEXT(vnum(face, vert))]; }
    int prevVert(int face, int vert) const { return F[face].v[PREV(vnum(face, vert))]; }
    int otherVert(int face, int v0, int v1) const {
        for (int i = 0; i < 3; ++i) if (F[face].v[i] != v0 && F[face].v[i] != v1) return F[face].v[i];
        Error("Basic logic error in SDVertex::otherVert()");
        return 0;
    }
    int valence(int vi) const {   // loopsubdiv.cpp:117-133
        const SDVertex &vert = V[vi];
        int f = vert.startFace;
        if (!vert.boundary) {
            int nf = 1;
            while ((f = nextFace(f, vi)) != vert.startFace) ++nf;
            return nf;
        }
        int nf = 1;
        while ((f = nextFace(f, vi)) != -1) ++nf;
        f = vert.startFace;
        while ((f = prevFace(f, vi)) != -1) ++nf;
        return nf + 1;
    }
    void oneRing(int vi, Vec3 *p) const {   // loopsubdiv.cpp:436-454
        const SDVertex &vert = V[vi];
        if (!vert.boundary) {
            int face = vert.startFace;
            do {
                *p++ = V[nextVert(face, vi)].p;
                face = nextFace(face, vi);
            } while (face != vert.startFace);
        } else {
            int face = vert.startFace, f2;
            while ((f2 = nextFace(face, vi)) != -1) face = f2;
            *p++ = V[nextVert(face, vi)].p;
            do {
                *p++ = V[prevVert(face, vi)].p;
                face = prevFace(face, vi);
            } while (face != -1);
        }
    }
    Vec3 weightOneRing(int vi, Float beta) const {   // loopsubdiv.cpp:425-434
        int val = valence(vi);
        std::vector<Vec3> ring(val);
        oneRing(vi, ring.data());
        Vec3 p = (1 - val * beta) * V[vi].p;
        for (int i = 0; i < val; ++i) p = p + beta * ring[i];
        return p;
    }
    Vec3 weightBoundary(int vi, Float beta) const {   // loopsubdiv.cpp:456-466
        int val = valence(vi);
        std::vector<Vec3> ring(val);
        oneRing(vi, ring.data());
        Vec3 p = (1 - 2 * beta) * V[vi].p;
        p = p + beta * ring[0];
        p = p + beta * ring[val - 1];
        return p;
    }
};

inline Float betaOf(int valence) { return valence == 3 ? 3.f / 16.f : 3.f / (8.f * valence); }
inline Float loopGamma(int valence) { return 1.f / (valence + 3.f / (8.f * betaOf(valence))); }
typedef std::pair<int, int> Edge;
inline Edge mkEdge(int a, int b) { return Edge(std::min(a, b), std::max(a, b)); }

}  // namespace

std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &o2w, bool ro, const ParamSet &params) {
    int nLevels = params.FindOneInt("levels", params.FindOneInt("nlevels", 3));
    int nps, nIndices;
    const int *vertexIndices = params.FindInt("indices", &nIndices);
    const Float *P = params.FindPoint3("P", &nps);
    if (!vertexIndices) { Error("Vertex indices \"indices\" not provided for LoopSubdiv shape."); return nullptr; }
    if (!P) { Error("Vertex positions \"P\" not provided for LoopSubdiv shape."); return nullptr; }
    params.FindOneString("scheme", "loop");

    Mesh cur;
    cur.V.resize(nps);
    for (int i = 0; i < nps; ++i) cur.V[i].p = Vec3(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
    int nFaces = nIndices / 3;
    cur.F.resize(nFaces);
    for (int i = 0; i < nFaces; ++i)
        for (int j = 0; j < 3; ++j) {
            int v = vertexIndices[3 * i + j];
            cur.F[i].v[j] = v;
            cur.V[v].startFace = i;
        }
    {   // neighbour pointers (loopsubdiv.cpp:177-197)
        std::map<Edge, std::pair<int, int>> edges;   // edge -> (first face, its edge number)
        for (int i = 0; i < nFaces; ++i)
            for (int e = 0; e < 3; ++e) {
                Edge key = mkEdge(cur.F[i].v[e], cur.F[i].v[NEXT(e)]);
                auto it = edges.find(key);
                if (it == edges.end()) edges[key] = std::make_pair(i, e);
                else {
                    cur.F[it->second.first].f[it->second.second] = i;
                    cur.F[i].f[e] = it->second.first;
                    edges.erase(it);
                }
            }
    }
    for (int i = 0; i < nps; ++i) {   // finish vertex initialisation (:200-213)
        SDVertex &v = cur.V[i];
        if (v.startFace < 0) continue;   // unreferenced control vertex
        int f = v.startFace;
        do { f = cur.nextFace(f, i); } while (f != -1 && f != v.startFace);
        v.boundary = (f == -1);
        int val = cur.valence(i);
        v.regular = (!v.boundary && val == 6) || (v.boundary && val == 4);
    }
    // the reference walks `vertices` (all control vertices); unreferenced ones would crash it, so scenes do not have them
    for (int level = 0; level < nLevels; ++level) {
        Mesh nxt;
        size_t nV = cur.V.size(), nF = cur.F.size();
        nxt.V.resize(nV);
        for (size_t i = 0; i < nV; ++i) { cur.V[i].child = (int)i; nxt.V[i].regular = cur.V[i].regular; nxt.V[i].boundary = cur.V[i].boundary; }
        nxt.F.resize(4 * nF);
        for (size_t i = 0; i < nF; ++i) for (int k = 0; k < 4; ++k) cur.F[i].children[k] = (int)(4 * i + k);
        // even vertices (:237-250)
        for (size_t i = 0; i < nV; ++i) {
            if (!cur.V[i].boundary) {
                if (cur.V[i].regular) nxt.V[i].p = cur.weightOneRing((int)i, 1.f / 16.f);
                else nxt.V[i].p = cur.weightOneRing((int)i, betaOf(cur.valence((int)i)));
            } else
                nxt.V[i].p = cur.weightBoundary((int)i, 1.f / 8.f);
        }
        // odd (edge) vertices (:252-284)
        std::map<Edge, int> edgeVerts;
        for (size_t fi = 0; fi < nF; ++fi)
            for (int k = 0; k < 3; ++k) {
                int a = cur.F[fi].v[k], b = cur.F[fi].v[NEXT(k)];
                Edge edge = mkEdge(a, b);
                if (edgeVerts.count(edge)) continue;
                SDVertex vert;
                vert.regular = true;
                vert.boundary = (cur.F[fi].f[k] == -1);
                vert.startFace = cur.F[fi].children[3];
                const Vec3 &p0 = cur.V[edge.first].p, &p1 = cur.V[edge.second].p;
                if (vert.boundary) {
                    vert.p = 0.5f * p0;
                    vert.p = vert.p + 0.5f * p1;
                } else {
                    vert.p = (3.f / 8.f) * p0;
                    vert.p = vert.p + (3.f / 8.f) * p1;
                    vert.p = vert.p + (1.f / 8.f) * cur.V[cur.otherVert((int)fi, edge.first, edge.second)].p;
                    vert.p = vert.p + (1.f / 8.f) * cur.V[cur.otherVert(cur.F[fi].f[k], edge.first, edge.second)].p;
                }
                edgeVerts[edge] = (int)nxt.V.size();
                nxt.V.push_back(vert);
            }
        // topology (:288-326)
        for (size_t i = 0; i < nV; ++i) {
            int vertNum = cur.vnum(cur.V[i].startFace, (int)i);
            nxt.V[i].startFace = cur.F[cur.V[i].startFace].children[vertNum];
        }
        for (size_t fi = 0; fi < nF; ++fi) {
            const SDFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nxt.F[face.children[3]].f[j] = face.children[NEXT(j)];
                nxt.F[face.children[j]].f[NEXT(j)] = face.children[3];
                int f2 = face.f[j];
                nxt.F[face.children[j]].f[j] = f2 != -1 ? cur.F[f2].children[cur.vnum(f2, face.v[j])] : -1;
                f2 = face.f[PREV(j)];
                nxt.F[face.children[j]].f[PREV(j)] = f2 != -1 ? cur.F[f2].children[cur.vnum(f2, face.v[j])] : -1;
            }
        }
        for (size_t fi = 0; fi < nF; ++fi) {
            const SDFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nxt.F[face.children[j]].v[j] = cur.V[face.v[j]].child;
                int vert = edgeVerts[mkEdge(face.v[j], face.v[NEXT(j)])];
                nxt.F[face.children[j]].v[NEXT(j)] = vert;
                nxt.F[face.children[NEXT(j)]].v[j] = vert;
                nxt.F[face.children[3]].v[j] = vert;
            }
        }
        cur = std::move(nxt);
    }
    // limit surface (:333-341)
    size_t nV = cur.V.size();
    std::vector<Vec3> pLimit(nV);
    for (size_t i = 0; i < nV; ++i) {
        if (cur.V[i].boundary) pLimit[i] = cur.weightBoundary((int)i, 1.f / 5.f);
        else pLimit[i] = cur.weightOneRing((int)i, loopGamma(cur.valence((int)i)));
    }
    for (size_t i = 0; i < nV; ++i) cur.V[i].p = pLimit[i];
    // normals from limit-surface tangents (:343-381)
    std::vector<Vec3> Ns(nV);
    std::vector<Vec3> pRing(16);
    for (size_t i = 0; i < nV; ++i) {
        Vec3 S(0, 0, 0), T(0, 0, 0);
        int valence = cur.valence((int)i);
        if (valence > (int)pRing.size()) pRing.resize(valence);
        cur.oneRing((int)i, pRing.data());
        const SDVertex &vertex = cur.V[i];
        if (!vertex.boundary) {
            for (int j = 0; j < valence; ++j) {
                S = S + std::cos(2 * kPi * j / valence) * pRing[j];
                T = T + std::sin(2 * kPi * j / valence) * pRing[j];
            }
        } else {
            S = pRing[valence - 1] - pRing[0];
            if (valence == 2) T = pRing[0] + pRing[1] - (Float)2 * vertex.p;
            else if (valence == 3) T = pRing[1] - vertex.p;
            else if (valence == 4)
                T = (Float)-1 * pRing[0] + (Float)2 * pRing[1] + (Float)2 * pRing[2] + (Float)-1 * pRing[3] + (Float)-2 * vertex.p;
            else {
                Float theta = kPi / float(valence - 1);
                T = std::sin(theta) * (pRing[0] + pRing[valence - 1]);
                for (int k = 1; k < valence - 1; ++k) {
                    Float wt = (2 * std::cos(theta) - 2) * std::sin((k)*theta);
                    T = T + wt * pRing[k];
                }
                T = -T;
            }
        }
        Ns[i] = Cross(S, T);
    }
    std::vector<int> verts(3 * cur.F.size());
    for (size_t i = 0; i < cur.F.size(); ++i) for (int j = 0; j < 3; ++j) verts[3 * i + j] = cur.F[i].v[j];
    return CreateTriangleMesh(o2w, ro, (int)cur.F.size(), verts.data(), (int)nV, pLimit.data(), nullptr, Ns.data(), nullptr);
}

}  // namespace pbrt_amd
