// Shape factories that end in a TriangleMesh (host, scene-load time).
// Reference: shapes/triangle.cpp:60-110 (TriangleMesh ctor: vertices/normals/tangents are
// transformed to world space once), :648-744 (CreateTriangleMeshShape parameter handling),
// shapes/plymesh.cpp:157-290 (PLY loading; quads split (0,1,2),(3,0,2), :141-147).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "scene.h"

namespace pbrt_amd {

std::shared_ptr<TriangleMesh> CreateTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTris,
                                                 const int *indices, int nVerts, const Vec3 *P, const Vec3 *S,
                                                 const Vec3 *N, const Float *UV) {
    auto mesh = std::make_shared<TriangleMesh>();
    mesh->indices.assign(indices, indices + 3 * nTris);
    mesh->p.resize(nVerts);
    for (int i = 0; i < nVerts; ++i) mesh->p[i] = o2w.Point(P[i]);
    if (UV) mesh->uv.assign(UV, UV + 2 * nVerts);
    if (N) { mesh->n.resize(nVerts); for (int i = 0; i < nVerts; ++i) mesh->n[i] = o2w.Normal(N[i]); }
    if (S) { mesh->s.resize(nVerts); for (int i = 0; i < nVerts; ++i) mesh->s[i] = o2w.Vector(S[i]); }
    mesh->reverseOrientation = reverseOrientation;
    mesh->transformSwapsHandedness = o2w.SwapsHandedness();   // shape.cpp:47-50
    return mesh;
}

// "alpha" / "shadowalpha" (triangle.cpp:717-741, plymesh.cpp:259-286) are resolved by pbrtShape (host/api.cpp), which owns the texture name maps
std::shared_ptr<TriangleMesh> CreateTriangleMeshShape(const Transform &o2w, bool ro, const ParamSet &ps) {
    int nvi, npi, nuvi = 0, nsi, nni;
    const int *vi = ps.FindInt("indices", &nvi);
    const Float *P = ps.FindPoint3("P", &npi);
    const Float *uvs = ps.FindPoint2("uv", &nuvi);
    if (!uvs) uvs = ps.FindPoint2("st", &nuvi);
    if (!uvs) {
        uvs = ps.FindFloat("uv", &nuvi);
        if (!uvs) uvs = ps.FindFloat("st", &nuvi);
        if (uvs) nuvi /= 2;
    }
    if (uvs) {
        if (nuvi < npi) {
            Error("Not enough of \"uv\"s for triangle mesh.  Expected %d, found %d.  Discarding.", npi, nuvi);
            uvs = nullptr;
        } else if (nuvi > npi)
            Warning("More \"uv\"s provided than will be used for triangle mesh.  (%d expcted, %d found)", npi, nuvi);
    }
    if (!vi) { Error("Vertex indices \"indices\" not provided with triangle mesh shape"); return nullptr; }
    if (!P) { Error("Vertex positions \"P\" not provided with triangle mesh shape"); return nullptr; }
    const Float *S = ps.FindVector3("S", &nsi);
    if (S && nsi != npi) { Error("Number of \"S\"s for triangle mesh must match \"P\"s"); S = nullptr; }
    const Float *N = ps.FindNormal3("N", &nni);
    if (N && nni != npi) { Error("Number of \"N\"s for triangle mesh must match \"P\"s"); N = nullptr; }
    for (int i = 0; i < nvi; ++i)
        if (vi[i] >= npi || vi[i] < 0) {
            Error("trianglemesh has out of-bounds vertex index %d (%d \"P\" values were given", vi[i], npi);
            return nullptr;
        }
    int nfi;
    ps.FindInt("faceIndices", &nfi);   // only consumed by ptex textures (not on this path)
    return CreateTriangleMesh(o2w, ro, nvi / 3, vi, npi, (const Vec3 *)P, (const Vec3 *)S, (const Vec3 *)N, uvs);
}

// ---------------------------------------------------------------- PLY
namespace {
struct PlyProp { std::string name, type, countType, itemType; bool isList = false; };
struct PlyElem { std::string name; long count = 0; std::vector<PlyProp> props; };
int plyTypeSize(const std::string &t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
double plyReadBinary(const unsigned char *&p, const std::string &t, bool swap) {
    int n = plyTypeSize(t);
    unsigned char b[8];
    for (int i = 0; i < n; ++i) b[i] = swap ? p[n - 1 - i] : p[i];
    p += n;
    if (t == "char" || t == "int8") return (double)*(int8_t *)b;
    if (t == "uchar" || t == "uint8") return (double)*(uint8_t *)b;
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, b, 4); return v; }
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); return v; }
    double v; std::memcpy(&v, b, 8); return v;
}
}  // namespace

std::shared_ptr<TriangleMesh> CreatePLYMesh(const Transform &o2w, bool ro, const ParamSet &ps) {
    std::string filename = ps.FindOneFilename("filename", "");
    std::ifstream in(filename, std::ios::binary);
    if (!in) { Error("Couldn't open PLY file \"%s\"", filename.c_str()); return nullptr; }
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto nextLine = [&](std::string *line) {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        *line = data.substr(pos, e - pos);
        if (!line->empty() && line->back() == '\r') line->pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!nextLine(&line) || line != "ply") { Error("\"%s\" is not a PLY file", filename.c_str()); return nullptr; }
    std::string format;
    std::vector<PlyElem> elems;
    while (nextLine(&line)) {
        std::istringstream ls(line);
        std::string tok;
        ls >> tok;
        if (tok == "format") ls >> format;
        else if (tok == "element") { PlyElem e; ls >> e.name >> e.count; elems.push_back(e); }
        else if (tok == "property") {
            PlyProp p;
            ls >> p.type;
            if (p.type == "list") { p.isList = true; ls >> p.countType >> p.itemType; }
            ls >> p.name;
            if (elems.empty()) { Error("PLY property before element in \"%s\"", filename.c_str()); return nullptr; }
            elems.back().props.push_back(p);
        } else if (tok == "end_header") break;
    }
    bool ascii = format == "ascii", swap = format == "binary_big_endian";
    if (!ascii && format != "binary_little_endian" && !swap) { Error("Unknown PLY format in \"%s\"", filename.c_str()); return nullptr; }

    long vertexCount = 0, faceCount = 0;
    for (auto &e : elems) { if (e.name == "vertex") vertexCount = e.count; if (e.name == "face") faceCount = e.count; }
    if (vertexCount == 0 || faceCount == 0) {
        Error("PLY file \"%s\" is invalid! No face/vertex elements found!", filename.c_str());
        return nullptr;
    }
    std::vector<Vec3> P(vertexCount), N;
    std::vector<Float> UV;
    std::vector<int> indices;
    indices.reserve(faceCount * 3);
    bool hasN = false, hasUV = false;

    std::istringstream as;
    const unsigned char *bp = (const unsigned char *)data.data() + std::min(pos, data.size());
    const unsigned char *const bend = (const unsigned char *)data.data() + data.size();
    if (ascii) as.str(data.substr(std::min(pos, data.size())));
    // a file shorter than its header promises (cut off, or still being written by somebody else) is an error, never a smaller mesh: every value read is
    // checked against the end of the file (plymesh.cpp:196-202: rply's ply_read fails the same way and the shape is dropped with an Error)
    bool truncated = false;
    auto readVal = [&](const std::string &type) -> double {
        if (truncated) return 0;
        if (ascii) { double v = 0; if (!(as >> v)) truncated = true; return v; }
        int n = plyTypeSize(type);
        if (n == 0 || bend - bp < n) { truncated = true; return 0; }
        return plyReadBinary(bp, type, swap);
    };
    auto cutOff = [&]() {
        Error("Unable to read the contents of PLY file \"%s\": the file ends before the %ld vertices / %ld faces its header declares", filename.c_str(), vertexCount, faceCount);
        return nullptr;
    };
    for (auto &e : elems) {
        if (truncated) break;
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t k = 0; k < e.props.size(); ++k) {
                const std::string &n = e.props[k].name;
                if (n == "x") ix = k; else if (n == "y") iy = k; else if (n == "z") iz = k;
                else if (n == "nx") inx = k; else if (n == "ny") iny = k; else if (n == "nz") inz = k;
                else if (n == "u" || n == "s" || n == "texture_u" || n == "texture_s") iu = k;   // plymesh.cpp:218-231
                else if (n == "v" || n == "t" || n == "texture_v" || n == "texture_t") iv = k;
            }
            if (ix < 0 || iy < 0 || iz < 0) { Error("PLY file \"%s\": Vertex coordinate property not found!", filename.c_str()); return nullptr; }
            hasN = inx >= 0 && iny >= 0 && inz >= 0;
            hasUV = iu >= 0 && iv >= 0;
            if (hasN) N.resize(vertexCount);
            if (hasUV) UV.resize(2 * vertexCount);
            if (e.count != vertexCount) return cutOff();   // two vertex elements
            for (long i = 0; i < e.count && !truncated; ++i)
                for (size_t k = 0; k < e.props.size(); ++k) {
                    if (e.props[k].isList) { long n = (long)readVal(e.props[k].countType); for (long j = 0; j < n && !truncated; ++j) readVal(e.props[k].itemType); continue; }
                    float v = (float)readVal(e.props[k].type);
                    if ((int)k == ix) P[i].x = v; else if ((int)k == iy) P[i].y = v; else if ((int)k == iz) P[i].z = v;
                    else if (hasN && (int)k == inx) N[i].x = v; else if (hasN && (int)k == iny) N[i].y = v;
                    else if (hasN && (int)k == inz) N[i].z = v;
                    else if (hasUV && (int)k == iu) UV[2 * i] = v; else if (hasUV && (int)k == iv) UV[2 * i + 1] = v;
                }
        } else if (e.name == "face") {
            for (long i = 0; i < e.count && !truncated; ++i)
                for (auto &p : e.props) {
                    if (!p.isList) { readVal(p.type); continue; }
                    long n = (long)readVal(p.countType);
                    if (truncated || n < 0 || n > (1 << 24)) { truncated = true; break; }
                    std::vector<int> face(n);
                    for (long j = 0; j < n; ++j) face[j] = (int)readVal(p.itemType);
                    if (truncated) break;
                    if (p.name != "vertex_indices" && p.name != "vertex_index") continue;
                    if (n != 3 && n != 4) { Warning("plymesh: Ignoring face with %i vertices (only triangles and quads are supported!)", (int)n); continue; }
                    for (long j = 0; j < n; ++j)
                        if (face[j] < 0 || face[j] >= vertexCount) {
                            Error("plymesh: Vertex reference %i is out of bounds! Valid range is [0..%i)", face[j], (int)vertexCount);
                            return nullptr;
                        }
                    indices.insert(indices.end(), {face[0], face[1], face[2]});
                    if (n == 4) indices.insert(indices.end(), {face[3], face[0], face[2]});
                }
        } else {   // skip unknown elements
            for (long i = 0; i < e.count && !truncated; ++i)
                for (auto &p : e.props) {
                    if (p.isList) { long n = (long)readVal(p.countType); for (long j = 0; j < n && !truncated; ++j) readVal(p.itemType); }
                    else readVal(p.type);
                }
        }
    }
    if (truncated) return cutOff();
    return CreateTriangleMesh(o2w, ro, (int)indices.size() / 3, indices.data(), (int)vertexCount, P.data(), nullptr,
                              hasN ? N.data() : nullptr, hasUV ? UV.data() : nullptr);
}

std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &, bool, const ParamSet &);   // loopsubdiv.cpp

// CreateSphereShape shapes/sphere.cpp:319-328 + the Sphere constructor (sphere.h:50-58)
std::shared_ptr<SphereShape> CreateSphereShape(const Transform &o2w, bool ro, const ParamSet &ps) {
    auto sp = std::make_shared<SphereShape>();
    Float radius = ps.FindOneFloat("radius", 1.f);
    Float zmin = ps.FindOneFloat("zmin", -radius);
    Float zmax = ps.FindOneFloat("zmax", radius);
    Float phimax = ps.FindOneFloat("phimax", 360.f);
    sp->o2w = o2w;
    sp->w2o = Transform(o2w.mInv, o2w.m);
    sp->reverseOrientation = ro;
    sp->transformSwapsHandedness = o2w.SwapsHandedness();
    sp->radius = radius;
    sp->zMin = Clamp(std::min(zmin, zmax), -radius, radius);
    sp->zMax = Clamp(std::max(zmin, zmax), -radius, radius);
    sp->thetaMin = std::acos(Clamp(std::min(zmin, zmax) / radius, -1, 1));
    sp->thetaMax = std::acos(Clamp(std::max(zmin, zmax) / radius, -1, 1));
    sp->phiMax = Radians(Clamp(phimax, 0, 360));
    return sp;
}
Bounds3 SphereShape::WorldBound() const {   // Transform::operator()(Bounds3f) transform.cpp:240-249 of ObjectBound() sphere.cpp:43-46
    Vec3 lo(-radius, -radius, zMin), hi(radius, radius, zMax);
    Bounds3 ret;
    bool first = true;
    for (int k = 0; k < 8; ++k) {
        Vec3 c((k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z);
        Vec3 w = o2w.Point(c);
        if (first) { ret = Bounds3(w, w); first = false; } else ret = Union(ret, w);
    }
    return ret;
}

// CreateHeightfield (shapes/heightfield.cpp:41-88): the reference itself turns a height field into a TriangleMesh with uvs
static std::shared_ptr<TriangleMesh> CreateHeightfield(const Transform &o2w, bool ro, const ParamSet &ps) {
    int nx = ps.FindOneInt("nu", -1), ny = ps.FindOneInt("nv", -1), nitems = 0;
    const Float *z = ps.FindFloat("Pz", &nitems);
    if (nx < 2 || ny < 2 || !z || nitems != nx * ny) { Error("heightfield: \"nu\" x \"nv\" must match the number of \"Pz\" values (and be at least 2 x 2)"); return nullptr; }
    int ntris = 2 * (nx - 1) * (ny - 1), nverts = nx * ny;
    std::vector<int> indices(3 * (size_t)ntris);
    std::vector<Vec3> P(nverts);
    std::vector<Float> uvs(2 * (size_t)nverts);
    int pos = 0;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            P[pos].x = uvs[2 * pos] = (float)x / (float)(nx - 1);
            P[pos].y = uvs[2 * pos + 1] = (float)y / (float)(ny - 1);
            P[pos].z = z[pos];
            ++pos;
        }
    int *vp = indices.data();
    for (int y = 0; y < ny - 1; ++y)
        for (int x = 0; x < nx - 1; ++x) {
            auto VERT = [nx](int xx, int yy) { return xx + yy * nx; };
            *vp++ = VERT(x, y); *vp++ = VERT(x + 1, y); *vp++ = VERT(x + 1, y + 1);
            *vp++ = VERT(x, y); *vp++ = VERT(x + 1, y + 1); *vp++ = VERT(x, y + 1);
        }
    return CreateTriangleMesh(o2w, ro, ntris, indices.data(), nverts, P.data(), nullptr, nullptr, uvs.data());
}

std::shared_ptr<TriangleMesh> MakeShapes(const std::string &name, const Transform &o2w, bool ro, const ParamSet &ps) {
    if (name == "trianglemesh") return CreateTriangleMeshShape(o2w, ro, ps);
    if (name == "heightfield") return CreateHeightfield(o2w, ro, ps);
    if (name == "nurbs") return CreateNURBS(o2w, ro, ps);
    if (name == "plymesh") return CreatePLYMesh(o2w, ro, ps);
    if (name == "loopsubdiv") return CreateLoopSubdiv(o2w, ro, ps);
    // the other quadrics / curves: not on the triangle hot path.  Leaving them out would render a plausible but
    // wrong image, so the scene is refused; a name the reference does not know either is skipped with its warning (api.cpp:603)
    static const char *const known[] = {"cone", "cylinder", "disk", "paraboloid", "hyperboloid", "curve"};
    for (const char *k : known)
        if (name == k) {
            Unsupported("Shape \"%s\" has no counterpart on this path (triangle meshes, spheres, loopsubdiv / nurbs / heightfield tessellations are carried)", name.c_str());
            return nullptr;
        }
    Warning("Shape \"%s\" unknown.", name.c_str());
    return nullptr;
}

}  // namespace pbrt_amd
