// Shape "nurbs" (shapes/nurbs.cpp): the reference dices a NURBS patch into a 30 x 30 grid of points with normals and
// (u,v) coordinates and renders the resulting TriangleMesh.  Same here: de Boor evaluation of the rational surface and
// its partial derivatives with the reference's operation order, so that the mesh -- and with it every hit -- is the same.
#include <cmath>

#include "scene.h"

namespace pbrt_amd {
namespace {

struct H4 { Float x = 0, y = 0, z = 0, w = 0; };

// index of the knot span that contains t (nurbs.cpp:43-52)
int SpanOf(const Float *knot, int order, Float t) {
    int k = order - 1;
    while (t > knot[k + 1]) ++k;
    return k;
}
// value (and derivative) of the rational curve with `order` control points cp[0 .. order-1] (stride `stride`) that are active on the
// span starting at knot `span` (nurbs.cpp:70-120: the triangular de Boor scheme, last level kept for the derivative)
H4 EvalSpan(int order, const Float *knotAll, int span, const H4 *cp, int stride, Float t, Vec3 *deriv) {
    const Float *knot = knotAll + span;
    std::vector<H4> w(order);
    for (int i = 0; i < order; ++i) w[i] = cp[i * stride];
    for (int i = 0; i < order - 2; ++i)
        for (int j = 0; j < order - 1 - i; ++j) {
            Float alpha = (knot[1 + j] - t) / (knot[1 + j] - knot[j + 2 - order + i]);
            w[j].x = w[j].x * alpha + w[j + 1].x * (1 - alpha);
            w[j].y = w[j].y * alpha + w[j + 1].y * (1 - alpha);
            w[j].z = w[j].z * alpha + w[j + 1].z * (1 - alpha);
            w[j].w = w[j].w * alpha + w[j + 1].w * (1 - alpha);
        }
    Float alpha = (knot[1] - t) / (knot[1] - knot[0]);
    H4 val;
    val.x = w[0].x * alpha + w[1].x * (1 - alpha);
    val.y = w[0].y * alpha + w[1].y * (1 - alpha);
    val.z = w[0].z * alpha + w[1].z * (1 - alpha);
    val.w = w[0].w * alpha + w[1].w * (1 - alpha);
    if (deriv) {
        Float factor = (order - 1) / (knot[1] - knot[0]);
        H4 delta;
        delta.x = (w[1].x - w[0].x) * factor; delta.y = (w[1].y - w[0].y) * factor;
        delta.z = (w[1].z - w[0].z) * factor; delta.w = (w[1].w - w[0].w) * factor;
        deriv->x = delta.x / val.w - (val.x * delta.w / (val.w * val.w));
        deriv->y = delta.y / val.w - (val.y * delta.w / (val.w * val.w));
        deriv->z = delta.z / val.w - (val.z * delta.w / (val.w * val.w));
    }
    return val;
}
// nurbs.cpp:122-151: iso-curve in one direction, then the curve through the iso points in the other; twice for the two derivatives
Vec3 EvalSurface(int uOrder, const Float *uKnot, int ucp, Float u, int vOrder, const Float *vKnot, int vcp, Float v, const H4 *cp, Vec3 *dpdu, Vec3 *dpdv) {
    (void)vcp;
    std::vector<H4> iso(std::max(uOrder, vOrder));
    int uSpan = SpanOf(uKnot, uOrder, u), vSpan = SpanOf(vKnot, vOrder, v);
    int uFirst = uSpan - uOrder + 1, vFirst = vSpan - vOrder + 1;
    for (int i = 0; i < uOrder; ++i) iso[i] = EvalSpan(vOrder, vKnot, vSpan, &cp[uFirst + i + vFirst * ucp], ucp, v, nullptr);
    H4 P = EvalSpan(uOrder, uKnot, uSpan, iso.data(), 1, u, dpdu);
    for (int i = 0; i < vOrder; ++i) iso[i] = EvalSpan(uOrder, uKnot, uSpan, &cp[(vFirst + i) * ucp + uFirst], 1, u, nullptr);
    (void)EvalSpan(vOrder, vKnot, vSpan, iso.data(), 1, v, dpdv);
    return Vec3(P.x / P.w, P.y / P.w, P.z / P.w);
}

}  // namespace

std::shared_ptr<TriangleMesh> CreateNURBS(const Transform &o2w, bool ro, const ParamSet &ps) {   // nurbs.cpp:153-307
    int nu = ps.FindOneInt("nu", -1);
    if (nu == -1) { Error("Must provide number of control points \"nu\" with NURBS shape."); return nullptr; }
    int uorder = ps.FindOneInt("uorder", -1);
    if (uorder == -1) { Error("Must provide u order \"uorder\" with NURBS shape."); return nullptr; }
    int nuknots = 0, nvknots = 0;
    const Float *uknots = ps.FindFloat("uknots", &nuknots);
    if (!uknots) { Error("Must provide u knot vector \"uknots\" with NURBS shape."); return nullptr; }
    if (nuknots != nu + uorder) { Error("Number of knots in u knot vector %d doesn't match sum of number of u control points %d and u order %d.", nuknots, nu, uorder); return nullptr; }
    Float u0 = ps.FindOneFloat("u0", uknots[uorder - 1]), u1 = ps.FindOneFloat("u1", uknots[nu]);
    int nv = ps.FindOneInt("nv", -1);
    if (nv == -1) { Error("Must provide number of control points \"nv\" with NURBS shape."); return nullptr; }
    int vorder = ps.FindOneInt("vorder", -1);
    if (vorder == -1) { Error("Must provide v order \"vorder\" with NURBS shape."); return nullptr; }
    const Float *vknots = ps.FindFloat("vknots", &nvknots);
    if (!vknots) { Error("Must provide v knot vector \"vknots\" with NURBS shape."); return nullptr; }
    if (nvknots != nv + vorder) { Error("Number of knots in v knot vector %d doesn't match sum of number of v control points %d and v order %d.", nvknots, nv, vorder); return nullptr; }
    Float v0 = ps.FindOneFloat("v0", vknots[vorder - 1]), v1 = ps.FindOneFloat("v1", vknots[nv]);
    if (uorder < 2 || vorder < 2) { Error("NURBS orders below 2 are not supported"); return nullptr; }
    bool homogeneous = false;
    int npts = 0;
    const Float *P = ps.FindPoint3("P", &npts);
    if (!P) {
        P = ps.FindFloat("Pw", &npts);
        if (!P) { Error("Must provide control points via \"P\" or \"Pw\" parameter to NURBS shape."); return nullptr; }
        if ((npts % 4) != 0) { Error("Number of \"Pw\" control points provided to NURBS shape must be multiple of four"); return nullptr; }
        npts /= 4;
        homogeneous = true;
    }
    if (npts != nu * nv) { Error("NURBS shape was expecting %dx%d=%d control points, was given %d", nu, nv, nu * nv, npts); return nullptr; }
    std::vector<H4> Pw((size_t)nu * nv);
    for (int i = 0; i < nu * nv; ++i) {
        if (homogeneous) { Pw[i].x = P[4 * i]; Pw[i].y = P[4 * i + 1]; Pw[i].z = P[4 * i + 2]; Pw[i].w = P[4 * i + 3]; }
        else { Pw[i].x = P[3 * i]; Pw[i].y = P[3 * i + 1]; Pw[i].z = P[3 * i + 2]; Pw[i].w = 1.; }
    }
    const int diceu = 30, dicev = 30;   // fixed dicing rates (:234)
    std::vector<Float> ueval(diceu), veval(dicev);
    for (int i = 0; i < diceu; ++i) { Float t = (float)i / (float)(diceu - 1); ueval[i] = (1 - t) * u0 + t * u1; }
    for (int i = 0; i < dicev; ++i) { Float t = (float)i / (float)(dicev - 1); veval[i] = (1 - t) * v0 + t * v1; }
    std::vector<Vec3> pts((size_t)diceu * dicev), nrm((size_t)diceu * dicev);
    std::vector<Float> uvs((size_t)2 * diceu * dicev);
    for (int v = 0; v < dicev; ++v)
        for (int u = 0; u < diceu; ++u) {
            size_t k = (size_t)v * diceu + u;
            uvs[2 * k] = ueval[u]; uvs[2 * k + 1] = veval[v];
            Vec3 dpdu, dpdv;
            pts[k] = EvalSurface(uorder, uknots, nu, ueval[u], vorder, vknots, nv, veval[v], Pw.data(), &dpdu, &dpdv);
            nrm[k] = Normalize(Cross(dpdu, dpdv));
        }
    std::vector<int> idx;
    idx.reserve((size_t)6 * (diceu - 1) * (dicev - 1));
    for (int v = 0; v < dicev - 1; ++v)
        for (int u = 0; u < diceu - 1; ++u) {
            auto VN = [diceu](int uu, int vv) { return vv * diceu + uu; };
            idx.push_back(VN(u, v)); idx.push_back(VN(u + 1, v)); idx.push_back(VN(u + 1, v + 1));
            idx.push_back(VN(u, v)); idx.push_back(VN(u + 1, v + 1)); idx.push_back(VN(u, v + 1));
        }
    return CreateTriangleMesh(o2w, ro, (int)idx.size() / 3, idx.data(), diceu * dicev, pts.data(), nullptr, nrm.data(), uvs.data());
}

}  // namespace pbrt_amd
