// Sphere (shapes/sphere.{h,cpp}) on the device: the one quadric this path carries -- intersectable and as a diffuse area
// light, so that scenes like the reference's own killeroo-simple.pbrt run unmodified.  The intersection follows the
// reference op for op, including the running error bounds of EFloat (core/efloat.h).  All routines are out of line:
// only scenes that contain spheres execute them (k_trace / k_shade pick the code path per primitive / per light).
#pragma once
#include "pt_math.h"

struct EFloat {   // core/efloat.h:47-190 (release build)
    float v, low, high;
    PT_DEV EFloat() {}
    PT_DEV EFloat(float v_, float err = 0.f) : v(v_) {
        if (err == 0.f) low = high = v_;
        else { low = NextFloatDown(v_ - err); high = NextFloatUp(v_ + err); }
    }
    PT_DEV EFloat operator+(EFloat ef) const { EFloat r; r.v = v + ef.v; r.low = NextFloatDown(low + ef.low); r.high = NextFloatUp(high + ef.high); return r; }
    PT_DEV EFloat operator-(EFloat ef) const { EFloat r; r.v = v - ef.v; r.low = NextFloatDown(low - ef.high); r.high = NextFloatUp(high - ef.low); return r; }
    PT_DEV EFloat operator*(EFloat ef) const {
        EFloat r;
        r.v = v * ef.v;
        Float p0 = low * ef.low, p1 = high * ef.low, p2 = low * ef.high, p3 = high * ef.high;
        r.low = NextFloatDown(mn(mn(p0, p1), mn(p2, p3)));
        r.high = NextFloatUp(mx(mx(p0, p1), mx(p2, p3)));
        return r;
    }
    PT_DEV EFloat operator/(EFloat ef) const {
        EFloat r;
        r.v = v / ef.v;
        if (ef.low < 0 && ef.high > 0) { r.low = -PT_INFINITY; r.high = PT_INFINITY; }
        else {
            Float d0 = low / ef.low, d1 = high / ef.low, d2 = low / ef.high, d3 = high / ef.high;
            r.low = NextFloatDown(mn(mn(d0, d1), mn(d2, d3)));
            r.high = NextFloatUp(mx(mx(d0, d1), mx(d2, d3)));
        }
        return r;
    }
};
PT_DEV bool EQuadratic(EFloat A, EFloat B, EFloat C, EFloat *t0, EFloat *t1) {   // efloat.h:262-284
    double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    double rootDiscrim = sqrt(discrim);
    EFloat floatRootDiscrim((float)rootDiscrim, (float)((double)PT_MACHINE_EPS * rootDiscrim));
    EFloat q;
    if (B.v < 0) q = EFloat(-.5f) * (B - floatRootDiscrim);
    else q = EFloat(-.5f) * (B + floatRootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (t0->v > t1->v) { EFloat s = *t0; *t0 = *t1; *t1 = s; }
    return true;
}
// Transform applications on a row-major 4x4 (core/transform.h:223-250, 278-352)
PT_DEV V3 SXfPoint(const float *m, const V3 &p) {
    Float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11], wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
PT_DEV V3 SXfPointErr(const float *m, const V3 &p, V3 *pError) {
    Float x = p.x, y = p.y, z = p.z;
    Float xAbsSum = (absf(m[0] * x) + absf(m[1] * y) + absf(m[2] * z) + absf(m[3]));
    Float yAbsSum = (absf(m[4] * x) + absf(m[5] * y) + absf(m[6] * z) + absf(m[7]));
    Float zAbsSum = (absf(m[8] * x) + absf(m[9] * y) + absf(m[10] * z) + absf(m[11]));
    *pError = gamma_n(3) * V3(xAbsSum, yAbsSum, zAbsSum);
    return SXfPoint(m, p);
}
PT_DEV V3 SXfPointErr2(const float *m, const V3 &pt, const V3 &e, V3 *absError) {
    Float x = pt.x, y = pt.y, z = pt.z;
    absError->x = (gamma_n(3) + (Float)1) * (absf(m[0]) * e.x + absf(m[1]) * e.y + absf(m[2]) * e.z) +
                  gamma_n(3) * (absf(m[0] * x) + absf(m[1] * y) + absf(m[2] * z) + absf(m[3]));
    absError->y = (gamma_n(3) + (Float)1) * (absf(m[4]) * e.x + absf(m[5]) * e.y + absf(m[6]) * e.z) +
                  gamma_n(3) * (absf(m[4] * x) + absf(m[5] * y) + absf(m[6] * z) + absf(m[7]));
    absError->z = (gamma_n(3) + (Float)1) * (absf(m[8]) * e.x + absf(m[9]) * e.y + absf(m[10]) * e.z) +
                  gamma_n(3) * (absf(m[8] * x) + absf(m[9] * y) + absf(m[10] * z) + absf(m[11]));
    return SXfPoint(m, pt);
}
PT_DEV V3 SXfVector(const float *m, const V3 &v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
PT_DEV V3 SXfVectorErr(const float *m, const V3 &v, V3 *absError) {
    absError->x = gamma_n(3) * (absf(m[0] * v.x) + absf(m[1] * v.y) + absf(m[2] * v.z));
    absError->y = gamma_n(3) * (absf(m[4] * v.x) + absf(m[5] * v.y) + absf(m[6] * v.z));
    absError->z = gamma_n(3) * (absf(m[8] * v.x) + absf(m[9] * v.y) + absf(m[10] * v.z));
    return SXfVector(m, v);
}
PT_DEV V3 SXfNormal(const float *mInv, const V3 &n) {   // transpose of the inverse
    return V3(mInv[0] * n.x + mInv[4] * n.y + mInv[8] * n.z, mInv[1] * n.x + mInv[5] * n.y + mInv[9] * n.z, mInv[2] * n.x + mInv[6] * n.y + mInv[10] * n.z);
}

struct SphereHit { V3 o, d, pHit; Float t; bool hit; };   // object-space ray, refined hit point, tShapeHit
// the hit test of Sphere::Intersect / IntersectP (shapes/sphere.cpp:48-110 == :164-217)
PT_DEV SphereHit SphereHitTest(const mi_sphere &sp, const V3 &ro, const V3 &rd, Float tMax) {
    SphereHit h;
    h.hit = false;
    V3 oErr, dErr;
    V3 o = SXfPointErr(sp.w2o, ro, &oErr);   // Ray ray = (*WorldToObject)(r, &oErr, &dErr) transform.h:382-394
    V3 d = SXfVectorErr(sp.w2o, rd, &dErr);
    Float lengthSquared = d.LengthSquared();
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), oErr) / lengthSquared;
        o = o + d * dt;
    }
    h.o = o; h.d = d;
    const Float radius = sp.radius, zMin = sp.zmin, zMax = sp.zmax, phiMax = sp.phi_max;
    EFloat ox(o.x, oErr.x), oy(o.y, oErr.y), oz(o.z, oErr.z);
    EFloat dx(d.x, dErr.x), dy(d.y, dErr.y), dz(d.z, dErr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = EFloat(2.f) * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(radius) * EFloat(radius);
    EFloat t0, t1;
    if (!EQuadratic(a, b, c, &t0, &t1)) return h;
    if (t0.high > tMax || t1.low <= 0) return h;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > tMax) return h;
    }
    V3 pHit;
    Float phi;
#define PT_SPHERE_POINT()                                                \
    pHit = o + d * tShapeHit.v;                                          \
    pHit = pHit * (radius / pHit.Length());                              \
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;             \
    phi = atan2f_(pHit.y, pHit.x);                                      \
    if (phi < 0) phi += 2 * PT_PI;
    PT_SPHERE_POINT()
    if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) {
        if (tShapeHit.v == t1.v) return h;
        if (t1.high > tMax) return h;
        tShapeHit = t1;
        PT_SPHERE_POINT()
        if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) return h;
    }
#undef PT_SPHERE_POINT
    h.pHit = pHit; h.t = tShapeHit.v; h.hit = true;
    return h;
}
// traversal entry: t of the hit or a negative value
__device__ __noinline__ Float SphereIntersectT(const mi_sphere *sp, const V3 ro, const V3 rd, Float tMax) {
    SphereHit h = SphereHitTest(*sp, ro, rd, tMax);
    return h.hit ? h.t : -1.f;
}
// interaction of a (known) hit: p, pError, n, ns, dpdus as IsectCore + wo (Sphere::Intersect :112-160, SurfaceInteraction
// core/interaction.cpp:44-71, Transform::operator()(SurfaceInteraction) core/transform.cpp:262-297)
struct SphereIsectOut { V3 p, pError, n, ns, dpdus, wo; bool hit; };
__device__ __noinline__ void SphereIsect(const mi_sphere *spp, const V3 ro, const V3 rd, Float tMax, SphereIsectOut *out) {
    const mi_sphere &sp = *spp;
    SphereHit h = SphereHitTest(sp, ro, rd, tMax);
    out->hit = h.hit;
    if (!h.hit) return;
    const V3 pHit = h.pHit;
    Float theta = acosf_(clampf(pHit.z / sp.radius, -1, 1));
    Float zRadius = sqrtf_(pHit.x * pHit.x + pHit.y * pHit.y);
    Float invZRadius = 1 / zRadius;
    Float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    V3 dpdu(-sp.phi_max * pHit.y, sp.phi_max * pHit.x, 0);
    V3 dpdv = (sp.theta_max - sp.theta_min) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -sp.radius * sinf_(theta));
    V3 pError = gamma_n(5) * Abs(pHit);
    V3 nObj = Normalize(Cross(dpdu, dpdv)), woObj = Normalize(-h.d);
    if (((sp.flags & 1u) != 0) != ((sp.flags & 2u) != 0)) nObj = -nObj;
    out->p = SXfPointErr2(sp.o2w, pHit, pError, &out->pError);
    out->n = Normalize(SXfNormal(sp.w2o, nObj));
    out->wo = Normalize(SXfVector(sp.o2w, woObj));
    out->dpdus = SXfVector(sp.o2w, dpdu);
    out->ns = Faceforward(Normalize(SXfNormal(sp.w2o, nObj)), out->n);
}
// Sphere::Sample(u, pdf) sphere.cpp:221-234
PT_DEV void SphereSampleArea(const mi_sphere &sp, Float u0, Float u1, V3 *p, V3 *pError, V3 *n, Float *pdf) {
    Float z = 1 - 2 * u0;   // UniformSampleSphere core/sampling.cpp:97-102
    Float rr = sqrtf_(mx((Float)0, (Float)1 - z * z));
    Float ph = 2 * PT_PI * u1;
    V3 pObj = V3(0, 0, 0) + sp.radius * V3(rr * cosf_(ph), rr * sinf_(ph), z);
    *n = Normalize(SXfNormal(sp.w2o, V3(pObj.x, pObj.y, pObj.z)));
    if (sp.flags & 1u) *n = *n * -1.f;
    pObj = pObj * (sp.radius / pObj.Length());
    V3 pObjError = gamma_n(5) * Abs(pObj);
    *p = SXfPointErr2(sp.o2w, pObj, pObjError, pError);
    *pdf = 1 / sp.area;
}
// Sphere::Sample(ref, u, pdf) sphere.cpp:236-295
struct SphereSample { V3 p, pError, n; Float pdf; };
__device__ __noinline__ void SphereSampleRef(const mi_sphere *spp, const V3 refP, const V3 refPError, const V3 refN, Float u0, Float u1, SphereSample *out) {
    const mi_sphere &sp = *spp;
    V3 pCenter = SXfPoint(sp.o2w, V3(0, 0, 0));
    V3 pOrigin = OffsetRayOrigin(refP, refPError, refN, pCenter - refP);
    if (DistanceSquared(pOrigin, pCenter) <= sp.radius * sp.radius) {
        SphereSampleArea(sp, u0, u1, &out->p, &out->pError, &out->n, &out->pdf);
        V3 wi = out->p - refP;
        if (wi.LengthSquared() == 0) out->pdf = 0;
        else {
            wi = Normalize(wi);
            out->pdf *= DistanceSquared(refP, out->p) / AbsDot(out->n, -wi);
        }
        if (__builtin_isinf(out->pdf)) out->pdf = 0.f;
        return;
    }
    V3 wc = Normalize(pCenter - refP), wcX, wcY;
    CoordinateSystem(wc, &wcX, &wcY);
    Float radius = sp.radius;
    Float sinThetaMax2 = radius * radius / DistanceSquared(refP, pCenter);
    Float cosThetaMax = sqrtf_(mx((Float)0, 1 - sinThetaMax2));
    Float cosTheta = (1 - u0) + u0 * cosThetaMax;
    Float sinTheta = sqrtf_(mx((Float)0, 1 - cosTheta * cosTheta));
    Float phi = u1 * 2 * PT_PI;
    Float dc = (refP - pCenter).Length();
    Float ds = dc * cosTheta - sqrtf_(mx((Float)0, radius * radius - dc * dc * sinTheta * sinTheta));
    Float cosAlpha = (dc * dc + radius * radius - ds * ds) / (2 * dc * radius);
    Float sinAlpha = sqrtf_(mx((Float)0, 1 - cosAlpha * cosAlpha));
    V3 nWorld = sinAlpha * cosf_(phi) * (-wcX) + sinAlpha * sinf_(phi) * (-wcY) + cosAlpha * (-wc);   // SphericalDirection geometry.h:1473-1477
    V3 pWorld = pCenter + radius * V3(nWorld.x, nWorld.y, nWorld.z);
    out->p = pWorld;
    out->pError = gamma_n(5) * Abs(pWorld);
    out->n = nWorld;
    if (sp.flags & 1u) out->n = out->n * -1.f;
    out->pdf = 1 / (2 * PT_PI * (1 - cosThetaMax));
}
// Sphere::Pdf(ref, wi) sphere.cpp:297-309 (inside the sphere: Shape::Pdf core/shape.cpp:72-87)
__device__ __noinline__ Float SpherePdf(const mi_sphere *spp, const V3 refP, const V3 refPError, const V3 refN, const V3 wi) {
    const mi_sphere &sp = *spp;
    V3 pCenter = SXfPoint(sp.o2w, V3(0, 0, 0));
    V3 pOrigin = OffsetRayOrigin(refP, refPError, refN, pCenter - refP);
    if (DistanceSquared(pOrigin, pCenter) <= sp.radius * sp.radius) {
        SphereIsectOut is;
        SphereIsect(spp, OffsetRayOrigin(refP, refPError, refN, wi), wi, PT_INFINITY, &is);
        if (!is.hit) return 0;
        Float pdf = DistanceSquared(refP, is.p) / (AbsDot(is.n, -wi) * sp.area);
        if (__builtin_isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    Float sinThetaMax2 = sp.radius * sp.radius / DistanceSquared(refP, pCenter);
    Float cosThetaMax = sqrtf_(mx((Float)0, 1 - sinThetaMax2));
    return 1 / (2 * PT_PI * (1 - cosThetaMax));
}
