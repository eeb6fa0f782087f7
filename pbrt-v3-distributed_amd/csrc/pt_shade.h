// Surface interaction, BSDF / BxDFs, microfacet distributions and lights for the shading kernels.
// One lane = one path vertex; a wave is sorted so that its lanes share a material, which makes
// every switch on mi_bxdf::type below wave-uniform (no divergence) and lets the lobe parameters
// come through scalar loads.  Each routine cites the reference code it reproduces.
#pragma once
#include "pt_scene.h"

// big, wave-uniformly-branched routines are real functions (one copy each): keeps the shading kernel's
// register footprint and code size down compared with inlining them at every BSDF::f/Pdf/Sample_f site
#define PT_FN __device__ __noinline__

// ------------------------------------------------------------------ SurfaceInteraction (core/interaction.h:94-157)
struct Isect {
    V3 p, pError, wo, n;   // n = oriented geometric normal
    V3 ns, dpdus;          // shading.n, shading.dpdu
    uint32_t prim;
};

// The few scene tables the out-of-line routines need, passed BY VALUE (registers).  Handing them `const DevScene &`
// would force the whole kernel-argument struct into scratch memory and turn every field access into a memory round trip.
struct GeomTables {
    const uint4 *tri_info;
    const TriShade *tri_shade;
    PT_DEV GeomTables(const DevScene &s) : tri_info(s.tri_info), tri_shade(s.tri_shade) {}
};
// a TriShade record in registers
struct TriShadeRegs {
    float4 a, b, c, d;   // n0.xyz n1.x | n1.yz n2.xy | n2.z uv0.xy uv1.x | uv1.y uv2.xy pad
    PT_DEV V3 n0() const { return V3(a.x, a.y, a.z); }
    PT_DEV V3 n1() const { return V3(a.w, b.x, b.y); }
    PT_DEV V3 n2() const { return V3(b.z, b.w, c.x); }
};
PT_DEV TriShadeRegs LoadTriShade(const TriShade *ts, uint32_t prim) {
    const float4 *q = reinterpret_cast<const float4 *>(ts + prim);
    TriShadeRegs r;
    r.a = q[0]; r.b = q[1]; r.c = q[2]; r.d = q[3];
    return r;
}

// (round 6) what a shading kernel reloads of the hit triangle -- vertices, normals / uvs, tri_info -- from the triangle's ONE 128-byte line (DevScene::tri_rec) when the
// scene has it, else from the three per-triangle arrays.  Same words either way.
#ifndef PT_TRI_REC
#define PT_TRI_REC 1   /* 0: the three per-triangle arrays (rounds 1-5; the A/B partner) */
#endif
PT_DEV void LoadHitTriangle(const DevScene &sc, uint32_t prim, V3 *p0, V3 *p1, V3 *p2, uint32_t *flags, TriShadeRegs *tsr, uint4 *tinfo) {
#if PT_TRI_REC
    const float4 *tr = sc.tri_rec + 8 * (size_t)prim;   // eight 16-byte loads from one line
    float4 va = tr[0], vb = tr[1], vc = tr[2], i4 = tr[7];
    tsr->a = tr[3]; tsr->b = tr[4]; tsr->c = tr[5]; tsr->d = tr[6];
    Pin(va, vb, vc);
    *p0 = V3(va.x, va.y, va.z); *p1 = V3(vb.x, vb.y, vb.z); *p2 = V3(vc.x, vc.y, vc.z);
    *flags = __float_as_uint(va.w);
    *tinfo = make_uint4(__float_as_uint(i4.x), __float_as_uint(i4.y), __float_as_uint(i4.z), __float_as_uint(i4.w));
#else
    *tinfo = sc.tri_info[prim];
    *tsr = LoadTriShade(sc.tri_shade, prim);   // with the vertices: one memory round trip
    LoadTri(sc, prim, p0, p1, p2, flags);
#endif
}

// Second half of Triangle::Intersect (shapes/triangle.cpp:293-421): build the interaction from the
// barycentrics the traversal found.  rayD = direction of the ray that hit; mflags / tsr = the triangle's mesh flags
// and shading record (loaded by the caller together with the vertices: one memory round trip).
// Arguments and result travel in registers (by value): a pointer to a caller's local would put it in scratch memory.
// (29 argument registers, 15 result registers: more of either would go through the stack; wo and prim are the caller's)
struct IsectCore { V3 p, pError, n, ns, dpdus; };
PT_DEV Isect BuildIsectBody(uint32_t mflags, const TriShadeRegs &tsr, const V3 &p0, const V3 &p1, const V3 &p2, const V3 &bary);
PT_FN IsectCore BuildIsectPre(uint32_t mflags, const TriShadeRegs tsr, const V3 p0, const V3 p1, const V3 p2, const V3 bary) {
    Isect is = BuildIsectBody(mflags, tsr, p0, p1, p2, bary);
    IsectCore c;
    c.p = is.p; c.pError = is.pError; c.n = is.n; c.ns = is.ns; c.dpdus = is.dpdus;
    return c;
}
PT_DEV Isect MakeIsect(const IsectCore &c, const V3 &rayD, uint32_t prim) {
    Isect is;
    is.p = c.p; is.pError = c.pError; is.n = c.n; is.ns = c.ns; is.dpdus = c.dpdus;
    is.wo = Normalize(-rayD);   // Interaction ctor: wo(Normalize(wo)), core/interaction.h:60
    is.prim = prim;
    return is;
}
PT_DEV Isect BuildIsectBody(uint32_t mflags, const TriShadeRegs &tsr, const V3 &p0, const V3 &p1, const V3 &p2, const V3 &bary) {
    Isect isv;
    Isect *is = &isv;
    Float uv[3][2] = {{tsr.c.y, tsr.c.z}, {tsr.c.w, tsr.d.x}, {tsr.d.y, tsr.d.z}};   // Triangle::GetUVs
    Float duv02x = uv[0][0] - uv[2][0], duv02y = uv[0][1] - uv[2][1], duv12x = uv[1][0] - uv[2][0], duv12y = uv[1][1] - uv[2][1];
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    Float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = absf(determinant) < 1e-8;
    V3 dpdu, dpdv;
    if (!degenerateUV) {
        Float invdet = 1 / determinant;
        dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerateUV || Cross(dpdu, dpdv).LengthSquared() == 0) {
        V3 ng = Cross(p2 - p0, p1 - p0);   // non-zero: TRI_FLAG_REJECT triangles never get here
        CoordinateSystem(Normalize(ng), &dpdu, &dpdv);
    }
    Float b0 = bary.x, b1 = bary.y, b2 = bary.z;
    Float xAbsSum = (absf(b0 * p0.x) + absf(b1 * p1.x) + absf(b2 * p2.x));
    Float yAbsSum = (absf(b0 * p0.y) + absf(b1 * p1.y) + absf(b2 * p2.y));
    Float zAbsSum = (absf(b0 * p0.z) + absf(b1 * p1.z) + absf(b2 * p2.z));
    is->pError = gamma_n(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    is->p = b0 * p0 + b1 * p1 + b2 * p2;
    is->wo = V3(); is->prim = 0;
    is->n = is->ns = Normalize(Cross(dp02, dp12));   // triangle.cpp:346
    is->dpdus = dpdu;
    bool flip = (mflags & MI_MESH_FLIP) != 0;
    if (mflags & MI_MESH_HAS_N) {   // :347-415
        V3 n0 = tsr.n0(), n1 = tsr.n1(), n2 = tsr.n2();
        V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (ns.LengthSquared() > 0) ns = Normalize(ns); else ns = is->n;
        V3 ss = Normalize(dpdu);
        V3 ts = Cross(ss, ns);
        if (ts.LengthSquared() > 0.f) { ts = Normalize(ts); ss = Cross(ts, ns); }
        else CoordinateSystem(ns, &ss, &ts);
        V3 sn = Normalize(Cross(ss, ts));   // SetShadingGeometry core/interaction.cpp:73-92 (authoritative)
        if (flip) sn = -sn;
        is->n = Faceforward(is->n, sn);
        is->ns = sn;
        is->dpdus = ss;
        is->n = Faceforward(is->n, is->ns);   // triangle.cpp:417-419
    } else if (flip) {
        is->n = is->ns = -is->n;              // :420-421
    }
    return isv;
}
PT_DEV void BuildIsect(const GeomTables sc, uint32_t prim, const V3 p0, const V3 p1, const V3 p2, const TriHit th, const V3 rayD, Isect *is) {
    uint32_t mflags = sc.tri_info[prim].x;
    TriShadeRegs tsr = LoadTriShade(sc.tri_shade, prim);
    Pin(tsr.a, tsr.b, tsr.c); Pin(tsr.d);
    *is = MakeIsect(BuildIsectPre(mflags, tsr, p0, p1, p2, V3(th.b0, th.b1, th.b2)), rayD, prim);
}

// interaction of a ray with a sphere primitive it is known to hit (the traversal accepted it)
PT_DEV Isect SphereIsectToIsect(const mi_sphere *sp, const V3 &ro, const V3 &rd, uint32_t prim) {
    SphereIsectOut o;
    SphereIsect(sp, ro, rd, PT_INFINITY, &o);
    Isect is;
    is.p = o.p; is.pError = o.pError; is.n = o.n; is.ns = o.ns; is.dpdus = o.dpdus; is.wo = o.wo; is.prim = prim;
    return is;
}

// ------------------------------------------------------------------ BxDFs (core/reflection.{h,cpp})
#define BSDF_REFLECTION 1
#define BSDF_TRANSMISSION 2
#define BSDF_DIFFUSE 4
#define BSDF_GLOSSY 8
#define BSDF_SPECULAR 16
#define BSDF_ALL 31

PT_DEV Float CosTheta(const V3 &w) { return w.z; }
PT_DEV Float Cos2Theta(const V3 &w) { return w.z * w.z; }
PT_DEV Float AbsCosTheta(const V3 &w) { return absf(w.z); }
PT_DEV Float Sin2Theta(const V3 &w) { return mx((Float)0, (Float)1 - Cos2Theta(w)); }
PT_DEV Float SinTheta(const V3 &w) { return sqrtf_(Sin2Theta(w)); }
PT_DEV Float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
PT_DEV Float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
PT_DEV Float CosPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 1 : clampf(w.x / s, -1, 1); }
PT_DEV Float SinPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 0 : clampf(w.y / s, -1, 1); }
PT_DEV Float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
PT_DEV Float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
PT_DEV V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }   // reflection.h:91-93
PT_DEV bool Refract(const V3 &wi, const V3 &n, Float eta, V3 *wt) {                 // reflection.h:95-107
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = mx(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = sqrtf_(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
PT_DEV bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }

PT_DEV Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {   // reflection.cpp:47-68
    cosThetaI = clampf(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { Float t = etaI; etaI = etaT; etaT = t; cosThetaI = absf(cosThetaI); }
    Float sinThetaI = sqrtf_(mx((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = sqrtf_(mx((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
PT_DEV RGB FrConductor(Float cosThetaI, const RGB &etai, const RGB &etat, const RGB &k) {   // reflection.cpp:71-94
    cosThetaI = clampf(cosThetaI, -1, 1);
    RGB eta = etat / etai, etak = k / etai;
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = (Float)(1. - (double)cosThetaI2);   // `1. - cosThetaI2` is a double expression there
    RGB eta2 = eta * eta, etak2 = etak * etak;
    RGB t0 = eta2 - etak2 - RGB(sinThetaI2);
    RGB a2plusb2 = SqrtRGB(t0 * t0 + 4 * eta2 * etak2);
    RGB t1 = a2plusb2 + RGB(cosThetaI2);
    RGB a = SqrtRGB(0.5f * (a2plusb2 + t0));
    RGB t2 = (Float)2 * cosThetaI * a;
    RGB Rs = (t1 - t2) / (t1 + t2);
    RGB t3 = cosThetaI2 * a2plusb2 + RGB(sinThetaI2 * sinThetaI2);
    RGB t4 = t2 * sinThetaI2;
    RGB Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);   // `0.5 * Spectrum` converts 0.5 to Float
}
PT_DEV RGB FresnelEvaluate(const mi_bxdf &b, Float cosThetaI) {   // reflection.cpp:118-130, FresnelNoOp
    if (b.fresnel == MI_FRESNEL_DIELECTRIC) return RGB(FrDielectric(cosThetaI, b.etaA, b.etaB));
    if (b.fresnel == MI_FRESNEL_CONDUCTOR) return FrConductor(absf(cosThetaI), RGB(1.f), rgb3(b.eta_c), rgb3(b.k_c));
    return RGB(1.f);
}

// MicrofacetDistribution (core/microfacet.{h,cpp}): TrowbridgeReitz (all stock materials) / Beckmann D, Lambda
// SeparableBSSRDF::Sw (core/bssrdf.h:93-96) over FresnelMoment1 (core/bssrdf.cpp:43-52; one coefficient is a double constant in the reference, kept as such)
PT_DEV Float FresnelMoment1D(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) return (Float)(0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945 * eta3 + 2.49277f * eta4 - 0.68441f * eta5);
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
PT_DEV Float BssrdfSw(Float eta, const V3 &w) {
    Float c = 1 - 2 * FresnelMoment1D(1 / eta);
    return (1 - FrDielectric(CosTheta(w), 1, eta)) / (c * PT_PI);
}
struct Distrib {
    Float ax, ay;
    int beckmann;
    PT_DEV Float D(const V3 &wh) const {   // microfacet.cpp:146-163
        Float tan2Theta = Tan2Theta(wh);
        if (__builtin_isinf(tan2Theta)) return 0.;
        const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        if (beckmann)
            return expf_(-tan2Theta * (Cos2Phi(wh) / (ax * ax) + Sin2Phi(wh) / (ay * ay))) / (PT_PI * ax * ay * cos4Theta);
        Float e = (Cos2Phi(wh) / (ax * ax) + Sin2Phi(wh) / (ay * ay)) * tan2Theta;
        return 1 / (PT_PI * ax * ay * cos4Theta * (1 + e) * (1 + e));
    }
    PT_DEV Float Lambda(const V3 &w) const {   // microfacet.cpp:165-184
        Float absTanTheta = absf(TanTheta(w));
        if (__builtin_isinf(absTanTheta)) return 0.;
        Float alpha = sqrtf_(Cos2Phi(w) * ax * ax + Sin2Phi(w) * ay * ay);
        if (beckmann) {
            Float a = 1 / (alpha * absTanTheta);
            if (a >= 1.6f) return 0;
            return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
        }
        Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + sqrtf_(1.f + alpha2Tan2Theta)) / 2;
    }
    PT_DEV Float G1(const V3 &w) const { return 1 / (1 + Lambda(w)); }
    PT_DEV Float G(const V3 &wo, const V3 &wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    PT_DEV Float Pdf(const V3 &wo, const V3 &wh) const { return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo); }   // :338-344
    PT_DEV V3 Sample_wh(const V3 &wo, Float u0, Float u1) const;
};
// TrowbridgeReitzSample11 microfacet.cpp:238-282; the unqualified sqrt/cos/sin of its first branch are
// the double overloads in the reference build -- reproduced with fp64 ops.
PT_DEV void TrowbridgeReitzSample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) {
    if ((double)cosTheta > .9999) {
        Float r = (Float)sqrt((double)(U1 / (1 - U1)));
        Float phi = (Float)(6.28318530718 * (double)U2);
        double sd, cd;
        SinCosD((double)phi, &sd, &cd);   // phi in [0, 2 pi]
        *slope_x = (Float)((double)r * cd);
        *slope_y = (Float)((double)r * sd);
        return;
    }
    Float sinTheta = sqrtf_(mx((Float)0, (Float)1 - cosTheta * cosTheta));
    Float tanTheta = sinTheta / cosTheta;
    Float a = 1 / tanTheta;
    Float G1 = 2 / (1 + sqrtf_(1.f + 1.f / (a * a)));
    Float A = 2 * U1 / G1 - 1;
    Float tmp = 1.f / (A * A - 1.f);
    if ((double)tmp > 1e10) tmp = (Float)1e10;
    Float B = tanTheta;
    Float D = sqrtf_(mx(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
    Float slope_x_1 = B * tmp - D;
    Float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    Float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf_(1.f + *slope_x * *slope_x);
}
PT_DEV V3 Distrib::Sample_wh(const V3 &wo, Float u0, Float u1) const {   // microfacet.cpp:284-336 (visible-area sampling)
    bool flip = wo.z < 0;
    V3 wi = flip ? -wo : wo;
    V3 wiStretched = Normalize(V3(ax * wi.x, ay * wi.y, wi.z));
    Float slope_x, slope_y;
    TrowbridgeReitzSample11(CosTheta(wiStretched), u0, u1, &slope_x, &slope_y);
    Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
    slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
    slope_x = tmp;
    slope_x = ax * slope_x;
    slope_y = ay * slope_y;
    V3 wh = Normalize(V3(-slope_x, -slope_y, 1.f));
    if (flip) wh = -wh;
    return wh;
}

PT_DEV int BxdfFlags(int type) {
    switch (type) {
    case MI_BXDF_LAMBERT_R: case MI_BXDF_OREN_NAYAR: case MI_BXDF_BSSRDF_ADAPTER: return BSDF_REFLECTION | BSDF_DIFFUSE;
    case MI_BXDF_LAMBERT_T: return BSDF_TRANSMISSION | BSDF_DIFFUSE;
    case MI_BXDF_SPECULAR_R: return BSDF_REFLECTION | BSDF_SPECULAR;
    case MI_BXDF_SPECULAR_T: return BSDF_TRANSMISSION | BSDF_SPECULAR;
    case MI_BXDF_FRESNEL_SPEC: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
    case MI_BXDF_MICROFACET_R: case MI_BXDF_FRESNEL_BLEND: return BSDF_REFLECTION | BSDF_GLOSSY;
    case MI_BXDF_MICROFACET_T: return BSDF_TRANSMISSION | BSDF_GLOSSY;
    }
    return 0;
}
PT_DEV bool Matches(int bxdfFlags, int flags) { return (bxdfFlags & flags) == bxdfFlags; }   // reflection.h:215

// Material records are read through the CONSTANT address space with wave-uniform addresses, i.e. with scalar loads
// (s_load -> SGPRs, scalar cache): k_shade runs the BSDF code once per distinct material of a wave (waterfall loop),
// so lobe counts, types and parameters never occupy vector registers or the vector memory pipe.
typedef const __attribute__((address_space(4))) mi_material *MatConst;
typedef const __attribute__((address_space(4))) mi_bxdf *BxdfConst;
typedef const __attribute__((address_space(4))) uint32_t *WordConst;
PT_DEV const mi_bxdf *Generic(BxdfConst b) { return (const mi_bxdf *)(unsigned long long)b; }
PT_DEV void LoadBxdfUniform(mi_bxdf &dst, const mi_bxdf *p) {   // p: the same address in every active lane
    WordConst w = (WordConst)(unsigned long long)UniformPtr(p);
    uint32_t tmp[sizeof(mi_bxdf) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(mi_bxdf) / 4); ++i) tmp[i] = w[i];
    __builtin_memcpy(&dst, tmp, sizeof(mi_bxdf));
}

// U = true: `bp` is wave-uniform (constant materials, scalar loads).  U = false: a per-lane record (the materials of
// textured scenes, built per hit in private memory by pt_material.h) read with ordinary loads.
template <bool U> PT_DEV void LoadBxdf(mi_bxdf &dst, const mi_bxdf *p) {
    if (U) LoadBxdfUniform(dst, p);
    else dst = *p;
}
PT_DEV const mi_bxdf *Generic(const mi_bxdf *b) { return b; }
// (round 6) the evaluation of a lobe whose record is already loaded: BxdfF_unscaled / BxdfPdf load and call these; BxdfFPdf loads ONCE and answers both
template <bool U> PT_DEV RGB BxdfF_body(const mi_bxdf &b, const V3 &wo, const V3 &wi) {
    if constexpr (!U) {   // SeparableBSSRDFAdapter::f core/bssrdf.h:162-167 (TransportMode::Radiance): only per-lane lobe lists carry it (k_shade_vol)
        if (b.type == MI_BXDF_BSSRDF_ADAPTER) { RGB f(BssrdfSw(b.etaB, wi)); return f * (b.etaB * b.etaB); }
    }
    switch (b.type) {
    case MI_BXDF_LAMBERT_R: return rgb3(b.R) * PT_INV_PI;   // reflection.cpp:178
    case MI_BXDF_LAMBERT_T: return rgb3(b.T) * PT_INV_PI;   // :187
    case MI_BXDF_OREN_NAYAR: {                              // :197-219
        Float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
        Float maxCos = 0;
        if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
            Float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi), sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
            Float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = mx((Float)0, dCos);
        }
        Float sinAlpha, tanBeta;
        if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
        return rgb3(b.R) * PT_INV_PI * (b.A + b.B * maxCos * sinAlpha * tanBeta);
    }
    case MI_BXDF_MICROFACET_R: {                            // :226-236
        Distrib dist{b.alphax, b.alphay, b.distrib};
        Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return RGB(0.f);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return RGB(0.f);
        wh = Normalize(wh);
        RGB F = FresnelEvaluate(b, Dot(wi, wh));
        return rgb3(b.R) * dist.D(wh) * dist.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
    }
    case MI_BXDF_MICROFACET_T: {                            // :244-266
        Distrib dist{b.alphax, b.alphay, b.distrib};
        if (SameHemisphere(wo, wi)) return RGB(0.f);
        Float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return RGB(0.f);
        Float eta = CosTheta(wo) > 0 ? (b.etaB / b.etaA) : (b.etaA / b.etaB);
        V3 wh = Normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        RGB F(FrDielectric(Dot(wo, wh), b.etaA, b.etaB));
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float factor = 1 / eta;   // TransportMode::Radiance
        return (RGB(1.f) - F) * rgb3(b.T) *
               absf(dist.D(wh) * dist.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                    (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    case MI_BXDF_FRESNEL_BLEND: {                           // :279-298 (Rd = R, Rs = T)
        Distrib dist{b.alphax, b.alphay, b.distrib};
        RGB Rd = rgb3(b.R), Rs = rgb3(b.T);
        Float ci = 1 - .5f * AbsCosTheta(wi), co = 1 - .5f * AbsCosTheta(wo);
        Float pi5 = (ci * ci) * (ci * ci) * ci, po5 = (co * co) * (co * co) * co;
        RGB diffuse = (28.f / (23.f * PT_PI)) * Rd * (RGB(1.f) - Rs) * (1 - pi5) * (1 - po5);
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return RGB(0.f);
        wh = Normalize(wh);
        Float c = 1 - Dot(wi, wh);
        Float c5 = (c * c) * (c * c) * c;
        RGB schlick = Rs + c5 * (RGB(1.f) - Rs);   // reflection.h:485-488
        RGB specular = dist.D(wh) / (4 * AbsDot(wi, wh) * mx(AbsCosTheta(wi), AbsCosTheta(wo))) * schlick;
        return diffuse + specular;
    }
    default: return RGB(0.f);   // specular lobes evaluate to zero
    }
}
template <bool U = true> PT_FN RGB BxdfF_unscaled(const mi_bxdf *bp, const V3 wo, const V3 wi) {
    mi_bxdf b;
    LoadBxdf<U>(b, bp);
    return BxdfF_body<U>(b, wo, wi);
}
template <bool U = true, class BP = BxdfConst> PT_DEV RGB BxdfF(BP b, const V3 &wo, const V3 &wi) {
    RGB f = BxdfF_unscaled<U>(Generic(b), wo, wi);
    return b->scaled ? RGB(b->scale[0], b->scale[1], b->scale[2]) * f : f;   // ScaledBxDF::f reflection.cpp:97-99
}
template <bool U> PT_DEV Float BxdfPdf_body(const mi_bxdf &b, const V3 &wo, const V3 &wi) {
    if constexpr (!U) { if (b.type == MI_BXDF_BSSRDF_ADAPTER) return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * PT_INV_PI : 0; }   // BxDF::Pdf reflection.cpp:387-389
    switch (b.type) {
    case MI_BXDF_LAMBERT_R: case MI_BXDF_OREN_NAYAR: return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * PT_INV_PI : 0;   // :387-389
    case MI_BXDF_LAMBERT_T: return !SameHemisphere(wo, wi) ? AbsCosTheta(wi) * PT_INV_PI : 0;                          // :400-403
    case MI_BXDF_MICROFACET_R: {                                                                                       // :418-423
        if (!SameHemisphere(wo, wi)) return 0;
        Distrib dist{b.alphax, b.alphay, b.distrib};
        V3 wh = Normalize(wo + wi);
        return dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
    }
    case MI_BXDF_MICROFACET_T: {                                                                                       // :436-448
        if (SameHemisphere(wo, wi)) return 0;
        Distrib dist{b.alphax, b.alphay, b.distrib};
        Float eta = CosTheta(wo) > 0 ? (b.etaB / b.etaA) : (b.etaA / b.etaB);
        V3 wh = Normalize(wo + wi * eta);
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float dwh_dwi = absf((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return dist.Pdf(wo, wh) * dwh_dwi;
    }
    case MI_BXDF_FRESNEL_BLEND: {                                                                                      // :470-475
        if (!SameHemisphere(wo, wi)) return 0;
        Distrib dist{b.alphax, b.alphay, b.distrib};
        V3 wh = Normalize(wo + wi);
        Float pdf_wh = dist.Pdf(wo, wh);
        return .5f * (AbsCosTheta(wi) * PT_INV_PI + pdf_wh / (4 * Dot(wo, wh)));
    }
    default: return 0;
    }
}
template <bool U = true> PT_FN Float BxdfPdf(const mi_bxdf *bp, const V3 wo, const V3 wi) {
    mi_bxdf b;
    LoadBxdf<U>(b, bp);
    return BxdfPdf_body<U>(b, wo, wi);
}
// f (with the ScaledBxDF factor, as BxdfF) and / or Pdf of one lobe from ONE fetch of its record: want bit 0 = f, bit 1 = pdf.  The profile with sub-probes
// (profiles/r06_g_*) prices every out-of-line question to a lobe -- one call + one fetch of the 100-byte record through the scalar cache -- at 2.5-6.5 k wave cycles;
// BSDF::f + BSDF::Pdf of the light-sampling half and the pdf / f loops at the end of BSDF::Sample_f asked each lobe twice.
struct BxdfFPdfR { RGB f; Float pdf; };
template <bool U = true> PT_FN BxdfFPdfR BxdfFPdf(const mi_bxdf *bp, const V3 wo, const V3 wi, int want) {
    mi_bxdf b;
    LoadBxdf<U>(b, bp);
    BxdfFPdfR r;
    r.f = RGB(0.f); r.pdf = 0;
    if (want & 2) r.pdf = BxdfPdf_body<U>(b, wo, wi);
    if (want & 1) {
        RGB f = BxdfF_body<U>(b, wo, wi);
        r.f = b.scaled ? RGB(b.scale[0], b.scale[1], b.scale[2]) * f : f;   // ScaledBxDF::f reflection.cpp:97-99
    }
    return r;
}
// BxDF::Sample_f per lobe; *sampledType preset to the lobe's flags, FresnelSpecular narrows it
struct BxdfSample { RGB f; V3 wi; Float pdf; int sampledType; };
template <bool U> PT_DEV RGB BxdfSample_f_impl(const mi_bxdf *bp, const V3 wo, V3 *wi, Float u0, Float u1, Float *pdf, int *sampledType);
template <bool U = true> PT_FN BxdfSample BxdfSample_f(const mi_bxdf *bp, const V3 wo, Float u0, Float u1, int sampledTypeIn) {   // by value: registers, no scratch
    BxdfSample r;
    r.wi = V3(); r.pdf = 0; r.sampledType = sampledTypeIn;
    r.f = BxdfSample_f_impl<U>(bp, wo, &r.wi, u0, u1, &r.pdf, &r.sampledType);
    return r;
}
template <bool U> PT_DEV RGB BxdfSample_f_impl(const mi_bxdf *bp, const V3 wo, V3 *wi, Float u0, Float u1, Float *pdf, int *sampledType) {
    mi_bxdf b;
    LoadBxdf<U>(b, bp);
    PROBE(22)   // lobe sampling: record load
    RGB f;
    // (round 6) the lobe's generic Pdf / f are evaluated ONCE after the switch, on the record already loaded (rounds 1-5: BxdfPdf / BxdfF_unscaled calls per case, each
    // fetching the record again)
    bool gPdf = false, gF = false;
    if constexpr (!U) {
        if (b.type == MI_BXDF_BSSRDF_ADAPTER) {   // BxDF::Sample_f reflection.cpp:378-385
            *wi = CosineSampleHemisphere(u0, u1);
            if (wo.z < 0) wi->z *= -1;
            *pdf = BxdfPdf<U>(bp, wo, *wi);
            return BxdfF_unscaled<U>(bp, wo, *wi);
        }
    }
    switch (b.type) {
    case MI_BXDF_LAMBERT_R: case MI_BXDF_OREN_NAYAR:   // BxDF::Sample_f reflection.cpp:378-385
        *wi = CosineSampleHemisphere(u0, u1);
        if (wo.z < 0) wi->z *= -1;
        PROBE(23)   // lobe sampling (diffuse): cosine sample
        gPdf = true; gF = true;
        break;
    case MI_BXDF_LAMBERT_T:                            // :391-398
        *wi = CosineSampleHemisphere(u0, u1);
        if (wo.z > 0) wi->z *= -1;
        gPdf = true; gF = true;
        break;
    case MI_BXDF_SPECULAR_R:                           // :136-143
        *wi = V3(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        f = FresnelEvaluate(b, CosTheta(*wi)) * rgb3(b.R) / AbsCosTheta(*wi);
        break;
    case MI_BXDF_SPECULAR_T: {                         // :150-166
        bool entering = CosTheta(wo) > 0;
        Float etaI = entering ? b.etaA : b.etaB, etaT = entering ? b.etaB : b.etaA;
        if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return RGB(0.f);
        *pdf = 1;
        RGB ft = rgb3(b.T) * (RGB(1.f) - RGB(FrDielectric(CosTheta(*wi), b.etaA, b.etaB)));
        ft = ft * ((etaI * etaI) / (etaT * etaT));
        f = ft / AbsCosTheta(*wi);
        break;
    }
    case MI_BXDF_FRESNEL_SPEC: {                       // :477-511
        Float F = FrDielectric(CosTheta(wo), b.etaA, b.etaB);
        if (u0 < F) {
            *wi = V3(-wo.x, -wo.y, wo.z);
            *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
            *pdf = F;
            f = F * rgb3(b.R) / AbsCosTheta(*wi);
        } else {
            bool entering = CosTheta(wo) > 0;
            Float etaI = entering ? b.etaA : b.etaB, etaT = entering ? b.etaB : b.etaA;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return RGB(0.f);
            RGB ft = rgb3(b.T) * (1 - F);
            ft = ft * ((etaI * etaI) / (etaT * etaT));
            *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1 - F;
            f = ft / AbsCosTheta(*wi);
        }
        break;
    }
    case MI_BXDF_MICROFACET_R: {                       // :405-416
        if (wo.z == 0) return RGB(0.f);
        Distrib dist{b.alphax, b.alphay, b.distrib};
        V3 wh = dist.Sample_wh(wo, u0, u1);
        *wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, *wi)) return RGB(0.f);
        *pdf = dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
        gF = true;
        break;
    }
    case MI_BXDF_MICROFACET_T: {                       // :425-434
        if (wo.z == 0) return RGB(0.f);
        Distrib dist{b.alphax, b.alphay, b.distrib};
        V3 wh = dist.Sample_wh(wo, u0, u1);
        Float eta = CosTheta(wo) > 0 ? (b.etaA / b.etaB) : (b.etaB / b.etaA);
        if (!Refract(wo, wh, eta, wi)) return RGB(0.f);
        gPdf = true; gF = true;
        break;
    }
    case MI_BXDF_FRESNEL_BLEND: {                      // :450-468
        if ((double)u0 < .5) {
            u0 = mn(2 * u0, PT_ONE_MINUS_EPS);
            *wi = CosineSampleHemisphere(u0, u1);
            if (wo.z < 0) wi->z *= -1;
        } else {
            u0 = mn(2 * (u0 - .5f), PT_ONE_MINUS_EPS);
            Distrib dist{b.alphax, b.alphay, b.distrib};
            V3 wh = dist.Sample_wh(wo, u0, u1);
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return RGB(0.f);
        }
        gPdf = true; gF = true;
        break;
    }
    }
    if (gPdf) *pdf = BxdfPdf_body<U>(b, wo, *wi);
    if (gF) f = BxdfF_body<U>(b, wo, *wi);
    return b.scaled ? rgb3(b.scale) * f : f;   // ScaledBxDF::Sample_f reflection.cpp:101-106
}

// ------------------------------------------------------------------ BSDF (core/reflection.h:153-202)
template <bool U> struct MatPtrOf { typedef MatConst type; };
template <> struct MatPtrOf<false> { typedef const mi_material *type; };
#ifndef PT_LOBE_HEADER
#define PT_LOBE_HEADER 1
#endif
template <bool U> struct BSDF_T {
    typedef typename MatPtrOf<U>::type MatPtr;
    MatPtr m;   // U: wave-uniform (see above); !U: this lane's own record
    V3 ns, ng, ss, ts;
    // the lobe count and the lobe TYPES of the material, 4 bits each (PT_LOBE_HEADER, round 3): every loop below asks "which lobes match" before it
    // touches a lobe's parameters, and with the types only in the 100-byte lobe records each question was a chain of dependent scalar loads (one
    // per lobe, each in its own cache line: NumComponents alone 1 + n round trips, three times per vertex).  U: one s_load_dwordx2 from the
    // per-material table mi_scene_upload builds (DevScene::mat_pack); otherwise gathered once here.
    int nb;
    uint32_t tpk;
    PT_DEV int LobeType(int i) const { return (int)((tpk >> (4 * i)) & 15u); }
    PT_DEV BSDF_T(const Isect &si, const mi_material *mat, const uint2 *packTable = nullptr, int matIndex = 0)
        : m((MatPtr)(unsigned long long)mat), ns(si.ns), ng(si.n), ss(Normalize(si.dpdus)) {
        ts = Cross(ns, ss);
        if (PT_LOBE_HEADER && U && packTable) {
            const __attribute__((address_space(4))) uint32_t *pt = (const __attribute__((address_space(4))) uint32_t *)(unsigned long long)packTable;
            nb = (int)pt[2 * matIndex]; tpk = pt[2 * matIndex + 1];   // constant address space + wave-uniform index: one s_load_dwordx2
        } else {
            nb = m->n_bxdfs; tpk = 0;
            for (int i = 0; i < nb; ++i) tpk |= (uint32_t)(m->bxdfs[i].type & 15) << (4 * i);
        }
    }
    PT_DEV V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    PT_DEV V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    PT_DEV int NumComponents(int flags) const {
        int n = 0;
        for (int i = 0; i < nb; ++i) if (Matches(BxdfFlags(LobeType(i)), flags)) ++n;
        return n;
    }
    PT_DEV RGB f(const V3 &woW, const V3 &wiW, int flags) const {   // reflection.cpp:670-683
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return RGB(0.f);
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        RGB f(0.f);
        for (int i = 0; i < nb; ++i) {
            auto b = &m->bxdfs[i];
            int t = BxdfFlags(LobeType(i));
            if (Matches(t, flags) && ((reflect && (t & BSDF_REFLECTION)) || (!reflect && (t & BSDF_TRANSMISSION)))) f = f + BxdfF<U>(b, wo, wi);
        }
        PROBE(20)   // BSDF::f (NEE)
        return f;
    }
    PT_DEV Float Pdf(const V3 &woW, const V3 &wiW, int flags) const {   // reflection.cpp:770-785
        if (nb == 0) return 0.f;
        V3 wo = WorldToLocal(woW), wi = WorldToLocal(wiW);
        if (wo.z == 0) return 0.f;
        Float pdf = 0.f;
        int matchingComps = 0;
        for (int i = 0; i < nb; ++i)
            if (Matches(BxdfFlags(LobeType(i)), flags)) { ++matchingComps; pdf += BxdfPdf<U>(Generic(&m->bxdfs[i]), wo, wi); }
        return matchingComps > 0 ? pdf / matchingComps : 0.f;
    }
    // BSDF::f and BSDF::Pdf for the same pair of directions (the light-sampling half of EstimateDirect, integrator.cpp:134-137): the two loops above as one, every
    // matching lobe asked once (BxdfFPdf).  Same terms added in the same (index) order.
    PT_DEV RGB fPdf(const V3 &woW, const V3 &wiW, int flags, Float *pdfOut) const {
        *pdfOut = 0.f;
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return RGB(0.f);
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        RGB f(0.f);
        Float pdf = 0.f;
        int matchingComps = 0;
        for (int i = 0; i < nb; ++i) {
            int t = BxdfFlags(LobeType(i));
            if (!Matches(t, flags)) continue;
            ++matchingComps;
            const bool wantF = (reflect && (t & BSDF_REFLECTION)) || (!reflect && (t & BSDF_TRANSMISSION));
            BxdfFPdfR r = BxdfFPdf<U>(Generic(&m->bxdfs[i]), wo, wi, wantF ? 3 : 2);
            pdf += r.pdf;
            if (wantF) f = f + r.f;
        }
        *pdfOut = matchingComps > 0 ? pdf / matchingComps : 0.f;
        PROBE(20)   // BSDF::f + Pdf (NEE)
        return f;
    }
    PT_DEV RGB Sample_f(const V3 &woWorld, V3 *wiWorld, Float u0, Float u1, Float *pdf, int type, int *sampledType) const {   // reflection.cpp:703-768
        int matchingComps = NumComponents(type);
        if (matchingComps == 0) { *pdf = 0; *sampledType = 0; return RGB(0.f); }
        int comp = mni((int)__builtin_floorf(u0 * matchingComps), matchingComps - 1);
        // the comp-th matching lobe (per lane: u0 differs) -- found with a wave-uniform loop over the lobes
        int chosen = 0, count = comp;
        bool have = false;
        for (int i = 0; i < nb; ++i)
            if (!have && Matches(BxdfFlags(LobeType(i)), type) && count-- == 0) { chosen = i; have = true; }
        Float ur0 = mn(u0 * matchingComps - comp, PT_ONE_MINUS_EPS);
        V3 wi, wo = WorldToLocal(woWorld);
        if (wo.z == 0) return RGB(0.f);
        *pdf = 0;
        int bt = 0;
        RGB f(0.f);
        PROBE(16)   // Sample_f: component choice
        // lanes may have chosen different lobes: each lobe's sampling routine runs for the lanes that picked it,
        // every time with a wave-uniform lobe pointer
        for (int i = 0; i < nb; ++i)
            if (chosen == i) {
                bt = BxdfFlags(LobeType(i));
                BxdfSample bs = BxdfSample_f<U>(Generic(&m->bxdfs[i]), wo, ur0, u1, bt);   // with the lobe's own f: it is the chosen lobe's term of the sum below (same routine, same wo / wi)
                f = bs.f; wi = bs.wi; *pdf = bs.pdf; *sampledType = bs.sampledType;
            }
        PROBE(17)   // Sample_f: the chosen lobe's sampling routine
        if (*pdf == 0) { *sampledType = 0; return RGB(0.f); }
        *wiWorld = LocalToWorld(wi);
        // reflection.cpp:739-763: the pdfs of the other matching lobes, then f over all matching lobes -- one pass, every lobe asked once for what is wanted of it
        if (!(bt & BSDF_SPECULAR)) {
            bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
            const RGB fChosen = f;
            f = RGB(0.f);
            for (int i = 0; i < nb; ++i) {
                int t = BxdfFlags(LobeType(i));
                if (!Matches(t, type)) continue;
                const bool wantPdf = matchingComps > 1 && i != chosen;
                const bool wantF = (reflect && (t & BSDF_REFLECTION)) || (!reflect && (t & BSDF_TRANSMISSION));
                if (i == chosen) { if (wantF) f = f + fChosen; continue; }   // at its place in the index order
                if (!wantPdf && !wantF) continue;
                BxdfFPdfR r = BxdfFPdf<U>(Generic(&m->bxdfs[i]), wo, wi, (wantF ? 1 : 0) | (wantPdf ? 2 : 0));
                if (wantPdf) *pdf += r.pdf;
                if (wantF) f = f + r.f;
            }
        }
        if (matchingComps > 1) *pdf /= matchingComps;
        PROBE(19)   // Sample_f: f over the matching lobes
        return f;
    }
};
typedef BSDF_T<true> BSDF;

// ------------------------------------------------------------------ lights
PT_DEV RGB AreaL(const DevLight &l, const V3 &n, const V3 &w) {   // DiffuseAreaLight::L lights/diffuse.h:56-58
    return (l.two_sided || Dot(n, w) > 0) ? rgb3(l.L) : RGB(0.f);
}

struct ShadowRay { V3 o, d; Float tMax; };
// Interaction::SpawnRayTo(const Interaction&) core/interaction.h:72-78
PT_DEV ShadowRay SpawnRayTo(const Isect &ref, const V3 &p2, const V3 &p2Error, const V3 &n2) {
    ShadowRay r;
    r.o = OffsetRayOrigin(ref.p, ref.pError, ref.n, p2 - ref.p);
    V3 target = OffsetRayOrigin(p2, p2Error, n2, r.o - p2);
    r.d = target - r.o;
    r.tMax = 1 - PT_SHADOW_EPS;
    return r;
}

// Number of leading cdf entries <= u of a non-decreasing cdf[size] -- what FindInterval's bisection (core/pbrt.h:398-411)
// computes -- found with up to 16 independent probes per round: 2-3 memory round trips instead of log2(size) dependent ones.
PT_DEV int CdfCountLE(const float *cdf, int size, Float u) {
    int lo = 0, hi = size;   // entries below lo are <= u; entries from hi on are > u
    while (lo < hi) {
        int step = (hi - lo + 15) >> 4;
        float pv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int idx = lo + (k + 1) * step - 1;
            pv[k] = idx < hi ? cdf[idx] : PT_INFINITY;
        }
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) cnt += (pv[k] <= u) ? 1 : 0;   // a prefix of the probes
        int nlo = lo + cnt * step;
        int nhi = lo + (cnt + 1) * step - 1;
        hi = nhi < hi ? nhi : hi;
        lo = nlo < hi ? nlo : hi;
    }
    return lo;
}

// Distribution1D::SampleDiscrete (core/sampling.h:90-100) on the voxel `vox` of the spatial light table (lightdistrib.cpp:139-152), with a guide
// table in front of the search (round 3).  sp_guide[vox][j] = G | H << 16 for the cell u in [j / M, (j + 1) / M), M a power of two (so u * M and the
// cell bounds are exact): G = #{cdf[k] <= j / M}, H = #{cdf[k] < (j + 1) / M}; the count `first` = #{cdf[k] <= u} that FindInterval's bisection finds
// lies in [G, H] and only cdf[G .. H) decides it.  Round trip 1 = the guide word; round trip 2 = those (usually 0-2) cdf entries AND the candidates
// for func[offset], which depend on G / H only -- two dependent fetches of a few words instead of the search's 24 probes in two rounds + func[offset].
// Same comparisons on the same cdf values, so `offset` and func[offset] are the search's, bit for bit.
// (the voxel of a point: SpatialLightDistribution::Lookup's Bounds3::Offset + clamp, lightdistrib.cpp:139-152, geometry.h:786-792)
PT_DEV size_t SpatialVoxel(const DevScene &sc, const V3 &p) {
    V3 bmin = v3(sc.sp_bmin), bmax = v3(sc.sp_bmax);
    V3 off = p - bmin;
    if (bmax.x > bmin.x) off.x /= bmax.x - bmin.x;
    if (bmax.y > bmin.y) off.y /= bmax.y - bmin.y;
    if (bmax.z > bmin.z) off.z /= bmax.z - bmin.z;
    int v0 = (int)(off.x * sc.sp_nvox[0]), v1 = (int)(off.y * sc.sp_nvox[1]), v2 = (int)(off.z * sc.sp_nvox[2]);
    v0 = v0 < 0 ? 0 : (v0 > sc.sp_nvox[0] - 1 ? sc.sp_nvox[0] - 1 : v0);
    v1 = v1 < 0 ? 0 : (v1 > sc.sp_nvox[1] - 1 ? sc.sp_nvox[1] - 1 : v1);
    v2 = v2 < 0 ? 0 : (v2 > sc.sp_nvox[2] - 1 ? sc.sp_nvox[2] - 1 : v2);
    return ((size_t)v0 * sc.sp_nvox[1] + v1) * sc.sp_nvox[2] + v2;
}
// the guide word of (voxel, u): the first of SpatialPick's two round trips, which k_shade issues together with the voxel's funcInt as soon as the hit point is known
PT_DEV uint32_t SpatialGuideWord(const DevScene &sc, size_t vox, Float u) {
    const uint32_t M = sc.sp_guide_m;
    uint32_t j = (uint32_t)(u * (Float)M);
    j = j < M - 1 ? j : M - 1;
    return sc.sp_guide[vox * M + j];
}
PT_DEV void SpatialPick(const DevScene &sc, size_t vox, Float u, int *offset, Float *funcAt, bool haveGuide = false, uint32_t guideWord = 0) {
    const int size = (int)sc.n_lights + 1;
    const float *cdf = sc.sp_cdf + vox * (size_t)size, *func = sc.sp_func + vox * (size_t)sc.n_lights;
    int first;
    if (sc.sp_guide_m) {
        const uint32_t gh = haveGuide ? guideWord : SpatialGuideWord(sc, vox, u);
        const int lo = (int)(gh & 0xffffu), hi = (int)(gh >> 16);
        if (hi - lo <= 4) {
            float c[4], f[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = lo + k < hi ? cdf[lo + k] : PT_INFINITY;
#pragma unroll
            for (int k = 0; k < 5; ++k) { int o = lo - 1 + k; o = o < 0 ? 0 : (o > size - 2 ? size - 2 : o); f[k] = func[o]; }
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt += (c[k] <= u) ? 1 : 0;   // the cdf is non-decreasing: a prefix
            first = lo + cnt;
            *offset = first - 1 < 0 ? 0 : (first - 1 > size - 2 ? size - 2 : first - 1);
            *funcAt = cnt == 0 ? f[0] : (cnt == 1 ? f[1] : (cnt == 2 ? f[2] : (cnt == 3 ? f[3] : f[4])));
            return;
        }
        first = lo + CdfCountLE(cdf + lo, hi - lo, u);
    } else
        first = CdfCountLE(cdf, size, u);
    *offset = first - 1 < 0 ? 0 : (first - 1 > size - 2 ? size - 2 : first - 1);
    *funcAt = func[*offset];
}

// ---- InfiniteAreaLight with a radiance map (lights/infinite.cpp:92-143)
PT_DEV int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }   // core/pbrt.h:310-313
PT_DEV RGB EnvTexel(const DevEnvMap &e, int s, int t) {   // MIPMap::Texel, ImageWrap::Repeat (mipmap.h:206-228)
    const float *p = e.rgb + 3 * ((size_t)ModI(t, e.height) * e.width + ModI(s, e.width));
    return RGB(p[0], p[1], p[2]);
}
PT_DEV RGB EnvLookup(const DevEnvMap &e, Float s_, Float t_) {   // Lookup(st) = triangle(0, st) (mipmap.h:244-250, 263-275)
    Float s = s_ * e.width - 0.5f, t = t_ * e.height - 0.5f;
    int s0 = (int)__builtin_floorf(s), t0 = (int)__builtin_floorf(t);
    Float ds = s - s0, dt = t - t0;
    RGB a = EnvTexel(e, s0, t0), b = EnvTexel(e, s0, t0 + 1), c = EnvTexel(e, s0 + 1, t0), d = EnvTexel(e, s0 + 1, t0 + 1);
    return ((1 - ds) * (1 - dt)) * a + ((1 - ds) * dt) * b + (ds * (1 - dt)) * c + (ds * dt) * d;
}
PT_DEV V3 Mul3(const V3 &r0, const V3 &r1, const V3 &r2, const V3 &v) {   // Transform::operator()(Vector3f), 3x3 part
    return V3(r0.x * v.x + r0.y * v.y + r0.z * v.z, r1.x * v.x + r1.y * v.y + r1.z * v.z, r2.x * v.x + r2.y * v.y + r2.z * v.z);
}
PT_DEV Float SphericalTheta(const V3 &v) { return acosf_(clampf(v.z, -1, 1)); }                                   // geometry.h:1479-1481
PT_DEV Float SphericalPhi(const V3 &v) { Float p = atan2f_(v.y, v.x); return (p < 0) ? (p + 2 * PT_PI) : p; }      // :1483-1486
#define PT_INV_2PI 0.15915494309189533577f
// InfiniteAreaLight::Le (infinite.cpp:92-96)
__device__ __noinline__ RGB InfiniteLe(const DevLight *dl, const V3 rayD) {
    if (!dl->ext) return rgb3(dl->L);
    V3 w = Normalize(Mul3(v3(dl->p0), v3(dl->p1), v3(dl->p2), rayD));
    return EnvLookup(*(const DevEnvMap *)dl->ext, SphericalPhi(w) * PT_INV_2PI, SphericalTheta(w) * PT_INV_PI);
}
// Distribution1D::SampleContinuous core/sampling.h:72-89
PT_DEV Float SampleContinuous1D(const float *func, const float *cdf, Float funcInt, int n, Float u, Float *pdf, int *off) {
    int first = CdfCountLE(cdf, n + 1, u);
    int offset = first - 1 < 0 ? 0 : (first - 1 > n - 1 ? n - 1 : first - 1);
    *off = offset;
    Float c0 = cdf[offset], c1 = cdf[offset + 1];
    Float du = u - c0;
    if ((c1 - c0) > 0) du /= (c1 - c0);
    *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}

// Distribution2D::SampleContinuous core/sampling.h:127-134 -> (u, v, pdf); out of line: only scenes with a radiance map pay for it
__device__ __noinline__ V3 SampleEnvMap(const DevEnvMap *envp, Float u0, Float u1) {
    const DevEnvMap &env = *envp;
    int nu = 2 * env.width, nv = 2 * env.height, v, dummy;
    Float pdf0, pdf1;
    Float d1 = SampleContinuous1D(env.marg_func, env.marg_cdf, env.marg_func_int, nv, u1, &pdf1, &v);
    Float d0 = SampleContinuous1D(env.cond_func + (size_t)v * nu, env.cond_cdf + (size_t)v * (nu + 1), env.cond_func_int[v], nu, u0, &pdf0, &dummy);
    return V3(d0, d1, pdf0 * pdf1);
}

struct LightSample { RGB Li; V3 wi; Float pdf; ShadowRay shadow; bool delta; };

// A DevLight record in registers: fetched with 7 independent 16-byte loads (one memory round trip)
struct LightRegs {
    int type, tri, two_sided;
    RGB L; Float area;
    V3 pos; Float world_radius;
    Float cos_total, cos_falloff;
    const void *ext;   // DevEnvMap* (infinite light) or mi_sphere* (sphere light)
    V3 p0, p1, p2; uint32_t mesh_flags;
};
PT_DEV LightRegs LoadLight(const DevLight *dl) {
    const float4 *q = reinterpret_cast<const float4 *>(dl);
    float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
    Pin(a, b, c); Pin(e, f, g); Pin(d);
    LightRegs r;
    r.type = (int)__float_as_uint(a.x); r.tri = (int)__float_as_uint(a.y); r.two_sided = (int)__float_as_uint(a.z);
    r.mesh_flags = __float_as_uint(a.w);
    r.L = RGB(b.x, b.y, b.z); r.area = b.w;
    r.pos = V3(c.x, c.y, c.z); r.world_radius = c.w;
    r.cos_total = d.x; r.cos_falloff = d.y; r.ext = (const void *)(((unsigned long long)__float_as_uint(d.w) << 32) | (unsigned long long)__float_as_uint(d.z));
    r.p0 = V3(e.x, e.y, e.z); r.p1 = V3(f.x, f.y, f.z); r.p2 = V3(g.x, g.y, g.z);
    return r;
}
PT_DEV RGB AreaL(const LightRegs &l, const V3 &n, const V3 &w) { return (l.two_sided || Dot(n, w) > 0) ? l.L : RGB(0.f); }

// Light::Sample_Li.  ref*: the reference point's p, pError, n (what Sample_Li and VisibilityTester's SpawnRayTo use).
PT_FN LightSample SampleLi(const GeomTables sc, const DevLight *dl, const V3 refP, const V3 refPError, const V3 refN, Float u0, Float u1) {
    LightSample lsv;
    LightSample *ls = &lsv;
    Isect ref;
    ref.p = refP; ref.pError = refPError; ref.n = refN;
    const LightRegs l = LoadLight(dl);
    ls->pdf = 0; ls->Li = RGB(0.f);
    ls->delta = l.type == MI_LIGHT_POINT || l.type == MI_LIGHT_DISTANT || l.type == MI_LIGHT_SPOT;
    if (l.type == MI_LIGHT_AREA_TRI) {
        // DiffuseAreaLight::Sample_Li lights/diffuse.cpp:68-81 -> Shape::Sample(ref,u) core/shape.cpp:56-70
        // -> Triangle::Sample(u) shapes/triangle.cpp:583-608
        Float su0 = sqrtf_(u0);
        Float b0 = 1 - su0, b1 = u1 * su0;   // UniformSampleTriangle core/sampling.cpp:154-157
        V3 p0 = l.p0, p1 = l.p1, p2 = l.p2;
        V3 p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        V3 n = Normalize(Cross(p1 - p0, p2 - p0));
        uint32_t mflags = l.mesh_flags;
        if (mflags & MI_MESH_HAS_N) {
            TriShadeRegs tsr = LoadTriShade(sc.tri_shade, (uint32_t)l.tri);
            V3 n0 = tsr.n0(), n1 = tsr.n1(), n2 = tsr.n2();
            V3 ns = b0 * n0 + b1 * n1 + (1 - b0 - b1) * n2;
            n = Faceforward(n, ns);
        } else if (mflags & MI_MESH_FLIP)
            n = n * -1.f;
        V3 pAbsSum = Abs(b0 * p0) + Abs(b1 * p1) + Abs((1 - b0 - b1) * p2);
        V3 pError = gamma_n(6) * pAbsSum;
        Float pdf = 1 / l.area;
        V3 wi = p - ref.p;
        if (wi.LengthSquared() == 0) pdf = 0;
        else {
            wi = Normalize(wi);
            pdf *= DistanceSquared(ref.p, p) / AbsDot(n, -wi);
            if (__builtin_isinf(pdf)) pdf = 0.f;
        }
        if (pdf == 0 || (p - ref.p).LengthSquared() == 0) { ls->pdf = 0; ls->Li = RGB(0.f); return lsv; }
        ls->wi = Normalize(p - ref.p);
        ls->pdf = pdf;
        ls->shadow = SpawnRayTo(ref, p, pError, n);
        ls->Li = AreaL(l, n, -ls->wi);
        return lsv;
    }
    if (l.type == MI_LIGHT_POINT) {   // lights/point.cpp:44-53
        V3 pLight = l.pos;
        ls->wi = Normalize(pLight - ref.p);
        ls->pdf = 1.f;
        ls->shadow = SpawnRayTo(ref, pLight, V3(), V3());
        ls->Li = l.L / DistanceSquared(pLight, ref.p);
        return lsv;
    }
    if (l.type == MI_LIGHT_SPOT) {   // lights/spot.cpp:54-72
        V3 pLight = l.pos;
        ls->wi = Normalize(pLight - ref.p);
        ls->pdf = 1.f;
        ls->shadow = SpawnRayTo(ref, pLight, V3(), V3());
        V3 w = -ls->wi;   // Falloff(-wi): wl = Normalize(WorldToLight(w)), rows of WorldToLight in p0..p2
        V3 wl = Normalize(V3(l.p0.x * w.x + l.p0.y * w.y + l.p0.z * w.z, l.p1.x * w.x + l.p1.y * w.y + l.p1.z * w.z,
                             l.p2.x * w.x + l.p2.y * w.y + l.p2.z * w.z));
        Float cosTheta = wl.z, falloff;
        if (cosTheta < l.cos_total) falloff = 0;
        else if (cosTheta >= l.cos_falloff) falloff = 1;
        else {
            Float delta = (cosTheta - l.cos_total) / (l.cos_falloff - l.cos_total);
            falloff = (delta * delta) * (delta * delta);
        }
        ls->Li = l.L * falloff / DistanceSquared(pLight, ref.p);
        return lsv;
    }
    if (l.type == MI_LIGHT_DISTANT) {   // lights/distant.cpp:49-59
        V3 wLight = l.pos;
        ls->wi = wLight;
        ls->pdf = 1;
        V3 pOutside = ref.p + wLight * (2 * l.world_radius);
        ls->shadow = SpawnRayTo(ref, pOutside, V3(), V3());
        ls->Li = l.L;
        return lsv;
    }
    {   // InfiniteAreaLight::Sample_Li lights/infinite.cpp:98-126; without a map the constructor's distribution is uniform (uv = u, mapPdf = 1)
        Float uv0 = u0, uv1 = u1, mapPdf = 1;
        const DevEnvMap *lenv = (const DevEnvMap *)l.ext;
        const bool hasMap = lenv != nullptr;
        if (hasMap) {
            V3 smp = SampleEnvMap(lenv, u0, u1);
            uv0 = smp.x; uv1 = smp.y; mapPdf = smp.z;
            if (mapPdf == 0) { ls->pdf = 0; ls->Li = RGB(0.f); return lsv; }
        }
        Float theta = uv1 * PT_PI, phi = uv0 * 2 * PT_PI;
        Float cosTheta, sinTheta, sinPhi, cosPhi;
        sincosf_(theta, &sinTheta, &cosTheta);
        sincosf_(phi, &sinPhi, &cosPhi);
        V3 wl(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        ls->wi = hasMap ? Mul3(v3(dl->l2w0), v3(dl->l2w1), v3(dl->l2w2), wl) : wl;
        ls->pdf = mapPdf / (2 * PT_PI * PT_PI * sinTheta);
        if (sinTheta == 0) ls->pdf = 0;
        V3 pOutside = ref.p + ls->wi * (2 * l.world_radius);
        ls->shadow = SpawnRayTo(ref, pOutside, V3(), V3());
        ls->Li = hasMap ? EnvLookup(*lenv, uv0, uv1) : l.L;
    }
    return lsv;
}

// Light::Pdf_Li
PT_FN Float PdfLi(const GeomTables sc, const DevLight *dl, const V3 refP, const V3 refPError, const V3 refN, const V3 wi) {
    const LightRegs l = LoadLight(dl);
    if (l.type == MI_LIGHT_AREA_TRI) {   // Shape::Pdf(ref, wi) core/shape.cpp:72-87: intersect that one triangle
        V3 o = OffsetRayOrigin(refP, refPError, refN, wi);
        V3 p0 = l.p0, p1 = l.p1, p2 = l.p2;
        TriHit th;
        if ((l.mesh_flags & 0x80000000u) || !TriangleTest(p0, p1, p2, o, wi, PT_INFINITY, &th)) return 0;   // bit 31: TRI_FLAG_REJECT of that triangle
        // Of the interaction Triangle::Intersect builds (shapes/triangle.cpp:293-421) the pdf reads the hit point and |n . wi|: the point is the barycentric sum and the
        // normal Normalize(Cross(dp02, dp12)) up to its SIGN (shading normals / ReverseOrientation only flip it, :346-421), which AbsDot drops -- the same values bit for
        // bit as through BuildIsect (rounds 1-4), without the shading record's fetch, dpdu / dpdv and the shading frame (round 5).
        const V3 lp = th.b0 * p0 + th.b1 * p1 + th.b2 * p2;
        const V3 ln = Normalize(Cross(p0 - p2, p1 - p2));
        Float pdf = DistanceSquared(refP, lp) / (AbsDot(ln, -wi) * l.area);
        if (__builtin_isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    if (l.type == MI_LIGHT_INFINITE) {   // lights/infinite.cpp:128-137
        V3 wl = l.ext ? Mul3(l.p0, l.p1, l.p2, wi) : wi;
        Float theta = SphericalTheta(wl), phi = SphericalPhi(wl);
        Float sinTheta = sinf_(theta);
        if (sinTheta == 0) return 0;
        Float mapPdf = 1;
        if (l.ext) {   // Distribution2D::Pdf core/sampling.h:135-141
            const DevEnvMap &env = *(const DevEnvMap *)l.ext;
            int nu = 2 * env.width, nv = 2 * env.height;
            int iu = (int)(phi * PT_INV_2PI * nu), iv = (int)(theta * PT_INV_PI * nv);
            iu = iu < 0 ? 0 : (iu > nu - 1 ? nu - 1 : iu); iv = iv < 0 ? 0 : (iv > nv - 1 ? nv - 1 : iv);
            mapPdf = env.cond_func[(size_t)iv * nu + iu] / env.marg_func_int;
        }
        return mapPdf / (2 * PT_PI * PT_PI * sinTheta);
    }
    return 0;
}

// DiffuseAreaLight on a Sphere: Sample_Li (lights/diffuse.cpp:68-81 over Sphere::Sample(ref, u)) and Pdf_Li.  Separate
// out-of-line routines, dispatched on the light type by the caller, so that SampleLi / PdfLi keep their register budget.
PT_FN LightSample SampleLiSphere(const DevLight *dl, const V3 refP, const V3 refPError, const V3 refN, Float u0, Float u1) {
    LightSample lsv;
    LightSample *ls = &lsv;
    Isect ref;
    ref.p = refP; ref.pError = refPError; ref.n = refN;
    ls->delta = false; ls->pdf = 0; ls->Li = RGB(0.f);
    SphereSample ss;
    SphereSampleRef((const mi_sphere *)dl->ext, refP, refPError, refN, u0, u1, &ss);
    if (ss.pdf == 0 || (ss.p - refP).LengthSquared() == 0) return lsv;
    ls->wi = Normalize(ss.p - refP);
    ls->pdf = ss.pdf;
    ls->shadow = SpawnRayTo(ref, ss.p, ss.pError, ss.n);
    ls->Li = AreaL(*dl, ss.n, -ls->wi);
    return lsv;
}
PT_DEV LightSample SampleLiAny(const GeomTables sc, const DevLight *dl, const V3 &refP, const V3 &refPError, const V3 &refN, Float u0, Float u1) {
    if (dl->type == MI_LIGHT_AREA_SPHERE) return SampleLiSphere(dl, refP, refPError, refN, u0, u1);
    return SampleLi(sc, dl, refP, refPError, refN, u0, u1);
}
PT_DEV Float PdfLiAny(const GeomTables sc, const DevLight *dl, const V3 &refP, const V3 &refPError, const V3 &refN, const V3 &wi) {
    if (dl->type == MI_LIGHT_AREA_SPHERE) return SpherePdf((const mi_sphere *)dl->ext, refP, refPError, refN, wi);
    return PdfLi(sc, dl, refP, refPError, refN, wi);
}
