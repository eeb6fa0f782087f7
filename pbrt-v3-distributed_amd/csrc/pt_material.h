// Per-hit materials of textured scenes: SurfaceInteraction::ComputeDifferentials
// (core/interaction.cpp:104-153), Material::Bump (core/material.cpp:46-83) and Material::ComputeScatteringFunctions of
// the nine materials (materials/*.cpp) evaluated per lane into an mi_material in private memory.  Only the kernels of
// textured scenes (k_shade<..., TEX = true>) include any of this; constant materials keep the pre-evaluated lists and
// the wave-uniform scalar path.
#pragma once
#include "pt_shade.h"
#include "pt_texture.h"

// what textures / bump mapping need of the interaction beyond Isect
struct IsectX {
    Float u, v;
    V3 dpdu, dpdv;            // geometric partial derivatives (ComputeDifferentials)
    V3 dpdvs, dndus, dndvs;   // shading.dpdv, shading.dndu, shading.dndv
    bool flipN;               // shape->reverseOrientation ^ shape->transformSwapsHandedness
    V3 dpdx, dpdy;
    Float dudx, dvdx, dudy, dvdy;
};

// the parts of Triangle::Intersect (shapes/triangle.cpp:293-415) that BuildIsectBody does not keep
__device__ __noinline__ IsectX BuildIsectTex(uint32_t mflags, const TriShadeRegs tsr, const V3 p0, const V3 p1, const V3 p2, const V3 bary) {
    IsectX x;
    Float uv[3][2] = {{tsr.c.y, tsr.c.z}, {tsr.c.w, tsr.d.x}, {tsr.d.y, tsr.d.z}};
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
        V3 ng = Cross(p2 - p0, p1 - p0);
        CoordinateSystem(Normalize(ng), &dpdu, &dpdv);
    }
    Float b0 = bary.x, b1 = bary.y, b2 = bary.z;
    x.u = b0 * uv[0][0] + b1 * uv[1][0] + b2 * uv[2][0];
    x.v = b0 * uv[0][1] + b1 * uv[1][1] + b2 * uv[2][1];
    x.dpdu = dpdu; x.dpdv = x.dpdvs = dpdv;
    x.dndus = x.dndvs = V3(0, 0, 0);
    x.flipN = (mflags & MI_MESH_FLIP) != 0;
    if (mflags & MI_MESH_HAS_N) {
        V3 n0 = tsr.n0(), n1 = tsr.n1(), n2 = tsr.n2();
        V3 ng = Normalize(Cross(dp02, dp12));
        V3 ns = (b0 * n0 + b1 * n1 + b2 * n2);
        if (ns.LengthSquared() > 0) ns = Normalize(ns); else ns = ng;
        V3 ss = Normalize(dpdu);
        V3 ts = Cross(ss, ns);
        if (ts.LengthSquared() > 0.f) { ts = Normalize(ts); ss = Cross(ts, ns); }
        else CoordinateSystem(ns, &ss, &ts);
        x.dpdvs = ts;
        V3 dn1 = n0 - n2, dn2 = n1 - n2;   // :381-414
        if (degenerateUV) {
            V3 dn = Cross(n2 - n0, n1 - n0);
            if (dn.LengthSquared() == 0) x.dndus = x.dndvs = V3(0, 0, 0);
            else CoordinateSystem(dn, &x.dndus, &x.dndvs);
        } else {
            Float invDet = 1 / determinant;
            x.dndus = (duv12y * dn1 - duv02y * dn2) * invDet;
            x.dndvs = (-duv12x * dn1 + duv02x * dn2) * invDet;
        }
    }
    x.dpdx = x.dpdy = V3(0, 0, 0);
    x.dudx = x.dvdx = x.dudy = x.dvdy = 0;
    return x;
}

// The same for a Sphere hit (shapes/sphere.cpp:108-145): (u, v), dpdu / dpdv, dndu / dndv from the fundamental forms (Weingarten), carried to
// world space as Transform::operator()(SurfaceInteraction) does (core/transform.cpp:262-297) -- read by textures and bump mapping only.
__device__ __noinline__ IsectX SphereIsectTex(const mi_sphere *spp, const V3 ro, const V3 rd) {
    const mi_sphere &sp = *spp;
    IsectX x;
    x.u = x.v = 0;
    x.dpdx = x.dpdy = V3(0, 0, 0);
    x.dudx = x.dvdx = x.dudy = x.dvdy = 0;
    x.flipN = ((sp.flags & 1u) != 0) != ((sp.flags & 2u) != 0);
    SphereHit h = SphereHitTest(sp, ro, rd, PT_INFINITY);
    if (!h.hit) return x;   // (cannot happen: the traversal found this hit with the same code)
    const V3 pHit = h.pHit;
    Float phi = atan2f_(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PT_PI;
    const Float phiMax = sp.phi_max, dTheta = sp.theta_max - sp.theta_min;
    Float theta = acosf_(clampf(pHit.z / sp.radius, -1, 1));
    x.u = phi / phiMax;
    x.v = (theta - sp.theta_min) / dTheta;
    Float zRadius = sqrtf_(pHit.x * pHit.x + pHit.y * pHit.y);
    Float invZRadius = 1 / zRadius;
    Float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
    V3 dpdv = dTheta * V3(pHit.z * cosPhi, pHit.z * sinPhi, -sp.radius * sinf_(theta));
    V3 d2Pduu = -phiMax * phiMax * V3(pHit.x, pHit.y, 0);
    V3 d2Pduv = dTheta * pHit.z * phiMax * V3(-sinPhi, cosPhi, 0.);
    V3 d2Pdvv = -dTheta * dTheta * V3(pHit.x, pHit.y, pHit.z);
    Float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
    V3 N = Normalize(Cross(dpdu, dpdv));
    Float e = Dot(N, d2Pduu), f = Dot(N, d2Pduv), g = Dot(N, d2Pdvv);
    Float invEGF2 = 1 / (E * G - F * F);
    V3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
    V3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
    x.dpdu = SXfVector(sp.o2w, dpdu);
    x.dpdv = x.dpdvs = SXfVector(sp.o2w, dpdv);
    x.dndus = SXfNormal(sp.w2o, dndu);
    x.dndvs = SXfNormal(sp.w2o, dndv);
    return x;
}

// the offset rays of PerspectiveCamera::GenerateRayDifferential (cameras/perspective.cpp:117-139), through CameraToWorld
// (core/transform.h:396-405) and ScaleDifferentials(1 / sqrt(spp)) (core/geometry.h:908-913, integrator.cpp:285-286).
// (o, d) = the camera ray as raygen stored it.
struct RayDiffT { V3 rxO, ryO, rxD, ryD; };
__device__ __noinline__ RayDiffT CameraDifferentials(const mi_camera *cam, Float pFilmX, Float pFilmY, Float l0, Float l1, int spp, const V3 o, const V3 d) {
    RayDiffT r;
    V3 pCamera = XfPointT(cam->raster_to_camera, V3(pFilmX, pFilmY, 0));
    V3 dxC = v3(cam->dx_camera), dyC = v3(cam->dy_camera);
    if (cam->lens_radius > 0) {
        Float ddx, ddy;
        ConcentricSampleDisk(l0, l1, &ddx, &ddy);
        Float lx = cam->lens_radius * ddx, ly = cam->lens_radius * ddy;
        V3 dx = Normalize(pCamera + dxC);
        Float ft = cam->focal_distance / dx.z;
        V3 pFocus = V3(0, 0, 0) + (ft * dx);
        r.rxO = V3(lx, ly, 0);
        r.rxD = Normalize(pFocus - r.rxO);
        V3 dy = Normalize(pCamera + dyC);
        ft = cam->focal_distance / dy.z;
        pFocus = V3(0, 0, 0) + (ft * dy);
        r.ryO = V3(lx, ly, 0);
        r.ryD = Normalize(pFocus - r.ryO);
    } else {
        r.rxO = r.ryO = V3(0, 0, 0);
        r.rxD = Normalize(pCamera + dxC);
        r.ryD = Normalize(pCamera + dyC);
    }
    const float *m = cam->camera_to_world;
    r.rxO = XfPointT(m, r.rxO); r.ryO = XfPointT(m, r.ryO);
    r.rxD = XfVectorT(m, r.rxD); r.ryD = XfVectorT(m, r.ryD);
    Float sc = 1 / sqrtf_((Float)spp);
    r.rxO = o + (r.rxO - o) * sc;
    r.ryO = o + (r.ryO - o) * sc;
    r.rxD = d + (r.rxD - d) * sc;
    r.ryD = d + (r.ryD - d) * sc;
    return r;
}

PT_DEV bool SolveLinearSystem2x2(Float a00, Float a01, Float a10, Float a11, Float B0, Float B1, Float *x0, Float *x1) {   // core/transform.cpp:41-49
    Float det = a00 * a11 - a01 * a10;
    if (absf(det) < 1e-10f) return false;
    *x0 = (a11 * B0 - a01 * B1) / det;
    *x1 = (a00 * B1 - a10 * B0) / det;
    if (__builtin_isnan(*x0) || __builtin_isnan(*x1)) return false;
    return true;
}
// SurfaceInteraction::ComputeDifferentials core/interaction.cpp:104-153 (has == ray.hasDifferentials)
__device__ __noinline__ void ComputeDifferentials(const V3 p, const V3 n, IsectX *x, const RayDiffT rd) {
    Float d = Dot(n, V3(p.x, p.y, p.z));
    Float tx = -(Dot(n, rd.rxO) - d) / Dot(n, rd.rxD);
    bool ok = !(__builtin_isinf(tx) || __builtin_isnan(tx));
    Float ty = 0;
    if (ok) {
        ty = -(Dot(n, rd.ryO) - d) / Dot(n, rd.ryD);
        ok = !(__builtin_isinf(ty) || __builtin_isnan(ty));
    }
    if (!ok) return;   // the caller zeroed the differentials
    V3 px = rd.rxO + tx * rd.rxD;
    V3 py = rd.ryO + ty * rd.ryD;
    x->dpdx = px - p;
    x->dpdy = py - p;
    int d0, d1;
    if (absf(n.x) > absf(n.y) && absf(n.x) > absf(n.z)) { d0 = 1; d1 = 2; }
    else if (absf(n.y) > absf(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    Float a00 = x->dpdu[d0], a01 = x->dpdv[d0], a10 = x->dpdu[d1], a11 = x->dpdv[d1];
    Float Bx0 = px[d0] - p[d0], Bx1 = px[d1] - p[d1], By0 = py[d0] - p[d0], By1 = py[d1] - p[d1];
    if (!SolveLinearSystem2x2(a00, a01, a10, a11, Bx0, Bx1, &x->dudx, &x->dvdx)) x->dudx = x->dvdx = 0;
    if (!SolveLinearSystem2x2(a00, a01, a10, a11, By0, By1, &x->dudy, &x->dvdy)) x->dudy = x->dvdy = 0;
}
PT_DEV TexCtx TexCtxOf(const Isect &si, const IsectX &x) {
    TexCtx c;
    c.p = si.p; c.u = x.u; c.v = x.v; c.dpdx = x.dpdx; c.dpdy = x.dpdy;
    c.dudx = x.dudx; c.dvdx = x.dvdx; c.dudy = x.dudy; c.dvdy = x.dvdy;
    return c;
}
// Material::Bump core/material.cpp:46-83
template <bool U> __device__ __noinline__ void BumpT(int tex, Isect *si, IsectX *x) {
    TexCtx ev = TexCtxOf(*si, *x);
    Float du = .5f * (absf(x->dudx) + absf(x->dudy));
    if (du == 0) du = .0005f;
    ev.p = si->p + du * si->dpdus;
    ev.u = x->u + du; ev.v = x->v + 0.f;
    Float uDisplace = TexEval<U>(tex, ev).r;
    Float dv = .5f * (absf(x->dvdx) + absf(x->dvdy));
    if (dv == 0) dv = .0005f;
    ev.p = si->p + dv * x->dpdvs;
    ev.u = x->u + 0.f; ev.v = x->v + dv;
    Float vDisplace = TexEval<U>(tex, ev).r;
    Float displace = TexEval<U>(tex, TexCtxOf(*si, *x)).r;
    V3 dpdu = si->dpdus + (uDisplace - displace) / du * si->ns + displace * x->dndus;
    V3 dpdv = x->dpdvs + (vDisplace - displace) / dv * si->ns + displace * x->dndvs;
    V3 sn = Normalize(Cross(dpdu, dpdv));   // SetShadingGeometry(..., false) core/interaction.cpp:73-92
    if (x->flipN) sn = -sn;
    si->ns = Faceforward(sn, si->n);
    si->dpdus = dpdu; x->dpdvs = dpdv;
}

// Triangle::Intersect / IntersectP alpha tests (shapes/triangle.cpp:333-338, 532-570): alphaMask for every ray,
// shadowAlphaMask for IntersectP only; the local interaction has no differentials (point-sampled / level-0 lookups)
// (round 5: through the per-mesh DevMaskFast records -- constants, `dots` and image maps over (u, v) are answered without the node / program tables)
PT_DEV Float MaskValue(const DevMaskFast &m, int node, const TexCtx &tc) {
    if (m.kind == 2) return m.v_out;
    if (m.kind == 3) return DotsInside(m.su * tc.u + m.du, m.sv * tc.v + m.dv) ? m.v_in : m.v_out;   // UVMapping2D texture.cpp:86-94 + dots.h:59-80
    if (m.kind == 4) {                                                                                // imagemap.h:87-94 with zero differentials
        if (m.image < 0 || (uint32_t)m.image >= c_tex.n_images) return 0;
        const V2 st{m.su * tc.u + m.du, m.sv * tc.v + m.dv};
        return MipLookup<false>(c_tex.images + m.image, st, V2{m.su * tc.dudx, m.sv * tc.dvdx}, V2{m.su * tc.dudy, m.sv * tc.dvdy}).r;
    }
    return TexEval(node, tc).r;
}
__device__ __noinline__ bool TriAlphaRejects(const uint4 *tri_info, const TriShade *tri_shade, uint32_t prim, const V3 p0, const V3 p1, const V3 p2, Float b0,
                                             Float b1, Float b2, bool anyHit) {
    uint32_t mesh = tri_info[prim].w;
    const DevMaskFast ma = c_tex.mask_fast[2 * mesh];
    DevMaskFast ms = c_tex.mask_fast[2 * mesh + 1];
    if (!anyHit) ms.kind = 0;
    if (ma.kind == 0 && ms.kind == 0) return false;
    TriShadeRegs tsr = LoadTriShade(tri_shade, prim);
    TexCtx tc;
    tc.p = b0 * p0 + b1 * p1 + b2 * p2;
    tc.u = b0 * tsr.c.y + b1 * tsr.c.w + b2 * tsr.d.y;
    tc.v = b0 * tsr.c.z + b1 * tsr.d.x + b2 * tsr.d.z;
    tc.dpdx = tc.dpdy = V3(0, 0, 0);
    tc.dudx = tc.dvdx = tc.dudy = tc.dvdy = 0;
    if (ma.kind != 0 && MaskValue(ma, ma.kind == 1 ? c_tex.mesh_alpha[2 * mesh] : -1, tc) == 0) return true;
    if (ms.kind != 0 && MaskValue(ms, ms.kind == 1 ? c_tex.mesh_alpha[2 * mesh + 1] : -1, tc) == 0) return true;
    return false;
}

PT_DEV Float RoughnessToAlphaT(Float roughness) {   // core/microfacet.h:123-128
    roughness = mx(roughness, (Float)1e-3);
    Float x = logf_(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
PT_DEV RGB ClampRGB(const RGB &v) { return RGB(clampf(v.r, 0, PT_INFINITY), clampf(v.g, 0, PT_INFINITY), clampf(v.b, 0, PT_INFINITY)); }
PT_DEV void Set3(float *d, const RGB &c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; }
PT_DEV mi_bxdf *AddLobe(mi_material *m, int type) {
    int i = m->n_bxdfs < MI_MAX_BXDFS ? m->n_bxdfs++ : MI_MAX_BXDFS - 1;
    mi_bxdf *b = &m->bxdfs[i];
#if defined(PT_HOST_EMU) && PT_HOST_EMU
    //  emulator builds poison the record: a reader of a word its lobe kind never set sees a signalling pattern (NaN as a float, a huge index as an
    // integer) instead of whatever the stack held, so the emulator's fixture tests fail on it -- the device build leaves the unread words alone
    __builtin_memset(b, 0xff, sizeof(*b));
#endif
    // the header words every consumer reads; the parameters a lobe kind reads are written by the code that adds it, the others are never looked at (a ScaledBxDF's
    // scale is set where `scaled` is).  Rounds 1-4 zero-filled all 25 words of the record in private memory per lobe and hit.
    b->type = type; b->fresnel = 0; b->scaled = 0; b->distrib = 0;
    return b;
}
PT_DEV void AddMicroR(mi_material *m, const RGB &r, Float ax, Float ay, int fresnel, Float etaI, Float etaT) {
    mi_bxdf *b = AddLobe(m, MI_BXDF_MICROFACET_R);
    Set3(b->R, r); b->alphax = ax; b->alphay = ay; b->fresnel = fresnel; b->etaA = etaI; b->etaB = etaT;
}
PT_DEV void AddMicroT(mi_material *m, const RGB &t, Float ax, Float ay, Float etaA, Float etaB) {
    mi_bxdf *b = AddLobe(m, MI_BXDF_MICROFACET_T);
    Set3(b->T, t); b->alphax = ax; b->alphay = ay; b->etaA = etaA; b->etaB = etaB; b->fresnel = MI_FRESNEL_DIELECTRIC;
}
PT_DEV void AddSpecR(mi_material *m, const RGB &r, int fresnel, Float etaI, Float etaT) {
    mi_bxdf *b = AddLobe(m, MI_BXDF_SPECULAR_R);
    Set3(b->R, r); b->fresnel = fresnel; b->etaA = etaI; b->etaB = etaT;
}
PT_DEV void AddSpecT(mi_material *m, const RGB &t, Float etaA, Float etaB) {
    mi_bxdf *b = AddLobe(m, MI_BXDF_SPECULAR_T);
    Set3(b->T, t); b->etaA = etaA; b->etaB = etaB; b->fresnel = MI_FRESNEL_DIELECTRIC;
}
template <bool U> PT_DEV void CopyMaterial(mi_material *dst, const mi_material *src) {
    const typename UPtr<U, uint32_t>::P s = UPtr<U, uint32_t>::of(reinterpret_cast<const uint32_t *>(src));   // word 0 = n_bxdfs
    uint32_t *d = (uint32_t *)dst;
    int words = 2 + (int)s[0] * (int)(sizeof(mi_bxdf) / 4);
    for (int k = 0; k < words; ++k) d[k] = s[k];
}

// Material::ComputeScatteringFunctions(si, arena, TransportMode::Radiance, allowMultipleLobes = true) of material `mat`
// into *out (n_bxdfs = 0, eta = 1 on entry).  D bounds the nesting of mix materials.
template <int D, bool U> struct MaterialEvalD { static __device__ __noinline__ void eval(const mi_material *materials, int mat, Isect *si, IsectX *x, mi_material *out); };
template <bool U> struct MaterialEvalD<0, U> { static PT_DEV void eval(const mi_material *, int, Isect *, IsectX *, mi_material *) {} };
template <int D, bool U> __device__ __noinline__ void MaterialEvalD<D, U>::eval(const mi_material *materials, int mat, Isect *si, IsectX *x, mi_material *out) {
    mat = UIdx<U>(mat);
    const typename UPtr<U, mi_material_desc>::P md = UPtr<U, mi_material_desc>::of(c_tex.descs + mat);
    if (!md->textured) { CopyMaterial<U>(out, materials + mat); return; }
    if (md->type == MI_MAT_MIX) {   // mixmat.cpp:45-64
        RGB s1 = ClampRGB(TexEval<U>(md->amount, TexCtxOf(*si, *x)));
        RGB s2 = ClampRGB(RGB(1.f) - s1);
        Isect si2 = *si;
        IsectX x2 = *x;
        mi_material l2;
        l2.n_bxdfs = 0; l2.eta = 1;
        MaterialEvalD<D - 1, U>::eval(materials, md->m1, si, x, out);
        MaterialEvalD<D - 1, U>::eval(materials, md->m2, &si2, &x2, &l2);
        for (int i = 0; i < out->n_bxdfs; ++i) {
            mi_bxdf *b = &out->bxdfs[i];
            if (b->scaled) { b->scale[0] = s1.r * b->scale[0]; b->scale[1] = s1.g * b->scale[1]; b->scale[2] = s1.b * b->scale[2]; }
            else { b->scaled = 1; Set3(b->scale, s1); }
        }
        for (int i = 0; i < l2.n_bxdfs && out->n_bxdfs < MI_MAX_BXDFS; ++i) {
            mi_bxdf *b = &out->bxdfs[out->n_bxdfs++];
            *b = l2.bxdfs[i];
            if (b->scaled) { b->scale[0] = s2.r * b->scale[0]; b->scale[1] = s2.g * b->scale[1]; b->scale[2] = s2.b * b->scale[2]; }
            else { b->scaled = 1; Set3(b->scale, s2); }
        }
        return;
    }
    PROBE(14)   // (profiler builds) k_shade<TEX>: up to here = differentials + the descriptor fetch
    if (md->bump >= 0) BumpT<U>(md->bump, si, x);
    PROBE(21)   // Material::Bump (three evaluations of the displacement texture)
    const TexCtx tc = TexCtxOf(*si, *x);
    const bool remap = md->remap_roughness != 0;
    switch (md->type) {
    case MI_MAT_MATTE: {   // matte.cpp:45-62
        RGB r = ClampRGB(TexEval<U>(md->Kd, tc));
        Float sig = clampf(TexEval<U>(md->sigma, tc).r, 0, 90);
        if (!r.IsBlack()) {
            if (sig == 0) Set3(AddLobe(out, MI_BXDF_LAMBERT_R)->R, r);
            else {   // OrenNayar ctor reflection.h:414-420
                mi_bxdf *b = AddLobe(out, MI_BXDF_OREN_NAYAR);
                Set3(b->R, r);
                Float sigma = sig * (PT_PI / 180);
                Float sigma2 = sigma * sigma;
                b->A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                b->B = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
        }
        break;
    }
    case MI_MAT_PLASTIC: {   // plastic.cpp:45-70
        RGB kd = ClampRGB(TexEval<U>(md->Kd, tc));
        if (!kd.IsBlack()) Set3(AddLobe(out, MI_BXDF_LAMBERT_R)->R, kd);
        RGB ks = ClampRGB(TexEval<U>(md->Ks, tc));
        if (!ks.IsBlack()) {
            Float rough = TexEval<U>(md->roughness, tc).r;
            if (remap) rough = RoughnessToAlphaT(rough);
            AddMicroR(out, ks, rough, rough, MI_FRESNEL_DIELECTRIC, 1.5f, 1.f);
        }
        break;
    }
    case MI_MAT_GLASS: {   // glass.cpp:45-92
        Float eta = TexEval<U>(md->eta_f, tc).r, urough = TexEval<U>(md->uroughness, tc).r, vrough = TexEval<U>(md->vroughness, tc).r;
        RGB R = ClampRGB(TexEval<U>(md->Kr, tc)), T = ClampRGB(TexEval<U>(md->Kt, tc));
        out->eta = eta;
        if (R.IsBlack() && T.IsBlack()) break;
        if (urough == 0 && vrough == 0) {
            mi_bxdf *b = AddLobe(out, MI_BXDF_FRESNEL_SPEC);
            Set3(b->R, R); Set3(b->T, T); b->etaA = 1.f; b->etaB = eta;
        } else {
            if (remap) { urough = RoughnessToAlphaT(urough); vrough = RoughnessToAlphaT(vrough); }
            if (!R.IsBlack()) AddMicroR(out, R, urough, vrough, MI_FRESNEL_DIELECTRIC, 1.f, eta);
            if (!T.IsBlack()) AddMicroT(out, T, urough, vrough, 1.f, eta);
        }
        break;
    }
    case MI_MAT_MIRROR: {   // mirror.cpp:45-56
        RGB R = ClampRGB(TexEval<U>(md->Kr, tc));
        if (!R.IsBlack()) AddSpecR(out, R, MI_FRESNEL_NOOP, 1, 1);
        break;
    }
    case MI_MAT_METAL: {   // metal.cpp:59-80
        Float uRough = md->uroughness >= 0 ? TexEval<U>(md->uroughness, tc).r : TexEval<U>(md->roughness, tc).r;
        Float vRough = md->vroughness >= 0 ? TexEval<U>(md->vroughness, tc).r : TexEval<U>(md->roughness, tc).r;
        if (remap) { uRough = RoughnessToAlphaT(uRough); vRough = RoughnessToAlphaT(vRough); }
        RGB eta = TexEval<U>(md->eta_s, tc), k = TexEval<U>(md->k_s, tc);
        AddMicroR(out, RGB(1.f), uRough, vRough, MI_FRESNEL_CONDUCTOR, 1.f, 1.f);
        mi_bxdf *b = &out->bxdfs[out->n_bxdfs - 1];
        Set3(b->eta_c, eta); Set3(b->k_c, k);
        break;
    }
    case MI_MAT_UBER: {   // uber.cpp:45-101
        Float e = TexEval<U>(md->eta_f, tc).r;
        RGB op = ClampRGB(TexEval<U>(md->opacity, tc));
        RGB t = ClampRGB(RGB(0.f) - op + RGB(1.f));
        if (!t.IsBlack()) { out->eta = 1.f; AddSpecT(out, t, 1.f, 1.f); }
        else out->eta = e;
        RGB kd = op * ClampRGB(TexEval<U>(md->Kd, tc));
        if (!kd.IsBlack()) Set3(AddLobe(out, MI_BXDF_LAMBERT_R)->R, kd);
        RGB ks = op * ClampRGB(TexEval<U>(md->Ks, tc));
        if (!ks.IsBlack()) {
            Float roughu = md->uroughness >= 0 ? TexEval<U>(md->uroughness, tc).r : TexEval<U>(md->roughness, tc).r;
            Float roughv = md->vroughness >= 0 ? TexEval<U>(md->vroughness, tc).r : roughu;
            if (remap) { roughu = RoughnessToAlphaT(roughu); roughv = RoughnessToAlphaT(roughv); }
            AddMicroR(out, ks, roughu, roughv, MI_FRESNEL_DIELECTRIC, 1.f, e);
        }
        RGB kr = op * ClampRGB(TexEval<U>(md->Kr, tc));
        if (!kr.IsBlack()) AddSpecR(out, kr, MI_FRESNEL_DIELECTRIC, 1.f, e);
        RGB kt = op * ClampRGB(TexEval<U>(md->Kt, tc));
        if (!kt.IsBlack()) AddSpecT(out, kt, 1.f, e);
        break;
    }
    case MI_MAT_SUBSTRATE: {   // substrate.cpp:45-65
        RGB dd = ClampRGB(TexEval<U>(md->Kd, tc)), ss = ClampRGB(TexEval<U>(md->Ks, tc));
        Float roughu = TexEval<U>(md->uroughness, tc).r, roughv = TexEval<U>(md->vroughness, tc).r;
        if (!dd.IsBlack() || !ss.IsBlack()) {
            if (remap) { roughu = RoughnessToAlphaT(roughu); roughv = RoughnessToAlphaT(roughv); }
            mi_bxdf *b = AddLobe(out, MI_BXDF_FRESNEL_BLEND);
            Set3(b->R, dd); Set3(b->T, ss); b->alphax = roughu; b->alphay = roughv;
        }
        break;
    }
    case MI_MAT_TRANSLUCENT: {   // translucent.cpp:45-80
        const Float eta = 1.5f;
        out->eta = eta;
        RGB r = ClampRGB(TexEval<U>(md->reflect, tc)), t = ClampRGB(TexEval<U>(md->transmit, tc));
        if (r.IsBlack() && t.IsBlack()) break;
        RGB kd = ClampRGB(TexEval<U>(md->Kd, tc));
        if (!kd.IsBlack()) {
            if (!r.IsBlack()) Set3(AddLobe(out, MI_BXDF_LAMBERT_R)->R, r * kd);
            if (!t.IsBlack()) Set3(AddLobe(out, MI_BXDF_LAMBERT_T)->T, t * kd);
        }
        RGB ks = ClampRGB(TexEval<U>(md->Ks, tc));
        if (!ks.IsBlack() && (!r.IsBlack() || !t.IsBlack())) {
            Float rough = TexEval<U>(md->roughness, tc).r;
            if (remap) rough = RoughnessToAlphaT(rough);
            if (!r.IsBlack()) AddMicroR(out, r * ks, rough, rough, MI_FRESNEL_DIELECTRIC, 1.f, eta);
            if (!t.IsBlack()) AddMicroT(out, t * ks, rough, rough, 1.f, eta);
        }
        break;
    }
    }
}
#define PT_MIX_MAX_DEPTH 3
// U: `mat` is the same in every active lane (k_shade's material waterfall; see UPtr in pt_texture.h)
template <bool U = false> PT_DEV void ComputeScatteringFunctionsT(const mi_material *materials, int mat, Isect *si, IsectX *x, mi_material *out) {
    out->n_bxdfs = 0; out->eta = 1;
    MaterialEvalD<PT_MIX_MAX_DEPTH, U>::eval(materials, mat, si, x, out);
}

// ------------------------------------------------------------------ two-level instancing, shading side
// The ray a TransformedPrimitive hands to its object: Inverse(PrimitiveToWorld)(r) (transform.h:252-264), as EnterInstance computes it
struct InstRay { V3 o, d; };
__device__ __noinline__ InstRay InstanceRay(const DevInstance *in, const V3 ro, const V3 rd) {
    V3 oErr;
    InstRay r;
    r.o = SXfPointErr(in->w2i, ro, &oErr);
    r.d = SXfVector(in->w2i, rd);
    Float lengthSquared = r.d.LengthSquared();
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(r.d), oErr) / lengthSquared;
        r.o = r.o + r.d * dt;
    }
    return r;
}
// Transform::operator()(const SurfaceInteraction &) core/transform.cpp:262-297 with PrimitiveToWorld
__device__ __noinline__ void InstanceToWorld(const DevInstance *in, Isect *is, IsectX *ix) {
    V3 pe;
    is->p = SXfPointErr2(in->i2w, is->p, is->pError, &pe);
    is->pError = pe;
    is->n = Normalize(SXfNormal(in->w2i, is->n));
    is->wo = Normalize(SXfVector(in->i2w, is->wo));
    is->ns = Normalize(SXfNormal(in->w2i, is->ns));
    is->dpdus = SXfVector(in->i2w, is->dpdus);
    ix->dpdu = SXfVector(in->i2w, ix->dpdu); ix->dpdv = SXfVector(in->i2w, ix->dpdv);
    ix->dpdvs = SXfVector(in->i2w, ix->dpdvs);
    ix->dndus = SXfNormal(in->w2i, ix->dndus); ix->dndvs = SXfNormal(in->w2i, ix->dndvs);
    is->ns = Faceforward(is->ns, is->n);
}
