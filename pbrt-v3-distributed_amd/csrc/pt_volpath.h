// k_shade_vol -- the shading kernel of scenes rendered with Integrator "volpath" (VolPathIntegrator::Li, integrators/volpath.cpp:55-190)
// and of scenes whose materials carry a BSSRDF (the subsurface branch, integrators/path.cpp:153-174 / volpath.cpp:153-180).
//
// Place in the wavefront pipeline: path-extension rays still go through k_trace<0> and the material sort -- that is where the time
// goes -- and this kernel replaces k_shade.  What differs from k_shade is WHERE the secondary rays are traced: the reference draws
// sampler dimensions inside its visibility code (ratio tracking in GridDensityMedium::Tr, core/light.cpp:63-82, core/scene.cpp:56-70), BEFORE the
// continuation direction is sampled, and the count is data dependent; and a BSSRDF vertex traces a chain of probe rays before its entry vertex
// is shaded.  Forms of the kernel (mi_scene_upload chooses):
//   general (WAVE = false): every lane walks the general BVH4 steps the traversal kernels use (TravNodeStep / TravLeafStep with spheres, alpha
//     masks and instances) for its own transmittance, MIS and probe rays, on an LDS stack of its own -- the rays are resolved in the
//     reference's sequence by construction.  Every combination has a wavefront form by now; this one is the
//     A/B partner of the other forms (PBRT_AMD_VOL_INLINE=1, PBRT_AMD_VOL_TR_QUEUES=0, PBRT_AMD_VOL_SPLIT=0).
//   wavefront (WAVE = true): the direct-lighting rays go through the shadow / MIS queues (NeeOut) --
//     * homogeneous media only, no interfaces or masks: k_trace<2> / <1> as in k_shade, closed-form transmittance folded into the terms;
//     * BSDF-less interfaces or alpha masks (DevVol::tr_queues): the rays are WALKED segment by segment, k_trace<..., TR> + k_vol_tr_step;
//     * a grid medium (DevVol::tr_dims, the split form): the walk draws its ratio-tracking dimensions from the path's sampler, and a vertex
//       with direct-lighting rays is shaded in two stages around it (this kernel up to the light sample, k_vol_continue for the rest);
//     * BSSRDF materials (DevVol::sss_wave): the path parks at the subsurface vertex, its probe chain is walked through the queues
//       (k_sss_probe_step) and k_sss_entry shades the entry vertex; the direct-lighting rays of both vertices take the plain traversals or the
//       walk, and with a grid medium both vertices are split like any other (k_vol_continue finishes the subsurface vertex AND the entry vertex).
// Included by pbrt_amd.hip after PathState / ChunkIter / DynIter / wave_append.
#pragma once
#include "pt_volume.h"

#ifndef PT_VOL_LANE_QN
#define PT_VOL_LANE_QN 1
#endif
struct LaneTracer {
    const DevScene *scp;
    LdsStackEntry *lds;
    StackEntry *spill;
    uint32_t nClosest, nAny, guardTrips;
};
// one ray, traced by this lane alone: closest hit (BVHAccel::Intersect) or any hit (IntersectP).  INST: two-level scenes (TransformedPrimitive
// leaves, see TravStateI); `inst` = the instance the hit primitive was reached through (TRAV_NO_INSTANCE: a top-level primitive)
struct LaneHit { uint32_t prim; Float t; uint32_t inst; };
template <bool ANY, bool INST>
__device__ __noinline__ LaneHit TraceLane(LaneTracer *lt, const V3 o, const V3 d, Float tMax) {
    const DevScene &sc = *lt->scp;
    TravStack st;
    st.lds = lt->lds; st.spill = lt->spill;
#if PT_VOL_LANE_QN
    if constexpr (!INST) {
        if (sc.nodesq) {   // single-level scenes: the 64-byte quantised nodes (four requests per interior step instead of seven; same hits, pt_bvh4q.h)
            TravStateQ tq;
            tq.init(sc, o, d, tMax, st);
            TraceCounters tcq = {0, 0, 0};
            uint32_t stepsq = 0;
            while (!tq.done()) {
                if (++stepsq > (1u << 22)) { ++lt->guardTrips; tq.prim = TRAV_MISS; break; }
                if (tq.atNode()) TravNodeStepQ<false>(sc, tq, st, &tcq);
                else TravLeafStep<ANY, false, true, true, TravStateQ, TravStack, false>(sc, tq, st, &tcq);
            }
            if (ANY) ++lt->nAny; else ++lt->nClosest;
            LaneHit hq;
            hq.prim = tq.prim; hq.t = tq.tHit; hq.inst = TRAV_NO_INSTANCE;
            return hq;
        }
    }
#endif
    typename std::conditional<INST, TravStateI, TravState>::type ts;
    ts.init(sc, o, d, tMax, st);
    TraceCounters tc = {0, 0, 0};
    uint32_t steps = 0;
    while (!ts.done()) {
        if (++steps > (1u << 22)) { ++lt->guardTrips; ts.prim = TRAV_MISS; break; }   // non-termination guard, reported through MI_CNT_TRACE_GUARD_TRIPS
        if (ts.atNode()) TravNodeStep<false>(sc, ts, st, &tc);
        else TravLeafStep<ANY, false, true, true, typename std::conditional<INST, TravStateI, TravState>::type, TravStack, INST>(sc, ts, st, &tc);
    }
    if (ANY) ++lt->nAny; else ++lt->nClosest;
    LaneHit h;
    h.prim = ts.prim; h.t = ts.tHit; h.inst = TRAV_NO_INSTANCE;
    if constexpr (INST) h.inst = ts.hitInst;
    return h;
}

// SurfaceInteraction of a hit + the medium interface GeometricPrimitive::Intersect leaves on it (core/primitive.cpp:122-126)
struct VHit {
    Isect is;
    IsectX ix;
    uint4 tinfo;
    int mIn, mOut;
};
// inst: the instance the primitive was reached through (two-level scenes; TRAV_NO_INSTANCE otherwise): the interaction is built in the
// object's space from the transformed ray and carried back to world space (TransformedPrimitive::Intersect, core/primitive.cpp:76-111)
__device__ __noinline__ void HitToIsect(const DevScene *scp, const DevVol *vol, uint32_t prim, V3 o, V3 d, uint32_t inst, int rayMedium, bool wantTex, VHit *out) {
    const DevScene &sc = *scp;
    const DevInstance *hitInst = nullptr;
    if (inst != TRAV_NO_INSTANCE) {
        hitInst = c_instances + inst;
        InstRay ir = InstanceRay(hitInst, o, d);
        o = ir.o; d = ir.d;
    }
    uint4 tinfo;
    TriShadeRegs tsr;
    V3 p0, p1, p2;
    uint32_t tf;
    LoadHitTriangle(sc, prim, &p0, &p1, &p2, &tf, &tsr, &tinfo);
    out->ix = IsectX();
    if (tf & TRI_FLAG_SPHERE) {
        out->is = SphereIsectToIsect(sc.spheres + __float_as_uint(p0.x), o, d, prim);
        if (wantTex) out->ix = SphereIsectTex(sc.spheres + __float_as_uint(p0.x), o, d);
    }
    else {
        TriHit th;
        TriangleTest(p0, p1, p2, o, d, PT_INFINITY, &th);   // same code, same inputs as the traversal: same b0, b1, b2, t
        out->is = MakeIsect(BuildIsectPre(tinfo.x, tsr, p0, p1, p2, V3(th.b0, th.b1, th.b2)), d, prim);
        if (wantTex) out->ix = BuildIsectTex(tinfo.x, tsr, p0, p1, p2, V3(th.b0, th.b1, th.b2));
    }
    if (hitInst && !hitInst->identity) InstanceToWorld(hitInst, &out->is, &out->ix);
    out->tinfo = tinfo;
    int in = -1, outm = -1;
    if (vol->mesh_medium) { in = vol->mesh_medium[2 * tinfo.w]; outm = vol->mesh_medium[2 * tinfo.w + 1]; }
    if (in != outm) { out->mIn = in; out->mOut = outm; }   // MediumInterface::IsMediumTransition
    else out->mIn = out->mOut = rayMedium;
}
PT_DEV int GetMediumOf(const V3 &n, int mIn, int mOut, const V3 &w) { return Dot(w, n) > 0 ? mOut : mIn; }   // Interaction::GetMedium(w) interaction.h:80-82

// the point Light::Sample_Li put into the VisibilityTester (p1): only needed when a transmittance ray has to be re-spawned
// behind a BSDF-less interface (VisibilityTester::Tr, core/light.cpp:63-82), so it is re-derived there instead of carried along
struct LightPoint { V3 p, pError, n; };
__device__ __noinline__ LightPoint LightPointOf(const DevScene *scp, const DevLight *dl, const V3 refP, const V3 refPError, const V3 refN, Float u0, Float u1, const V3 wi) {
    LightPoint lp;
    lp.pError = V3(); lp.n = V3();
    const int type = dl->type;
    if (type == MI_LIGHT_AREA_TRI) {   // Triangle::Sample(u) shapes/triangle.cpp:583-608, as SampleLi evaluates it
        Float su0 = sqrtf_(u0);
        Float b0 = 1 - su0, b1 = u1 * su0;
        V3 p0 = v3(dl->p0), p1 = v3(dl->p1), p2 = v3(dl->p2);
        lp.p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        V3 n = Normalize(Cross(p1 - p0, p2 - p0));
        if (dl->mesh_flags & MI_MESH_HAS_N) {
            TriShadeRegs tsr = LoadTriShade(scp->tri_shade, (uint32_t)dl->tri);
            V3 ns = b0 * tsr.n0() + b1 * tsr.n1() + (1 - b0 - b1) * tsr.n2();
            n = Faceforward(n, ns);
        } else if (dl->mesh_flags & MI_MESH_FLIP)
            n = n * -1.f;
        lp.n = n;
        lp.pError = gamma_n(6) * (Abs(b0 * p0) + Abs(b1 * p1) + Abs((1 - b0 - b1) * p2));
    } else if (type == MI_LIGHT_AREA_SPHERE) {
        SphereSample ss;
        SphereSampleRef((const mi_sphere *)dl->ext, refP, refPError, refN, u0, u1, &ss);
        lp.p = ss.p; lp.pError = ss.pError; lp.n = ss.n;
    } else if (type == MI_LIGHT_POINT || type == MI_LIGHT_SPOT)
        lp.p = v3(dl->pos);
    else   // distant / infinite: Interaction(ref.p + wi * (2 * worldRadius))
        lp.p = refP + wi * (2 * dl->world_radius);
    return lp;
}

// Wavefront form of the two direct-lighting rays (k_shade_vol<true>): scenes whose media are all homogeneous and that have no BSSRDF materials
// draw no sampler dimension inside their visibility code.  Without BSDF-less interfaces and alpha masks the first surface a shadow / MIS ray
// meets ends it -- so the rays go through the shadow and MIS queues to k_trace<2> / k_trace<1> as in k_shade, with the closed-form
// transmittance folded into the shadow term here and applied over the hit distance by k_trace<1> (PathState::vol_tr); with them the rays are
// walked segment by segment (DevVol::tr_queues: k_trace<..., TR> + k_vol_tr_step below).
struct NeeOut {
    bool wantShadow, wantMis;
    ShadowRay sh;
    RGB shTerm;            // f Li Tr weight / lightPdf
    V3 miO, miD;
    RGB miTerm, miSigmaT;  // f weight / scatteringPdf; sigma_t of the medium the MIS ray travels in (0: vacuum)
    int lightNum;
    Float selPdf;
    LightPoint lp;         // DevVol::tr_queues: the sampled point on the light (VisibilityTester::p1) and the media the two rays start in
    int shMedium, miMedium;
};
struct VolCtx {
    const DevScene *scp;
    const DevVol *vol;
    LaneTracer *lt;
    VSampler *smp;
    NeeOut *nee;   // null: the lane traces its own rays
};

// VisibilityTester::Tr (core/light.cpp:63-82) with media, VisibilityTester::Unoccluded (:59-61) without
template <bool INST>
__device__ __noinline__ RGB VisibilityTrD(VolCtx cx, V3 o, V3 d, Float tMax, int medium, const DevLight *dl, const V3 refP, const V3 refPError, const V3 refN, Float u0,
                                          Float u1, const V3 wi, const LightPoint *stored = nullptr) {
    if (!cx.vol->handle_media) {
        LaneHit h = TraceLane<true, INST>(cx.lt, o, d, tMax);
        return h.prim == TRAV_MISS ? RGB(1.f) : RGB(0.f);
    }
    RGB Tr(1.f);
    for (int guard = 0; guard < 65536; ++guard) {
        LaneHit h = TraceLane<false, INST>(cx.lt, o, d, tMax);
        const bool hitSurface = h.prim != TRAV_MISS;
        if (hitSurface && (int)cx.scp->tri_info[h.prim].y >= 0) return RGB(0.f);
        if (medium >= 0) Tr = Tr * MediumTr(cx.scp, cx.vol->media + medium, o, d, hitSurface ? h.t : tMax, cx.smp);
        if (!hitSurface) break;
        VHit vh;
        HitToIsect(cx.scp, cx.vol, h.prim, o, d, h.inst, medium, false, &vh);
        LightPoint lp = stored ? *stored : LightPointOf(cx.scp, dl, refP, refPError, refN, u0, u1, wi);
        ShadowRay sr = SpawnRayTo(vh.is, lp.p, lp.pError, lp.n);   // isect.SpawnRayTo(p1)
        o = sr.o; d = sr.d; tMax = sr.tMax;
        medium = GetMediumOf(vh.is.n, vh.mIn, vh.mOut, d);
    }
    return Tr;
}
// Scene::IntersectTr (core/scene.cpp:56-70) with media, Scene::Intersect without.  *oOut: origin of the ray segment that found the hit
template <bool INST>
__device__ __noinline__ LaneHit IntersectTrD(VolCtx cx, V3 o, const V3 d, int medium, RGB *Tr, V3 *oOut) {
    *Tr = RGB(1.f);
    if (!cx.vol->handle_media) {
        LaneHit h = TraceLane<false, INST>(cx.lt, o, d, PT_INFINITY);
        *oOut = o;
        return h;
    }
    for (int guard = 0; guard < 65536; ++guard) {
        LaneHit h = TraceLane<false, INST>(cx.lt, o, d, PT_INFINITY);
        const bool hitSurface = h.prim != TRAV_MISS;
        if (medium >= 0) *Tr = *Tr * MediumTr(cx.scp, cx.vol->media + medium, o, d, hitSurface ? h.t : PT_INFINITY, cx.smp);
        *oOut = o;
        if (!hitSurface) return h;
        if ((int)cx.scp->tri_info[h.prim].y >= 0) return h;
        VHit vh;
        HitToIsect(cx.scp, cx.vol, h.prim, o, d, h.inst, medium, false, &vh);
        o = OffsetRayOrigin(vh.is.p, vh.is.pError, vh.is.n, d);   // isect.SpawnRay(ray.d)
        medium = GetMediumOf(vh.is.n, vh.mIn, vh.mOut, d);
    }
    LaneHit miss;
    miss.prim = TRAV_MISS; miss.t = 0; miss.inst = TRAV_NO_INSTANCE;
    return miss;
}

typedef BSDF_T<false> LaneBSDF;
// EstimateDirect (core/integrator.cpp:108-215); bsdf == nullptr: `it` is a MediumInteraction with phase function HenyeyGreenstein(g)
template <bool INST, class BS>
__device__ __noinline__ RGB EstimateDirectD(VolCtx cx, const Isect *itp, int mIn, int mOut, const BS *bsdf, Float g, Float uS0, Float uS1, int lightNum, Float uL0,
                                            Float uL1) {
    const DevScene &sc = *cx.scp;
    const Isect &it = *itp;
    const DevLight *light = sc.lights + lightNum;
    const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    RGB Ld(0.f);
    LightSample ls = SampleLiAny(GeomTables(sc), light, it.p, it.pError, it.n, uL0, uL1);
    Float lightPdf = ls.pdf, scatteringPdf = 0;
    if (lightPdf > 0 && !ls.Li.IsBlack()) {
        RGB f;
        if (bsdf) {
            f = bsdf->f(it.wo, ls.wi, bsdfFlags) * AbsDot(ls.wi, it.ns);
            scatteringPdf = bsdf->Pdf(it.wo, ls.wi, bsdfFlags);
        } else {
            Float p = HGp(g, it.wo, ls.wi);
            f = RGB(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack()) {
            const int shMedium = GetMediumOf(it.n, mIn, mOut, ls.shadow.d);
            RGB Li;
            if (cx.nee) {   // unoccluded transmittance in closed form (HomogeneousMedium::Tr); occlusion is k_trace<2>'s answer
                Li = ls.Li;
                if (!cx.vol->tr_queues && shMedium >= 0) Li = Li * ExpRGB(-rgb3(cx.vol->media[shMedium].sigma_t) * mn(ls.shadow.tMax * ls.shadow.d.Length(), PT_MAX_FLOAT));
                // tr_queues: the segment may cross BSDF-less interfaces into other media -- k_vol_tr<2> multiplies the transmittance in, interface by interface
            } else
                Li = ls.Li * VisibilityTrD<INST>(cx, ls.shadow.o, ls.shadow.d, ls.shadow.tMax, shMedium, light, it.p, it.pError, it.n, uL0, uL1, ls.wi);
            if (cx.nee) {
                if (!Li.IsBlack()) {
                    cx.nee->wantShadow = true;
                    cx.nee->sh = ls.shadow;
                    if (cx.vol->tr_queues) { cx.nee->lp = LightPointOf(cx.scp, light, it.p, it.pError, it.n, uL0, uL1, ls.wi); cx.nee->shMedium = shMedium; }
                    cx.nee->shTerm = ls.delta ? f * Li / lightPdf : f * Li * PowerHeuristic(lightPdf, scatteringPdf) / lightPdf;
                }
            } else if (!Li.IsBlack()) {
                if (ls.delta) Ld = Ld + f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(lightPdf, scatteringPdf);
                    Ld = Ld + f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!ls.delta) {
        RGB f;
        V3 wi;
        bool sampledSpecular = false;
        if (bsdf) {
            int sampledType;
            f = bsdf->Sample_f(it.wo, &wi, uS0, uS1, &scatteringPdf, bsdfFlags, &sampledType);
            f = f * AbsDot(wi, it.ns);
            sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        } else {
            Float p = HGSample_p(g, it.wo, &wi, uS0, uS1);
            f = RGB(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = PdfLiAny(GeomTables(sc), light, it.p, it.pError, it.n, wi);
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(scatteringPdf, lightPdf);
            }
            V3 ro = OffsetRayOrigin(it.p, it.pError, it.n, wi), segO;   // it.SpawnRay(wi)
            if (cx.nee) {
                const int miMedium = GetMediumOf(it.n, mIn, mOut, wi);
                cx.nee->wantMis = true;
                cx.nee->miO = ro; cx.nee->miD = wi;
                cx.nee->miTerm = f * weight / scatteringPdf;
                cx.nee->miSigmaT = miMedium >= 0 ? rgb3(cx.vol->media[miMedium].sigma_t) : RGB(0.f);
                cx.nee->miMedium = miMedium;
                cx.nee->lightNum = lightNum;
                return Ld;
            }
            RGB Tr(1.f);
            const LaneHit lh = IntersectTrD<INST>(cx, ro, wi, GetMediumOf(it.n, mIn, mOut, wi), &Tr, &segO);
            const uint32_t prim = lh.prim;
            RGB Li(0.f);
            if (prim != TRAV_MISS) {
                if ((int)sc.tri_info[prim].z == lightNum) {   // lightIsect.primitive->GetAreaLight() == &light
                    VHit vh;
                    HitToIsect(cx.scp, cx.vol, prim, segO, wi, lh.inst, -1, false, &vh);
                    Li = AreaL(*light, vh.is.n, -wi);   // lightIsect.Le(-wi)
                }
            } else if (light->type == MI_LIGHT_INFINITE)
                Li = InfiniteLe(light, wi);
            if (!Li.IsBlack()) Ld = Ld + f * Li * Tr * weight / scatteringPdf;
        }
    }
    return Ld;
}

// UniformSampleOneLight (core/integrator.cpp:85-106) over the light distribution looked up at it.p (path.cpp:125-127 / volpath.cpp:96,125)
template <bool INST, class BS>
__device__ __noinline__ RGB UniformSampleOneLightD(VolCtx cx, const Isect *itp, int mIn, int mOut, const BS *bsdf, Float g) {
    const DevScene &sc = *cx.scp;
    if (sc.n_lights == 0) return RGB(0.f);
    Float funcInt = sc.light_func_int;
    const bool spatial = sc.light_strategy == MI_LIGHT_STRATEGY_SPATIAL;
    size_t vox = 0;
    if (spatial) {   // SpatialLightDistribution::Lookup lightdistrib.cpp:139-152; Bounds3::Offset geometry.h:786-792
        V3 bmin = v3(sc.sp_bmin), bmax = v3(sc.sp_bmax);
        V3 off = itp->p - bmin;
        if (bmax.x > bmin.x) off.x /= bmax.x - bmin.x;
        if (bmax.y > bmin.y) off.y /= bmax.y - bmin.y;
        if (bmax.z > bmin.z) off.z /= bmax.z - bmin.z;
        int v0 = (int)(off.x * sc.sp_nvox[0]), v1 = (int)(off.y * sc.sp_nvox[1]), v2 = (int)(off.z * sc.sp_nvox[2]);
        v0 = v0 < 0 ? 0 : (v0 > sc.sp_nvox[0] - 1 ? sc.sp_nvox[0] - 1 : v0);
        v1 = v1 < 0 ? 0 : (v1 > sc.sp_nvox[1] - 1 ? sc.sp_nvox[1] - 1 : v1);
        v2 = v2 < 0 ? 0 : (v2 > sc.sp_nvox[2] - 1 ? sc.sp_nvox[2] - 1 : v2);
        vox = ((size_t)v0 * sc.sp_nvox[1] + v1) * sc.sp_nvox[2] + v2;
        funcInt = sc.sp_func_int[vox];
    }
    Float ul = cx.smp->Get1D(sc);
    int lightNum;
    Float funcAt;
    if (spatial) SpatialPick(sc, vox, ul, &lightNum, &funcAt);   // Distribution1D::SampleDiscrete core/sampling.h:90-100
    else {
        const int size = (int)sc.n_lights + 1, first = CdfCountLE(sc.light_cdf, size, ul);
        lightNum = first - 1 < 0 ? 0 : (first - 1 > size - 2 ? size - 2 : first - 1);
        funcAt = sc.light_func[lightNum];
    }
    Float selPdf = (funcInt > 0) ? funcAt / (funcInt * (int)sc.n_lights) : 0;
    if (selPdf == 0) return RGB(0.f);
    Float uL0, uL1, uS0, uS1;
    cx.smp->Get2D(sc, &uL0, &uL1);
    cx.smp->Get2D(sc, &uS0, &uS1);
    if (cx.nee) cx.nee->selPdf = selPdf;
    return EstimateDirectD<INST, BS>(cx, itp, mIn, mOut, bsdf, g, uS0, uS1, lightNum, uL0, uL1) / selPdf;
}

// si->bssrdf of SubsurfaceMaterial / KdSubsurfaceMaterial::ComputeScatteringFunctions (materials/subsurface.cpp:95-99, kdsubsurface.cpp:87-93),
// evaluated on the interaction the BSDF evaluation left behind (bump-mapped shading frame)
__device__ __noinline__ void ComputeBSSRDFD(const DevVol *vol, int mat, const Isect *si, const IsectX *ix, DevBSSRDF *out) {
    out->table = nullptr;
    if (!vol->bssrdf) return;
    // MixMaterial (mixmat.cpp:45-64) evaluates m1 on *si itself: si->bssrdf is then m1's, carrying m1 as its material -- which no primitive of
    // the mix has, so its probe rays accept nothing and the path ends at the first transmission (the reference's behaviour, kept)
    for (int guard = 0; guard < 64 && c_tex.descs && c_tex.descs[mat].type == MI_MAT_MIX && c_tex.descs[mat].textured; ++guard) mat = c_tex.descs[mat].m1;
    const mi_bssrdf_desc b = vol->bssrdf[mat];
    if (b.kind == MI_BSSRDF_NONE) return;
    const DevBssrdfTable *tb = vol->tables + b.table;
    RGB sig_a, sig_s;
    TexCtx tc = TexCtxOf(*si, *ix);
    if (b.kind == MI_BSSRDF_SUBSURFACE) {
        sig_a = b.scale * ClampRGB(TexEval(b.sigma_a, tc));
        sig_s = b.scale * ClampRGB(TexEval(b.sigma_s, tc));
    } else {   // SubsurfaceFromDiffuse core/bssrdf.cpp:178-188
        RGB mfree = b.scale * ClampRGB(TexEval(b.mfp, tc));
        RGB kd = ClampRGB(TexEval(b.Kd, tc));
        Float rr = InvertCatmullRom(tb->n_rho, tb->rho_samples, tb->rho_eff, kd.r);
        Float rg = InvertCatmullRom(tb->n_rho, tb->rho_samples, tb->rho_eff, kd.g);
        Float rb = InvertCatmullRom(tb->n_rho, tb->rho_samples, tb->rho_eff, kd.b);
        sig_s = RGB(rr / mfree.r, rg / mfree.g, rb / mfree.b);
        sig_a = RGB((1 - rr) / mfree.r, (1 - rg) / mfree.g, (1 - rb) / mfree.b);
    }
    out->table = tb;
    out->poP = si->p; out->ns = si->ns; out->ss = Normalize(si->dpdus); out->ts = Cross(out->ns, out->ss);
    out->eta = b.eta; out->material = mat;
    out->sigma_t = sig_a + sig_s;
    out->rho = RGB(out->sigma_t.r != 0 ? sig_s.r / out->sigma_t.r : 0, out->sigma_t.g != 0 ? sig_s.g / out->sigma_t.g : 0, out->sigma_t.b != 0 ? sig_s.b / out->sigma_t.b : 0);
}
// the probe segment of Sample_Sp (bssrdf.cpp:249-283): projection axis and spectral channel from u1, radius and angle from u2, the segment
// baseP -> pTarget through the sphere of radius rMax; false: the sample yields nothing (r < 0 or r >= rMax).  *u1Left: u1 after the two remaps
// (it picks among the chain's hits).  One function for the per-lane form and the walked form (k_shade_vol<WAVE> + k_sss_probe_step): same arithmetic
struct SssProbe { V3 baseP, pTarget; Float u1Left; };
PT_DEV bool BssrdfProbeSetup(const DevBSSRDF *bs, Float u1, Float u20, Float u21, SssProbe *pr) {
    V3 vx, vy, vz;
    if (u1 < .5f) { vx = bs->ss; vy = bs->ts; vz = bs->ns; u1 *= 2; }
    else if (u1 < .75f) { vx = bs->ts; vy = bs->ns; vz = bs->ss; u1 = (u1 - .5f) * 4; }
    else { vx = bs->ns; vy = bs->ss; vz = bs->ts; u1 = (u1 - .75f) * 4; }
    int ch = (int)(u1 * 3);
    ch = ch < 0 ? 0 : (ch > 2 ? 2 : ch);
    u1 = u1 * 3 - ch;
    Float r = BssrdfSample_Sr(bs, ch, u20);
    if (r < 0) return false;
    Float phi = 2 * PT_PI * u21;
    Float rMax = BssrdfSample_Sr(bs, ch, 0.999f);
    if (r >= rMax) return false;
    Float l = 2 * sqrtf_(rMax * rMax - r * r);
    pr->baseP = bs->poP + r * (vx * cosf_(phi) + vy * sinf_(phi)) - l * vz * 0.5f;
    pr->pTarget = pr->baseP + l * vz;
    pr->u1Left = u1;
    return true;
}
// SeparableBSSRDF::Sample_Sp (core/bssrdf.cpp:249-326): a probe segment through the sphere of radius rMax around po; every hit on a primitive
// of the same material object counts, one of them is chosen.  The reference collects the chain in a list; here the chain is walked twice
// (count, then stop at the chosen one) -- the same rays, hence the same hits.
template <bool INST>
__device__ __noinline__ RGB BssrdfSample_Sp(VolCtx cx, const DevBSSRDF *bs, Float u1, Float u20, Float u21, VHit *pi, Float *pdf) {
    SssProbe pr;
    if (!BssrdfProbeSetup(bs, u1, u20, u21, &pr)) return RGB(0.f);
    const V3 baseP = pr.baseP, pTarget = pr.pTarget;
    u1 = pr.u1Left;
    int nFound = 0, selected = 0;
    for (int pass = 0; pass < 2; ++pass) {
        V3 p = baseP, pErr, n;   // a plain Interaction: no normal, no error bounds, no media
        int mIn = -1, mOut = -1, seen = 0;
        for (int guard = 0; guard < 65536; ++guard) {
            V3 dir = pTarget - p;
            if (dir.x == 0 && dir.y == 0 && dir.z == 0) break;
            V3 origin = OffsetRayOrigin(p, pErr, n, dir);   // Interaction::SpawnRayTo(const Point3f &) interaction.h:68-72
            LaneHit h = TraceLane<false, INST>(cx.lt, origin, dir, 1 - PT_SHADOW_EPS);
            if (h.prim == TRAV_MISS) break;
            VHit vh;
            HitToIsect(cx.scp, cx.vol, h.prim, origin, dir, h.inst, GetMediumOf(n, mIn, mOut, dir), false, &vh);
            p = vh.is.p; pErr = vh.is.pError; n = vh.is.n; mIn = vh.mIn; mOut = vh.mOut;
            if ((int)vh.tinfo.y == bs->material) {
                if (pass == 1 && seen == selected) { *pi = vh; break; }
                ++seen;
            }
        }
        if (pass == 0) {
            nFound = seen;
            if (nFound == 0) return RGB(0.0f);
            selected = (int)(u1 * nFound);
            selected = selected < 0 ? 0 : (selected > nFound - 1 ? nFound - 1 : selected);
        }
    }
    *pdf = BssrdfPdf_Sp(bs, pi->is.p, pi->is.n) / nFound;
    return BssrdfSr(bs, (bs->poP - pi->is.p).Length());   // Sp(pi) = Sr(Distance(po.p, pi.p))
}

// ---- BSSRDF probe chains in wavefront form (DevVol::sss_wave: Integrator "path" scenes with subsurface materials; round 3).
// Sample_S draws its three numbers before the chain is traced and nothing while it is traced (bssrdf.cpp:249-326), so the dimension stream of a path
// does not depend on the chain: the vertex's own direct-lighting rays go through the shadow / MIS queues as in every wavefront form, the path PARKS
// (SssRec: the BSSRDF at po, the probe segment, the chain's state), k_sss_probe_step + k_trace<2, ..., TR> walk the chain hit by hit -- count the hits on
// the material object, choose; a second walk up to the chosen hit (BssrdfSample_Sp's two walks) only if it is not among the first few, which are kept --
// and k_sss_entry shades the entry vertex pi (Sp, pdf, the
// adapter lobe's light sample and continuation, Russian roulette).  Same rays, same arithmetic as the per-lane form (PBRT_AMD_VOL_INLINE=1).
#ifndef PT_SSS_KEEP
#define PT_SSS_KEEP 3
#endif
struct __attribute__((aligned(16))) SssRec {
    float4 po_eta;       // po.p | eta
    float4 ns_mat;       // po's shading normal | material slot (bits)
    float4 ss_tab;       // po's shading tangent | table index (bits)
    float4 sigt_u1;      // sigma_t | u1 after the axis / channel remaps (picks among the hits)
    float4 rho_found;    // rho | hits of the first walk (bits)
    float4 base_seen;    // baseP | hits seen in this walk, bit 31: second walk (bits)
    float4 target_sel;   // pTarget | index of the chosen hit (bits)
    float4 p_ex;         // the chain's current point p | pError.x
    float4 e_n;          // pError.y, pError.z, n.x, n.y
    float4 nz_pi;        // n.z | pi: primitive, instance (bits) | the chain point's media: (mIn + 1) | (mOut + 1) << 16 (bits)
    float4 pi_o, pi_d;   // the segment that found pi
    // the first PT_SSS_KEEP counted hits of the first walk (segment origin | primitive, segment direction | its medium, instance): when the chosen hit is
    // among them -- a probe through a closed object has two -- the second walk is not needed (the reference keeps the whole chain in a list)
    float4 keep_o[PT_SSS_KEEP], keep_d[PT_SSS_KEEP];
    uint32_t keep_inst[4];
};
PT_DEV void SssPark(SssRec *S, const DevVol &vol, const DevBSSRDF &b, const SssProbe &pr) {
    S->po_eta = make_float4(b.poP.x, b.poP.y, b.poP.z, b.eta);
    S->ns_mat = make_float4(b.ns.x, b.ns.y, b.ns.z, __uint_as_float((uint32_t)b.material));
    S->ss_tab = make_float4(b.ss.x, b.ss.y, b.ss.z, __uint_as_float((uint32_t)(b.table - vol.tables)));
    S->sigt_u1 = make_float4(b.sigma_t.r, b.sigma_t.g, b.sigma_t.b, pr.u1Left);
    S->rho_found = make_float4(b.rho.r, b.rho.g, b.rho.b, __uint_as_float(0u));
    S->base_seen = make_float4(pr.baseP.x, pr.baseP.y, pr.baseP.z, __uint_as_float(0u));
    S->target_sel = make_float4(pr.pTarget.x, pr.pTarget.y, pr.pTarget.z, __uint_as_float(0u));
}
PT_DEV void SssBssrdfOf(const SssRec *S, const DevVol &vol, DevBSSRDF *b) {
    const float4 a = S->po_eta, n = S->ns_mat, t = S->ss_tab, g = S->sigt_u1, r = S->rho_found;
    b->poP = V3(a.x, a.y, a.z); b->eta = a.w;
    b->ns = V3(n.x, n.y, n.z); b->material = (int)__float_as_uint(n.w);
    b->ss = V3(t.x, t.y, t.z); b->table = vol.tables + __float_as_uint(t.w);
    b->ts = Cross(b->ns, b->ss);   // as ComputeBSSRDFD formed it
    b->sigma_t = RGB(g.x, g.y, g.z); b->rho = RGB(r.x, r.y, r.z);
}

// the two direct-lighting rays of a vertex into its NeeRec (wavefront forms): the terms are added by k_trace<2> iff unoccluded / by k_trace<1> times
// Le times the transmittance up to the hit (or by k_vol_tr_step at the end of a walk)
PT_DEV void WriteNeeRecords(const PathState &ps, const DevVol &vol, uint32_t slot, const NeeOut &nee, const RGB &betaNee) {
    if (nee.wantShadow) {
        RGB c = betaNee * (nee.shTerm / nee.selPdf);
        ps.nee[slot].sh_o = make_float4(nee.sh.o.x, nee.sh.o.y, nee.sh.o.z, nee.sh.tMax);
        ps.nee[slot].sh_d = make_float4(nee.sh.d.x, nee.sh.d.y, nee.sh.d.z, 0);
        ps.nee[slot].sh_c = make_float4(c.r, c.g, c.b, vol.tr_queues ? nee.lp.n.z : 0);
        if (vol.tr_queues) {   // the light point + start medium ride in the record's free words (k_vol_tr_step<2>)
            ps.trs[slot].acc[0] = make_float4(1, 1, 1, 0);
            ps.nee[slot].sh_d.w = __int_as_float(nee.shMedium);
            ps.nee[slot].pad[0] = make_float4(nee.lp.p.x, nee.lp.p.y, nee.lp.p.z, nee.lp.pError.x);
            ps.nee[slot].pad[1] = make_float4(nee.lp.pError.y, nee.lp.pError.z, nee.lp.n.x, nee.lp.n.y);
        }
    }
    if (nee.wantMis) {
        RGB c = betaNee * (nee.miTerm / nee.selPdf);
        ps.nee[slot].mi_o = make_float4(nee.miO.x, nee.miO.y, nee.miO.z, vol.tr_queues ? __int_as_float(nee.miMedium) : 0);
        ps.nee[slot].mi_d = make_float4(nee.miD.x, nee.miD.y, nee.miD.z, __uint_as_float((uint32_t)nee.lightNum));
        ps.nee[slot].mi_c = make_float4(c.r, c.g, c.b, 0);
        if (vol.tr_queues) ps.trs[slot].acc[1] = make_float4(1, 1, 1, 0);
        if (!vol.tr_queues) ps.nee[slot].pad[0] = make_float4(nee.miSigmaT.r, nee.miSigmaT.g, nee.miSigmaT.b, 0);
    }
}

// The out-of-line routines this kernel shares with k_shade<..., TEX> (BSDF, texture and light code) are compiled ONCE, with the loosest
// register bound of their callers: a lower occupancy target here would take the textured shading kernel from 168 to 244 VGPRs (measured).
#ifndef PT_VOL_SHADE_WAVES
#define PT_VOL_SHADE_WAVES PT_TEX_SHADE_WAVES
#endif
// one path vertex per lane (material-sorted queue, as k_shade): Medium::Sample on the segment, then either the medium interaction or the
// surface interaction of VolPathIntegrator::Li's loop body; PathIntegrator::Li's body when vol.handle_media == 0 (scenes with a BSSRDF)
// WAVE: the direct-lighting rays go through the shadow / MIS queues (see NeeOut) instead of being traced by the lane
// INST: two-level scenes (the hit primitive may have been reached through an instance, PathRec::pad0)
// UMAT: no material of the scene is textured -- lanes of a wave share a material almost always (sorted queue), and the surface branch runs once per
// distinct material of the wave with a WAVE-UNIFORM material pointer (lobe lists through scalar loads, lobe switches as scalar branches), as in k_shade
template <bool WAVE, bool INST, bool UMAT>
__global__ void __launch_bounds__(PT_BLOCK, PT_VOL_SHADE_WAVES) k_shade_vol(const DevScene *scp, PathState ps, DevVol vol, uint32_t qout) {
    __shared__ StackEntry lds_stack[WAVE ? 1 : PT_LDS_STACK * PT_BLOCK];
    NoiseLdsInit();
    const DevScene &sc = *scp;
    LaneTracer lt;
    lt.scp = scp;
    lt.lds = (LdsStackEntry *)&lds_stack[WAVE ? 0 : threadIdx.x];
    lt.spill = reinterpret_cast<StackEntry *>(ps.spill) + (size_t)(blockIdx.x * PT_BLOCK + threadIdx.x) * ps.spill_per_thread;
    lt.nClosest = lt.nAny = lt.guardTrips = 0;
    uint32_t n, sbase;
    ShadeRange(ps, &sbase, &n);
    uint32_t nseg = 0;
    for (DynIter it(n, ps.cursor); it.more(); it.next()) {
        const uint32_t i = it.item();
        const bool active = it.valid();
        bool cont = false, wantProbe = false;   // wantProbe (walked BSSRDF probes, DevVol::sss_wave): the path waits for its probe chain instead of continuing
        int wantSplit = 0;                      // split form (DevVol::tr_dims): 1 = a surface vertex, 2 = a medium vertex waits for its direct-lighting rays' transmittances (k_vol_continue)
        V3 splitP;
        uint32_t slot = 0;
        NeeOut nee;
        nee.wantShadow = nee.wantMis = false;
        if (active) {
            slot = ps.q_sorted[sbase + i];
            const uint2 hr = ps.rec[slot].hit;
            const float4 o4 = ps.rec[slot].ray_o, d4 = ps.rec[slot].ray_d, b4 = ps.rec[slot].beta, L4 = ps.rec[slot].L;
            const uint4 s4 = ps.rec[slot].smp;
            const V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            RGB beta(b4.x, b4.y, b4.z), L(L4.x, L4.y, L4.z);
            Float etaScale = b4.w;
            int bounces = (int)(s4.w & 0xffffu);
            bool specularBounce = (s4.w >> 16) & 1u;
            bool noDiff = (s4.w >> 17) & 1u;
            int medium = vol.handle_media ? (int)__float_as_uint(ps.rec[slot].pad2.x) : -1;
            VSampler smp;
            smp.index = (uint64_t)s4.x | ((uint64_t)s4.y << 32);
            smp.dimension = (int)s4.z;
            smp.px = smp.py = 0;   // only dimensions 0 / 1 (camera sample) look at the pixel
            smp.Prefetch(sc);
            VolCtx cx;
            cx.scp = scp; cx.vol = &vol; cx.lt = &lt; cx.smp = &smp; cx.nee = WAVE ? &nee : nullptr;
            ++nseg;
            const bool found = hr.x != MISS_PRIM;
            V3 no = ro, nd = rd;   // the next ray
            int nmedium = medium;
            bool alive = true, scattered = false, nullCrossing = false;
            RGB betaNee(0.f);   // beta at the vertex's light sample
            // volpath.cpp:82-83: if (ray.medium) beta *= ray.medium->Sample(ray, sampler, arena, &mi)
            MediumSampleOut ms;
            ms.valid = false;
            if (medium >= 0) {
                ms = MediumSample(scp, vol.media + medium, ro, rd, found ? __uint_as_float(hr.y) : o4.w, &smp);
                beta = beta * ms.w;
            }
            if (beta.IsBlack()) alive = false;
            if (alive && ms.valid) {   // volpath.cpp:87-105: scattering inside the medium
                if (bounces >= sc.max_depth) alive = false;
                else {
                    Isect mi;   // the MediumInteraction as an Interaction: no normal, no error bounds, the same medium on both sides
                    mi.p = ms.p; mi.pError = V3(); mi.n = V3(); mi.ns = V3(); mi.dpdus = V3(); mi.wo = -rd; mi.prim = MISS_PRIM;
                    const Float g = vol.media[medium].g;
                    L = L + beta * UniformSampleOneLightD<INST, LaneBSDF>(cx, &mi, medium, medium, nullptr, g);
                    betaNee = beta;
                    if (WAVE && vol.tr_dims && (nee.wantShadow || nee.wantMis)) {   // split form: the visibility queries draw dimensions -- k_vol_continue samples the phase function after them
                        wantSplit = 2; splitP = mi.p;
                    } else {
                    Float u0, u1;
                    smp.Get2D(sc, &u0, &u1);
                    V3 wi;
                    HGSample_p(g, -rd, &wi, u0, u1);
                    no = OffsetRayOrigin(mi.p, mi.pError, mi.n, wi);   // mi.SpawnRay(wi)
                    nd = wi;
                    specularBounce = false;
                    scattered = true;
                    }
                }
            } else if (alive) {
                VHit vh;
                if (found) HitToIsect(scp, &vol, hr.x, ro, rd, INST ? ps.rec[slot].pad0 : TRAV_NO_INSTANCE, medium, vol.textured != 0, &vh);
                if (bounces == 0 || specularBounce) {   // volpath.cpp:110-116 / path.cpp:91-101
                    if (found) {
                        int li = (int)vh.tinfo.z;
                        if (li >= 0) L = L + beta * AreaL(sc.lights[li], vh.is.n, -rd);
                    } else
                        for (uint32_t k = 0; k < sc.n_infinite; ++k) L = L + beta * InfiniteLe(&sc.lights[sc.infinite_lights[k]], rd);
                }
                if (!found || bounces >= sc.max_depth) alive = false;
                else {
                    const int matIdx = (int)vh.tinfo.y;
                    if (matIdx < 0) {   // a surface without a BSDF (medium boundary): step through, same bounce count (volpath.cpp:117-121 / path.cpp:108-113)
                        no = OffsetRayOrigin(vh.is.p, vh.is.pError, vh.is.n, rd);
                        nmedium = GetMediumOf(vh.is.n, vh.mIn, vh.mOut, rd);
                        nullCrossing = true;
                        noDiff = true;
                    } else {
                        typedef typename std::conditional<UMAT, BSDF_T<true>, LaneBSDF>::type SurfBSDF;
                        bool matTodo = true;
                        while (matTodo) {   // material waterfall (one trip for per-lane lobe lists)
                        const int matU = UMAT ? UniformInt(matIdx) : matIdx;
                        if (UMAT ? SameAs(matIdx, matU) : true) {
                        matTodo = false;
                        mi_material laneMat;
                        if (!UMAT && vol.textured) {   // isect.ComputeScatteringFunctions(ray, arena, true): differentials of camera rays, then the material
                            if (bounces == 0 && !noDiff) {
                                float2 pf = ps.rec[slot].pfilm, ln = ps.rec[slot].lens;
                                RayDiffT rdf = CameraDifferentials(&c_tex.camera, pf.x, pf.y, ln.x, ln.y, c_tex.spp, ro, rd);
                                ComputeDifferentials(vh.is.p, vh.is.n, &vh.ix, rdf);
                            }
                            if constexpr (!UMAT) ComputeScatteringFunctionsT(sc.materials, matIdx, &vh.is, &vh.ix, &laneMat);
                        }
                        const mi_material *matPtr = (!UMAT && vol.textured) ? &laneMat : sc.materials + matU;   // constant lobe lists are read in place
                        DevBSSRDF bssrdf;
                        bssrdf.table = nullptr;
                        // (si->bssrdf is a pure function of the interaction: evaluated below, only when the sampled lobe transmits -- the coefficient textures and the three
                        //  spline inversions of a kdsubsurface material are wasted on every other vertex)
                        SurfBSDF bsdf(vh.is, matPtr);
                        // volpath.cpp:125-128 samples a light unconditionally; path.cpp:122 only for surfaces with a non-specular lobe
                        if (vol.handle_media || bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) L = L + beta * UniformSampleOneLightD<INST, SurfBSDF>(cx, &vh.is, vh.mIn, vh.mOut, &bsdf, 0);
                        betaNee = beta;
                        V3 wo = -rd, wi;
                        Float pdf = 0, u0, u1;
                        int flags = 0;
                        RGB f(0.f);
                        // split form (DevVol::tr_dims: a grid medium's Tr draws sampler dimensions, core/light.cpp:63-82, media/grid.cpp:89-118): the continuation is
                        // sampled by k_vol_continue AFTER this vertex's shadow and MIS rays have been walked
                        const bool split = WAVE && vol.tr_dims && (nee.wantShadow || nee.wantMis);
                        if (split) wantSplit = 1;
                        else {
                            smp.Get2D(sc, &u0, &u1);
                            f = bsdf.Sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                        }
                        if (split) {
                        } else if (f.IsBlack() || pdf == 0.f) alive = false;
                        else {
                            beta = beta * (f * AbsDot(wi, vh.is.ns) / pdf);
                            specularBounce = (flags & BSDF_SPECULAR) != 0;
                            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                                Float eta = bsdf.m->eta;
                                etaScale *= (Dot(wo, vh.is.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                            }
                            no = OffsetRayOrigin(vh.is.p, vh.is.pError, vh.is.n, wi);   // isect.SpawnRay(wi)
                            nd = wi;
                            nmedium = GetMediumOf(vh.is.n, vh.mIn, vh.mOut, wi);
                            scattered = true;
                            if constexpr (!UMAT) { if (vol.bssrdf && (flags & BSDF_TRANSMISSION)) ComputeBSSRDFD(&vol, matIdx, &vh.is, &vh.ix, &bssrdf); }
                            if (!UMAT && bssrdf.table && (flags & BSDF_TRANSMISSION)) {   // path.cpp:153-174 / volpath.cpp:153-180
                                // S = bssrdf->Sample_S(scene, sampler.Get1D(), sampler.Get2D(), ...): the two calls are function ARGUMENTS and
                                // g++ evaluates them right to left -- the 2-D sample takes the earlier dimensions (pinned by the oracle's fixtures)
                                Float u20, u21, u1s, spdf = 0;
                                smp.Get2D(sc, &u20, &u21);
                                u1s = smp.Get1D(sc);
                                VHit pi;
                                RGB S(0.f);
                                if constexpr (WAVE) {
                                    // walked form (DevVol::sss_wave): the probe chain goes through the queues (k_sss_probe_step + k_trace<2, ..., TR>) and the
                                    // entry vertex is shaded by k_sss_entry, which also takes this iteration's Russian roulette and ++bounces
                                    SssProbe pr;
                                    if (!BssrdfProbeSetup(&bssrdf, u1s, u20, u21, &pr)) alive = false;
                                    else { wantProbe = true; SssPark(&ps.sss[slot], vol, bssrdf, pr); }
                                } else
                                    S = BssrdfSample_Sp<INST>(cx, &bssrdf, u1s, u20, u21, &pi, &spdf);
                                if constexpr (WAVE) {   // (k_sss_entry goes on from here)
                                } else if (S.IsBlack() || spdf == 0) alive = false;
                                else {
                                    beta = beta * (S / spdf);
                                    // the entry vertex (bssrdf.cpp:235-247): a BSDF of the adapter lobe alone on pi's own shading frame, wo = shading.n
                                    mi_material piMat;
                                    piMat.n_bxdfs = 1; piMat.eta = 1;
                                    __builtin_memset(&piMat.bxdfs[0], 0, sizeof(mi_bxdf));
                                    piMat.bxdfs[0].type = MI_BXDF_BSSRDF_ADAPTER;
                                    piMat.bxdfs[0].etaB = bssrdf.eta;
                                    pi.is.wo = pi.is.ns;
                                    LaneBSDF piBsdf(pi.is, &piMat);
                                    L = L + beta * UniformSampleOneLightD<INST, LaneBSDF>(cx, &pi.is, pi.mIn, pi.mOut, &piBsdf, 0);
                                    smp.Get2D(sc, &u0, &u1);
                                    f = piBsdf.Sample_f(pi.is.wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                                    if (f.IsBlack() || pdf == 0) alive = false;
                                    else {
                                        beta = beta * (f * AbsDot(wi, pi.is.ns) / pdf);
                                        specularBounce = (flags & BSDF_SPECULAR) != 0;
                                        no = OffsetRayOrigin(pi.is.p, pi.is.pError, pi.is.n, wi);   // pi.SpawnRay(wi)
                                        nd = wi;
                                        nmedium = GetMediumOf(pi.is.n, pi.mIn, pi.mOut, wi);
                                    }
                                }
                            }
                        }
                        }   // matIdx == matU
                        }   // material waterfall
                    }
                }
            }
            if (!alive) wantSplit = 0;
            if (alive && (wantProbe || wantSplit)) {   // parked: beta, etaScale, the sampler state and the bounce count as they are now (k_sss_entry / k_vol_continue resume)
                ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17) | ((uint32_t)(wantSplit == 2) << 18));
                if (wantSplit == 2) ps.rec[slot].pad2 = make_float4(__uint_as_float((uint32_t)medium), splitP.x, splitP.y, splitP.z);
            } else if (alive && scattered) {   // Russian roulette (volpath.cpp:183-189 / path.cpp:176-184); the loop's ++bounces
                cont = true;
                RGB rrBeta = beta * etaScale;
                if (rrBeta.MaxComponentValue() < sc.rr_threshold && bounces > 3) {
                    Float q = mx((Float).05, 1 - rrBeta.MaxComponentValue());
                    if (smp.Get1D(sc) < q) cont = false;
                    else beta = beta / (1 - q);
                }
                ++bounces;
            } else if (alive && nullCrossing)
                cont = true;
            ps.rec[slot].L = make_float4(L.r, L.g, L.b, 0);
            if (WAVE) WriteNeeRecords(ps, vol, slot, nee, betaNee);
            if (cont) {
                ps.rec[slot].ray_o = make_float4(no.x, no.y, no.z, PT_INFINITY);
                ps.rec[slot].ray_d = make_float4(nd.x, nd.y, nd.z, 0);
                ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17));
                if (vol.handle_media) ps.rec[slot].pad2 = make_float4(__uint_as_float((uint32_t)nmedium), 0, 0, 0);
            }
        }
        const uint32_t qseg = blockIdx.x & 7, qbase = qseg * ps.seg_cap;
        if (WAVE) {
            uint32_t posE, posS, posM;
            PT_WAVE_APPEND3(&ps.qcount[QCI(qout, qseg)], &ps.qcount[QCI(QC_SHADOW, qseg)], &ps.qcount[QCI(QC_MIS, qseg)], cont, nee.wantShadow, nee.wantMis, &posE, &posS, &posM);
            if (cont) { ps.q_ext[qout][qbase + posE] = slot; }
            if (nee.wantShadow) ps.q_shadow[qbase + posS] = slot;
            if (nee.wantMis) ps.q_mis[qbase + posM] = slot;
            if (vol.tr_dims) {   // split form: the vertices that wait for their transmittances (k_vol_continue's queue)
                const uint32_t posB = wave_append(&ps.qcount[QCI(QC_CONT, qseg)], wantSplit != 0);
                if (wantSplit) ps.q_cont[qbase + posB] = slot;
            }
            if constexpr (!UMAT) {
                if (vol.sss_wave) {   // the probe chains of this bounce: first queue of the walk (k_sss_probe_step)
                    const uint32_t posP = wave_append(&ps.qcount[QCI(QC_PROBE0, qseg)], wantProbe);
                    if (wantProbe) ps.q_probe[0][qbase + posP] = slot;
                }
            }
        } else {
            uint32_t posE = wave_append(&ps.qcount[QCI(qout, qseg)], cont);
            if (cont) { ps.q_ext[qout][qbase + posE] = slot; }
        }
    }
    wave_count(&ps.counters[MI_CNT_PATH_SEGMENTS], nseg);
    wave_count(&ps.counters[MI_CNT_CLOSEST_RAYS], lt.nClosest);
    wave_count(&ps.counters[MI_CNT_SHADOW_RAYS], lt.nAny);
    if (lt.guardTrips) atomicAdd(&ps.counters[MI_CNT_TRACE_GUARD_TRIPS], (unsigned long long)lt.guardTrips);
}
// Camera::medium onto the camera rays of a pass (Camera::GenerateRayDifferential sets ray->medium = medium, cameras/perspective.cpp:203)
__global__ void __launch_bounds__(PT_BLOCK) k_vol_camera_medium(PathState ps, uint32_t n, int32_t medium) {
    for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += gridDim.x * PT_BLOCK) ps.rec[i].pad2 = make_float4(__uint_as_float((uint32_t)medium), 0, 0, 0);
}

// ---- the direct-lighting rays of the wavefront form when BSDF-less interfaces separate homogeneous media (DevVol::tr_queues; round 3).
// k_trace<2> / k_trace<1> end a ray at the first surface; with interfaces a surface without a BSDF has to be stepped through instead
// (VisibilityTester::Tr core/light.cpp:63-82, Scene::IntersectTr core/scene.cpp:56-70).  The rays are WALKED segment by segment through the queues:
// k_trace<2 / 1, ..., TR> finds the segment's closest hit with the persistent-lane machinery (a first version that let every lane loop through
// VisibilityTrD on its own cost what the shading kernel had saved: profiles/r03_o_*), this kernel multiplies the segment's closed-form transmittance
// in (HomogeneousMedium::Tr draws no sampler dimension: the path's dimension stream does not depend on these rays), ends the ray -- an opaque
// surface, the light, nothing -- or re-aims it behind the interface (shadow rays: isect.SpawnRayTo(p1); MIS rays: isect.SpawnRay(d)) and appends it
// to the other queue for the next round.  Most rays end in the first round.
// MODE 2: shadow rays (NeeRec::sh_*; light point in pad[0..1] + sh_c.w).  MODE 1: the BSDF-sampled ray of the MIS estimator (NeeRec::mi_*).
// DIMS: the split form (DevVol::tr_dims) -- a separate instance, so that scenes with homogeneous media only keep the lean kernel that was measured (profiles/r03_q_bench_c3_fogbox.json)
template <int MODE, bool INST, bool DIMS>
__global__ void __launch_bounds__(PT_BLOCK) k_vol_tr_step(const DevScene *scp, PathState ps, DevVol vol, const uint32_t *qIn, uint32_t rowIn, uint32_t *qOut, uint32_t rowOut) {
    const DevScene &sc = *scp;
    const int w = MODE == 1 ? 1 : 0;
    for (SegIter it(ps.qcount, rowIn, ps.seg_cap); it.more(); it.next()) {
        const bool active = it.valid();
        bool again = false;
        uint32_t slot = 0;
        if (active) {
            slot = qIn[it.item()];
            const uint4 hit = ps.trs[slot].hit[w];
            const float4 acc = ps.trs[slot].acc[w];
            RGB Tr(acc.x, acc.y, acc.z);
            const bool hitSurface = hit.x != TRAV_MISS;
            const Float tHit = __uint_as_float(hit.y);
            const bool opaque = hitSurface && (int)sc.tri_info[hit.x].y >= 0;
            // split form (DevVol::tr_dims): Medium::Tr draws from the path's sampler (ratio tracking, media/grid.cpp:89-118) -- the shadow walk of a vertex runs
            // to its end before its MIS walk starts (run_pass), k_vol_continue goes on from the dimension they leave behind
            VSampler smp;
            VSampler *sp = nullptr;
            auto samplerFor = [&](int m) {   // only a segment inside a grid medium draws: the others leave the path's sampler alone
                if constexpr (DIMS) if (m >= 0 && vol.media[m].type != MI_MEDIUM_HOMOGENEOUS) {
                    const uint4 s4 = ps.rec[slot].smp;
                    smp.index = (uint64_t)s4.x | ((uint64_t)s4.y << 32);
                    smp.dimension = (int)s4.z;
                    smp.px = smp.py = 0;
                    smp.Prefetch(sc);
                    sp = &smp;
                }
            };
            if (MODE == 2) {
                const float4 o4 = ps.nee[slot].sh_o, d4 = ps.nee[slot].sh_d;
                const V3 o(o4.x, o4.y, o4.z), d(d4.x, d4.y, d4.z);
                const int medium = __float_as_int(d4.w);
                if (!opaque) {   // (an opaque surface in between: the light sample contributes nothing)
                    samplerFor(medium);
                    if (medium >= 0) Tr = Tr * MediumTr(scp, vol.media + medium, o, d, hitSurface ? tHit : o4.w, sp);
                    if (!hitSurface) {
                        if (!Tr.IsBlack()) {
                            const float4 c4 = ps.nee[slot].sh_c;
                            float4 L = ps.rec[slot].L;
                            L.x += c4.x * Tr.r; L.y += c4.y * Tr.g; L.z += c4.z * Tr.b;
                            ps.rec[slot].L = L;
                        }
                    } else {   // a BSDF-less interface: re-aim at the light point from behind it
                        VHit vh;
                        HitToIsect(scp, &vol, hit.x, o, d, hit.z, medium, false, &vh);
                        const float4 c4 = ps.nee[slot].sh_c, a4 = ps.nee[slot].pad[0], b4 = ps.nee[slot].pad[1];
                        ShadowRay sr = SpawnRayTo(vh.is, V3(a4.x, a4.y, a4.z), V3(a4.w, b4.x, b4.y), V3(b4.z, b4.w, c4.w));
                        ps.nee[slot].sh_o = make_float4(sr.o.x, sr.o.y, sr.o.z, sr.tMax);
                        ps.nee[slot].sh_d = make_float4(sr.d.x, sr.d.y, sr.d.z, __int_as_float(GetMediumOf(vh.is.n, vh.mIn, vh.mOut, sr.d)));
                        ps.trs[slot].acc[0] = make_float4(Tr.r, Tr.g, Tr.b, 0);
                        again = true;
                    }
                }
            } else {
                const float4 o4 = ps.nee[slot].mi_o, d4 = ps.nee[slot].mi_d;
                const V3 o(o4.x, o4.y, o4.z), wi(d4.x, d4.y, d4.z);
                const int medium = __float_as_int(o4.w);
                samplerFor(medium);
                if (medium >= 0) Tr = Tr * MediumTr(scp, vol.media + medium, o, wi, hitSurface ? tHit : PT_INFINITY, sp);
                if (hitSurface && !opaque) {   // interface: isect.SpawnRay(ray.d), medium = isect.GetMedium(d)
                    VHit vh;
                    HitToIsect(scp, &vol, hit.x, o, wi, hit.z, medium, false, &vh);
                    const V3 no = OffsetRayOrigin(vh.is.p, vh.is.pError, vh.is.n, wi);
                    ps.nee[slot].mi_o = make_float4(no.x, no.y, no.z, __int_as_float(GetMediumOf(vh.is.n, vh.mIn, vh.mOut, wi)));
                    ps.trs[slot].acc[1] = make_float4(Tr.r, Tr.g, Tr.b, 0);
                    again = true;
                } else {
                    const int lightNum = (int)__float_as_uint(d4.w);
                    const DevLight *light = sc.lights + lightNum;
                    RGB Li(0.f);
                    if (hitSurface) {
                        if ((int)sc.tri_info[hit.x].z == lightNum) {   // lightIsect.primitive->GetAreaLight() == &light (integrator.cpp:207)
                            VHit vh;
                            HitToIsect(scp, &vol, hit.x, o, wi, hit.z, -1, false, &vh);
                            Li = AreaL(*light, vh.is.n, -wi);
                        }
                    } else if (light->type == MI_LIGHT_INFINITE)
                        Li = InfiniteLe(light, wi);
                    Li = Li * Tr;
                    if (!Li.IsBlack()) {
                        const float4 c4 = ps.nee[slot].mi_c;
                        float4 L = ps.rec[slot].L;
                        L.x += c4.x * Li.r; L.y += c4.y * Li.g; L.z += c4.z * Li.b;
                        ps.rec[slot].L = L;
                    }
                }
            }
            if constexpr (DIMS) { if (sp) ps.rec[slot].smp.z = (uint32_t)smp.dimension; }
        }
        const uint32_t qseg = blockIdx.x & 7;
        const uint32_t pos = wave_append(&ps.qcount[QCI(rowOut, qseg)], again);
        if (again) qOut[qseg * ps.seg_cap + pos] = slot;
    }
}

// One round of the probe walk.  first: the paths k_shade_vol<WAVE> parked in this bounce (no segment traced yet); otherwise TrState::hit[0] holds the
// closest hit of the segment in NeeRec::sh_o / sh_d (k_trace<2, ..., TR>: Scene::Intersect with tMax = 1 - ShadowEpsilon, alphaMask at candidate
// hits).  Per path: take the hit into the chain (Interaction::SpawnRayTo(pTarget) from the hit point, interaction.h:68-72), count it if its primitive
// carries the BSSRDF's material object (bssrdf.cpp:302), at the end of the first walk choose (bssrdf.cpp:311-314) and start the second; the second
// walk ends at the chosen hit -> QC_SSS for k_sss_entry.  A chain without such a hit ends the path (path.cpp:160: S.IsBlack()).
// One probe-walk step of ONE path (the body of a round): *again = the chain goes on with the segment now in NeeRec::sh_o / sh_d, *done = pi is chosen (QC_SSS)
// The tail kernel's own list of the counted hits of a first walk (k_sss_probe_tail): entry k = the (first + k)-th counted hit, in the format of SssRec::keep_*.  With it the
// chain of a long probe ray is walked ONCE, as the reference walks it (bssrdf.cpp:285-314 keeps every hit in a list and steps to the chosen one): without, the walk is
// repeated up to the chosen hit, on average half its length again -- and the tail launch lasts as long as its longest chain.
struct SssLog {
    float4 *o, *d;      // segment origin | primitive (bits), segment direction | its medium (bits)
    uint32_t *inst;
    uint32_t first, cap;
};
template <bool INST>
PT_DEV void SssProbeStepOne(const DevScene *scp, const PathState &ps, const DevVol &vol, uint32_t slot, bool first, bool *againOut, bool *doneOut, const SssLog *lg = nullptr) {
    bool again = false, done = false;
    SssRec *S = &ps.sss[slot];
    const float4 bs4 = S->base_seen, tg4 = S->target_sel;
    const V3 baseP(bs4.x, bs4.y, bs4.z), pTarget(tg4.x, tg4.y, tg4.z);
    const int material = (int)__float_as_uint(S->ns_mat.w);
    uint32_t seen = __float_as_uint(bs4.w) & 0x7fffffffu, pass = __float_as_uint(bs4.w) >> 31;
    uint32_t nFound = __float_as_uint(S->rho_found.w), selected = __float_as_uint(tg4.w);
    V3 p = baseP, pErr, n;   // a plain Interaction: no normal, no error bounds, no media
    int mIn = -1, mOut = -1;
    bool endOfWalk = false;
    if (!first) {
        const uint4 hit = ps.trs[slot].hit[0];
        if (hit.x == TRAV_MISS) endOfWalk = true;
        else {
            const float4 o4 = ps.nee[slot].sh_o, d4 = ps.nee[slot].sh_d;
            const int rayMedium = __float_as_int(d4.w);   // GetMedium(dir) of the chain point the segment left from
            VHit vh;
            HitToIsect(scp, &vol, hit.x, V3(o4.x, o4.y, o4.z), V3(d4.x, d4.y, d4.z), hit.z, rayMedium, false, &vh);
            p = vh.is.p; pErr = vh.is.pError; n = vh.is.n; mIn = vh.mIn; mOut = vh.mOut;
            if ((int)vh.tinfo.y == material) {
                if (pass == 1 && seen == selected) {   // pi: k_sss_entry rebuilds the interaction from the segment, its medium and the primitive
                    S->nz_pi = make_float4(0, __uint_as_float(hit.x), __uint_as_float(hit.z), d4.w);
                    S->pi_o = o4; S->pi_d = d4;
                    done = true;
                } else {
                    if (pass == 0 && seen < PT_SSS_KEEP) {
                        S->keep_o[seen] = make_float4(o4.x, o4.y, o4.z, __uint_as_float(hit.x));
                        S->keep_d[seen] = d4;
                        S->keep_inst[seen] = hit.z;
                    }
                    if (lg && pass == 0 && seen >= lg->first && seen - lg->first < lg->cap) {
                        const uint32_t k = seen - lg->first;
                        lg->o[k] = make_float4(o4.x, o4.y, o4.z, __uint_as_float(hit.x));
                        lg->d[k] = d4;
                        lg->inst[k] = hit.z;
                    }
                    ++seen;
                }
            }
        }
    }
    if (!done) {
        V3 dir = pTarget - p;
        if (!endOfWalk && dir.x == 0 && dir.y == 0 && dir.z == 0) endOfWalk = true;
        bool walking = !endOfWalk;
        if (endOfWalk && pass == 0 && seen > 0) {   // choose, then the same walk again up to the chosen hit
            nFound = seen;
            int sel = (int)(S->sigt_u1.w * (int)nFound);
            selected = (uint32_t)(sel < 0 ? 0 : (sel > (int)nFound - 1 ? (int)nFound - 1 : sel));
            if (selected < PT_SSS_KEEP) {   // the chosen hit is one of the kept ones
                const float4 ko = S->keep_o[selected], kd = S->keep_d[selected];
                S->nz_pi = make_float4(0, ko.w, __uint_as_float(S->keep_inst[selected]), kd.w);
                S->pi_o = ko; S->pi_d = kd;
                S->rho_found.w = __uint_as_float(nFound);
                done = true;
            } else if (lg && selected >= lg->first && selected - lg->first < lg->cap) {   // ... or one the tail kernel listed
                const uint32_t k = selected - lg->first;
                const float4 ko = lg->o[k], kd = lg->d[k];
                S->nz_pi = make_float4(0, ko.w, __uint_as_float(lg->inst[k]), kd.w);
                S->pi_o = ko; S->pi_d = kd;
                S->rho_found.w = __uint_as_float(nFound);
                done = true;
            } else {
                pass = 1; seen = 0;
                p = baseP; pErr = V3(); n = V3(); mIn = mOut = -1;
                dir = pTarget - p;
                walking = !(dir.x == 0 && dir.y == 0 && dir.z == 0);
            }
        }
        if (walking && !done) {
            const V3 origin = OffsetRayOrigin(p, pErr, n, dir);
            ps.nee[slot].sh_o = make_float4(origin.x, origin.y, origin.z, 1 - PT_SHADOW_EPS);
            ps.nee[slot].sh_d = make_float4(dir.x, dir.y, dir.z, __int_as_float(GetMediumOf(n, mIn, mOut, dir)));
            S->rho_found.w = __uint_as_float(nFound);
            S->base_seen.w = __uint_as_float(seen | (pass << 31));
            S->target_sel.w = __uint_as_float(selected);
            again = true;
        }
    }
    *againOut = again; *doneOut = done;
}
template <bool INST>
__global__ void __launch_bounds__(PT_BLOCK) k_sss_probe_step(const DevScene *scp, PathState ps, DevVol vol, const uint32_t *qIn, uint32_t rowIn, uint32_t *qOut, uint32_t rowOut, int first) {
    for (SegIter it(ps.qcount, rowIn, ps.seg_cap); it.more(); it.next()) {
        const bool active = it.valid();
        bool again = false, done = false;
        uint32_t slot = 0;
        if (active) {
            slot = qIn[it.item()];
            SssProbeStepOne<INST>(scp, ps, vol, slot, first != 0, &again, &done);
        }
        const uint32_t qseg = blockIdx.x & 7;
        const uint32_t pos = wave_append(&ps.qcount[QCI(rowOut, qseg)], again);
        if (again) qOut[qseg * ps.seg_cap + pos] = slot;
        const uint32_t posD = wave_append(&ps.qcount[QCI(QC_SSS, qseg)], done);
        if (done) ps.q_sss[qseg * ps.seg_cap + posD] = slot;
    }
}
// The TAIL of the probe walk (round 4).  Chains have very different lengths: after a few rounds most paths are done and the queue holds the few whose probe ray
// crosses many surfaces -- each further round is two launches and a host read-back for a handful of rays (915 ms of the 10 M-triangle
// subsurface frame).  Once the queue is below PBRT_AMD_SSS_TAIL rays (default 131072) the host hands it to this kernel: every lane finishes its own chain --
// the segment's closest hit by the per-lane tracer (TraceLane: Scene::Intersect with alphaMask at candidate hits, as k_trace<2, ..., TR> finds it), then
// the same step (SssProbeStepOne) -- until pi is chosen or the chain ends.  Same rays, same arithmetic, same counters.
template <bool INST>
__global__ void __launch_bounds__(PT_BLOCK) k_sss_probe_tail(const DevScene *scp, PathState ps, DevVol vol, const uint32_t *qIn, uint32_t rowIn) {
    __shared__ StackEntry lds_stack[PT_LDS_STACK * PT_BLOCK];
    NoiseLdsInit();
    LaneTracer lt;
    lt.scp = scp;
    lt.lds = (LdsStackEntry *)&lds_stack[threadIdx.x];
    lt.spill = reinterpret_cast<StackEntry *>(ps.spill) + (size_t)(blockIdx.x * PT_BLOCK + threadIdx.x) * ps.spill_per_thread;
    lt.nClosest = lt.nAny = lt.guardTrips = 0;
    for (SegIter it(ps.qcount, rowIn, ps.seg_cap); it.more(); it.next()) {
        const bool active = it.valid();
        bool done = false;
        uint32_t slot = 0;
        if (active) {
            slot = qIn[it.item()];
            // this lane's list of counted hits (PathState::sss_log_*: one slice per thread of the launch, null beyond the slices or without PBRT_AMD_SSS_LOG=1), from the
            // hit count the chain arrives with; a chain that arrives in its second walk has no use for one
            SssLog lg;
            const SssLog *lgp = nullptr;
            const uint32_t thread = blockIdx.x * PT_BLOCK + threadIdx.x;
            if (ps.sss_log_o && thread < ps.sss_log_threads) {
                const uint32_t seenBits = __float_as_uint(ps.sss[slot].base_seen.w);
                if ((seenBits >> 31) == 0) {
                    lg.o = ps.sss_log_o + (size_t)thread * ps.sss_log_cap; lg.d = ps.sss_log_d + (size_t)thread * ps.sss_log_cap; lg.inst = ps.sss_log_inst + (size_t)thread * ps.sss_log_cap;
                    lg.first = seenBits & 0x7fffffffu; lg.cap = ps.sss_log_cap;
                    lgp = &lg;
                }
            }
            bool again = true;
            for (uint32_t seg = 0; again && seg < 16384u; ++seg) {   // (the host's own bound on the rounds of a walk)
                const float4 o4 = ps.nee[slot].sh_o, d4 = ps.nee[slot].sh_d;
                const LaneHit h = TraceLane<false, INST>(&lt, V3(o4.x, o4.y, o4.z), V3(d4.x, d4.y, d4.z), o4.w);
                ps.trs[slot].hit[0] = make_uint4(h.prim, __float_as_uint(h.t), h.inst, 0u);
                SssProbeStepOne<INST>(scp, ps, vol, slot, false, &again, &done, lgp);
            }
            if (again) ++lt.guardTrips;   // a chain that did not end: reported like a traversal that did not (the host fails the render)
        }
        const uint32_t qseg = blockIdx.x & 7;
        const uint32_t posD = wave_append(&ps.qcount[QCI(QC_SSS, qseg)], done);
        if (done) ps.q_sss[qseg * ps.seg_cap + posD] = slot;
    }
    wave_count(&ps.counters[MI_CNT_CLOSEST_RAYS], lt.nClosest);
    wave_count(&ps.counters[MI_CNT_TRACE_GUARD_TRIPS], lt.guardTrips);
}

// The entry vertex of a subsurface path (path.cpp:160-174 after Sample_S; bssrdf.cpp:235-247, 316-326): Sp and its pdf at the chosen hit, beta *= S / pdf,
// one light sample for the adapter lobe on pi's own shading frame (its two rays into the shadow / MIS queues), the adapter's Sample_f for the
// continuation, then the iteration's Russian roulette and ++bounces -- what the per-lane form does after BssrdfSample_Sp, from the parked state.
template <bool INST>
__global__ void __launch_bounds__(PT_BLOCK, PT_VOL_SHADE_WAVES) k_sss_entry(const DevScene *scp, PathState ps, DevVol vol, uint32_t qout) {
    __shared__ StackEntry lds_stack[1];
    NoiseLdsInit();
    const DevScene &sc = *scp;
    LaneTracer lt;
    lt.scp = scp; lt.lds = (LdsStackEntry *)&lds_stack[0]; lt.spill = nullptr;
    lt.nClosest = lt.nAny = lt.guardTrips = 0;
    for (SegIter it(ps.qcount, QC_SSS, ps.seg_cap); it.more(); it.next()) {
        const bool active = it.valid();
        bool cont = false, wantSplit = false;
        uint32_t slot = 0;
        NeeOut nee;
        nee.wantShadow = nee.wantMis = false;
        if (active) {
            slot = ps.q_sss[it.item()];
            const SssRec *S = &ps.sss[slot];
            const float4 b4 = ps.rec[slot].beta, L4 = ps.rec[slot].L;
            const uint4 s4 = ps.rec[slot].smp;
            RGB beta(b4.x, b4.y, b4.z), L(L4.x, L4.y, L4.z), betaNee(0.f);
            const Float etaScale = b4.w;
            int bounces = (int)(s4.w & 0xffffu);
            bool specularBounce = (s4.w >> 16) & 1u;
            const bool noDiff = (s4.w >> 17) & 1u;
            VSampler smp;
            smp.index = (uint64_t)s4.x | ((uint64_t)s4.y << 32);
            smp.dimension = (int)s4.z;
            smp.px = smp.py = 0;
            smp.Prefetch(sc);
            VolCtx cx;
            cx.scp = scp; cx.vol = &vol; cx.lt = &lt; cx.smp = &smp; cx.nee = &nee;
            DevBSSRDF bssrdf;
            SssBssrdfOf(S, vol, &bssrdf);
            const float4 zp = S->nz_pi, po4 = S->pi_o, pd4 = S->pi_d;
            const uint32_t nFound = __float_as_uint(S->rho_found.w);
            VHit pi;
            HitToIsect(scp, &vol, __float_as_uint(zp.y), V3(po4.x, po4.y, po4.z), V3(pd4.x, pd4.y, pd4.z), __float_as_uint(zp.z), __float_as_int(zp.w), false, &pi);
            const Float spdf = BssrdfPdf_Sp(&bssrdf, pi.is.p, pi.is.n) / (int)nFound;
            const RGB Sp = BssrdfSr(&bssrdf, (bssrdf.poP - pi.is.p).Length());   // Sp(pi) = Sr(Distance(po.p, pi.p))
            bool alive = !(Sp.IsBlack() || spdf == 0);
            V3 no, nd;
            int nmedium = -1;
            if (alive) {
                beta = beta * (Sp / spdf);
                mi_material piMat;
                piMat.n_bxdfs = 1; piMat.eta = 1;
                __builtin_memset(&piMat.bxdfs[0], 0, sizeof(mi_bxdf));
                piMat.bxdfs[0].type = MI_BXDF_BSSRDF_ADAPTER;
                piMat.bxdfs[0].etaB = bssrdf.eta;
                pi.is.wo = pi.is.ns;
                LaneBSDF piBsdf(pi.is, &piMat);
                L = L + beta * UniformSampleOneLightD<INST, LaneBSDF>(cx, &pi.is, pi.mIn, pi.mOut, &piBsdf, 0);
                betaNee = beta;
                if (vol.tr_dims && (nee.wantShadow || nee.wantMis)) {   // split form: the entry vertex's visibility queries draw dimensions -- k_vol_continue samples the adapter lobe after them
                    wantSplit = true;
                    ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                    ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17) | (1u << 19));
                    alive = false;
                } else {
                Float u0, u1, pdf;
                int flags;
                V3 wi;
                smp.Get2D(sc, &u0, &u1);
                const RGB f = piBsdf.Sample_f(pi.is.wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                if (f.IsBlack() || pdf == 0) alive = false;
                else {
                    beta = beta * (f * AbsDot(wi, pi.is.ns) / pdf);
                    specularBounce = (flags & BSDF_SPECULAR) != 0;
                    no = OffsetRayOrigin(pi.is.p, pi.is.pError, pi.is.n, wi);   // pi.SpawnRay(wi)
                    nd = wi;
                    nmedium = GetMediumOf(pi.is.n, pi.mIn, pi.mOut, wi);
                }
                }
            }
            if (alive) {   // Russian roulette (path.cpp:176-184 / volpath.cpp:183-189); the loop's ++bounces
                cont = true;
                const RGB rrBeta = beta * etaScale;
                if (rrBeta.MaxComponentValue() < sc.rr_threshold && bounces > 3) {
                    const Float q = mx((Float).05, 1 - rrBeta.MaxComponentValue());
                    if (smp.Get1D(sc) < q) cont = false;
                    else beta = beta / (1 - q);
                }
                ++bounces;
            }
            ps.rec[slot].L = make_float4(L.r, L.g, L.b, 0);
            WriteNeeRecords(ps, vol, slot, nee, betaNee);
            if (cont) {
                ps.rec[slot].ray_o = make_float4(no.x, no.y, no.z, PT_INFINITY);
                ps.rec[slot].ray_d = make_float4(nd.x, nd.y, nd.z, 0);
                ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17));
                if (vol.handle_media) ps.rec[slot].pad2 = make_float4(__uint_as_float((uint32_t)nmedium), 0, 0, 0);
            }
        }
        const uint32_t qseg = blockIdx.x & 7, qbase = qseg * ps.seg_cap;
        uint32_t posE, posS, posM;
        PT_WAVE_APPEND3(&ps.qcount[QCI(qout, qseg)], &ps.qcount[QCI(QC_SHADOW, qseg)], &ps.qcount[QCI(QC_MIS, qseg)], cont, nee.wantShadow, nee.wantMis, &posE, &posS, &posM);
        if (cont) { ps.q_ext[qout][qbase + posE] = slot; }
        if (nee.wantShadow) ps.q_shadow[qbase + posS] = slot;
        if (nee.wantMis) ps.q_mis[qbase + posM] = slot;
        if (vol.tr_dims) {
            const uint32_t posB = wave_append(&ps.qcount[QCI(QC_CONT, qseg)], wantSplit);
            if (wantSplit) ps.q_cont[qbase + posB] = slot;
        }
    }
}

// Second stage of a vertex in the split form (DevVol::tr_dims: some medium is a grid).  VolPathIntegrator::Li samples the continuation AFTER
// UniformSampleOneLight (volpath.cpp:96-103, 125-135), and with a grid medium the visibility queries in between -- VisibilityTester::Tr, Scene::IntersectTr --
// draw a data-dependent number of sampler dimensions (ratio tracking).  k_shade_vol<WAVE> therefore stops a vertex that has direct-lighting rays
// after the light sample (beta, L, the sampler state and, for a medium vertex, the scattering point are parked in the PathRec; ray and hit are still
// there), the walks consume their dimensions (k_vol_tr_step), and this kernel rebuilds the vertex the way k_shade_vol built it -- same code, same inputs --
// and does the rest of the loop body: phase function / BSDF sample, throughput, eta scale, Russian roulette, ++bounces.
template <bool INST>
__global__ void __launch_bounds__(PT_BLOCK, PT_VOL_SHADE_WAVES) k_vol_continue(const DevScene *scp, PathState ps, DevVol vol, uint32_t qout) {
    NoiseLdsInit();
    const DevScene &sc = *scp;
    for (SegIter it(ps.qcount, QC_CONT, ps.seg_cap); it.more(); it.next()) {
        const bool active = it.valid();
        bool cont = false, wantProbe = false;
        uint32_t slot = 0;
        if (active) {
            slot = ps.q_cont[it.item()];
            const uint2 hr = ps.rec[slot].hit;
            const float4 o4 = ps.rec[slot].ray_o, d4 = ps.rec[slot].ray_d, b4 = ps.rec[slot].beta, m4 = ps.rec[slot].pad2;
            const uint4 s4 = ps.rec[slot].smp;
            const V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            RGB beta(b4.x, b4.y, b4.z);
            Float etaScale = b4.w;
            int bounces = (int)(s4.w & 0xffffu);
            bool specularBounce = false;
            const bool noDiff = (s4.w >> 17) & 1u, mediumVertex = (s4.w >> 18) & 1u, entryVertex = (s4.w >> 19) & 1u;
            const int medium = (int)__float_as_uint(m4.x);
            VSampler smp;
            smp.index = (uint64_t)s4.x | ((uint64_t)s4.y << 32);
            smp.dimension = (int)s4.z;
            smp.px = smp.py = 0;
            smp.Prefetch(sc);
            V3 no, nd;
            int nmedium = medium;
            bool alive = true;
            if (mediumVertex) {   // volpath.cpp:99-103
                const V3 p(m4.y, m4.z, m4.w);
                const Float g = vol.media[medium].g;
                Float u0, u1;
                smp.Get2D(sc, &u0, &u1);
                V3 wi;
                HGSample_p(g, -rd, &wi, u0, u1);
                no = OffsetRayOrigin(p, V3(), V3(), wi);   // mi.SpawnRay(wi)
                nd = wi;
            } else if (entryVertex) {   // the entry vertex of a subsurface path, parked by k_sss_entry after its light sample (volpath.cpp:170-180): the adapter lobe's Sample_f
                const SssRec *S = &ps.sss[slot];
                DevBSSRDF bssrdf;
                SssBssrdfOf(S, vol, &bssrdf);
                const float4 zp = S->nz_pi, po4 = S->pi_o, pd4 = S->pi_d;
                VHit pi;
                HitToIsect(scp, &vol, __float_as_uint(zp.y), V3(po4.x, po4.y, po4.z), V3(pd4.x, pd4.y, pd4.z), __float_as_uint(zp.z), __float_as_int(zp.w), false, &pi);
                mi_material piMat;
                piMat.n_bxdfs = 1; piMat.eta = 1;
                __builtin_memset(&piMat.bxdfs[0], 0, sizeof(mi_bxdf));
                piMat.bxdfs[0].type = MI_BXDF_BSSRDF_ADAPTER;
                piMat.bxdfs[0].etaB = bssrdf.eta;
                pi.is.wo = pi.is.ns;
                LaneBSDF piBsdf(pi.is, &piMat);
                Float u0, u1, pdf;
                int flags;
                V3 wi;
                smp.Get2D(sc, &u0, &u1);
                const RGB f = piBsdf.Sample_f(pi.is.wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                if (f.IsBlack() || pdf == 0) alive = false;
                else {
                    beta = beta * (f * AbsDot(wi, pi.is.ns) / pdf);
                    specularBounce = (flags & BSDF_SPECULAR) != 0;
                    no = OffsetRayOrigin(pi.is.p, pi.is.pError, pi.is.n, wi);   // pi.SpawnRay(wi)
                    nd = wi;
                    nmedium = GetMediumOf(pi.is.n, pi.mIn, pi.mOut, wi);
                }
            } else {              // volpath.cpp:130-150
                VHit vh;
                HitToIsect(scp, &vol, hr.x, ro, rd, INST ? ps.rec[slot].pad0 : TRAV_NO_INSTANCE, medium, vol.textured != 0, &vh);
                const int matIdx = (int)vh.tinfo.y;
                mi_material laneMat;
                if (vol.textured) {
                    if (bounces == 0 && !noDiff) {
                        float2 pf = ps.rec[slot].pfilm, ln = ps.rec[slot].lens;
                        RayDiffT rdf = CameraDifferentials(&c_tex.camera, pf.x, pf.y, ln.x, ln.y, c_tex.spp, ro, rd);
                        ComputeDifferentials(vh.is.p, vh.is.n, &vh.ix, rdf);
                    }
                    ComputeScatteringFunctionsT(sc.materials, matIdx, &vh.is, &vh.ix, &laneMat);
                }
                const mi_material *matPtr = vol.textured ? &laneMat : sc.materials + matIdx;
                LaneBSDF bsdf(vh.is, matPtr);
                const V3 wo = -rd;
                V3 wi;
                Float pdf, u0, u1;
                int flags;
                smp.Get2D(sc, &u0, &u1);
                const RGB f = bsdf.Sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                if (f.IsBlack() || pdf == 0.f) alive = false;
                else {
                    beta = beta * (f * AbsDot(wi, vh.is.ns) / pdf);
                    specularBounce = (flags & BSDF_SPECULAR) != 0;
                    if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                        Float eta = bsdf.m->eta;
                        etaScale *= (Dot(wo, vh.is.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                    }
                    no = OffsetRayOrigin(vh.is.p, vh.is.pError, vh.is.n, wi);   // isect.SpawnRay(wi)
                    nd = wi;
                    nmedium = GetMediumOf(vh.is.n, vh.mIn, vh.mOut, wi);
                    if (vol.sss_wave && (flags & BSDF_TRANSMISSION)) {   // volpath.cpp:153-180: a BSSRDF here sends the path into its probe chain (as k_shade_vol<WAVE> does for a vertex it finishes itself)
                        DevBSSRDF bssrdf;
                        ComputeBSSRDFD(&vol, matIdx, &vh.is, &vh.ix, &bssrdf);
                        if (bssrdf.table) {
                            Float u20, u21, u1s;
                            smp.Get2D(sc, &u20, &u21);   // Sample_S(scene, sampler.Get1D(), sampler.Get2D(), ...): arguments evaluated right to left
                            u1s = smp.Get1D(sc);
                            SssProbe pr;
                            if (!BssrdfProbeSetup(&bssrdf, u1s, u20, u21, &pr)) alive = false;
                            else { wantProbe = true; SssPark(&ps.sss[slot], vol, bssrdf, pr); }
                        }
                    }
                }
            }
            if (alive && wantProbe) {   // parked for k_sss_probe_step / k_sss_entry, which also take this iteration's Russian roulette and ++bounces
                ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17));
                alive = false;
            } else
                wantProbe = false;
            if (alive) {   // Russian roulette (volpath.cpp:183-189); the loop's ++bounces
                cont = true;
                const RGB rrBeta = beta * etaScale;
                if (rrBeta.MaxComponentValue() < sc.rr_threshold && bounces > 3) {
                    const Float q = mx((Float).05, 1 - rrBeta.MaxComponentValue());
                    if (smp.Get1D(sc) < q) cont = false;
                    else beta = beta / (1 - q);
                }
                ++bounces;
            }
            if (cont) {
                ps.rec[slot].ray_o = make_float4(no.x, no.y, no.z, PT_INFINITY);
                ps.rec[slot].ray_d = make_float4(nd.x, nd.y, nd.z, 0);
                ps.rec[slot].beta = make_float4(beta.r, beta.g, beta.b, etaScale);
                ps.rec[slot].smp = make_uint4(s4.x, s4.y, (uint32_t)smp.dimension, (uint32_t)bounces | ((uint32_t)specularBounce << 16) | ((uint32_t)noDiff << 17));
                ps.rec[slot].pad2 = make_float4(__uint_as_float((uint32_t)nmedium), 0, 0, 0);
            }
        }
        const uint32_t qseg = blockIdx.x & 7, qbase = qseg * ps.seg_cap;
        const uint32_t posE = wave_append(&ps.qcount[QCI(qout, qseg)], cont);
        if (cont) { ps.q_ext[qout][qbase + posE] = slot; }
        if (vol.sss_wave) {
            const uint32_t posP = wave_append(&ps.qcount[QCI(QC_PROBE0, qseg)], wantProbe);
            if (wantProbe) ps.q_probe[0][qbase + posP] = slot;
        }
    }
}
