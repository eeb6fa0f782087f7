// Participating media and subsurface scattering on the device: what VolPathIntegrator::Li
// (integrators/volpath.cpp:55-190) and the BSSRDF branch of PathIntegrator::Li (integrators/path.cpp:153-174) need beyond
// the surface path -- Medium::Sample / Medium::Tr for HomogeneousMedium and GridDensityMedium, the Henyey-Greenstein phase
// function, TabulatedBSSRDF over SeparableBSSRDF, and the Catmull-Rom spline helpers of core/interpolation.cpp.
// One lane = one path vertex (k_shade_vol, pt_volpath.h); every loop below is a per-lane loop with a data-dependent trip
// count (delta / ratio tracking, spline inversion), which is why these scenes run their own shading kernel.
#pragma once
#include "pt_material.h"

#define PT_MAX_FLOAT 3.402823466e+38f
#define PT_INV_4PI 0.07957747154594766788f

struct DevBssrdfTable {   // mi_bssrdf_table with device pointers
    int32_t n_rho, n_radius;
    const float *rho_samples, *radius_samples, *profile, *rho_eff, *profile_cdf;
};
// what the volumetric / subsurface shading kernel needs beyond DevScene (one kernel argument; the kernels of other scenes never see it)
struct DevVol {
    const mi_medium *media;          // density pointers patched to device memory
    const int32_t *mesh_medium;      // [2 * mesh] inside, [2 * mesh + 1] outside; null: no primitive names a medium
    const mi_bssrdf_desc *bssrdf;    // per material slot; null: no subsurface material
    const DevBssrdfTable *tables;
    int32_t camera_medium;
    int32_t handle_media;            // Integrator "volpath": media attenuate and scatter (PathIntegrator ignores them)
    int32_t textured;                // some material is textured (c_tex.descs is set): lobe lists are built per hit
    int32_t tr_queues;               // wavefront form with BSDF-less interfaces between homogeneous media: shadow / MIS rays carry their light point and start
                                     // medium through the queues and k_vol_tr walks them interface by interface (pt_volpath.h)
    int32_t tr_dims;                 // split form: some medium is a grid, whose Tr draws sampler dimensions -- tr_queues is on, k_vol_tr_step carries the path's sampler through the walks,
                                     // k_shade_vol<WAVE> parks a vertex with direct-lighting rays after its light sample and k_vol_continue samples the continuation (pt_volpath.h)
    int32_t sss_wave;                // wavefront form of scenes with BSSRDF materials under Integrator "path": probe chains walked through the queues (pt_volpath.h: SssRec)
};

// The path's sampler with the next PT_VOL_PRE dimensions drawn ahead in ONE batch (SamplerBatch: wave-uniform matrix rows through the scalar
// cache, as k_shade does) -- Sampler::SampleDimension fetches its 20-50 matrix words per dimension with per-lane vector loads.  Dimensions
// past the batch (long delta / ratio tracking chains) fall back to it.
#ifndef PT_VOL_PRE
#define PT_VOL_PRE 12
#endif
struct VSampler : Sampler {
    Float pre[PT_VOL_PRE];
    int preBase, preN;
    PT_DEV void Prefetch(const DevScene &sc) {
        if (Pix(sc)) { preBase = 0; preN = 0; return; }   // tile-serial samplers: nothing can be drawn ahead of a stream
        preBase = dimension;
        const int limit = sc.sampler_type == MI_SAMPLER_HALTON ? 1000 : PBRT_AMD_SOBOL_NDIM;
        preN = limit - dimension < PT_VOL_PRE ? (limit - dimension < 0 ? 0 : limit - dimension) : PT_VOL_PRE;
        if (preN > 0) SamplerBatch<PT_VOL_PRE>(sc, index, dimension, pre);   // (past the tables -- long tracking chains -- nothing is drawn ahead: SampleDimension clamps)
    }
    PT_DEV Float Get1D(const DevScene &sc) {
        if (Pix(sc)) return PixGet1D(sc);
        const int k = dimension - preBase;
        Float v;
        if (k >= 0 && k < preN) v = pre[k];
        else v = SampleDimension(sc, dimension);
        ++dimension;
        return v;
    }
    PT_DEV void Get2D(const DevScene &sc, Float *u0, Float *u1) {
        if (Pix(sc)) { PixGet2D(sc, u0, u1); return; }
        *u0 = Get1D(sc); *u1 = Get1D(sc);
    }
};

PT_DEV RGB ExpRGB(const RGB &s) { return RGB(expf_(s.r), expf_(s.g), expf_(s.b)); }   // Exp(Spectrum) core/spectrum.h:253-258

// ------------------------------------------------------------------ HenyeyGreenstein (core/medium.h:69-72, medium.cpp:189-213)
PT_DEV Float PhaseHG(Float cosTheta, Float g) {
    Float denom = 1 + g * g + 2 * g * cosTheta;
    return PT_INV_4PI * (1 - g * g) / (denom * sqrtf_(denom));
}
PT_DEV Float HGp(Float g, const V3 &wo, const V3 &wi) { return PhaseHG(Dot(wo, wi), g); }
PT_DEV Float HGSample_p(Float g, const V3 &wo, V3 *wi, Float u0, Float u1) {
    Float cosTheta;
    if ((double)absf(g) < 1e-3) cosTheta = 1 - 2 * u0;
    else {
        Float sqrTerm = (1 - g * g) / (1 - g + 2 * g * u0);
        cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    Float sinTheta = sqrtf_(mx((Float)0, 1 - cosTheta * cosTheta));
    Float phi = 2 * PT_PI * u1;
    V3 v1, v2;
    CoordinateSystem(wo, &v1, &v2);
    Float sp, cp;
    sincosf_(phi, &sp, &cp);
    *wi = sinTheta * cp * v1 + sinTheta * sp * v2 + cosTheta * (-wo);   // SphericalDirection geometry.h:1467-1472
    return PhaseHG(-cosTheta, g);
}

// ------------------------------------------------------------------ GridDensityMedium (media/grid.{h,cpp})
PT_DEV Float GridD(const mi_medium *m, int x, int y, int z) {   // grid.h:79-84
    if (x < 0 || y < 0 || z < 0 || x >= m->nx || y >= m->ny || z >= m->nz) return 0;
    return m->density[((size_t)z * m->ny + y) * m->nx + x];
}
PT_DEV Float GridDensity(const mi_medium *m, const V3 &p) {   // grid.cpp:44-59
    V3 ps(p.x * m->nx - .5f, p.y * m->ny - .5f, p.z * m->nz - .5f);
    int px = (int)__builtin_floorf(ps.x), py = (int)__builtin_floorf(ps.y), pz = (int)__builtin_floorf(ps.z);
    V3 d = ps - V3((Float)px, (Float)py, (Float)pz);
    Float d00 = Lerp(d.x, GridD(m, px, py, pz), GridD(m, px + 1, py, pz));
    Float d10 = Lerp(d.x, GridD(m, px, py + 1, pz), GridD(m, px + 1, py + 1, pz));
    Float d01 = Lerp(d.x, GridD(m, px, py, pz + 1), GridD(m, px + 1, py, pz + 1));
    Float d11 = Lerp(d.x, GridD(m, px, py + 1, pz + 1), GridD(m, px + 1, py + 1, pz + 1));
    Float d0 = Lerp(d.y, d00, d10);
    Float d1 = Lerp(d.y, d01, d11);
    return Lerp(d.z, d0, d1);
}
struct MRay { V3 o, d; Float tMax; };
// Bounds3f((0,0,0),(1,1,1)).IntersectP(ray, &t0, &t1) core/geometry.h:1388-1409
PT_DEV bool UnitBoxIntersectP(const MRay &ray, Float *hitt0, Float *hitt1) {
    Float t0 = 0, t1 = ray.tMax;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Float invRayDir = 1 / ray.d[i];
        Float tNear = (0 - ray.o[i]) * invRayDir;
        Float tFar = (1 - ray.o[i]) * invRayDir;
        if (tNear > tFar) { Float s = tNear; tNear = tFar; tFar = s; }
        tFar *= 1 + 2 * gamma_n(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    *hitt0 = t0; *hitt1 = t1;
    return true;
}
// WorldToMedium(Ray(rWorld.o, Normalize(rWorld.d), rWorld.tMax * rWorld.d.Length())): Transform::operator()(Ray) core/transform.h:252-264
PT_DEV MRay GridRay(const mi_medium *m, const V3 &wo_, const V3 &wd_, Float wtMax) {
    const float *M = m->world_to_medium;
    V3 nd = Normalize(wd_);
    Float tMax = wtMax * wd_.Length();
    V3 o = XfPointT(M, wo_);
    Float xAbs = (absf(M[0] * wo_.x) + absf(M[1] * wo_.y) + absf(M[2] * wo_.z) + absf(M[3]));
    Float yAbs = (absf(M[4] * wo_.x) + absf(M[5] * wo_.y) + absf(M[6] * wo_.z) + absf(M[7]));
    Float zAbs = (absf(M[8] * wo_.x) + absf(M[9] * wo_.y) + absf(M[10] * wo_.z) + absf(M[11]));
    V3 oError = gamma_n(3) * V3(xAbs, yAbs, zAbs);
    V3 d(M[0] * nd.x + M[1] * nd.y + M[2] * nd.z, M[4] * nd.x + M[5] * nd.y + M[6] * nd.z, M[8] * nd.x + M[9] * nd.y + M[10] * nd.z);
    Float lengthSquared = d.LengthSquared();
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), oError) / lengthSquared;
        o = o + d * dt;
        tMax -= dt;
    }
    MRay r;
    r.o = o; r.d = d; r.tMax = tMax;
    return r;
}

// ------------------------------------------------------------------ Medium::Tr / Medium::Sample
// HomogeneousMedium media/homogeneous.cpp:41-74; GridDensityMedium: ratio tracking grid.cpp:90-118, delta tracking grid.cpp:61-88.
// `smp` advances by however many dimensions the tracking loop draws (none for homogeneous Tr).
__device__ __noinline__ RGB MediumTr(const DevScene *scp, const mi_medium *m, const V3 ro, const V3 rd, Float tMax, VSampler *smp) {
    if (m->type == MI_MEDIUM_HOMOGENEOUS) return ExpRGB(-rgb3(m->sigma_t) * mn(tMax * rd.Length(), PT_MAX_FLOAT));
    MRay ray = GridRay(m, ro, rd, tMax);
    Float tMin, tEnd;
    if (!UnitBoxIntersectP(ray, &tMin, &tEnd)) return RGB(1.f);
    Float Tr = 1, t = tMin;
    const Float sigma_t = m->sigma_t[0];
    while (true) {
        t -= logf_(1 - smp->Get1D(*scp)) * m->inv_max_density / sigma_t;
        if (t >= tEnd) break;
        Float density = GridDensity(m, ray.o + ray.d * t);
        Tr *= 1 - mx((Float)0, density * m->inv_max_density);
        const Float rrThreshold = .1;
        if (Tr < rrThreshold) {
            Float q = mx((Float).05, 1 - Tr);
            if (smp->Get1D(*scp) < q) return RGB(0.f);
            Tr /= 1 - q;
        }
    }
    return RGB(Tr);
}
struct MediumSampleOut { RGB w; V3 p; bool valid; };
__device__ __noinline__ MediumSampleOut MediumSample(const DevScene *scp, const mi_medium *m, const V3 ro, const V3 rd, Float tMax, VSampler *smp) {
    MediumSampleOut out;
    out.valid = false; out.w = RGB(1.f);
    if (m->type == MI_MEDIUM_HOMOGENEOUS) {
        const RGB sigma_t = rgb3(m->sigma_t), sigma_s = rgb3(m->sigma_s);
        int channel = mni((int)(smp->Get1D(*scp) * 3), 2);
        Float dist = -logf_(1 - smp->Get1D(*scp)) / m->sigma_t[channel];
        Float t = mn(dist / rd.Length(), tMax);
        bool sampledMedium = t < tMax;
        if (sampledMedium) { out.valid = true; out.p = ro + rd * t; }
        RGB Tr = ExpRGB(-sigma_t * mn(t, PT_MAX_FLOAT) * rd.Length());
        RGB density = sampledMedium ? (sigma_t * Tr) : Tr;
        Float pdf = 0;
        pdf += density.r; pdf += density.g; pdf += density.b;
        pdf *= 1 / (Float)3;
        if (pdf == 0) pdf = 1;
        out.w = sampledMedium ? (Tr * sigma_s / pdf) : (Tr / pdf);
        return out;
    }
    MRay ray = GridRay(m, ro, rd, tMax);
    Float tMin, tEnd;
    if (!UnitBoxIntersectP(ray, &tMin, &tEnd)) return out;
    const Float sigma_t = m->sigma_t[0];
    Float t = tMin;
    while (true) {
        t -= logf_(1 - smp->Get1D(*scp)) * m->inv_max_density / sigma_t;
        if (t >= tEnd) break;
        Float dens = GridDensity(m, ray.o + ray.d * t) * m->inv_max_density;
        if (dens > smp->Get1D(*scp)) {
            out.valid = true; out.p = ro + rd * t;   // rWorld(t): the reference mixes the two parametrisations (grid.cpp:80)
            out.w = rgb3(m->sigma_s) / sigma_t;
            return out;
        }
    }
    return out;
}

// ------------------------------------------------------------------ spline helpers (core/interpolation.cpp)
// FindInterval (core/pbrt.h:398-411) over "nodes[i] <= x"
PT_DEV int FindIntervalLE(int size, const float *nodes, Float x) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (nodes[middle] <= x) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int r = first - 1;
    return r < 0 ? 0 : (r > size - 2 ? size - 2 : r);
}
struct CRWeights { Float w[4]; int offset; bool ok; };
__device__ __noinline__ CRWeights CatmullRomWeights(int size, const float *nodes, Float x) {   // interpolation.cpp:61-102
    CRWeights o;
    o.w[0] = o.w[1] = o.w[2] = o.w[3] = 0; o.offset = 0; o.ok = false;
    if (!(x >= nodes[0] && x <= nodes[size - 1])) return o;
    int idx = FindIntervalLE(size, nodes, x);
    o.offset = idx - 1;
    Float x0 = nodes[idx], x1 = nodes[idx + 1];
    Float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    o.w[1] = 2 * t3 - 3 * t2 + 1;
    o.w[2] = -2 * t3 + 3 * t2;
    if (idx > 0) {
        Float w0 = (t3 - 2 * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        o.w[0] = -w0;
        o.w[2] += w0;
    } else {
        Float w0 = t3 - 2 * t2 + t;
        o.w[0] = 0;
        o.w[1] -= w0;
        o.w[2] += w0;
    }
    if (idx + 2 < size) {
        Float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        o.w[1] -= w3;
        o.w[3] = w3;
    } else {
        Float w3 = t3 - t2;
        o.w[1] -= w3;
        o.w[2] += w3;
        o.w[3] = 0;
    }
    o.ok = true;
    return o;
}
PT_DEV Float CRInterp(const float *array, int size2, const CRWeights &cw, int idx) {   // the `interpolate` lambda of SampleCatmullRom2D
    Float value = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (cw.w[i] != 0) value += array[(cw.offset + i) * size2 + idx] * cw.w[i];
    return value;
}
// SampleCatmullRom2D interpolation.cpp:172-258 (fval / pdf outputs unused on this path)
__device__ __noinline__ Float SampleCatmullRom2D(int size1, int size2, const float *nodes1, const float *nodes2, const float *values, const float *cdf, Float alpha, Float u) {
    CRWeights cw = CatmullRomWeights(size1, nodes1, alpha);
    if (!cw.ok) return 0;
    Float maximum = CRInterp(cdf, size2, cw, size2 - 1);
    u *= maximum;
    int idx;
    {
        int first = 0, len = size2;
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (CRInterp(cdf, size2, cw, middle) <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        idx = first - 1;
        idx = idx < 0 ? 0 : (idx > size2 - 2 ? size2 - 2 : idx);
    }
    Float f0 = CRInterp(values, size2, cw, idx), f1 = CRInterp(values, size2, cw, idx + 1);
    Float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    Float width = x1 - x0;
    Float d0, d1;
    u = (u - CRInterp(cdf, size2, cw, idx)) / width;
    if (idx > 0) d0 = width * (f1 - CRInterp(values, size2, cw, idx - 1)) / (x1 - nodes2[idx - 1]);
    else d0 = f1 - f0;
    if (idx + 2 < size2) d1 = width * (CRInterp(values, size2, cw, idx + 2) - f0) / (nodes2[idx + 2] - x0);
    else d1 = f1 - f0;
    Float t;
    if (f0 != f1) t = (f0 - sqrtf_(mx((Float)0, f0 * f0 + 2 * u * (f1 - f0)))) / (f0 - f1);
    else t = u / f0;
    Float a = 0, b = 1, Fhat, fhat;
    int guard = 0;
    while (true) {
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        Fhat = t * (f0 + t * (.5f * d0 + t * ((1.f / 3.f) * (-2 * d0 - d1) + f1 - f0 + t * (.25f * (d0 + d1) + .5f * (f0 - f1)))));
        fhat = f0 + t * (d0 + t * (-2 * d0 - d1 + 3 * (f1 - f0) + t * (d0 + d1 + 2 * (f0 - f1))));
        if (absf(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (++guard > 4096) break;   // NaN inputs: the reference would spin; never taken on finite tables
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}
// InvertCatmullRom interpolation.cpp:288-345
__device__ __noinline__ Float InvertCatmullRom(int n, const float *x, const float *values, Float u) {
    if (!(u > values[0])) return x[0];
    else if (!(u < values[n - 1])) return x[n - 1];
    int i = FindIntervalLE(n, values, u);
    Float x0 = x[i], x1 = x[i + 1];
    Float f0 = values[i], f1 = values[i + 1];
    Float width = x1 - x0;
    Float d0, d1;
    if (i > 0) d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]);
    else d0 = f1 - f0;
    if (i + 2 < n) d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0);
    else d1 = f1 - f0;
    Float a = 0, b = 1, t = .5f;
    Float Fhat, fhat;
    int guard = 0;
    while (true) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        Float t2 = t * t, t3 = t2 * t;
        Fhat = (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
        fhat = (6 * t2 - 6 * t) * f0 + (-6 * t2 + 6 * t) * f1 + (3 * t2 - 4 * t + 1) * d0 + (3 * t2 - 2 * t) * d1;
        if (absf(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (++guard > 4096) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}

// ------------------------------------------------------------------ TabulatedBSSRDF over SeparableBSSRDF (core/bssrdf.h:73-140, bssrdf.cpp:199-392), TransportMode::Radiance
struct DevBSSRDF {
    const DevBssrdfTable *table;   // null: the material has no BSSRDF
    V3 poP, ns, ss, ts;            // po.p and po's shading frame (after bump mapping)
    Float eta;
    RGB sigma_t, rho;
    int material;                  // slot = Material object (Sample_Sp accepts probe hits on primitives of the same object only, bssrdf.cpp:302)
};
PT_DEV Float ch3(const RGB &c, int i) { return i == 0 ? c.r : (i == 1 ? c.g : c.b); }
__device__ __noinline__ RGB BssrdfSr(const DevBSSRDF *bs, Float r) {   // bssrdf.cpp:199-233
    const DevBssrdfTable *tb = bs->table;
    Float out[3] = {0, 0, 0};
    for (int ch = 0; ch < 3; ++ch) {
        Float rOptical = r * ch3(bs->sigma_t, ch);
        CRWeights rw = CatmullRomWeights(tb->n_rho, tb->rho_samples, ch3(bs->rho, ch));
        if (!rw.ok) continue;
        CRWeights dw = CatmullRomWeights(tb->n_radius, tb->radius_samples, rOptical);
        if (!dw.ok) continue;
        Float sr = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                Float weight = rw.w[i] * dw.w[j];
                if (weight != 0) sr += weight * tb->profile[(rw.offset + i) * tb->n_radius + (dw.offset + j)];
            }
        if (rOptical != 0) sr /= 2 * PT_PI * rOptical;
        out[ch] = sr;
    }
    RGB Sr(out[0], out[1], out[2]);
    Sr = Sr * (bs->sigma_t * bs->sigma_t);
    return ClampRGB(Sr);
}
PT_DEV Float BssrdfSample_Sr(const DevBSSRDF *bs, int ch, Float u) {   // bssrdf.cpp:353-360
    Float st = ch3(bs->sigma_t, ch);
    if (st == 0) return -1;
    const DevBssrdfTable *tb = bs->table;
    return SampleCatmullRom2D(tb->n_rho, tb->n_radius, tb->rho_samples, tb->radius_samples, tb->profile, tb->profile_cdf, ch3(bs->rho, ch), u) / st;
}
__device__ __noinline__ Float BssrdfPdf_Sr(const DevBSSRDF *bs, int ch, Float r) {   // bssrdf.cpp:362-390
    const DevBssrdfTable *tb = bs->table;
    Float st = ch3(bs->sigma_t, ch);
    Float rOptical = r * st;
    CRWeights rw = CatmullRomWeights(tb->n_rho, tb->rho_samples, ch3(bs->rho, ch));
    if (!rw.ok) return 0.f;
    CRWeights dw = CatmullRomWeights(tb->n_radius, tb->radius_samples, rOptical);
    if (!dw.ok) return 0.f;
    Float sr = 0, rhoEff = 0;
    for (int i = 0; i < 4; ++i) {
        if (rw.w[i] == 0) continue;
        rhoEff += tb->rho_eff[rw.offset + i] * rw.w[i];
        for (int j = 0; j < 4; ++j) {
            if (dw.w[j] == 0) continue;
            sr += tb->profile[(rw.offset + i) * tb->n_radius + (dw.offset + j)] * rw.w[i] * dw.w[j];
        }
    }
    if (rOptical != 0) sr /= 2 * PT_PI * rOptical;
    return mx((Float)0, sr * st * st / rhoEff);
}
__device__ __noinline__ Float BssrdfPdf_Sp(const DevBSSRDF *bs, const V3 piP, const V3 piN) {   // bssrdf.cpp:328-351
    V3 d = bs->poP - piP;
    V3 dLocal(Dot(bs->ss, d), Dot(bs->ts, d), Dot(bs->ns, d));
    V3 nLocal(Dot(bs->ss, piN), Dot(bs->ts, piN), Dot(bs->ns, piN));
    Float rProj[3] = {sqrtf_(dLocal.y * dLocal.y + dLocal.z * dLocal.z), sqrtf_(dLocal.z * dLocal.z + dLocal.x * dLocal.x),
                      sqrtf_(dLocal.x * dLocal.x + dLocal.y * dLocal.y)};
    Float pdf = 0;
    const Float axisProb[3] = {.25f, .25f, .5f};
    Float chProb = 1 / (Float)3;
    for (int axis = 0; axis < 3; ++axis)
        for (int ch = 0; ch < 3; ++ch) pdf += BssrdfPdf_Sr(bs, ch, rProj[axis]) * absf(nLocal[axis]) * chProb * axisProb[axis];
    return pdf;
}
