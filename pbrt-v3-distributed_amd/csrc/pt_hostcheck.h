// Host-side pieces shared by the layout validators (mi_bvh4q_validate): the per-ray slab constants of the quantised node step
// (host + device) and a host copy of the watertight triangle test, so that the emulation of the kernel's per-ray state machine
// (pt_bvh4q.h, bvh4q::traverse) can be compared with the oracle's BVH2 walk bit for bit on a machine without a GPU.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include "pbrt_amd.h"

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

struct SlabRay {   // per-ray constants of a quantised node step
    float o[3], inv[3];   // inv: 1/d, with +-1e30 standing in for the infinities of axis-parallel rays (see SlabRayInit)
    int neg[3];
};
PT_HD void SlabRayInit(SlabRay &r, const float o[3], const float d[3]) {
    for (int a = 0; a < 3; ++a) {
        r.o[a] = o[a];
        // d == 0: the reference works with inv = +-inf and lets NaN comparisons fall through (conservative accept on that axis iff the
        // origin lies inside the slab); a huge finite inverse gives exactly "inside the (wider) quantised slab" without NaNs
        r.inv[a] = d[a] == 0 ? __builtin_copysignf(1e30f, d[a]) : 1 / d[a];
        r.neg[a] = r.inv[a] < 0;
    }
}
namespace hostcheck {
// ---- host copy of the watertight test (Triangle::Intersect shapes/triangle.cpp:188-291, as TriangleTest in pt_scene.h)
struct Shear { int kz; float Sx, Sy, Sz; };
inline void shearInit(Shear &rs, const float d[3]) {
    float ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
    rs.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    float dx, dy, dz;
    if (rs.kz == 0) { dx = d[1]; dy = d[2]; dz = d[0]; }
    else if (rs.kz == 1) { dx = d[2]; dy = d[0]; dz = d[1]; }
    else { dx = d[0]; dy = d[1]; dz = d[2]; }
    rs.Sx = -dx / dz; rs.Sy = -dy / dz; rs.Sz = 1.f / dz;
}
inline void permute(const Shear &rs, const float v[3], float o[3]) {
    if (rs.kz == 0) { o[0] = v[1]; o[1] = v[2]; o[2] = v[0]; }
    else if (rs.kz == 1) { o[0] = v[2]; o[1] = v[0]; o[2] = v[1]; }
    else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
}
inline float gammaN(int n) { const float e = 5.9604644775390625e-08f; return (n * e) / (1 - n * e); }
inline float max3(float a, float b, float c) { return std::max(a, std::max(b, c)); }
inline bool triangleTest(const float *P0, const float *P1, const float *P2, const float o[3], const Shear &rs, float tMax, float *tOut, float bOut[3]) {
    float a[3], b[3], c[3], p0t[3], p1t[3], p2t[3];
    for (int k = 0; k < 3; ++k) { a[k] = P0[k] - o[k]; b[k] = P1[k] - o[k]; c[k] = P2[k] - o[k]; }
    permute(rs, a, p0t); permute(rs, b, p1t); permute(rs, c, p2t);
    p0t[0] += rs.Sx * p0t[2]; p0t[1] += rs.Sy * p0t[2];
    p1t[0] += rs.Sx * p1t[2]; p1t[1] += rs.Sy * p1t[2];
    p2t[0] += rs.Sx * p2t[2]; p2t[1] += rs.Sy * p2t[2];
    float e0 = p1t[0] * p2t[1] - p1t[1] * p2t[0];
    float e1 = p2t[0] * p0t[1] - p2t[1] * p0t[0];
    float e2 = p0t[0] * p1t[1] - p0t[1] * p1t[0];
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t[0] * (double)p1t[1], p2typ1tx = (double)p2t[1] * (double)p1t[0];
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t[0] * (double)p2t[1], p0typ2tx = (double)p0t[1] * (double)p2t[0];
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t[0] * (double)p0t[1], p1typ0tx = (double)p1t[1] * (double)p0t[0];
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t[2] *= rs.Sz; p1t[2] *= rs.Sz; p2t[2] *= rs.Sz;
    float tScaled = e0 * p0t[2] + e1 * p1t[2] + e2 * p2t[2];
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = max3(std::fabs(p0t[2]), std::fabs(p1t[2]), std::fabs(p2t[2]));
    float deltaZ = gammaN(3) * maxZt;
    float maxXt = max3(std::fabs(p0t[0]), std::fabs(p1t[0]), std::fabs(p2t[0]));
    float maxYt = max3(std::fabs(p0t[1]), std::fabs(p1t[1]), std::fabs(p2t[1]));
    float deltaX = gammaN(5) * (maxXt + maxZt);
    float deltaY = gammaN(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammaN(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max3(std::fabs(e0), std::fabs(e1), std::fabs(e2));
    float deltaT = 3 * (gammaN(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * std::fabs(invDet);
    if (t <= deltaT) return false;
    *tOut = t; bOut[0] = b0; bOut[1] = b1; bOut[2] = b2;
    return true;
}
// per-triangle rejection of shapes/triangle.cpp:300-315 (what TRI_FLAG_REJECT records at upload time)
inline bool triangleRejected(const mi_scene_desc *d, uint32_t t) {
    const uint32_t *v = d->tri_indices + 3 * (size_t)t;
    const float *p0 = d->P + 3 * (size_t)v[0], *p1 = d->P + 3 * (size_t)v[1], *p2 = d->P + 3 * (size_t)v[2];
    uint32_t mflags = d->meshes[d->tri_mesh[t]].flags;
    float uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
    if (d->UV && (mflags & MI_MESH_HAS_UV)) for (int k = 0; k < 3; ++k) { uv[k][0] = d->UV[2 * (size_t)v[k]]; uv[k][1] = d->UV[2 * (size_t)v[k] + 1]; }
    float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
    float dp02[3], dp12[3];
    for (int k = 0; k < 3; ++k) { dp02[k] = p0[k] - p2[k]; dp12[k] = p1[k] - p2[k]; }
    float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    bool degenerateUV = std::abs(determinant) < 1e-8;
    auto crossLen2 = [](const float a[3], const float b[3]) {
        double ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
        float cx = (float)((ay * bz) - (az * by)), cy = (float)((az * bx) - (ax * bz)), cz = (float)((ax * by) - (ay * bx));
        return cx * cx + cy * cy + cz * cz;
    };
    bool needNg = degenerateUV;
    if (!degenerateUV) {
        float invdet = 1 / determinant, dpdu[3], dpdv[3];
        for (int k = 0; k < 3; ++k) {
            dpdu[k] = (duv12[1] * dp02[k] - duv02[1] * dp12[k]) * invdet;
            dpdv[k] = (-duv12[0] * dp02[k] + duv02[0] * dp12[k]) * invdet;
        }
        if (crossLen2(dpdu, dpdv) == 0) needNg = true;
    }
    if (needNg) {
        float a[3], b[3];
        for (int k = 0; k < 3; ++k) { a[k] = p2[k] - p0[k]; b[k] = p1[k] - p0[k]; }
        if (crossLen2(a, b) == 0) return true;
    }
    return false;
}

}  // namespace hostcheck
