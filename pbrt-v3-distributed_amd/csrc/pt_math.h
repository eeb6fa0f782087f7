// Device-side scalar/vector math for the gfx950 wavefront path tracer.
// Every routine reproduces the reference's float semantics (operation order, the places where
// pbrt computes in double, reciprocal-multiply divisions, std::min/max NaN behaviour); the build
// uses -ffp-contract=off and HIP's correctly-rounded fp32 divide/sqrt, so +,-,*,/,sqrt agree
// bit-for-bit with the reference's x86-64 SSE2 code.  Reference lines are cited per function.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
// ---- developer instrumentation (PT_SHADE_PROF builds only): wave time between consecutive PROBE(k) points of k_shade, k = 0 .. 23; usable from any device routine
// the kernel calls (round 6: the accumulators are file-scope LDS, one set per wave of the block).  All memory is drained at each probe; the sums stay per wave in LDS and
// reach the global counters once, when the block ends -- rounds 2-5 did two global atomics per probe and wave, whose contention (every wave of the chip on the same two
// words) was most of what the profile measured: every phase cost about the same and the kernel ran 3-7 x slower than unprofiled (profiles/r06_d_*).
#ifndef PT_SHADE_PROF
#define PT_SHADE_PROF 0
#endif
#if PT_SHADE_PROF
__shared__ long long s_prof[4];
__shared__ unsigned long long s_pacc[4][24], s_pcnt[4][24], s_plan[4][24];   // cycles, visits, active lanes at the probe
#define PROBE(k)                                                                                     \
    {                                                                                                \
        __builtin_amdgcn_s_waitcnt(0);                                                               \
        unsigned long long pm_ = __ballot(1);                                                        \
        if (__lane_id() == (uint32_t)(__ffsll((long long)pm_) - 1)) {                                \
            long long now_ = clock64();                                                              \
            s_pacc[(threadIdx.x >> 6) & 3][(k)] += (unsigned long long)(now_ - s_prof[(threadIdx.x >> 6) & 3]);  \
            s_pcnt[(threadIdx.x >> 6) & 3][(k)] += 1ull;                                             \
            s_plan[(threadIdx.x >> 6) & 3][(k)] += (unsigned long long)__popcll(pm_);                \
            s_prof[(threadIdx.x >> 6) & 3] = now_;                                                   \
        }                                                                                            \
    }
#else
#define PROBE(k)
#endif

#include "pt_libm.h"

typedef float Float;
#define PT_PI 3.14159265358979323846f        /* core/pbrt.h:201 (rounded to float there too) */
#define PT_INV_PI 0.31830988618379067154f
#define PT_PI_OVER2 1.57079632679489661923f
#define PT_PI_OVER4 0.78539816339744830961f
#define PT_INFINITY __builtin_huge_valf()
#define PT_MACHINE_EPS 5.9604644775390625e-08f /* numeric_limits<float>::epsilon() * 0.5, core/pbrt.h:197 */
#define PT_SHADOW_EPS 0.0001f                  /* core/pbrt.h:200 */
#define PT_ONE_MINUS_EPS 0x1.fffffep-1f        /* core/rng.h:52 */

// gamma(n), core/pbrt.h:285, evaluated in float exactly as there
PT_DEV Float gamma_n(int n) { return (n * PT_MACHINE_EPS) / (1 - n * PT_MACHINE_EPS); }

// std::min / std::max semantics (second argument wins only if strictly smaller / larger)
PT_DEV Float mn(Float a, Float b) { return (b < a) ? b : a; }
PT_DEV Float mx(Float a, Float b) { return (a < b) ? b : a; }
PT_DEV int mni(int a, int b) { return (b < a) ? b : a; }
PT_DEV Float clampf(Float v, Float lo, Float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // core/pbrt.h:303-311
PT_DEV Float absf(Float v) { return __builtin_fabsf(v); }
PT_DEV Float sqrtf_(Float v) { return __builtin_sqrtf(v); }

// libm calls: the reference's std::sin / cos / acos / atan2 / exp / log on floats are glibc's float routines; pt_libm.h performs
// the same operation sequences, so the device returns the same bits (all 2^32 inputs of each routine checked against the
// installed libm: tools/libm_check, tests/test_libm.py).
#ifndef PT_FAST_MATH
#define PT_FAST_MATH 0
#endif
#if PT_FAST_MATH
// MEASUREMENT BUILD ONLY (`make variant NAME=fastmath FLAGS="-DPT_FAST_MATH=1 ..."`, never built by build()): the hardware's transcendental
// instructions (v_sin / v_cos / v_exp / v_log through the __*f intrinsics, OCML's float acos / atan2) instead of the glibc-identical fp64 polynomials -- what
// bit-identity with the reference's libm costs the shading kernels.  profiles/r06_*_fast_math_study.txt holds the result and the decision.
PT_DEV Float sinf_(Float v) { return __sinf(v); }
PT_DEV Float cosf_(Float v) { return __cosf(v); }
PT_DEV void sincosf_(Float v, Float *s, Float *c) { *s = __sinf(v); *c = __cosf(v); }
PT_DEV Float acosf_(Float v) { return acosf(v); }
PT_DEV Float expf_(Float v) { return __expf(v); }
PT_DEV Float logf_(Float v) { return __logf(v); }
PT_DEV Float atan2f_(Float y, Float x) { return atan2f(y, x); }
#else
PT_DEV Float sinf_(Float v) { return pt_sinf(v); }
PT_DEV Float cosf_(Float v) { return pt_cosf(v); }
PT_DEV void sincosf_(Float v, Float *s, Float *c) { pt_sincosf(v, s, c); }
PT_DEV Float acosf_(Float v) { return pt_acosf(v); }
PT_DEV Float expf_(Float v) { return pt_expf(v); }
PT_DEV Float logf_(Float v) { return pt_logf(v); }
PT_DEV Float atan2f_(Float y, Float x) { return pt_atan2f(y, x); }
#endif
// sin and cos of a DOUBLE (the one place the reference calls the double overloads on this path: TrowbridgeReitzSample11's first branch,
// core/microfacet.cpp:243-248, |x| <= 2 pi): two-term Cody-Waite reduction by pi/2 and the two fdlibm kernels (error < 1 ulp of double; the
// product with r is rounded to float afterwards, so a last-place difference from glibc's own < 1 ulp sin / cos shows in ~1e-9 of the cases).
PT_DEV void SinCosD(double x, double *sn, double *cs) {
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
    const int n = (int)fn;
    const double r = (x - fn * 1.57079632673412561417e+00) - fn * 6.07710050650619224932e-11;   // pi/2 = pio2_1 (33 bits) + pio2_1t
    const double z = r * r, v = z * r;
    const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + v * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double c = 1.0 - (0.5 * z - z * rc);
    const bool swap = n & 1;
    const double a = swap ? c : s, b = swap ? s : c;
    *sn = (n & 2) ? -a : a;
    *cs = ((n + 1) & 2) ? -b : b;
}

PT_DEV uint32_t f2u(Float f) { return __float_as_uint(f); }
PT_DEV Float u2f(uint32_t u) { return __uint_as_float(u); }
PT_DEV Float NextFloatUp(Float v) {   // core/pbrt.h:237-248
    if (__builtin_isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2u(v);
    if (v >= 0) ++ui; else --ui;
    return u2f(ui);
}
PT_DEV Float NextFloatDown(Float v) {   // core/pbrt.h:250-261
    if (__builtin_isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2u(v);
    if (v > 0) --ui; else ++ui;
    return u2f(ui);
}

// Wave-uniform values.  readfirstlane puts the value of the first active lane into an SGPR; the empty asm makes it
// opaque, so that the compiler cannot "simplify" it back to the (equal, but per-lane) VGPR value inside an
// `if (v == uniform)` block -- which would turn scalar loads (s_load, scalar cache) into per-lane vector loads.
PT_DEV int Opaque(int s) { asm volatile("" : "+s"(s)); return s; }
PT_DEV int UniformInt(int v) { return Opaque(__builtin_amdgcn_readfirstlane(v)); }
// v == u, computed so that the compiler learns no equality it could use to substitute v for u
PT_DEV bool SameAs(int v, int u) { int d = v - u; asm volatile("" : "+v"(d)); return d == 0; }
template <typename T> PT_DEV const T *UniformPtr(const T *p) {
    unsigned long long a = (unsigned long long)p;
    int lo = UniformInt((int)(uint32_t)a), hi = UniformInt((int)(uint32_t)(a >> 32));
    return (const T *)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo);
}

struct V3 {
    Float x, y, z;
    PT_DEV V3() : x(0), y(0), z(0) {}
    PT_DEV V3(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    PT_DEV Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    PT_DEV V3 operator+(const V3 &v) const { return V3(x + v.x, y + v.y, z + v.z); }
    PT_DEV V3 operator-(const V3 &v) const { return V3(x - v.x, y - v.y, z - v.z); }
    PT_DEV V3 operator-() const { return V3(-x, -y, -z); }
    PT_DEV V3 operator*(Float s) const { return V3(x * s, y * s, z * s); }
    PT_DEV V3 operator/(Float f) const { Float inv = (Float)1 / f; return V3(x * inv, y * inv, z * inv); }   // core/geometry.h:244-248
    PT_DEV Float LengthSquared() const { return x * x + y * y + z * z; }
    PT_DEV Float Length() const { return sqrtf_(LengthSquared()); }
};
PT_DEV V3 operator*(Float s, const V3 &v) { return V3(v.x * s, v.y * s, v.z * s); }
PT_DEV Float Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV Float AbsDot(const V3 &a, const V3 &b) { return absf(Dot(a, b)); }
PT_DEV V3 Abs(const V3 &v) { return V3(absf(v.x), absf(v.y), absf(v.z)); }
PT_DEV V3 Normalize(const V3 &v) { return v / v.Length(); }
PT_DEV V3 Cross(const V3 &a, const V3 &b) {   // core/geometry.h:957-963: double arithmetic, one rounding
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx)));
}
PT_DEV V3 Faceforward(const V3 &n, const V3 &v) { return (Dot(n, v) < 0.f) ? -n : n; }   // core/geometry.h:1213
PT_DEV Float DistanceSquared(const V3 &a, const V3 &b) { return (a - b).LengthSquared(); }
PT_DEV Float MaxComponent(const V3 &v) { return mx(v.x, mx(v.y, v.z)); }
PT_DEV void CoordinateSystem(const V3 &v1, V3 *v2, V3 *v3) {   // core/geometry.h:1020-1027
    if (absf(v1.x) > absf(v1.y)) *v2 = V3(-v1.z, 0, v1.x) / sqrtf_(v1.x * v1.x + v1.z * v1.z);
    else *v2 = V3(0, v1.z, -v1.y) / sqrtf_(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
PT_DEV V3 OffsetRayOrigin(const V3 &p, const V3 &pError, const V3 &n, const V3 &w) {   // core/geometry.h:1440-1460
    Float d = Dot(Abs(n), pError);
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    if (offset.x > 0) po.x = NextFloatUp(po.x); else if (offset.x < 0) po.x = NextFloatDown(po.x);
    if (offset.y > 0) po.y = NextFloatUp(po.y); else if (offset.y < 0) po.y = NextFloatDown(po.y);
    if (offset.z > 0) po.z = NextFloatUp(po.z); else if (offset.z < 0) po.z = NextFloatDown(po.z);
    return po;
}

struct RGB {   // RGBSpectrum, core/spectrum.h:429-488: per-component true arithmetic incl. division
    Float r, g, b;
    PT_DEV RGB() : r(0), g(0), b(0) {}
    PT_DEV explicit RGB(Float v) : r(v), g(v), b(v) {}
    PT_DEV RGB(Float r, Float g, Float b) : r(r), g(g), b(b) {}
    PT_DEV bool IsBlack() const { return r == 0 && g == 0 && b == 0; }
    PT_DEV RGB operator+(const RGB &o) const { return RGB(r + o.r, g + o.g, b + o.b); }
    PT_DEV RGB operator-(const RGB &o) const { return RGB(r - o.r, g - o.g, b - o.b); }
    PT_DEV RGB operator-() const { return RGB(-r, -g, -b); }
    PT_DEV RGB operator*(const RGB &o) const { return RGB(r * o.r, g * o.g, b * o.b); }
    PT_DEV RGB operator/(const RGB &o) const { return RGB(r / o.r, g / o.g, b / o.b); }
    PT_DEV RGB operator*(Float s) const { return RGB(r * s, g * s, b * s); }
    PT_DEV RGB operator/(Float s) const { return RGB(r / s, g / s, b / s); }
    PT_DEV Float y() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }   // spectrum.h:462-465
    PT_DEV Float MaxComponentValue() const { return mx(r, mx(g, b)); }
    PT_DEV bool HasNaNs() const { return __builtin_isnan(r) || __builtin_isnan(g) || __builtin_isnan(b); }
};
PT_DEV RGB operator*(Float s, const RGB &v) { return RGB(v.r * s, v.g * s, v.b * s); }
PT_DEV RGB SqrtRGB(const RGB &s) { return RGB(sqrtf_(s.r), sqrtf_(s.g), sqrtf_(s.b)); }
PT_DEV RGB rgb3(const float *p) { return RGB(p[0], p[1], p[2]); }
PT_DEV V3 v3(const float *p) { return V3(p[0], p[1], p[2]); }

// ---- sampling helpers
PT_DEV void ConcentricSampleDisk(Float u0, Float u1, Float *dx, Float *dy) {   // core/sampling.cpp:113-130
    Float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    Float theta, r;
    if (absf(ox) > absf(oy)) { r = ox; theta = PT_PI_OVER4 * (oy / ox); }
    else { r = oy; theta = PT_PI_OVER2 - PT_PI_OVER4 * (ox / oy); }
    Float sn, cs;
    sincosf_(theta, &sn, &cs);
    *dx = r * cs;
    *dy = r * sn;
}
PT_DEV V3 CosineSampleHemisphere(Float u0, Float u1) {   // core/sampling.h:159-163
    Float dx, dy;
    ConcentricSampleDisk(u0, u1, &dx, &dy);
    Float z = sqrtf_(mx((Float)0, 1 - dx * dx - dy * dy));
    return V3(dx, dy, z);
}
PT_DEV Float PowerHeuristic(Float fPdf, Float gPdf) {   // core/sampling.h:171-174 with nf = ng = 1
    Float f = 1 * fPdf, g = 1 * gPdf;
    return (f * f) / (f * f + g * g);
}
PT_DEV Float Lerp(Float t, Float v1, Float v2) { return (1 - t) * v1 + t * v2; }
