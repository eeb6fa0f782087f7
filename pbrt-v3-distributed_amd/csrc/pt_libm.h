// The float libm routines of the path, restated so that the device returns what the REFERENCE's build returns.
//
// Provenance / licence: the ALGORITHMS and constants below restate the GNU C Library 2.35 (sysdeps/ieee754/flt-32/{s_sinf,s_cosf,e_expf,e_logf,e_acosf,
// s_atanf,e_atan2f}.c and s_sincosf.h; glibc is a THIRD party, not the reference).  glibc is distributed under the GNU Lesser General Public License
// v2.1 or later; sinf / cosf / sincosf / expf / logf there derive from ARM's Optimized Routines (MIT / Apache-2.0 WITH LLVM-exception, contributed to
// glibc under the LGPL) and acosf / atanf / atan2f from Sun Microsystems' FDLIBM ("Copyright (C) 1993 by Sun Microsystems, Inc.  Permission to use, copy,
// modify, and distribute this software is freely granted, provided that this notice is preserved").  No glibc source text is copied here: the code was
// written against the published algorithms and checked against the installed binary's bits (see "Proof" below).
//
// pbrt-v3 calls std::sin / cos / acos / atan2 / exp / log on `Float` = float (core/sampling.cpp:93-150, core/reflection.cpp:572,
// core/microfacet.cpp:146-163, core/geometry.h:1463-1486, shapes/sphere.cpp:85-120, lights/infinite.cpp:109-172, core/texture.h:95,
// media/homogeneous.cpp:56, media/grid.cpp:76-104, core/medium.cpp, textures/marble.h:66).  Compiled here (oracle/ref_build, g++ -O2) those
// are calls into the image's glibc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11), x86-64, whose IFUNC resolvers pick the FMA builds of
// sinf / cosf / sincosf / expf / logf on every CPU with AVX2 + FMA (sysdeps/x86_64/fpu/multiarch/ifunc-fma.h) and the generic
// SSE2 builds of acosf / atanf / atan2f.  None of them is correctly rounded: sinf differs from the correctly rounded value on
// 1.3 % of its inputs, which 30 specular bounces amplify into different paths (C4).  So each routine below performs
// the SAME operation sequence as the installed binary -- algorithm from glibc's published sources (sysdeps/ieee754/flt-32/
// s_sinf.c, s_cosf.c, s_sincosf.h, e_expf.c, e_logf.c, e_acosf.c, s_atanf.c, e_atan2f.c), the places where the FMA build
// contracts a multiply-add read off the disassembly of libm.so.6, and every constant / table checked against the bytes of the
// installed libm.so.6 by tools/libm_check/gen_tables.py.  IEEE +, -, *, /, sqrt, fma and the conversions are correctly rounded
// on both machines, so equal sequences give equal bits.
//
// Proof: tools/libm_check/check.cpp compiles THIS header for the host and compares with the host's libm over all 2^32 inputs of
// each one-argument routine (and 10^9 pairs for atan2f); tests/test_libm.py runs it (`-m "not gpu"`), and
// test_device_libm_matches_host_glibc (`-m gpu`) runs the device build through the stage entry mi_libm_eval against the GPU
// box's own glibc.  NaN results are compared as "both NaN" (payloads are not part of the contract).
//
// The header compiles for the device (hipcc, PT_DEV = __device__ __forceinline__) and for the host checker (no HIP).
#pragma once
#include <stdint.h>

#ifndef PT_DEV
#define PT_DEV static inline
#define PT_LM_TABLE static const
#else
#define PT_LM_TABLE __device__ static const
#endif

#define PT_LM_FMA(a, b, c) __builtin_fma((a), (b), (c))
PT_DEV uint32_t pt_lm_asu32(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_DEV float pt_lm_asf32(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_DEV uint64_t pt_lm_asu64(double d) { return __builtin_bit_cast(uint64_t, d); }
PT_DEV double pt_lm_asf64(uint64_t u) { return __builtin_bit_cast(double, u); }
PT_DEV float pt_lm_nan() { return pt_lm_asf32(0x7fc00000u); }

// ---------------------------------------------------------------------------------------------------------------------------
// sinf / cosf / sincosf   (s_sinf.c, s_cosf.c, s_sincosf.c, s_sincosf.h, s_sincosf_data.c; FMA build)
// |y| < 0x1p-12: sin = y, cos = 1.  abstop12(y) < abstop12(pi/4) (i.e. |y| < 0.75): polynomials on y itself.  Below 120:
// n = round(y * 2/pi) through a 2^24-scaled truncation, r = y - n * pi/2 as ONE fused operation.  Above: 4/pi from a 192-bit
// table, integer arithmetic.  Quadrant n picks the sine (even) or cosine (odd) polynomial; signs by the sign[] / negated-table
// rule of the source -- an exact negation either way, applied here to the polynomial's value.
PT_LM_TABLE uint32_t pt_lm_inv_pio4[24] = {
    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
    0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

PT_DEV double pt_lm_sin_poly(double x, double x2) {   // sinf_poly, (n & 1) == 0
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double t = PT_LM_FMA(s3, x2, s2);
    const double x7 = x3 * x2;
    const double s = PT_LM_FMA(x3, s1, x);
    return PT_LM_FMA(t, x7, s);
}
PT_DEV double pt_lm_cos_poly(double x2) {   // sinf_poly, (n & 1) == 1, table 0
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double hi = PT_LM_FMA(c4, x2, c3);
    const double lo = PT_LM_FMA(c1, x2, c0);
    const double x6 = x4 * x2;
    const double c = PT_LM_FMA(x4, c2, lo);
    return PT_LM_FMA(hi, x6, c);
}
// Reduction: returns r, sets n (polynomial choice) and q (sign / table choice).  class: 0 tiny, 1 valid, 2 invalid (inf / NaN).
PT_DEV int pt_lm_sincos_reduce(float y, double *xr, int *np, int *qp) {
    const uint32_t iy = pt_lm_asu32(y), top = (iy >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {                       // abstop12(y) < abstop12(pio4)
        if (top < 0x398) return 0;           // |y| < 0x1p-12
        *xr = x; *np = 0; *qp = 0;
        return 1;
    }
    if (top < 0x42f) {                       // < 120: reduce_fast
        const double r = x * 0x1.45F306DC9C883p+23;
        const int n = ((int32_t)r + 0x800000) >> 24;
        *xr = PT_LM_FMA(-(double)n, 0x1.921FB54442D18p0, x);
        *np = n; *qp = n;
        return 1;
    }
    if (top < 0x7f8) {                       // reduce_large
        const uint32_t *arr = &pt_lm_inv_pio4[(iy >> 26) & 15];
        const int shift = (iy >> 23) & 7;
        uint32_t xi = (iy & 0xffffff) | 0x800000;
        xi <<= shift;
        uint64_t res0 = (uint32_t)(xi * arr[0]);
        const uint64_t res1 = (uint64_t)xi * arr[4], res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t n = (res0 + (1ULL << 61)) >> 62;
        res0 -= n << 62;
        *xr = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
        *np = (int)n; *qp = (int)n + (int)(iy >> 31);
        return 1;
    }
    return 2;
}
PT_DEV float pt_sinf(float y) {
    double x; int n, q;
    const int cls = pt_lm_sincos_reduce(y, &x, &n, &q);
    if (cls == 0) return y;
    if (cls == 2) return pt_lm_nan();
    const double x2 = x * x;
    if ((n & 1) == 0) { const double v = pt_lm_sin_poly(x, x2); return (float)((((q & 3) == 1) | ((q & 3) == 2)) ? -v : v); }
    const double v = pt_lm_cos_poly(x2);
    return (float)((q & 2) ? -v : v);
}
PT_DEV float pt_cosf(float y) {
    double x; int n, q;
    const int cls = pt_lm_sincos_reduce(y, &x, &n, &q);
    if (cls == 0) return 1.0f;
    if (cls == 2) return pt_lm_nan();
    const double x2 = x * x;
    if ((n & 1) == 1) { const double v = pt_lm_sin_poly(x, x2); return (float)((((q & 3) == 1) | ((q & 3) == 2)) ? -v : v); }
    const double v = pt_lm_cos_poly(x2);
    return (float)((q & 2) ? -v : v);
}
// One reduction, both polynomials: what g++ -O2 turns a sin(x), cos(x) pair of the reference into (sincosf; same values as the two calls).
PT_DEV void pt_sincosf(float y, float *sp, float *cp) {
    double x; int n, q;
    const int cls = pt_lm_sincos_reduce(y, &x, &n, &q);
    if (cls == 0) { *sp = y; *cp = 1.0f; return; }
    if (cls == 2) { *sp = pt_lm_nan(); *cp = pt_lm_nan(); return; }
    const double x2 = x * x;
    const double S = pt_lm_sin_poly(x, x2), C = pt_lm_cos_poly(x2);
    const double sv = (((q & 3) == 1) | ((q & 3) == 2)) ? -S : S, cv = (q & 2) ? -C : C;
    const bool odd = n & 1;
    *sp = (float)(odd ? cv : sv);
    *cp = (float)(odd ? sv : cv);
}

// ---------------------------------------------------------------------------------------------------------------------------
// expf   (e_expf.c + e_exp2f_data.c, EXP2F_TABLE_BITS = 5; FMA build: z = InvLn2N * x is never rounded on its own -- both its
// uses, z + SHIFT and z - kd, are fused)
PT_LM_TABLE uint64_t pt_lm_exp2f_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
PT_DEV float pt_expf(float x) {
    const uint32_t ix = pt_lm_asu32(x), abstop = (ix >> 20) & 0x7ff;
    if (abstop > 0x42a) {                                    // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop > 0x7f7) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_huge_valf();  // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                  // underflow
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;             // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f
    }
    const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const double xd = (double)x;
    double kd = PT_LM_FMA(InvLn2N, xd, SHIFT);
    const uint64_t ki = pt_lm_asu64(kd);
    kd -= SHIFT;
    const double r = PT_LM_FMA(InvLn2N, xd, -kd);
    const double s = pt_lm_asf64(pt_lm_exp2f_tab[ki & 31] + (ki << 47));
    const double z = PT_LM_FMA(C0, r, C1);
    const double r2 = r * r;
    double y = PT_LM_FMA(C2, r, 1.0);
    y = PT_LM_FMA(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---------------------------------------------------------------------------------------------------------------------------
// logf   (e_logf.c + e_logf_data.c, LOGF_TABLE_BITS = 4; FMA build)
PT_LM_TABLE double pt_lm_logf_tab[32] = {   // {invc, logc} x 16
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2,
    0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5,
    0x1p+0, 0x0p+0, 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3, 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,
    0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};
PT_DEV float pt_logf(float x) {
    uint32_t ix = pt_lm_asu32(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, inf or NaN
        if (ix * 2 == 0) return -__builtin_huge_valf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return pt_lm_nan();
        ix = pt_lm_asu32(x * 0x1p23f);                        // subnormal: normalise
        ix -= 23u << 23;
    }
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = pt_lm_logf_tab[2 * i], logc = pt_lm_logf_tab[2 * i + 1];
    const double z = (double)pt_lm_asf32(iz);
    const double r = PT_LM_FMA(z, invc, -1.0);
    const double y0 = PT_LM_FMA((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = PT_LM_FMA(A1, r, A2);
    y = PT_LM_FMA(A0, r2, y);
    y = PT_LM_FMA(y, r2, y0 + r);
    return (float)y;
}

// ---------------------------------------------------------------------------------------------------------------------------
// acosf   (e_acosf.c: the fdlibm routine in float arithmetic; generic build, no contraction)
PT_DEV float pt_acosf(float x) {
    const float one = 1.0f, pi = pt_lm_asf32(0x40490fda), pio2_hi = pt_lm_asf32(0x3fc90fda), pio2_lo = pt_lm_asf32(0x33a22168);
    const float pS0 = pt_lm_asf32(0x3e2aaaab), pS1 = pt_lm_asf32(0xbea6b090), pS2 = pt_lm_asf32(0x3e4e0aa8), pS3 = pt_lm_asf32(0xbd241146);
    const float pS4 = pt_lm_asf32(0x3a4f7f04), pS5 = pt_lm_asf32(0x3811ef08);
    const float qS1 = pt_lm_asf32(0xc019d139), qS2 = pt_lm_asf32(0x4001572d), qS3 = pt_lm_asf32(0xbf303361), qS4 = pt_lm_asf32(0x3d9dc62e);
    const int32_t hx = (int32_t)pt_lm_asu32(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) {                                   // |x| == 1
        if (hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    }
    if (ix > 0x3f800000) return pt_lm_nan();                  // |x| > 1 or NaN: (x - x) / (x - x)
    if (ix < 0x3f000000) {                                    // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - r * x));
    }
    if (hx < 0) {                                             // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = __builtin_sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;                         // x > 0.5
    const float s = __builtin_sqrtf(z);
    const float df = pt_lm_asf32(pt_lm_asu32(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// ---------------------------------------------------------------------------------------------------------------------------
// atanf, atan2f   (s_atanf.c, e_atan2f.c: fdlibm in float arithmetic; generic build)
PT_DEV float pt_atanf(float x) {
    const float aT0 = pt_lm_asf32(0x3eaaaaab), aT1 = pt_lm_asf32(0xbe4ccccd), aT2 = pt_lm_asf32(0x3e124925), aT3 = pt_lm_asf32(0xbde38e38);
    const float aT4 = pt_lm_asf32(0x3dba2e6e), aT5 = pt_lm_asf32(0xbd9d8795), aT6 = pt_lm_asf32(0x3d886b35), aT7 = pt_lm_asf32(0xbd6ef16b);
    const float aT8 = pt_lm_asf32(0x3d4bda59), aT9 = pt_lm_asf32(0xbd15a221), aT10 = pt_lm_asf32(0x3c8569d7);
    const float one = 1.0f;
    const int32_t hx = (int32_t)pt_lm_asu32(x), ix = hx & 0x7fffffff;
    float hi, lo;
    int id;
    if (ix >= 0x4c000000) {                                   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        const float ahi = pt_lm_asf32(0x3fc90fda), alo = pt_lm_asf32(0x33a22168);
        return hx > 0 ? ahi + alo : -ahi - alo;
    }
    if (ix < 0x3ee00000) {                                    // |x| < 0.4375
        if (ix < 0x31000000) return x;                        // |x| < 2^-29
        id = -1; hi = 0; lo = 0;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {                                // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); hi = pt_lm_asf32(0x3eed6338); lo = pt_lm_asf32(0x31ac3769); }
            else { id = 1; x = (x - one) / (x + one); hi = pt_lm_asf32(0x3f490fda); lo = pt_lm_asf32(0x33222168); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); hi = pt_lm_asf32(0x3f7b985e); lo = pt_lm_asf32(0x33140fb4); }
            else { id = 3; x = -1.0f / x; hi = pt_lm_asf32(0x3fc90fda); lo = pt_lm_asf32(0x33a22168); }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -r : r;
}
PT_DEV float pt_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = pt_lm_asf32(0x3f490fdb), pi_o_2 = pt_lm_asf32(0x3fc90fdb), pi = pt_lm_asf32(0x40490fdb), pi_lo = pt_lm_asf32(0xb3bbbd2e);
    const int32_t hx = (int32_t)pt_lm_asu32(x), ix = hx & 0x7fffffff, hy = (int32_t)pt_lm_asu32(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return pt_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            if (m == 2) return 3.0f * pi_o_4 + tiny;
            return -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = pt_atanf(__builtin_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return pt_lm_asf32(pt_lm_asu32(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
