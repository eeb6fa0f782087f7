// Texture evaluation on the device: Texture<T>::Evaluate for every class of src/textures
// (Ptex excepted), the texture mappings and Perlin noise of core/texture.{h,cpp}, MIPMap trilinear / EWA lookups
// (core/mipmap.h:223-353).  One lane = one evaluation; the node graph of a material is wave-uniform in the shading
// kernel (lanes are sorted by material), so the switches below are scalar branches there.  The texture tables hang
// off a __constant__ block (c_tex): no kernel-argument growth for the kernels of untextured scenes.
// A Float texture is evaluated on an RGB triple with equal channels (every operator involved is componentwise).
#pragma once
#include "pt_scene.h"

struct DevImage {   // mi_image + per-level offsets (in floats) into `texels`
    int32_t width, height, levels, channels, trilinear, wrap;
    float max_aniso;
    uint32_t pad;
    const float *texels;
    uint32_t level_off[16];
};
// What an alpha mask evaluates to at a candidate hit, pre-resolved per mesh at upload (round 5).  The traversal's alpha phase reached a mask through a chain of DEPENDENT
// fetches -- tri_info -> mesh_alpha -> node type -> program offsets -> program steps -> node fields -- before any arithmetic; for the mask kinds real scenes use the record
// below carries everything after the first hop.  kind: 0 no mask, 1 any other texture graph (TexEval as before), 2 a constant, 3 `dots` over the (u, v) mapping with constant
// children, 4 an image map over the (u, v) mapping (point lookup: the hit has no differentials).  The values are EvalNode's, by the same statements.
struct DevMaskFast { int32_t kind, image; float su, sv, du, dv, v_out, v_in; };   // 32 bytes
struct DevTex {
    const DevMaskFast *mask_fast;    // 2 per mesh (alphaMask, shadowAlphaMask), beside mesh_alpha; NULL when no mesh has a mask
    const mi_texture *nodes;
    const DevImage *images;
    const mi_material_desc *descs;   // per material; NULL when no material is textured
    const int32_t *mesh_alpha;       // 2 per mesh; NULL when no mesh has a mask
    const int32_t *prog_off;         // n_nodes + 1: the evaluation program of node i is prog[prog_off[i] .. prog_off[i+1])
    const int4 *prog;                // post-order steps {node, step of tex1, step of tex2, step of amount} (-1: no such child)
    uint32_t n_nodes, n_images;
    mi_camera camera;                // for the ray differentials of camera rays (rebuilt at the first hit)
    int32_t spp, pad;
    float ewa_lut[128];              // MIPMap::weightLut (mipmap.h:187-195), computed on the host with the reference's expression
};
extern __constant__ DevTex c_tex;

struct TexCtx {   // what textures read of a SurfaceInteraction
    V3 p;
    Float u, v;
    V3 dpdx, dpdy;
    Float dudx, dvdx, dudy, dvdy;
};
struct V2 { Float x, y; };

// Wave-uniform evaluation (U = true; built in round 4, measured -2.3 % / -3 % of the textured shading time in profiles/r04_v_*, r05_i_*, on since round 5).  k_shade<..., TEX> runs the material code once per distinct material of a wave
// (waterfall loop), so inside it every active lane evaluates the SAME material, hence the same texture nodes, programs and images -- only the interaction differs
// per lane.  The out-of-line routines below receive their node / image as ordinary (vector-register) arguments, which made every field of a node a 64-lane gather of one
// address.  With U they take the argument from the first active lane and read the tables through the CONSTANT address space: scalar loads, scalar branches.
// U = false: per-lane nodes (alpha masks inside the traversal, k_shade_vol's per-lane materials, mi_texture_eval).
template <bool U, class T> struct UPtr { typedef const T *P; static PT_DEV P of(const T *p) { return p; } };
template <class T> struct UPtr<true, T> {
    typedef const __attribute__((address_space(4))) T *P;
    static PT_DEV P of(const T *p) { return (P)(unsigned long long)UniformPtr(p); }
};
template <bool U> PT_DEV int UIdx(int i) { if constexpr (U) return UniformInt(i); else return i; }
template <class A> PT_DEV RGB Rgb3P(A a) { return RGB(a[0], a[1], a[2]); }
template <class A> PT_DEV V3 V3P(A a) { return V3(a[0], a[1], a[2]); }

PT_DEV Float Log2T(Float x) { const Float invLog2 = 1.442695040888963387004650940071f; return logf_(x) * invLog2; }   // core/pbrt.h:324-327
PT_DEV int ModT(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }                                 // core/pbrt.h:310-313

// ---- Perlin noise (core/texture.cpp:47-220); Ken Perlin's reference permutation ("Improving Noise", SIGGRAPH 2002)
__device__ const uint8_t c_noise_perm[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
// Round 6: the table is read from LDS.  A noise value is 14 table look-ups in three DEPENDENT levels (P[P[P[x] + y] + z] for the eight corners); from global memory every
// level is a per-lane gather at vector-memory latency, and the kernels that evaluate procedural textures run at 2-4 waves per SIMD, so a bump map (twelve noise calls) or
// a `dots` alpha mask inside the traversal (up to three) was latency, not arithmetic.  256 bytes = one dword per LDS bank: no bank conflicts whatever the indices.
// Every kernel that can reach a noise call copies the table at its start (NoiseLdsInit); which kernels those are is read off the compiler's own per-kernel LDS sizes
// (tools/debug/kernel_resources.sh: exactly the instances whose group segment grew by 256 bytes).
__shared__ uint8_t s_noise_perm[256];
PT_DEV void NoiseLdsInit() {
    for (uint32_t k = threadIdx.x; k < 64; k += blockDim.x) ((uint32_t *)s_noise_perm)[k] = ((const uint32_t *)c_noise_perm)[k];
    __syncthreads();
}
PT_DEV int NoisePerm(int i) { return s_noise_perm[i & 255]; }
PT_DEV Float NoiseGrad(int x, int y, int z, Float dx, Float dy, Float dz) {   // texture.cpp:186-192
    int h = NoisePerm(NoisePerm(NoisePerm(x) + y) + z);
    h &= 15;
    Float u = h < 8 || h == 12 || h == 13 ? dx : dy;
    Float v = h < 4 || h == 12 || h == 13 ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
PT_DEV Float NoiseWeight(Float t) { Float t3 = t * t * t; Float t4 = t3 * t; return 6 * t4 * t - 15 * t4 + 10 * t3; }
__device__ __noinline__ Float Noise3(Float x, Float y, Float z) {   // texture.cpp:155-184
    int ix = (int)__builtin_floorf(x), iy = (int)__builtin_floorf(y), iz = (int)__builtin_floorf(z);
    Float dx = x - ix, dy = y - iy, dz = z - iz;
    ix &= 255; iy &= 255; iz &= 255;
    Float w000 = NoiseGrad(ix, iy, iz, dx, dy, dz);
    Float w100 = NoiseGrad(ix + 1, iy, iz, dx - 1, dy, dz);
    Float w010 = NoiseGrad(ix, iy + 1, iz, dx, dy - 1, dz);
    Float w110 = NoiseGrad(ix + 1, iy + 1, iz, dx - 1, dy - 1, dz);
    Float w001 = NoiseGrad(ix, iy, iz + 1, dx, dy, dz - 1);
    Float w101 = NoiseGrad(ix + 1, iy, iz + 1, dx - 1, dy, dz - 1);
    Float w011 = NoiseGrad(ix, iy + 1, iz + 1, dx, dy - 1, dz - 1);
    Float w111 = NoiseGrad(ix + 1, iy + 1, iz + 1, dx - 1, dy - 1, dz - 1);
    Float wx = NoiseWeight(dx), wy = NoiseWeight(dy), wz = NoiseWeight(dz);
    Float x00 = Lerp(wx, w000, w100), x10 = Lerp(wx, w010, w110), x01 = Lerp(wx, w001, w101), x11 = Lerp(wx, w011, w111);
    Float y0 = Lerp(wy, x00, x10), y1 = Lerp(wy, x01, x11);
    return Lerp(wz, y0, y1);
}
// DotsTexture::Evaluate's decision (textures/dots.h:59-80): is (s, t) inside the cell's dot?
PT_DEV bool DotsInside(Float s, Float t) {
    int sCell = (int)__builtin_floorf(s + .5f), tCell = (int)__builtin_floorf(t + .5f);
    if (Noise3(sCell + .5f, tCell + .5f, .5f) > 0) {
        Float radius = .35f;
        Float maxShift = 0.5f - radius;
        Float sCenter = sCell + maxShift * Noise3(sCell + 1.5f, tCell + 2.8f, .5f);
        Float tCenter = tCell + maxShift * Noise3(sCell + 4.5f, tCell + 9.8f, .5f);
        Float dx = s - sCenter, dy = t - tCenter;
        if (dx * dx + dy * dy < radius * radius) return true;
    }
    return false;
}
PT_DEV Float SmoothStepT(Float lo, Float hi, Float value) {   // texture.cpp:41-44
    Float v = clampf((value - lo) / (hi - lo), 0, 1);
    return v * v * (-2 * v + 3);
}
// FBm (turb == false, texture.cpp:200-217) and Turbulence (turb == true, :219-245)
__device__ __noinline__ Float FBmT(const V3 p, const V3 dpdx, const V3 dpdy, Float omega, int maxOctaves, bool turb) {
    Float len2 = mx(dpdx.LengthSquared(), dpdy.LengthSquared());
    Float n = clampf(-1 - .5f * Log2T(len2), 0, (Float)maxOctaves);
    int nInt = (int)__builtin_floorf(n);
    Float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        Float nz = Noise3(lambda * p.x, lambda * p.y, lambda * p.z);
        sum += o * (turb ? absf(nz) : nz);
        lambda *= 1.99f;
        o *= omega;
    }
    Float nPartial = n - nInt;
    // the partial octave's weight is exactly 0 for nPartial <= .3 -- always on rays without differentials (every ray after the camera's: len2 = 0, n = maxOctaves) -- and the
    // reference then adds o * 0 * noise = +-0 (fbm) or o * ((1 - 0) * .2 + 0 * |noise|) = o * .2 (turbulence): the same bits without the noise call (noise is finite; the sum
    // cannot be -0), one Noise3 of five saved per evaluation, three per bump-mapped vertex
    const Float sPartial = SmoothStepT(.3f, .7f, nPartial);
    const Float nz = sPartial == 0 ? (Float)0 : Noise3(lambda * p.x, lambda * p.y, lambda * p.z);
    if (!turb) return sum + o * sPartial * nz;
    sum += o * Lerp(sPartial, (Float)0.2, absf(nz));
    for (int i = nInt; i < maxOctaves; ++i) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}

// ---- texture mappings (core/texture.cpp:84-153)
template <class M> PT_DEV V3 XfPointT(M m, const V3 &p) {   // core/transform.h:223-234
    Float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    Float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    Float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
template <class M> PT_DEV V3 XfVectorT(M m, const V3 &v) {   // core/transform.h:236-242
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
template <class TP> PT_DEV V2 SphereST(TP t, const V3 &P) {   // SphericalMapping2D::sphere :117-121
    V3 vec = Normalize(XfPointT(t->w2t, P) - V3(0, 0, 0));
    Float theta = acosf_(clampf(vec.z, -1, 1));
    Float phi = atan2f_(vec.y, vec.x);
    phi = (phi < 0) ? (phi + 2 * PT_PI) : phi;
    return V2{theta * PT_INV_PI, phi * 0.15915494309189533577f};
}
template <class TP> PT_DEV V2 CylinderST(TP t, const V3 &P) {   // CylindricalMapping2D::cylinder texture.h:93-96
    V3 vec = Normalize(XfPointT(t->w2t, P) - V3(0, 0, 0));
    return V2{(PT_PI + atan2f_(vec.y, vec.x)) * 0.15915494309189533577f, vec.z};
}
PT_DEV V2 DivV2(const V2 &a, const V2 &b, Float f) { Float inv = (Float)1 / f; return V2{(a.x - b.x) * inv, (a.y - b.y) * inv}; }
struct Map2DOut { V2 st, dstdx, dstdy; };
// (TexCtx travels BY REFERENCE through the out-of-line texture routines since round 5: by value its 22 words were copied to the callee's stack frame at every call --
// TexEval -> EvalNode -> Map2D -- while a routine reads two to six of them; the caller's copy lives in its frame once)
template <bool U> __device__ __noinline__ Map2DOut Map2DAny(const mi_texture *tp, const TexCtx &si) {
    const typename UPtr<U, mi_texture>::P t = UPtr<U, mi_texture>::of(tp);
    Map2DOut o;
    switch (t->mapping) {
    case MI_MAP_SPHERICAL: {   // texture.cpp:96-115
        o.st = SphereST(t, si.p);
        const Float delta = .1f;
        o.dstdx = DivV2(SphereST(t, si.p + delta * si.dpdx), o.st, delta);
        o.dstdy = DivV2(SphereST(t, si.p + delta * si.dpdy), o.st, delta);
        if ((double)o.dstdx.y > .5) o.dstdx.y = 1 - o.dstdx.y; else if (o.dstdx.y < -.5f) o.dstdx.y = -(o.dstdx.y + 1);
        if ((double)o.dstdy.y > .5) o.dstdy.y = 1 - o.dstdy.y; else if (o.dstdy.y < -.5f) o.dstdy.y = -(o.dstdy.y + 1);
        break;
    }
    case MI_MAP_CYLINDRICAL: {   // texture.cpp:123-140
        o.st = CylinderST(t, si.p);
        const Float delta = .01f;
        o.dstdx = DivV2(CylinderST(t, si.p + delta * si.dpdx), o.st, delta);
        if ((double)o.dstdx.y > .5) o.dstdx.y = 1.f - o.dstdx.y; else if (o.dstdx.y < -.5f) o.dstdx.y = -(o.dstdx.y + 1);
        o.dstdy = DivV2(CylinderST(t, si.p + delta * si.dpdy), o.st, delta);
        if ((double)o.dstdy.y > .5) o.dstdy.y = 1.f - o.dstdy.y; else if (o.dstdy.y < -.5f) o.dstdy.y = -(o.dstdy.y + 1);
        break;
    }
    case MI_MAP_PLANAR: {   // texture.cpp:142-148
        V3 vs = V3P(t->vs), vt = V3P(t->vt);
        o.dstdx = V2{Dot(si.dpdx, vs), Dot(si.dpdx, vt)};
        o.dstdy = V2{Dot(si.dpdy, vs), Dot(si.dpdy, vt)};
        o.st = V2{t->du + Dot(si.p, vs), t->dv + Dot(si.p, vt)};
        break;
    }
    default:   // UVMapping2D texture.cpp:86-94
        o.dstdx = V2{t->su * si.dudx, t->sv * si.dvdx};
        o.dstdy = V2{t->su * si.dudy, t->sv * si.dvdy};
        o.st = V2{t->su * si.u + t->du, t->sv * si.v + t->dv};
        break;
    }
    return o;
}
// the (u, v) mapping in line -- what nearly every image map uses; the three other mappings stay out of line
template <bool U> PT_DEV Map2DOut Map2D(const mi_texture *tp, const TexCtx &si) {
    const typename UPtr<U, mi_texture>::P t = UPtr<U, mi_texture>::of(tp);
    if (t->mapping == MI_MAP_UV) {   // UVMapping2D texture.cpp:86-94
        Map2DOut o;
        o.dstdx = V2{t->su * si.dudx, t->sv * si.dvdx};
        o.dstdy = V2{t->su * si.dudy, t->sv * si.dvdy};
        o.st = V2{t->su * si.u + t->du, t->sv * si.v + t->dv};
        return o;
    }
    return Map2DAny<U>(tp, si);
}

// ---- MIPMap<T> (core/mipmap.h:201-353)
template <class IP> PT_DEV int LevW(IP im, int l) { int w = im->width >> l; return w < 1 ? 1 : w; }
template <class IP> PT_DEV int LevH(IP im, int l) { int h = im->height >> l; return h < 1 ? 1 : h; }
template <class IP> PT_DEV RGB MipTexel(IP im, int level, int s, int t) {   // :201-221
    int w = LevW(im, level), h = LevH(im, level);
    if (im->wrap == 0) { s = ModT(s, w); t = ModT(t, h); }
    else if (im->wrap == 2) { s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s); t = t < 0 ? 0 : (t > h - 1 ? h - 1 : t); }
    else if (s < 0 || s >= w || t < 0 || t >= h) return RGB(0.f);
    const float *px = im->texels + im->level_off[level] + ((size_t)t * w + s) * im->channels;
    return im->channels == 1 ? RGB(px[0]) : RGB(px[0], px[1], px[2]);
}
template <bool U> __device__ __noinline__ RGB MipTriangle(const DevImage *imp, int level, Float s_, Float t_) {   // :263-275
    const typename UPtr<U, DevImage>::P im = UPtr<U, DevImage>::of(imp);
    level = level < 0 ? 0 : (level > im->levels - 1 ? im->levels - 1 : level);
    Float s = s_ * LevW(im, level) - 0.5f, t = t_ * LevH(im, level) - 0.5f;
    int s0 = (int)__builtin_floorf(s), t0 = (int)__builtin_floorf(t);
    Float ds = s - s0, dt = t - t0;
    RGB a = MipTexel(im, level, s0, t0), b = MipTexel(im, level, s0, t0 + 1), c = MipTexel(im, level, s0 + 1, t0), d = MipTexel(im, level, s0 + 1, t0 + 1);
    return ((1 - ds) * (1 - dt)) * a + ((1 - ds) * dt) * b + (ds * (1 - dt)) * c + (ds * dt) * d;
}
template <bool U> __device__ __noinline__ RGB MipEWA(const DevImage *imp, int level, V2 st, V2 dst0, V2 dst1) {   // :309-353
    const typename UPtr<U, DevImage>::P im = UPtr<U, DevImage>::of(imp);
    if (level >= im->levels) return MipTexel(im, im->levels - 1, 0, 0);
    int w = LevW(im, level), h = LevH(im, level);
    st.x = st.x * w - 0.5f; st.y = st.y * h - 0.5f;
    dst0.x *= w; dst0.y *= h;
    dst1.x *= w; dst1.y *= h;
    Float A = dst0.y * dst0.y + dst1.y * dst1.y + 1;
    Float B = -2 * (dst0.x * dst0.y + dst1.x * dst1.y);
    Float C = dst0.x * dst0.x + dst1.x * dst1.x + 1;
    Float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    Float det = -B * B + 4 * A * C;
    Float invDet = 1 / det;
    Float uSqrt = sqrtf_(det * C), vSqrt = sqrtf_(A * det);
    int s0 = (int)__builtin_ceilf(st.x - 2 * invDet * uSqrt), s1 = (int)__builtin_floorf(st.x + 2 * invDet * uSqrt);
    int t0 = (int)__builtin_ceilf(st.y - 2 * invDet * vSqrt), t1 = (int)__builtin_floorf(st.y + 2 * invDet * vSqrt);
    RGB sum(0.f);
    Float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        Float tt = it - st.y;
        for (int is = s0; is <= s1; ++is) {
            Float ss = is - st.x;
            Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = mni((int)(r2 * 128), 128 - 1);
                Float weight = c_tex.ewa_lut[index];
                sum = sum + MipTexel(im, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return sum / sumWts;
}
template <bool U> __device__ __noinline__ RGB MipLookup(const DevImage *imp, V2 st, V2 dst0, V2 dst1) {   // :277-307 (and :223-241 for the trilinear case)
    const typename UPtr<U, DevImage>::P im = UPtr<U, DevImage>::of(imp);
    if (im->trilinear) {
        Float width = 2 * mx(mx(absf(dst0.x), absf(dst0.y)), mx(absf(dst1.x), absf(dst1.y)));
        Float level = im->levels - 1 + Log2T(mx(width, (Float)1e-8));
        if (level < 0) return MipTriangle<U>(imp, 0, st.x, st.y);
        else if (level >= im->levels - 1) return MipTexel(im, im->levels - 1, 0, 0);
        int iLevel = (int)__builtin_floorf(level);
        Float delta = level - iLevel;
        return (1 - delta) * MipTriangle<U>(imp, iLevel, st.x, st.y) + delta * MipTriangle<U>(imp, iLevel + 1, st.x, st.y);
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { V2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
    Float majorLength = sqrtf_(dst0.x * dst0.x + dst0.y * dst0.y);
    Float minorLength = sqrtf_(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minorLength * im->max_aniso < majorLength && minorLength > 0) {
        Float scale = majorLength / (minorLength * im->max_aniso);
        dst1.x *= scale; dst1.y *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return MipTriangle<U>(imp, 0, st.x, st.y);
    Float lod = mx((Float)0, im->levels - (Float)1 + Log2T(minorLength));
    int ilod = (int)__builtin_floorf(lod);
    Float d = lod - ilod;
    return (1 - d) * MipEWA<U>(imp, ilod, st, dst0, dst1) + d * MipEWA<U>(imp, ilod + 1, st, dst0, dst1);
}

// ---- Texture<T>::Evaluate.  The node graph below a texture (scale / mix / checkerboard / dots refer to child textures)
// is evaluated WITHOUT recursion: mi_scene_upload flattens the graph below every node into a post-order program
// (c_tex.prog: {node, step of tex1, step of tex2, step of amount}); the loop below evaluates the steps in order into a
// small per-lane value array.  Both children of a checkerboard / dots node are evaluated and one is selected (or the
// two are blended), which gives the value the reference's lazy evaluation gives -- textures have no side effects.
// (A recursive formulation kept every level's live state in registers across the calls: 280 VGPRs at depth 6, one
// wave per SIMD in the shading kernel; the flat loop needs what one node needs.)
#define PT_TEX_MAX_PROG 24
template <bool U> __device__ __noinline__ RGB EvalNode(const mi_texture *tp, const RGB t1v, const RGB t2v, const RGB amtv, const TexCtx &si) {
    const typename UPtr<U, mi_texture>::P t = UPtr<U, mi_texture>::of(tp);
    switch (t->type) {
    case MI_TEX_CONSTANT: return Rgb3P(t->value);                     // constant.h:54
    case MI_TEX_SCALE: return t1v * t2v;                             // scale.h:57-59
    case MI_TEX_MIX: {                                               // mix.h:58-62
        Float amt = amtv.r;
        return (1 - amt) * t1v + amt * t2v;
    }
    case MI_TEX_BILERP: {                                            // bilerp.h:57-62
        Map2DOut m = Map2D<U>(tp, si);
        Float s = m.st.x, tt = m.st.y;
        return ((1 - s) * (1 - tt)) * Rgb3P(t->v00) + ((1 - s) * (tt)) * Rgb3P(t->v01) + ((s) * (1 - tt)) * Rgb3P(t->v10) + ((s) * (tt)) * Rgb3P(t->v11);
    }
    case MI_TEX_IMAGEMAP: {                                          // imagemap.h:87-94
        Map2DOut m = Map2D<U>(tp, si);
        if (t->image < 0 || (uint32_t)t->image >= c_tex.n_images) return RGB(0.f);
        return MipLookup<U>(c_tex.images + t->image, m.st, m.dstdx, m.dstdy);
    }
    case MI_TEX_UV: {                                                // uv.h:54-60
        Map2DOut m = Map2D<U>(tp, si);
        return RGB(m.st.x - __builtin_floorf(m.st.x), m.st.y - __builtin_floorf(m.st.y), 0);
    }
    case MI_TEX_CHECKERBOARD: {
        if (t->dim == 3) {                                           // checkerboard.h:116-126
            V3 p = XfPointT(t->w2t, si.p);
            return (((int)__builtin_floorf(p.x) + (int)__builtin_floorf(p.y) + (int)__builtin_floorf(p.z)) % 2 == 0) ? t1v : t2v;
        }
        Map2DOut m = Map2D<U>(tp, si);                                   // checkerboard.h:63-99
        bool even = (((int)__builtin_floorf(m.st.x) + (int)__builtin_floorf(m.st.y)) % 2 == 0);
        if (t->aa == 0) return even ? t1v : t2v;
        Float ds = mx(absf(m.dstdx.x), absf(m.dstdy.x));
        Float dt = mx(absf(m.dstdx.y), absf(m.dstdy.y));
        Float s0 = m.st.x - ds, s1 = m.st.x + ds;
        Float t0 = m.st.y - dt, t1 = m.st.y + dt;
        if (__builtin_floorf(s0) == __builtin_floorf(s1) && __builtin_floorf(t0) == __builtin_floorf(t1)) return even ? t1v : t2v;
        auto bumpInt = [](Float x) { return (int)__builtin_floorf(x / 2) + 2 * mx(x / 2 - (int)__builtin_floorf(x / 2) - (Float)0.5, (Float)0); };
        Float sint = (bumpInt(s1) - bumpInt(s0)) / (2 * ds);
        Float tint = (bumpInt(t1) - bumpInt(t0)) / (2 * dt);
        Float area2 = sint + tint - 2 * sint * tint;
        if (ds > 1 || dt > 1) area2 = .5f;
        return (1 - area2) * t1v + area2 * t2v;
    }
    case MI_TEX_DOTS: {                                              // dots.h:59-80 (tex1 = outsideDot, tex2 = insideDot)
        Map2DOut m = Map2D<U>(tp, si);
        return DotsInside(m.st.x, m.st.y) ? t2v : t1v;
    }
    case MI_TEX_FBM: case MI_TEX_WRINKLED: {                         // fbm.h:57-61, wrinkled.h:56-60
        V3 P = XfPointT(t->w2t, si.p);
        return RGB(FBmT(P, XfVectorT(t->w2t, si.dpdx), XfVectorT(t->w2t, si.dpdy), t->omega, t->octaves, t->type == MI_TEX_WRINKLED));
    }
    case MI_TEX_WINDY: {                                             // windy.h:55-61
        V3 P = XfPointT(t->w2t, si.p), dpdx = XfVectorT(t->w2t, si.dpdx), dpdy = XfVectorT(t->w2t, si.dpdy);
        Float windStrength = FBmT(.1f * P, .1f * dpdx, .1f * dpdy, .5f, 3, false);
        Float waveHeight = FBmT(P, dpdx, dpdy, .5f, 6, false);
        return RGB(absf(windStrength) * waveHeight);
    }
    case MI_TEX_MARBLE: {                                            // marble.h:60-90
        V3 p = XfPointT(t->w2t, si.p), dpdx = XfVectorT(t->w2t, si.dpdx), dpdy = XfVectorT(t->w2t, si.dpdy);
        p = p * t->scale;
        Float marble = p.y + t->variation * FBmT(p, t->scale * dpdx, t->scale * dpdy, t->omega, t->octaves, false);
        Float tt = .5f + .5f * sinf_(marble);
        const Float c[9][3] = {{.58f, .58f, .6f}, {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.5f, .5f, .5f}, {.6f, .59f, .58f},
                               {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.2f, .2f, .33f}, {.58f, .58f, .6f}};
        const Float NSEG = 6;
        int first = (int)__builtin_floorf(tt * NSEG);
        tt = (tt * NSEG - first);
        first = first > 5 ? 5 : (first < 0 ? 0 : first);
        RGB c0 = rgb3(c[first]), c1 = rgb3(c[first + 1]), c2 = rgb3(c[first + 2]), c3 = rgb3(c[first + 3]);
        RGB s0 = (1.f - tt) * c0 + tt * c1;
        RGB s1 = (1.f - tt) * c1 + tt * c2;
        RGB s2 = (1.f - tt) * c2 + tt * c3;
        s0 = (1.f - tt) * s0 + tt * s1;
        s1 = (1.f - tt) * s1 + tt * s2;
        return 1.5f * ((1.f - tt) * s0 + tt * s1);
    }
    }
    return RGB(0.f);
}
template <bool U = false> __device__ __noinline__ RGB TexEvalAny(int node, const TexCtx &si) {
    node = UIdx<U>(node);
    if (node < 0 || (uint32_t)node >= c_tex.n_nodes) return RGB(0.f);
    const typename UPtr<U, int32_t>::P progOff = UPtr<U, int32_t>::of(c_tex.prog_off);
    const int off = progOff[node], len = progOff[node + 1] - off;
    if (len == 1) return EvalNode<U>(c_tex.nodes + node, RGB(0.f), RGB(0.f), RGB(0.f), si);   // a leaf (the common case: constants, image maps)
    Float val[3 * PT_TEX_MAX_PROG];   // (plain words, left uninitialised: every step is written before a later step reads it)
    auto get = [&](int k) { return k >= 0 ? RGB(val[3 * k], val[3 * k + 1], val[3 * k + 2]) : RGB(0.f); };
    auto put = [&](int k, const RGB &v) { val[3 * k] = v.r; val[3 * k + 1] = v.g; val[3 * k + 2] = v.b; };
    // one step: the three node kinds that are plain arithmetic on their children -- CONSTANT, SCALE, MIX: EvalNode's own statements -- in line, the others out of line
    auto step = [&](int k, int sx, int sy, int sz, int sw) {
        const typename UPtr<U, mi_texture>::P t = UPtr<U, mi_texture>::of(c_tex.nodes + sx);
        const int type = t->type;
        if (type == MI_TEX_CONSTANT) put(k, Rgb3P(t->value));                                   // constant.h:54
        else if (type == MI_TEX_SCALE) put(k, get(sy) * get(sz));                               // scale.h:57-59
        else if (type == MI_TEX_MIX) { const RGB t1v = get(sy), t2v = get(sz); const Float amt = get(sw).r; put(k, (1 - amt) * t1v + amt * t2v); }   // mix.h:58-62
        else put(k, EvalNode<U>(c_tex.nodes + sx, get(sy), get(sz), get(sw), si));
    };
    if constexpr (U) {
        const typename UPtr<U, int32_t>::P prog = UPtr<U, int32_t>::of(reinterpret_cast<const int32_t *>(c_tex.prog));   // int4 steps, read word by word (scalar loads)
        for (int k = 0; k < len; ++k) step(k, prog[4 * (off + k)], prog[4 * (off + k) + 1], prog[4 * (off + k) + 2], prog[4 * (off + k) + 3]);
    } else {
        for (int k = 0; k < len; ++k) { const int4 st = c_tex.prog[off + k]; step(k, st.x, st.y, st.z, st.w); }
    }
    return get(len - 1);
}
// Texture<T>::Evaluate of node `node`.  A CONSTANT node -- most parameters of most materials -- is answered in line from the node table (EvalNode's own statement for it,
// constant.h:54): no call, no frame.  Everything else goes through the out-of-line program loop.
template <bool U = false> PT_DEV RGB TexEval(int node, const TexCtx &si) {
    const int nu = UIdx<U>(node);
    if (nu >= 0 && (uint32_t)nu < c_tex.n_nodes) {
        const typename UPtr<U, mi_texture>::P t = UPtr<U, mi_texture>::of(c_tex.nodes + nu);
        if (t->type == MI_TEX_CONSTANT) return Rgb3P(t->value);
    }
    return TexEvalAny<U>(node, si);
}
