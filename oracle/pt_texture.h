/*
 * pt_texture.h -- TEST INFRASTRUCTURE, part of the oracle (included by pt_oracle.cpp only).
 *
 * CPU restatement of the reference's Texture<T>::Evaluate(const SurfaceInteraction&) for every texture class of
 * src/textures (Ptex excepted), the 2D/3D texture mappings and Perlin noise of src/core/texture.{h,cpp}, and the
 * MIPMap lookups of src/core/mipmap.h, working on the POD node table of include/pbrt_amd.h.  A Float texture is
 * evaluated with the same expressions on an RGB triple whose channels are equal (every operator involved is
 * componentwise), and read back from channel 0.
 */
#pragma once

struct TexCtx {   // what the textures read of a SurfaceInteraction (core/interaction.h:94-157)
    V3 p;
    Float uv[2];
    V3 dpdx, dpdy;
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
};
struct V2 { Float x, y; };

inline Float Log2f_(Float x) { const Float invLog2 = 1.442695040888963387004650940071; return std::log(x) * invLog2; }   // core/pbrt.h:324-327
inline int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }                                       // core/pbrt.h:310-313

// ---- Perlin noise, core/texture.cpp:47-220.  The permutation is Ken Perlin's reference table ("Improving Noise",
// SIGGRAPH 2002), repeated twice as texture.cpp:51-90 stores it.
static const int NoisePermSize = 256;
static const uint8_t NoisePermBase[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
inline int NoisePerm(int i) { return NoisePermBase[i & 255]; }
inline Float NoiseGrad(int x, int y, int z, Float dx, Float dy, Float dz) {   // texture.cpp:186-192
    int h = NoisePerm(NoisePerm(NoisePerm(x) + y) + z);
    h &= 15;
    Float u = h < 8 || h == 12 || h == 13 ? dx : dy;
    Float v = h < 4 || h == 12 || h == 13 ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
inline Float NoiseWeight(Float t) { Float t3 = t * t * t; Float t4 = t3 * t; return 6 * t4 * t - 15 * t4 + 10 * t3; }   // :194-198
inline Float Noise(Float x, Float y = .5f, Float z = .5f) {   // texture.cpp:155-184
    int ix = (int)std::floor(x), iy = (int)std::floor(y), iz = (int)std::floor(z);
    Float dx = x - ix, dy = y - iy, dz = z - iz;
    ix &= NoisePermSize - 1; iy &= NoisePermSize - 1; iz &= NoisePermSize - 1;
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
inline Float NoiseP(const V3 &p) { return Noise(p.x, p.y, p.z); }
inline Float SmoothStep(Float mn, Float mx, Float value) {   // texture.cpp:41-44
    Float v = Clamp((value - mn) / (mx - mn), 0, 1);
    return v * v * (-2 * v + 3);
}
inline Float FBm(const V3 &p, const V3 &dpdx, const V3 &dpdy, Float omega, int maxOctaves) {   // texture.cpp:200-217
    Float len2 = std::max(dpdx.LengthSquared(), dpdy.LengthSquared());
    Float n = Clamp(-1 - .5f * Log2f_(len2), 0, maxOctaves);
    int nInt = (int)std::floor(n);
    Float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        sum += o * NoiseP(lambda * p);
        lambda *= 1.99f;
        o *= omega;
    }
    Float nPartial = n - nInt;
    sum += o * SmoothStep(.3f, .7f, nPartial) * NoiseP(lambda * p);
    return sum;
}
inline Float Turbulence(const V3 &p, const V3 &dpdx, const V3 &dpdy, Float omega, int maxOctaves) {   // texture.cpp:219-245
    Float len2 = std::max(dpdx.LengthSquared(), dpdy.LengthSquared());
    Float n = Clamp(-1 - .5f * Log2f_(len2), 0, maxOctaves);
    int nInt = (int)std::floor(n);
    Float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        sum += o * std::abs(NoiseP(lambda * p));
        lambda *= 1.99f;
        o *= omega;
    }
    Float nPartial = n - nInt;
    sum += o * Lerp(SmoothStep(.3f, .7f, nPartial), (Float)0.2, std::abs(NoiseP(lambda * p)));
    for (int i = nInt; i < maxOctaves; ++i) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}

// ---- texture mappings, core/texture.cpp:84-153
inline V2 SphereST(const mi_texture &t, const V3 &P) {   // SphericalMapping2D::sphere :117-121
    V3 vec = Normalize(XfPoint(t.w2t, P) - V3(0, 0, 0));
    Float theta = std::acos(Clamp(vec.z, -1, 1));           // SphericalTheta geometry.h:1474
    Float phi = std::atan2(vec.y, vec.x);                   // SphericalPhi  geometry.h:1478
    phi = (phi < 0) ? (phi + 2 * Pi) : phi;
    return V2{theta * InvPi, phi * Inv2Pi};
}
inline V2 CylinderST(const mi_texture &t, const V3 &P) {   // CylindricalMapping2D::cylinder texture.h:93-96
    V3 vec = Normalize(XfPoint(t.w2t, P) - V3(0, 0, 0));
    return V2{(Pi + std::atan2(vec.y, vec.x)) * Inv2Pi, vec.z};
}
inline V2 DivV2(const V2 &a, const V2 &b, Float f) { Float inv = (Float)1 / f; return V2{(a.x - b.x) * inv, (a.y - b.y) * inv}; }   // Vector2::operator/ geometry.h:120-124
inline V2 Map2D(const mi_texture &t, const TexCtx &si, V2 *dstdx, V2 *dstdy) {
    switch (t.mapping) {
    case MI_MAP_SPHERICAL: {   // texture.cpp:96-115
        V2 st = SphereST(t, si.p);
        const Float delta = .1f;
        *dstdx = DivV2(SphereST(t, si.p + delta * si.dpdx), st, delta);
        *dstdy = DivV2(SphereST(t, si.p + delta * si.dpdy), st, delta);
        if (dstdx->y > .5) dstdx->y = 1 - dstdx->y; else if (dstdx->y < -.5f) dstdx->y = -(dstdx->y + 1);
        if (dstdy->y > .5) dstdy->y = 1 - dstdy->y; else if (dstdy->y < -.5f) dstdy->y = -(dstdy->y + 1);
        return st;
    }
    case MI_MAP_CYLINDRICAL: {   // texture.cpp:123-140
        V2 st = CylinderST(t, si.p);
        const Float delta = .01f;
        *dstdx = DivV2(CylinderST(t, si.p + delta * si.dpdx), st, delta);
        if (dstdx->y > .5) dstdx->y = 1.f - dstdx->y; else if (dstdx->y < -.5f) dstdx->y = -(dstdx->y + 1);
        *dstdy = DivV2(CylinderST(t, si.p + delta * si.dpdy), st, delta);
        if (dstdy->y > .5) dstdy->y = 1.f - dstdy->y; else if (dstdy->y < -.5f) dstdy->y = -(dstdy->y + 1);
        return st;
    }
    case MI_MAP_PLANAR: {   // texture.cpp:142-148
        V3 vs(t.vs), vt(t.vt);
        *dstdx = V2{Dot(si.dpdx, vs), Dot(si.dpdx, vt)};
        *dstdy = V2{Dot(si.dpdy, vs), Dot(si.dpdy, vt)};
        return V2{t.du + Dot(si.p, vs), t.dv + Dot(si.p, vt)};
    }
    default:   // UVMapping2D texture.cpp:86-94
        *dstdx = V2{t.su * si.dudx, t.sv * si.dvdx};
        *dstdy = V2{t.su * si.dudy, t.sv * si.dvdy};
        return V2{t.su * si.uv[0] + t.du, t.sv * si.uv[1] + t.dv};
    }
}
inline V3 Map3D(const mi_texture &t, const TexCtx &si, V3 *dpdx, V3 *dpdy) {   // IdentityMapping3D::Map texture.cpp:150-155
    *dpdx = XfVector(t.w2t, si.dpdx);
    *dpdy = XfVector(t.w2t, si.dpdy);
    return XfPoint(t.w2t, si.p);
}

// ---- MIPMap<T>, core/mipmap.h:201-353
struct MipView {
    const mi_image &im;
    int lw(int l) const { return std::max(1, im.width >> l); }
    int lh(int l) const { return std::max(1, im.height >> l); }
    size_t off(int l) const { size_t o = 0; for (int i = 0; i < l; ++i) o += (size_t)lw(i) * lh(i) * im.channels; return o; }
    RGB Texel(int level, int s, int t) const {   // :201-221
        int w = lw(level), h = lh(level);
        switch (im.wrap) {
        case 0: s = ModI(s, w); t = ModI(t, h); break;
        case 2: s = Clamp(s, 0, w - 1); t = Clamp(t, 0, h - 1); break;
        default: if (s < 0 || s >= w || t < 0 || t >= h) return RGB(0.f); break;
        }
        const float *px = im.texels + off(level) + ((size_t)t * w + s) * im.channels;
        return im.channels == 1 ? RGB(px[0]) : RGB(px[0], px[1], px[2]);
    }
    RGB triangle(int level, Float s_, Float t_) const {   // :263-275
        level = Clamp(level, 0, im.levels - 1);
        Float s = s_ * lw(level) - 0.5f, t = t_ * lh(level) - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        Float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * Texel(level, s0, t0) + (1 - ds) * dt * Texel(level, s0, t0 + 1) +
               ds * (1 - dt) * Texel(level, s0 + 1, t0) + ds * dt * Texel(level, s0 + 1, t0 + 1);
    }
    RGB LookupWidth(Float s, Float t, Float width) const {   // :223-241
        Float level = im.levels - 1 + Log2f_(std::max(width, (Float)1e-8));
        if (level < 0) return triangle(0, s, t);
        else if (level >= im.levels - 1) return Texel(im.levels - 1, 0, 0);
        int iLevel = (int)std::floor(level);
        Float delta = level - iLevel;
        return (1 - delta) * triangle(iLevel, s, t) + delta * triangle(iLevel + 1, s, t);   // Lerp
    }
    static const Float *WeightLut() {   // :187-195
        static Float lut[128];
        static bool init = false;
        if (!init) {
            for (int i = 0; i < 128; ++i) {
                Float alpha = 2;
                Float r2 = Float(i) / Float(128 - 1);
                lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
            }
            init = true;
        }
        return lut;
    }
    RGB EWA(int level, V2 st, V2 dst0, V2 dst1) const {   // :309-353
        if (level >= im.levels) return Texel(im.levels - 1, 0, 0);
        st.x = st.x * lw(level) - 0.5f; st.y = st.y * lh(level) - 0.5f;
        dst0.x *= lw(level); dst0.y *= lh(level);
        dst1.x *= lw(level); dst1.y *= lh(level);
        Float A = dst0.y * dst0.y + dst1.y * dst1.y + 1;
        Float B = -2 * (dst0.x * dst0.y + dst1.x * dst1.y);
        Float C = dst0.x * dst0.x + dst1.x * dst1.x + 1;
        Float invF = 1 / (A * C - B * B * 0.25f);
        A *= invF; B *= invF; C *= invF;
        Float det = -B * B + 4 * A * C;
        Float invDet = 1 / det;
        Float uSqrt = std::sqrt(det * C), vSqrt = std::sqrt(A * det);
        int s0 = (int)std::ceil(st.x - 2 * invDet * uSqrt), s1 = (int)std::floor(st.x + 2 * invDet * uSqrt);
        int t0 = (int)std::ceil(st.y - 2 * invDet * vSqrt), t1 = (int)std::floor(st.y + 2 * invDet * vSqrt);
        RGB sum(0.f);
        Float sumWts = 0;
        const Float *lut = WeightLut();
        for (int it = t0; it <= t1; ++it) {
            Float tt = it - st.y;
            for (int is = s0; is <= s1; ++is) {
                Float ss = is - st.x;
                Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
                if (r2 < 1) {
                    int index = std::min((int)(r2 * 128), 128 - 1);
                    Float weight = lut[index];
                    sum += Texel(level, is, it) * weight;
                    sumWts += weight;
                }
            }
        }
        return sum / sumWts;
    }
    RGB Lookup(V2 st, V2 dst0, V2 dst1) const {   // :277-307
        if (im.trilinear) {
            Float width = std::max(std::max(std::abs(dst0.x), std::abs(dst0.y)), std::max(std::abs(dst1.x), std::abs(dst1.y)));
            return LookupWidth(st.x, st.y, 2 * width);
        }
        if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
        Float majorLength = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
        Float minorLength = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
        if (minorLength * im.max_aniso < majorLength && minorLength > 0) {
            Float scale = majorLength / (minorLength * im.max_aniso);
            dst1.x *= scale; dst1.y *= scale;
            minorLength *= scale;
        }
        if (minorLength == 0) return triangle(0, st.x, st.y);
        Float lod = std::max((Float)0, im.levels - (Float)1 + Log2f_(minorLength));
        int ilod = (int)std::floor(lod);
        Float d = lod - ilod;
        return (1 - d) * EWA(ilod, st, dst0, dst1) + d * EWA(ilod + 1, st, dst0, dst1);   // Lerp
    }
};

// ---- Texture<T>::Evaluate for every node type
inline RGB TexEval(const mi_scene_desc *d, int node, const TexCtx &si) {
    if (node < 0 || (uint32_t)node >= d->n_textures) return RGB(0.f);
    const mi_texture &t = d->textures[node];
    switch (t.type) {
    case MI_TEX_CONSTANT: return RGB(t.value);                                                      // constant.h:54
    case MI_TEX_SCALE: return TexEval(d, t.tex1, si) * TexEval(d, t.tex2, si);                      // scale.h:57-59
    case MI_TEX_MIX: {                                                                              // mix.h:58-62
        RGB t1 = TexEval(d, t.tex1, si), t2 = TexEval(d, t.tex2, si);
        Float amt = TexEval(d, t.amount, si).c[0];
        return (1 - amt) * t1 + amt * t2;
    }
    case MI_TEX_BILERP: {                                                                           // bilerp.h:57-62
        V2 dx, dy;
        V2 st = Map2D(t, si, &dx, &dy);
        return (1 - st.x) * (1 - st.y) * RGB(t.v00) + (1 - st.x) * (st.y) * RGB(t.v01) + (st.x) * (1 - st.y) * RGB(t.v10) + (st.x) * (st.y) * RGB(t.v11);
    }
    case MI_TEX_IMAGEMAP: {                                                                         // imagemap.h:87-94
        V2 dx, dy;
        V2 st = Map2D(t, si, &dx, &dy);
        if (t.image < 0 || (uint32_t)t.image >= d->n_images) return RGB(0.f);
        return MipView{d->images[t.image]}.Lookup(st, dx, dy);
    }
    case MI_TEX_UV: {                                                                               // uv.h:54-60
        V2 dx, dy;
        V2 st = Map2D(t, si, &dx, &dy);
        return RGB(st.x - std::floor(st.x), st.y - std::floor(st.y), 0);
    }
    case MI_TEX_CHECKERBOARD: {
        if (t.dim == 3) {                                                                           // checkerboard.h:116-126
            V3 dpdx, dpdy;
            V3 p = Map3D(t, si, &dpdx, &dpdy);
            if (((int)std::floor(p.x) + (int)std::floor(p.y) + (int)std::floor(p.z)) % 2 == 0) return TexEval(d, t.tex1, si);
            return TexEval(d, t.tex2, si);
        }
        V2 dstdx, dstdy;                                                                            // checkerboard.h:63-99
        V2 st = Map2D(t, si, &dstdx, &dstdy);
        auto point = [&]() { return (((int)std::floor(st.x) + (int)std::floor(st.y)) % 2 == 0) ? TexEval(d, t.tex1, si) : TexEval(d, t.tex2, si); };
        if (t.aa == 0) return point();
        Float ds = std::max(std::abs(dstdx.x), std::abs(dstdy.x));
        Float dt = std::max(std::abs(dstdx.y), std::abs(dstdy.y));
        Float s0 = st.x - ds, s1 = st.x + ds;
        Float t0 = st.y - dt, t1 = st.y + dt;
        if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1)) return point();
        auto bumpInt = [](Float x) { return (int)std::floor(x / 2) + 2 * std::max(x / 2 - (int)std::floor(x / 2) - (Float)0.5, (Float)0); };
        Float sint = (bumpInt(s1) - bumpInt(s0)) / (2 * ds);
        Float tint = (bumpInt(t1) - bumpInt(t0)) / (2 * dt);
        Float area2 = sint + tint - 2 * sint * tint;
        if (ds > 1 || dt > 1) area2 = .5f;
        return (1 - area2) * TexEval(d, t.tex1, si) + area2 * TexEval(d, t.tex2, si);
    }
    case MI_TEX_DOTS: {                                                                             // dots.h:59-80 (tex1 = outsideDot, tex2 = insideDot)
        V2 dstdx, dstdy;
        V2 st = Map2D(t, si, &dstdx, &dstdy);
        int sCell = (int)std::floor(st.x + .5f), tCell = (int)std::floor(st.y + .5f);
        if (Noise(sCell + .5f, tCell + .5f) > 0) {
            Float radius = .35f;
            Float maxShift = 0.5f - radius;
            Float sCenter = sCell + maxShift * Noise(sCell + 1.5f, tCell + 2.8f);
            Float tCenter = tCell + maxShift * Noise(sCell + 4.5f, tCell + 9.8f);
            Float dx = st.x - sCenter, dy = st.y - tCenter;
            if (dx * dx + dy * dy < radius * radius) return TexEval(d, t.tex2, si);
        }
        return TexEval(d, t.tex1, si);
    }
    case MI_TEX_FBM: {                                                                              // fbm.h:57-61
        V3 dpdx, dpdy;
        V3 P = Map3D(t, si, &dpdx, &dpdy);
        return RGB(FBm(P, dpdx, dpdy, t.omega, t.octaves));
    }
    case MI_TEX_WRINKLED: {                                                                         // wrinkled.h:56-60
        V3 dpdx, dpdy;
        V3 P = Map3D(t, si, &dpdx, &dpdy);
        return RGB(Turbulence(P, dpdx, dpdy, t.omega, t.octaves));
    }
    case MI_TEX_WINDY: {                                                                            // windy.h:55-61
        V3 dpdx, dpdy;
        V3 P = Map3D(t, si, &dpdx, &dpdy);
        Float windStrength = FBm(.1f * P, .1f * dpdx, .1f * dpdy, .5, 3);
        Float waveHeight = FBm(P, dpdx, dpdy, .5, 6);
        return RGB(std::abs(windStrength) * waveHeight);
    }
    case MI_TEX_MARBLE: {                                                                           // marble.h:60-90
        V3 dpdx, dpdy;
        V3 p = Map3D(t, si, &dpdx, &dpdy);
        p = p * t.scale;
        Float marble = p.y + t.variation * FBm(p, t.scale * dpdx, t.scale * dpdy, t.omega, t.octaves);
        Float tt = .5f + .5f * std::sin(marble);
        static const Float c[][3] = {{.58f, .58f, .6f}, {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.5f, .5f, .5f}, {.6f, .59f, .58f},
                                     {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.2f, .2f, .33f}, {.58f, .58f, .6f}};
        const size_t NSEG = 9 - 3;
        int first = (int)std::floor(tt * NSEG);
        tt = (tt * NSEG - first);
        first = std::min(first, 5);   // tt == 1 indexes past the table in the reference (sin() == 1 exactly)
        RGB c0(c[first]), c1(c[first + 1]), c2(c[first + 2]), c3(c[first + 3]);
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
