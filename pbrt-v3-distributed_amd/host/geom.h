// Host-side vector / bounds / transform math for scene construction.
// Behaviour follows the reference's core/geometry.h and core/transform.{h,cpp} operation
// by operation (float rounding included) so that world-space vertices, camera matrices
// and the SAH build see bit-identical inputs; citations give the reference lines.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>

namespace pbrt_amd {

typedef float Float;
static const Float kPi = 3.14159265358979323846;          // pbrt.h:201
static const Float kInvPi = 0.31830988618379067154;
static const Float kPiOver2 = 1.57079632679489661923;
static const Float kPiOver4 = 0.78539816339744830961;
static const Float kInfinity = std::numeric_limits<Float>::infinity();
static const Float kMachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5;  // pbrt.h:197
inline Float Gamma(int n) { return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon); }  // pbrt.h:285
inline Float Radians(Float deg) { return (kPi / 180) * deg; }                         // pbrt.h:330
template <typename T, typename U, typename V>
inline T Clamp(T v, U lo, V hi) { return v < lo ? T(lo) : (v > hi ? T(hi) : v); }
inline int32_t RoundUpPow2(int32_t v) {
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1;
}
inline int Log2Int(uint32_t v) { return 31 - __builtin_clz(v); }
inline bool IsPowerOf2(int64_t v) { return v && !(v & (v - 1)); }

struct Vec3 {
    Float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vec3 operator+(const Vec3 &v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator-(const Vec3 &v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 operator*(Float s) const { return Vec3(x * s, y * s, z * s); }
    // geometry.h:244-248: division multiplies by the rounded reciprocal
    Vec3 operator/(Float f) const { Float inv = (Float)1 / f; return Vec3(x * inv, y * inv, z * inv); }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
inline Vec3 operator*(Float s, const Vec3 &v) { return v * s; }
inline Float Dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 Normalize(const Vec3 &v) { return v / v.Length(); }
// geometry.h:957-963: cross product is evaluated in double and rounded once
inline Vec3 Cross(const Vec3 &a, const Vec3 &b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return Vec3((Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx)));
}
inline void CoordinateSystem(const Vec3 &v1, Vec3 *v2, Vec3 *v3) {   // geometry.h:1020-1027
    if (std::abs(v1.x) > std::abs(v1.y)) *v2 = Vec3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else *v2 = Vec3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
inline Vec3 Min(const Vec3 &a, const Vec3 &b) { return Vec3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline Vec3 Max(const Vec3 &a, const Vec3 &b) { return Vec3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }

struct Bounds3 {   // geometry.h:  default = inverted (max, lowest)
    Vec3 pMin, pMax;
    Bounds3() {
        Float lo = std::numeric_limits<Float>::lowest(), hi = std::numeric_limits<Float>::max();
        pMin = Vec3(hi, hi, hi); pMax = Vec3(lo, lo, lo);
    }
    explicit Bounds3(const Vec3 &p) : pMin(p), pMax(p) {}
    Bounds3(const Vec3 &a, const Vec3 &b) : pMin(Min(a, b)), pMax(Max(a, b)) {}
    Vec3 Diagonal() const { return pMax - pMin; }
    Float SurfaceArea() const { Vec3 d = Diagonal(); return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
    int MaximumExtent() const {
        Vec3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        return d.y > d.z ? 1 : 2;
    }
    Vec3 Offset(const Vec3 &p) const {   // geometry.h Bounds3::Offset
        Vec3 o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }
};
inline Bounds3 Union(const Bounds3 &b, const Vec3 &p) { Bounds3 r; r.pMin = Min(b.pMin, p); r.pMax = Max(b.pMax, p); return r; }
inline Bounds3 Union(const Bounds3 &a, const Bounds3 &b) { Bounds3 r; r.pMin = Min(a.pMin, b.pMin); r.pMax = Max(a.pMax, b.pMax); return r; }

struct Matrix4x4 {
    Float m[4][4];
    Matrix4x4() { std::memset(m, 0, sizeof(m)); m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1; }
    Matrix4x4(Float a00, Float a01, Float a02, Float a03, Float a10, Float a11, Float a12, Float a13, Float a20,
              Float a21, Float a22, Float a23, Float a30, Float a31, Float a32, Float a33) {
        Float t[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
        std::memcpy(m, t, sizeof(m));
    }
    bool operator==(const Matrix4x4 &o) const { return std::memcmp(m, o.m, sizeof(m)) == 0 || eq(o); }
    bool eq(const Matrix4x4 &o) const {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (m[i][j] != o.m[i][j]) return false;
        return true;
    }
    static Matrix4x4 Mul(const Matrix4x4 &a, const Matrix4x4 &b) {   // transform.h:83-90
        Matrix4x4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
        return r;
    }
};
Matrix4x4 Transpose(const Matrix4x4 &m);
Matrix4x4 Inverse(const Matrix4x4 &m);   // transform.cpp:82-139 (Gauss-Jordan, full pivoting)

struct Transform {
    Matrix4x4 m, mInv;
    Transform() {}
    explicit Transform(const Matrix4x4 &m) : m(m), mInv(Inverse(m)) {}
    Transform(const Matrix4x4 &m, const Matrix4x4 &mInv) : m(m), mInv(mInv) {}
    Transform operator*(const Transform &t) const { return Transform(Matrix4x4::Mul(m, t.m), Matrix4x4::Mul(t.mInv, mInv)); }
    bool IsIdentity() const { return m.eq(Matrix4x4()); }
    bool operator==(const Transform &t) const { return m.eq(t.m) && mInv.eq(t.mInv); }
    bool operator<(const Transform &t) const { return std::memcmp(m.m, t.m.m, sizeof(m.m)) < 0; }
    bool SwapsHandedness() const {   // transform.cpp:254-259
        Float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) -
                    m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                    m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0;
    }
    Vec3 Point(const Vec3 &p) const {   // transform.h:223-234
        Float x = p.x, y = p.y, z = p.z;
        Float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
        Float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
        Float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
        Float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
        if (wp == 1) return Vec3(xp, yp, zp);
        Float inv = (Float)1 / wp;                // Point3::operator/ (geometry.h:499-503)
        return Vec3(inv * xp, inv * yp, inv * zp);
    }
    Vec3 Vector(const Vec3 &v) const {  // transform.h:236-242
        Float x = v.x, y = v.y, z = v.z;
        return Vec3(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
                    m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
    }
    Vec3 Normal(const Vec3 &n) const {  // transform.h:244-250 (inverse transpose)
        Float x = n.x, y = n.y, z = n.z;
        return Vec3(mInv.m[0][0] * x + mInv.m[1][0] * y + mInv.m[2][0] * z,
                    mInv.m[0][1] * x + mInv.m[1][1] * y + mInv.m[2][1] * z,
                    mInv.m[0][2] * x + mInv.m[1][2] * y + mInv.m[2][2] * z);
    }
};
inline Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
Transform Translate(const Vec3 &d);
Transform Scale(Float x, Float y, Float z);
Transform Rotate(Float thetaDeg, const Vec3 &axis);
Transform LookAt(const Vec3 &pos, const Vec3 &look, const Vec3 &up, bool *ok);
Transform Perspective(Float fov, Float n, Float f);

}  // namespace pbrt_amd
