// Transform constructors; behaviour per reference core/transform.cpp (lines cited per function).
#include "geom.h"

namespace pbrt_amd {

Matrix4x4 Transpose(const Matrix4x4 &a) {
    Matrix4x4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

// transform.cpp:82-139.  Gauss-Jordan elimination with full pivoting; the pivot reciprocal is
// formed in double (`1. / pivot`) and rounded to float before the row scale, as there.
Matrix4x4 Inverse(const Matrix4x4 &src) {
    int colIdx[4], rowIdx[4];
    int used[4] = {0, 0, 0, 0};
    Float a[4][4];
    std::memcpy(a, src.m, sizeof(a));
    for (int step = 0; step < 4; ++step) {
        int prow = 0, pcol = 0;
        Float best = 0.f;
        for (int r = 0; r < 4; ++r) {
            if (used[r] == 1) continue;
            for (int c = 0; c < 4; ++c) {
                if (used[c] == 0) {
                    if (std::abs(a[r][c]) >= best) {
                        best = Float(std::abs(a[r][c]));
                        prow = r;
                        pcol = c;
                    }
                } else if (used[c] > 1) {
                    std::fprintf(stderr, "Error: Singular matrix in MatrixInvert\n");
                }
            }
        }
        ++used[pcol];
        if (prow != pcol)
            for (int k = 0; k < 4; ++k) std::swap(a[prow][k], a[pcol][k]);
        rowIdx[step] = prow;
        colIdx[step] = pcol;
        if (a[pcol][pcol] == 0.f) std::fprintf(stderr, "Error: Singular matrix in MatrixInvert\n");
        Float pivinv = 1. / a[pcol][pcol];
        a[pcol][pcol] = 1.;
        for (int j = 0; j < 4; ++j) a[pcol][j] *= pivinv;
        for (int r = 0; r < 4; ++r) {
            if (r == pcol) continue;
            Float save = a[r][pcol];
            a[r][pcol] = 0;
            for (int k = 0; k < 4; ++k) a[r][k] -= a[pcol][k] * save;
        }
    }
    for (int j = 3; j >= 0; --j)
        if (rowIdx[j] != colIdx[j])
            for (int k = 0; k < 4; ++k) std::swap(a[k][rowIdx[j]], a[k][colIdx[j]]);
    Matrix4x4 r;
    std::memcpy(r.m, a, sizeof(a));
    return r;
}

Transform Translate(const Vec3 &d) {   // transform.cpp:144-150
    return Transform(Matrix4x4(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1),
                     Matrix4x4(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1));
}

Transform Scale(Float x, Float y, Float z) {   // transform.cpp:152-156
    return Transform(Matrix4x4(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1),
                     Matrix4x4(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1));
}

Transform Rotate(Float theta, const Vec3 &axis) {   // transform.cpp:182-205
    Vec3 a = Normalize(axis);
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * c;
    m.m[0][1] = a.x * a.y * (1 - c) - a.z * s;
    m.m[0][2] = a.x * a.z * (1 - c) + a.y * s;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - c) + a.z * s;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * c;
    m.m[1][2] = a.y * a.z * (1 - c) - a.x * s;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - c) - a.y * s;
    m.m[2][1] = a.y * a.z * (1 - c) + a.x * s;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * c;
    m.m[2][3] = 0;
    return Transform(m, Transpose(m));
}

Transform LookAt(const Vec3 &pos, const Vec3 &look, const Vec3 &up, bool *ok) {   // transform.cpp:207-241
    Matrix4x4 c2w;
    c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1;
    Vec3 dir = Normalize(look - pos);
    if (Cross(Normalize(up), dir).Length() == 0) {
        if (ok) *ok = false;
        return Transform();
    }
    if (ok) *ok = true;
    Vec3 right = Normalize(Cross(Normalize(up), dir));
    Vec3 newUp = Cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x;   c2w.m[1][2] = dir.y;   c2w.m[2][2] = dir.z;   c2w.m[3][2] = 0.;
    return Transform(Inverse(c2w), c2w);
}

Transform Perspective(Float fov, Float n, Float f) {   // transform.cpp:302-310
    Matrix4x4 persp(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    Float invTanAng = 1 / std::tan(Radians(fov) / 2);
    return Scale(invTanAng, invTanAng, 1) * Transform(persp);
}

}  // namespace pbrt_amd
