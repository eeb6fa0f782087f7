// Static instruction-count probe (NOT product code, never launched): one interior-node step of the traversal for alternative
// node layouts of the same tree, compiled for gfx950 so that the per-step VALU / VMEM / LDS instruction counts can be read off the
// ISA (tools/isa_probe/count.py).  Each kernel performs ONE step for a per-lane ray and stores what the next step would need, so that
// nothing is optimised away.  Layouts:
//   bvh4_full    128 B  the round-1 node (6 x float4 planes SoA + 4 child refs)
//   bvh4_quant    64 B  4 children, 8-bit planes on a per-node power-of-two grid (origin 12 B, cell size 12 B, 24 B planes, 16 B refs)
//   bvh8_quant   128 B  8 children, same quantisation (12 + 12 + 48 + 32 B refs + pad): one cache line per node
//   bvh8_full    256 B  8 children, float planes (two cache lines)
// Children are visited nearest first; the others are pushed with their entry distance (culled at pop time), unsorted for the
// 8-wide layouts (sorting 8 keys costs 19 compare-exchanges), sorted for the 4-wide ones as in the product kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define INF __builtin_huge_valf()
struct Ray { float ox, oy, oz, ix, iy, iz, fx, fy, fz, tMax; uint32_t cur; };   // i = 1/d, f = (1/d) * (1 + 4 gamma(3))
struct Out { uint32_t next; uint32_t pushed[7]; float pt[7]; int n; };

#define BOXT(lo_x, lo_y, lo_z, hi_x, hi_y, hi_z, e, x)                                                                          \
    {                                                                                                                          \
        float nx = r.ix < 0 ? hi_x : lo_x, fx_ = r.ix < 0 ? lo_x : hi_x, ny = r.iy < 0 ? hi_y : lo_y, fy_ = r.iy < 0 ? lo_y : hi_y; \
        float nz = r.iz < 0 ? hi_z : lo_z, fz_ = r.iz < 0 ? lo_z : hi_z;                                                          \
        e = __builtin_fmaxf(__builtin_fmaxf((nx - r.ox) * r.ix, (ny - r.oy) * r.iy), (nz - r.oz) * r.iz);                         \
        x = __builtin_fminf(__builtin_fminf((fx_ - r.ox) * r.fx, (fy_ - r.oy) * r.fy), (fz_ - r.oz) * r.fz);                      \
    }

template <int W> __device__ __forceinline__ void finish(const float *t, const bool *h, const uint32_t *c, Out &o) {
    int best = -1;
    float tb = INF;
#pragma unroll
    for (int k = 0; k < W; ++k) if (h[k] && t[k] < tb) { tb = t[k]; best = k; }
    o.n = 0;
    o.next = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        if (!h[k]) continue;
        if (k == best) o.next = c[k];
        else { o.pushed[o.n] = c[k]; o.pt[o.n] = t[k]; ++o.n; }
    }
}

struct N4F { float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4]; uint32_t child[4], pad[4]; };
extern "C" __global__ void bvh4_full(const N4F *nodes, const Ray *rays, Out *out) {
    Ray r = rays[threadIdx.x];
    const float4 *q = reinterpret_cast<const float4 *>(nodes + r.cur);
    float4 lx = q[0], ly = q[1], lz = q[2], hx = q[3], hy = q[4], hz = q[5];
    uint4 ch = reinterpret_cast<const uint4 *>(nodes + r.cur)[6];
    float t[4]; bool h[4]; uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
    float e, x;
#define ONE(k, m) BOXT(lx.m, ly.m, lz.m, hx.m, hy.m, hz.m, e, x) h[k] = (e <= x) && (e < r.tMax) && (x > 0); t[k] = e;
    ONE(0, x) ONE(1, y) ONE(2, z) ONE(3, w)
#undef ONE
    Out o; finish<4>(t, h, c, o); out[threadIdx.x] = o;
}

struct N4Q { float p[3], s[3]; uint8_t q[6][4]; uint32_t child[4]; };   // 64 B
extern "C" __global__ void bvh4_quant(const N4Q *nodes, const Ray *rays, Out *out) {
    Ray r = rays[threadIdx.x];
    const uint4 *w = reinterpret_cast<const uint4 *>(nodes + r.cur);
    uint4 a = w[0], b = w[1], cc = w[2], ch = w[3];
    float px = __uint_as_float(a.x), py = __uint_as_float(a.y), pz = __uint_as_float(a.z), sx = __uint_as_float(a.w), sy = __uint_as_float(b.x), sz = __uint_as_float(b.y);
    uint32_t qlx = b.z, qly = b.w, qlz = cc.x, qhx = cc.y, qhy = cc.z, qhz = cc.w;
    float t[4]; bool h[4]; uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float lox = px + (float)((qlx >> (8 * k)) & 255u) * sx, loy = py + (float)((qly >> (8 * k)) & 255u) * sy, loz = pz + (float)((qlz >> (8 * k)) & 255u) * sz;
        float hix = px + (float)((qhx >> (8 * k)) & 255u) * sx, hiy = py + (float)((qhy >> (8 * k)) & 255u) * sy, hiz = pz + (float)((qhz >> (8 * k)) & 255u) * sz;
        float e, x;
        BOXT(lox, loy, loz, hix, hiy, hiz, e, x)
        h[k] = (e <= x) && (e < r.tMax) && (x > 0) && c[k] != 0xffffffffu; t[k] = e;
    }
    Out o; finish<4>(t, h, c, o); out[threadIdx.x] = o;
}

struct N8Q { float p[3], s[3]; uint32_t child[8]; uint8_t q[6][8]; uint32_t pad[6]; };   // 128 B
extern "C" __global__ void bvh8_quant(const N8Q *nodes, const Ray *rays, Out *out) {
    Ray r = rays[threadIdx.x];
    const uint4 *w = reinterpret_cast<const uint4 *>(nodes + r.cur);
    uint4 a = w[0], b = w[1], c0 = w[2], c1 = w[3], q0 = w[4], q1 = w[5], q2 = w[6];
    float px = __uint_as_float(a.x), py = __uint_as_float(a.y), pz = __uint_as_float(a.z), sx = __uint_as_float(a.w), sy = __uint_as_float(b.x), sz = __uint_as_float(b.y);
    uint32_t c[8] = {b.z, b.w, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};   // (layout detail irrelevant for the count)
    uint32_t ql[3][2] = {{c1.z, c1.w}, {q0.x, q0.y}, {q0.z, q0.w}}, qh[3][2] = {{q1.x, q1.y}, {q1.z, q1.w}, {q2.x, q2.y}};
    float t[8]; bool h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int wd = k >> 2, sh = 8 * (k & 3);
        float lox = px + (float)((ql[0][wd] >> sh) & 255u) * sx, loy = py + (float)((ql[1][wd] >> sh) & 255u) * sy, loz = pz + (float)((ql[2][wd] >> sh) & 255u) * sz;
        float hix = px + (float)((qh[0][wd] >> sh) & 255u) * sx, hiy = py + (float)((qh[1][wd] >> sh) & 255u) * sy, hiz = pz + (float)((qh[2][wd] >> sh) & 255u) * sz;
        float e, x;
        BOXT(lox, loy, loz, hix, hiy, hiz, e, x)
        h[k] = (e <= x) && (e < r.tMax) && (x > 0) && c[k] != 0xffffffffu; t[k] = e;
    }
    Out o; finish<8>(t, h, c, o); out[threadIdx.x] = o;
}

// the same layout with the decode folded into the ray:  t = q * (s / d) + (p - o) / d  -- three operations per plane instead of five.
// The rounding of this form is not the reference's, so conservativeness needs explicit slack: per node and axis
// delta = 4 eps (|B| + 255 |A|) is subtracted from the near offsets and added to the far ones (two B's per axis, nothing per plane).
extern "C" __global__ void bvh8_quant_folded(const N8Q *nodes, const Ray *rays, Out *out) {
    Ray r = rays[threadIdx.x];
    const uint4 *w = reinterpret_cast<const uint4 *>(nodes + r.cur);
    uint4 a = w[0], b = w[1], c0 = w[2], c1 = w[3], q0 = w[4], q1 = w[5], q2 = w[6];
    float px = __uint_as_float(a.x), py = __uint_as_float(a.y), pz = __uint_as_float(a.z), sx = __uint_as_float(a.w), sy = __uint_as_float(b.x), sz = __uint_as_float(b.y);
    uint32_t c[8] = {b.z, b.w, c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
    // near / far plane words picked per ray sign (uniform per ray: selects, not per-plane work)
    uint32_t nxw[2] = {r.ix < 0 ? q1.x : c1.z, r.ix < 0 ? q1.y : c1.w}, fxw[2] = {r.ix < 0 ? c1.z : q1.x, r.ix < 0 ? c1.w : q1.y};
    uint32_t nyw[2] = {r.iy < 0 ? q1.z : q0.x, r.iy < 0 ? q1.w : q0.y}, fyw[2] = {r.iy < 0 ? q0.x : q1.z, r.iy < 0 ? q0.y : q1.w};
    uint32_t nzw[2] = {r.iz < 0 ? q2.x : q0.z, r.iz < 0 ? q2.y : q0.w}, fzw[2] = {r.iz < 0 ? q0.z : q2.x, r.iz < 0 ? q0.w : q2.y};
    const float eps4 = 4 * 5.9604644775390625e-08f;
    float Ax = sx * r.ix, Bx = (px - r.ox) * r.ix, dx = eps4 * (__builtin_fabsf(Bx) + 255 * __builtin_fabsf(Ax));
    float Ay = sy * r.iy, By = (py - r.oy) * r.iy, dy = eps4 * (__builtin_fabsf(By) + 255 * __builtin_fabsf(Ay));
    float Az = sz * r.iz, Bz = (pz - r.oz) * r.iz, dz = eps4 * (__builtin_fabsf(Bz) + 255 * __builtin_fabsf(Az));
    float Bnx = Bx - dx, Bfx = Bx + dx, Bny = By - dy, Bfy = By + dy, Bnz = Bz - dz, Bfz = Bz + dz;
    float t[8]; bool h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int wd = k >> 2, sh = 8 * (k & 3);
        float e = __builtin_fmaxf(__builtin_fmaxf((float)((nxw[wd] >> sh) & 255u) * Ax + Bnx, (float)((nyw[wd] >> sh) & 255u) * Ay + Bny), (float)((nzw[wd] >> sh) & 255u) * Az + Bnz);
        float x = __builtin_fminf(__builtin_fminf((float)((fxw[wd] >> sh) & 255u) * Ax + Bfx, (float)((fyw[wd] >> sh) & 255u) * Ay + Bfy), (float)((fzw[wd] >> sh) & 255u) * Az + Bfz);
        h[k] = (e <= x) && (e < r.tMax) && (x > 0) && c[k] != 0xffffffffu; t[k] = e;
    }
    Out o; finish<8>(t, h, c, o); out[threadIdx.x] = o;
}
// the step without its bookkeeping: what `finish` costs alone (subtract from the others)
extern "C" __global__ void finish8_only(const float *tt, const uint32_t *cc, Out *out) {
    float t[8]; bool h[8]; uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { t[k] = tt[threadIdx.x * 8 + k]; c[k] = cc[threadIdx.x * 8 + k]; h[k] = t[k] < 1e30f; }
    Out o; finish<8>(t, h, c, o); out[threadIdx.x] = o;
}
extern "C" __global__ void finish4_only(const float *tt, const uint32_t *cc, Out *out) {
    float t[4]; bool h[4]; uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { t[k] = tt[threadIdx.x * 4 + k]; c[k] = cc[threadIdx.x * 4 + k]; h[k] = t[k] < 1e30f; }
    Out o; finish<4>(t, h, c, o); out[threadIdx.x] = o;
}

struct N8F { float lox[8], loy[8], loz[8], hix[8], hiy[8], hiz[8]; uint32_t child[8], pad[8]; };   // 256 B
extern "C" __global__ void bvh8_full(const N8F *nodes, const Ray *rays, Out *out) {
    Ray r = rays[threadIdx.x];
    const N8F &n = nodes[r.cur];
    float t[8]; bool h[8]; uint32_t c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float e, x;
        BOXT(n.lox[k], n.loy[k], n.loz[k], n.hix[k], n.hiy[k], n.hiz[k], e, x)
        c[k] = n.child[k];
        h[k] = (e <= x) && (e < r.tMax) && (x > 0); t[k] = e;
    }
    Out o; finish<8>(t, h, c, o); out[threadIdx.x] = o;
}

// ---- round-3 candidates on the product's node (pt_bvh4q.h: 16-bit planes on one grid, folded test t = q A + B with per-ray A / Bn / Bf)
struct RayQ { float Ax, Ay, Az, Bnx, Bny, Bnz, Bfx, Bfy, Bfz, tMax; int negx, negy, negz; uint32_t cur; };
// (a) today's shape: one ray per lane, 4 x 16-byte loads, four children tested per lane, hit children sorted and pushed by the lane
struct N4QF { uint16_t lo[3][4], hi[3][4]; uint32_t child[4]; };   // 64 B, plane-major
extern "C" __global__ void bvh4q_lane(const N4QF *nodes, const RayQ *rays, Out *out) {
    RayQ r = rays[threadIdx.x];
    const uint4 *wp = reinterpret_cast<const uint4 *>(nodes + r.cur);
    uint4 w0 = wp[0], w1 = wp[1], w2 = wp[2], ch = wp[3];
    const uint32_t w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    const uint32_t nx0 = r.negx ? w[6] : w[0], nx1 = r.negx ? w[7] : w[1], fx0 = r.negx ? w[0] : w[6], fx1 = r.negx ? w[1] : w[7];
    const uint32_t ny0 = r.negy ? w[8] : w[2], ny1 = r.negy ? w[9] : w[3], fy0 = r.negy ? w[2] : w[8], fy1 = r.negy ? w[3] : w[9];
    const uint32_t nz0 = r.negz ? w[10] : w[4], nz1 = r.negz ? w[11] : w[5], fz0 = r.negz ? w[4] : w[10], fz1 = r.negz ? w[5] : w[11];
    float t[4]; bool h[4]; uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sh = 16 * (k & 1);
        const uint32_t wnx = k < 2 ? nx0 : nx1, wny = k < 2 ? ny0 : ny1, wnz = k < 2 ? nz0 : nz1, wfx = k < 2 ? fx0 : fx1, wfy = k < 2 ? fy0 : fy1, wfz = k < 2 ? fz0 : fz1;
        float e = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)((wnx >> sh) & 65535u), r.Ax, r.Bnx), __builtin_fmaf((float)((wny >> sh) & 65535u), r.Ay, r.Bny)), __builtin_fmaf((float)((wnz >> sh) & 65535u), r.Az, r.Bnz));
        float x = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)((wfx >> sh) & 65535u), r.Ax, r.Bfx), __builtin_fmaf((float)((wfy >> sh) & 65535u), r.Ay, r.Bfy)), __builtin_fmaf((float)((wfz >> sh) & 65535u), r.Az, r.Bfz));
        h[k] = (e <= x) && (e < r.tMax) && (x > 0) && c[k] != 0xffffffffu; t[k] = e;
    }
    Out o; finish<4>(t, h, c, o); out[threadIdx.x] = o;
}
// (b) one ray per QUAD: child-major node (child j = six 16-bit planes + its reference = ONE 16-byte word), lane j of the quad loads and tests
// child j -- one coalesced load instruction per step (a quad reads one 64-byte line) -- the hits are ranked across the quad with quad-permute
// DPP moves, the nearest becomes the quad's next node, the others are stored far-to-near by their own lanes (the quad shares one stack).
struct N4QC { struct { uint16_t lo[3], hi[3]; uint32_t child; } c[4]; };   // 64 B, child-major
struct OutQ { uint32_t next, n; };
template <int CTRL> __device__ __forceinline__ float quadf(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ uint32_t quadu(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
extern "C" __global__ void bvh4q_quad(const N4QC *nodes, const RayQ *rays, OutQ *out, uint32_t *stack /* per quad: 24 entries */, float *stack_t) {
    const uint32_t lane = threadIdx.x, j = lane & 3u, quad = lane >> 2;
    RayQ r = rays[quad];   // the quad's ray, replicated in its four lanes
    const uint4 w = reinterpret_cast<const uint4 *>(nodes + r.cur)[j];
    const uint32_t lox = w.x & 65535u, loy = w.x >> 16, loz = w.y & 65535u, hix = w.y >> 16, hiy = w.z & 65535u, hiz = w.z >> 16, child = w.w;
    const float e = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)(r.negx ? hix : lox), r.Ax, r.Bnx), __builtin_fmaf((float)(r.negy ? hiy : loy), r.Ay, r.Bny)), __builtin_fmaf((float)(r.negz ? hiz : loz), r.Az, r.Bnz));
    const float x = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)(r.negx ? lox : hix), r.Ax, r.Bfx), __builtin_fmaf((float)(r.negy ? loy : hiy), r.Ay, r.Bfy)), __builtin_fmaf((float)(r.negz ? loz : hiz), r.Az, r.Bfz));
    const bool hit = (e <= x) && (e < r.tMax) && (x > 0) && child != 0xffffffffu;
    const float key = hit ? e : INF;
    // rank among the quad's four keys (ties: the lower child slot first, as the per-lane sort network resolves them)
    const float k1 = quadf<0x39>(key), k2 = quadf<0x4e>(key), k3 = quadf<0x93>(key);   // quad_perm [1,2,3,0], [2,3,0,1], [3,0,1,2]: lanes j+1, j+2, j+3 (mod 4)
    const uint32_t j1 = (j + 1) & 3u, j2 = (j + 2) & 3u, j3 = (j + 3) & 3u;
    const uint32_t rank = (uint32_t)(k1 < key || (k1 == key && j1 < j)) + (uint32_t)(k2 < key || (k2 == key && j2 < j)) + (uint32_t)(k3 < key || (k3 == key && j3 < j));
    const uint32_t hits4 = (uint32_t)hit + (uint32_t)(k1 < INF) + (uint32_t)(k2 < INF) + (uint32_t)(k3 < INF);
    // nearest hit child -> every lane of the quad (the lane of rank 0 contributes it, the others 0; OR over the quad)
    uint32_t nxt = (hit && rank == 0) ? child : 0u;
    nxt |= quadu<0x39>(nxt); nxt |= quadu<0x4e>(nxt);
    if (hit && rank > 0) { const uint32_t slot = quad * 24u + (hits4 - 1u - rank); stack[slot] = child; stack_t[slot] = e; }   // far to near: rank hits4-1 at the bottom
    if (j == 0) { OutQ o; o.next = hits4 ? nxt : 0xffffffffu; o.n = hits4 ? hits4 - 1u : 0u; out[quad] = o; }
}

// ---- round 6 (last session) candidate: f16 planes in a NODE-LOCAL frame, taken straight into the slab FMAs by v_fma_mix_f32 (an f16 half of a packed word as a
// multiplicand of an f32 FMA: no conversion instruction; hipcc emits it for fmaf((float)half, A, B) on gfx950).  A global grid does not fit f16 (11 significant
// bits), so the node carries its own origin (3 x f32) and ONE power-of-two cell size: 80 bytes = five 16-byte words.  Per node: A = s / d (3 mul),
// B = p / d - o / d (3 fma); the evaluation slack is a per-RAY constant folded into the entry test (e - x <= 2 delta, one subtract per child).
// Compare with bvh4q_lane (today's shape) through the same `finish`.
#include <hip/hip_fp16.h>
struct RayH { float ix, iy, iz, oix, oiy, oiz, delta2, tMax; int negx, negy, negz; uint32_t cur; };
struct N4H { float p[3], s; __half2 lo[3][2], hi[3][2]; uint32_t child[4]; };   // 80 B, plane-major, children (0,1) and (2,3) packed
extern "C" __global__ void bvh4h_lane(const N4H *nodes, const RayH *rays, Out *out) {
    RayH r = rays[threadIdx.x];
    const uint4 *wp = reinterpret_cast<const uint4 *>(nodes + r.cur);
    uint4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], ch = wp[4];
    const float px = __uint_as_float(w0.x), py = __uint_as_float(w0.y), pz = __uint_as_float(w0.z), s = __uint_as_float(w0.w);
    const float Ax = s * r.ix, Ay = s * r.iy, Az = s * r.iz;
    const float Bx = __builtin_fmaf(px, r.ix, -r.oix), By = __builtin_fmaf(py, r.iy, -r.oiy), Bz = __builtin_fmaf(pz, r.iz, -r.oiz);
    const uint32_t w[12] = {w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
    const uint32_t nx0 = r.negx ? w[6] : w[0], nx1 = r.negx ? w[7] : w[1], fx0 = r.negx ? w[0] : w[6], fx1 = r.negx ? w[1] : w[7];
    const uint32_t ny0 = r.negy ? w[8] : w[2], ny1 = r.negy ? w[9] : w[3], fy0 = r.negy ? w[2] : w[8], fy1 = r.negy ? w[3] : w[9];
    const uint32_t nz0 = r.negz ? w[10] : w[4], nz1 = r.negz ? w[11] : w[5], fz0 = r.negz ? w[4] : w[10], fz1 = r.negz ? w[5] : w[11];
    float t[4]; bool h[4]; uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
#define HALF_OF(word, hi_) ((float)((hi_) ? __high2half(*reinterpret_cast<const __half2 *>(&(word))) : __low2half(*reinterpret_cast<const __half2 *>(&(word)))))
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool hi_ = k & 1;
        const uint32_t wnx = k < 2 ? nx0 : nx1, wny = k < 2 ? ny0 : ny1, wnz = k < 2 ? nz0 : nz1, wfx = k < 2 ? fx0 : fx1, wfy = k < 2 ? fy0 : fy1, wfz = k < 2 ? fz0 : fz1;
        float e = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(HALF_OF(wnx, hi_), Ax, Bx), __builtin_fmaf(HALF_OF(wny, hi_), Ay, By)), __builtin_fmaf(HALF_OF(wnz, hi_), Az, Bz));
        float x = __builtin_fminf(__builtin_fminf(__builtin_fmaf(HALF_OF(wfx, hi_), Ax, Bx), __builtin_fmaf(HALF_OF(wfy, hi_), Ay, By)), __builtin_fmaf(HALF_OF(wfz, hi_), Az, Bz));
        h[k] = (e - x <= r.delta2) && (e < r.tMax) && (x > 0) && c[k] != 0xffffffffu; t[k] = e;
    }
#undef HALF_OF
    Out o; finish<4>(t, h, c, o); out[threadIdx.x] = o;
}
