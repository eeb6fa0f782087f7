// Lean BVH4 traversal steps for the persistent-lane scheduler of k_trace (round 2).
//
// Round 1's PMC passes read the traversal as waiting on memory; the round-2 A/B against the quantised BVH8 (fewer
// dependent fetches, more arithmetic per step: 35 % SLOWER) showed that with 5 waves per SIMD the fetch latency is
// covered and the kernel is bound by instruction ISSUE: ~230 instructions per interior step and ~340 per triangle step,
// a third of them exec-mask bookkeeping around early returns, LDS-or-spill pushes and the per-lane axis permutation.
// The steps below do the same work as TravNodeStep / TravLeafStep (pt_scene.h) in straight-line predicated code:
//   * slab distances in the folded form t = plane * inv - o * inv (one FMA per plane instead of a subtract and a
//     multiply), with the constants widened so that the test stays a SUPERSET of Bounds3::IntersectP (below);
//   * pushes are unconditional LDS stores above the stack top, the stack pointer moves by predicate; the pop value is
//     prefetched together with the node / triangle so that no step has two dependent memory round trips;
//   * the watertight test reads the triangle from the copy that is already permuted for the ray's dominant axis
//     (three copies of the 48-byte records: 288 GB of HBM buys 18 v_cndmask per test), evaluates every rejection
//     as a predicate and combines them at the end; only the fp64 edge fallback stays a (wave-uniform) branch.
// The arithmetic that decides a hit -- Triangle::Intersect, shapes/triangle.cpp:188-291 -- is op for op the one of
// TriangleTest; hits stay bit-identical to the reference's (parity tests: closest hits vs the oracle's BVH2 traversal).
//
// Conservative folded slab test.  Reference (core/geometry.h:1412-1438), per axis with I = invDir:
//     tNear_ref = fl(fl(pNear - o) * I)                      within u (1 +- 2.1 eps) of u = (pNear - o) I
//     tFar_ref  = fl(fl(fl(pFar - o) * I) * fl(1 + 2 gamma(3)))   <= u (1 + 11.2 eps) for u > 0
// Here, with eps = 2^-24:
//     In = fl(I (1 - 8 eps)), cN = fl(o In) pushed up by 4 eps |cN|:  tNear = fma(pNear, In, -cN) <= u (1 - 5.8 eps)  (u > 0)
//     If = fl(I (1 + 24 eps)), cF = fl(o If) pushed down by 4 eps |cF|: tFar = fma(pFar, If, -cF) >= u (1 + 21.8 eps) (u > 0)
// so tNear <= tNear_ref and tFar >= tFar_ref wherever the reference's values are positive; a negative near distance
// stays <= 0 (it can only be non-binding, as in the reference, because acceptance needs tExit > 0) and a negative far
// distance makes the reference reject.  A constant that overflowed (|o I| = inf) is replaced by NaN: the FMA then
// yields NaN and max3 / min3 skip that axis -- never a rejection the reference would not make.
// Zero direction components (I = +-inf; NOT rare: Sobol' values such as 0.5 give exact zeros in the local frame of an
// axis-aligned surface) must keep their culling -- such a ray lives in a plane and would otherwise visit every node that
// plane cuts (measured: a 35 ms tail per launch on the 10 M-triangle scene).  The reference gets -inf / +inf from
// (p - o) * inf by the SIGN of p - o; here I is replaced by +-2^100 and c = o * 2^100 is exact (a power of two), so
// fma(p, 2^100, -c) = fl((p - o) 2^100) has exactly that sign and a magnitude that dwarfs every finite distance on the
// other axes.  Where p == o the reference evaluates 0 * inf = NaN and (y / z axes) IGNORES that bound, so equality must
// land on the accepting side: c is moved by max(|c| 2^-22, 1e20) (FastZeroAxis) -- planes within 4 ulp of the origin
// count as "inside the slab".
// Every box the reference enters is entered here; extra boxes cost time, not correctness.
#pragma once

#define PT_FAST_STACK_GUARD 4   /* the fast steps need room for three pushes above the top inside the LDS part of the stack */

struct FastRay {
    V3 inN, cN, inF, cF;          // folded slab constants (see above)
    uint32_t offX, offY, offZ;    // byte offsets of the near planes in a BVH4Node (lo or hi arrays by ray sign)
    V3 op;                        // ray origin, permuted (kx, ky, kz)
    Float Sx, Sy, Sz;             // shear (triangle.cpp:210-216)
    const float4 *tri;            // the triangle copy permuted for this ray's kz
    Float tMax;
    uint32_t prim, cur;
    PT_DEV bool done() const { return cur == TRAV_DONE; }
    PT_DEV bool atLeaf() const { return cur != TRAV_DONE && (cur & BVH4_LEAF); }
    PT_DEV bool atNode() const { return !(cur & BVH4_LEAF); }
};

PT_DEV Float FastFoldC(Float c, Float push) {   // c moved by 4 eps |c| in the direction `push`; non-finite -> NaN (axis ignored)
    Float r = c + push * (4 * PT_MACHINE_EPS) * absf(c);
    return (absf(r) < PT_INFINITY) ? r : __builtin_nanf("");
}
// d == 0 on this axis: I -> +-2^100, c = o I exactly (power of two), then moved by delta = max(|c| 2^-22, 1e20) to the accepting side:
// a plane within 4 ulp of the origin counts as "origin inside the slab".  The reference gets 0 * inf = NaN when the origin lies exactly
// ON a plane and (on the y / z axes) ignores that bound, so equality must not reject here.
PT_DEV void FastZeroAxis(Float o, Float d, Float *inN, Float *inF, Float *cN, Float *cF) {
    const Float I = __builtin_copysignf(0x1p100f, d), c = o * I;
    const Float delta = __builtin_fmaxf(absf(c) * 0x1p-22f, 1e20f);
    *inN = *inF = I;
    const bool fin = absf(c) < PT_INFINITY;
    *cN = fin ? c + delta : __builtin_nanf("");
    *cF = fin ? c - delta : __builtin_nanf("");
}
PT_DEV void FastRayInit(const DevScene &sc, const float4 *triPerm, size_t triCopyStride, FastRay &fr, const V3 &o, const V3 &d, Float tMax, TravStack &st) {
    const V3 inv(1 / d.x, 1 / d.y, 1 / d.z);
    const Float wn = 1 - 8 * PT_MACHINE_EPS, wf = 1 + 24 * PT_MACHINE_EPS;
    fr.inN = V3(inv.x * wn, inv.y * wn, inv.z * wn);
    fr.inF = V3(inv.x * wf, inv.y * wf, inv.z * wf);
    fr.cN = V3(FastFoldC(o.x * fr.inN.x, 1), FastFoldC(o.y * fr.inN.y, 1), FastFoldC(o.z * fr.inN.z, 1));
    fr.cF = V3(FastFoldC(o.x * fr.inF.x, -1), FastFoldC(o.y * fr.inF.y, -1), FastFoldC(o.z * fr.inF.z, -1));
    // zero direction components: sign-exact stand-in for +-inf (header comment)
    if (d.x == 0) FastZeroAxis(o.x, d.x, &fr.inN.x, &fr.inF.x, &fr.cN.x, &fr.cF.x);
    if (d.y == 0) FastZeroAxis(o.y, d.y, &fr.inN.y, &fr.inF.y, &fr.cN.y, &fr.cF.y);
    if (d.z == 0) FastZeroAxis(o.z, d.z, &fr.inN.z, &fr.inF.z, &fr.cN.z, &fr.cF.z);
    fr.offX = inv.x < 0 ? 48u : 0u;
    fr.offY = inv.y < 0 ? 64u : 16u;
    fr.offZ = inv.z < 0 ? 80u : 32u;
    RayShear rs;
    rs.init(d);
    fr.op = rs.permute(o);
    fr.Sx = rs.Sx; fr.Sy = rs.Sy; fr.Sz = rs.Sz;
    fr.tri = triPerm + (size_t)rs.kz * triCopyStride;
    fr.tMax = tMax;
    fr.prim = TRAV_MISS;
    st.sp = 0;
    fr.cur = sc.n_nodes ? 0u : TRAV_DONE;
}

// 32-bit byte offsets from a wave-uniform base: the loads take the base from SGPRs (global_load ... v_off, s[base:base+1])
PT_DEV float4 LdF4(const char *base, uint32_t off) { return *reinterpret_cast<const float4 *>(base + off); }
PT_DEV uint4 LdU4(const char *base, uint32_t off) { return *reinterpret_cast<const uint4 *>(base + off); }

// one interior step.  SAFE == false requires st.sp <= PT_LDS_STACK - PT_FAST_STACK_GUARD in every lane that takes the step (pushes and the
// pop stay inside the LDS part of the stack); SAFE == true is the same step with TravStack's spilling push / pop
template <bool COUNT, bool ORDERED, bool SAFE>
PT_DEV void FastNodeStep(const DevScene &sc, FastRay &fr, TravStack &st, TraceCounters *cnt) {
    const char *base = reinterpret_cast<const char *>(sc.nodes);
    const uint32_t nb = fr.cur << 7;
    float4 nx = LdF4(base, nb + fr.offX), fx = LdF4(base, nb + (48u - fr.offX));
    float4 ny = LdF4(base, nb + fr.offY), fy = LdF4(base, nb + (80u - fr.offY));
    float4 nz = LdF4(base, nb + fr.offZ), fz = LdF4(base, nb + (112u - fr.offZ));
    uint4 ch = LdU4(base, nb + 96u);
#ifdef PT_FAST_EXTRA_LOADS   /* experiment (profiles/r02 notes): extra 16-byte loads from the SAME cache line -- do L1 look-ups bound the step? */
    {
        // three more per-lane requests to the node's own cache line, at distinct addresses (identical ones would be merged by the compiler)
        uint32_t a0 = nb + 112u, a1 = nb + 100u, a2 = nb + 104u;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
        uint4 x0 = LdU4(base, a0);
        uint32_t x1 = *reinterpret_cast<const uint32_t *>(base + a1), x2 = *reinterpret_cast<const uint32_t *>(base + a2);
        asm volatile("" :: "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1), "v"(x2));
    }
#endif
    uint32_t top = 0;
    if (!SAFE) top = st.lds[(st.sp > 0 ? st.sp - 1 : 0) * PT_BLOCK];   // the entry a pop would return, fetched alongside the node
    if (COUNT) ++cnt->nodes;
    Float t0, t1, t2, t3;
#define PT_FBOX(c, tk)                                                                                                                    \
    {                                                                                                                                     \
        Float e = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(nx.c, fr.inN.x, -fr.cN.x), __builtin_fmaf(ny.c, fr.inN.y, -fr.cN.y)),   \
                                  __builtin_fmaf(nz.c, fr.inN.z, -fr.cN.z));                                                              \
        Float x = __builtin_fminf(__builtin_fminf(__builtin_fmaf(fx.c, fr.inF.x, -fr.cF.x), __builtin_fmaf(fy.c, fr.inF.y, -fr.cF.y)),   \
                                  __builtin_fmaf(fz.c, fr.inF.z, -fr.cF.z));                                                              \
        tk = ((e <= x) && (e < fr.tMax) && (x > 0)) ? e : PT_INFINITY;                                                                    \
    }
    PT_FBOX(x, t0) PT_FBOX(y, t1) PT_FBOX(z, t2) PT_FBOX(w, t3)
#undef PT_FBOX
    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
    if (ORDERED) {
#define PT_FSWAP(ta, ca, tb, cb) { const bool s = tb < ta; const Float tl = s ? tb : ta, th = s ? ta : tb; const uint32_t cl = s ? cb : ca, chh = s ? ca : cb; ta = tl; tb = th; ca = cl; cb = chh; }
        PT_FSWAP(t0, c0, t1, c1) PT_FSWAP(t2, c2, t3, c3) PT_FSWAP(t0, c0, t2, c2) PT_FSWAP(t1, c1, t3, c3) PT_FSWAP(t1, c1, t2, c2)
#undef PT_FSWAP
    } else {   // any-hit: only the misses have to sink to the end (the order among hits cannot change the answer)
#define PT_FSINK(ta, ca, tb, cb) { const bool s = !(ta < PT_INFINITY) && (tb < PT_INFINITY); const Float tl = s ? tb : ta, th = s ? ta : tb; const uint32_t cl = s ? cb : ca, chh = s ? ca : cb; ta = tl; tb = th; ca = cl; cb = chh; }
        PT_FSINK(t0, c0, t1, c1) PT_FSINK(t2, c2, t3, c3) PT_FSINK(t0, c0, t2, c2) PT_FSINK(t1, c1, t3, c3) PT_FSINK(t1, c1, t2, c2)
#undef PT_FSINK
    }
    if (SAFE) {
        if (t3 < PT_INFINITY) st.push(c3, t3);
        if (t2 < PT_INFINITY) st.push(c2, t2);
        if (t1 < PT_INFINITY) st.push(c1, t1);
        fr.cur = (t0 < PT_INFINITY) ? c0 : st.pop(fr.tMax);
        return;
    }
    // pushes far to near: the store always happens (above the top it is harmless), the stack pointer moves by predicate
    int sp = st.sp;
    st.lds[sp * PT_BLOCK] = c3; sp += (t3 < PT_INFINITY) ? 1 : 0;
    st.lds[sp * PT_BLOCK] = c2; sp += (t2 < PT_INFINITY) ? 1 : 0;
    st.lds[sp * PT_BLOCK] = c1; sp += (t1 < PT_INFINITY) ? 1 : 0;
    const bool any = t0 < PT_INFINITY;
    const bool popOk = !any && sp > 0;   // nothing hit: nothing was pushed either, sp is still the value the prefetch used
    fr.cur = any ? c0 : (popOk ? top : TRAV_DONE);
    st.sp = sp - (popOk ? 1 : 0);
}

// one leaf step = one triangle (see TravLeafStep).  SAFE == false requires st.sp <= PT_LDS_STACK (the pop reads the LDS part)
template <bool ANY, bool COUNT, bool SAFE>
PT_DEV void FastLeafStep(const DevScene &sc, FastRay &fr, TravStack &st, TraceCounters *cnt) {
    const uint32_t first = fr.cur & BVH4_FIRST_MASK, left = (fr.cur >> 27) & 0xfu;
    const float4 *tv = fr.tri + 3 * (size_t)first;
    const float4 a = tv[0], b = tv[1], c = tv[2];
    uint32_t top = 0;
    if (!SAFE) top = st.lds[(st.sp > 0 ? st.sp - 1 : 0) * PT_BLOCK];
    if (COUNT) ++cnt->tris;
    const uint32_t flags = __float_as_uint(a.w);
    // Triangle::Intersect, triangle.cpp:197-291, on the pre-permuted vertices
    V3 p0t(a.x - fr.op.x, a.y - fr.op.y, a.z - fr.op.z), p1t(b.x - fr.op.x, b.y - fr.op.y, b.z - fr.op.z), p2t(c.x - fr.op.x, c.y - fr.op.y, c.z - fr.op.z);
    const Float Sx = fr.Sx, Sy = fr.Sy, Sz = fr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    const bool edge = e0 == 0.0f || e1 == 0.0f || e2 == 0.0f;
    if (__any(edge)) {   // :234-245 fp64 re-evaluation at triangle edges (rare: the whole wave skips it otherwise)
        if (edge) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
    }
    bool ok = !((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0));
    const Float det = e0 + e1 + e2;
    ok = ok && det != 0;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    const Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const Float tmd = fr.tMax * det;
    ok = ok && !(det < 0 && (tScaled >= 0 || tScaled < tmd)) && !(det > 0 && (tScaled <= 0 || tScaled > tmd));
    const Float invDet = 1 / det;
    const Float t = tScaled * invDet;
    const Float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    const Float deltaZ = gamma_n(3) * maxZt;
    const Float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    const Float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    const Float deltaX = gamma_n(5) * (maxXt + maxZt);
    const Float deltaY = gamma_n(5) * (maxYt + maxZt);
    const Float deltaE = 2 * (gamma_n(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    const Float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    const Float deltaT = 3 * (gamma_n(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absf(invDet);
    ok = ok && (t > deltaT) && !(flags & TRI_FLAG_REJECT);
    if (ok) { fr.prim = first; fr.tMax = t; }   // GeometricPrimitive::Intersect shrinks ray.tMax (core/primitive.cpp:120)
    if (SAFE) {
        if (ANY && ok) fr.cur = TRAV_DONE;
        else if (left) fr.cur = BVH4_LEAF | ((left - 1) << 27) | (first + 1);
        else fr.cur = st.pop(fr.tMax);
        return;
    }
    const bool popOk = !left && st.sp > 0;
    uint32_t nxt = left ? (BVH4_LEAF | ((left - 1) << 27) | (first + 1)) : (popOk ? top : TRAV_DONE);
    if (ANY && ok) nxt = TRAV_DONE;
    fr.cur = nxt;
    st.sp -= (popOk && !(ANY && ok)) ? 1 : 0;
}

// ------------------------------------------------------------------ BVH8C (80-byte compressed 8-wide nodes, pt_bvh8c.h)
// Same lean per-lane state machine over the compressed layout: 5 x 16-byte loads per interior step, nearest hit child next, the
// other hit children pushed with their entry distances (8-byte stack entries; an entry whose box lies beyond the hit found in the
// meantime is dropped at pop time without being fetched).  Triangles are read from the traversal-order copy of the records
// (sc.tri_trav; one copy: the axis permutation is done in registers -- with the memory side as the bound, 18 v_cndmask are
// cheaper than three times the triangle working set) and the hit is reported as the REFERENCE primitive index (sc.trav2prim).
struct Fast8Ray {
    V3 o, inv;                    // inv: +-1e30 stands in for the infinities of zero direction components (Ray8Init, pt_bvh8.h)
    RayShear shear;
    Float tMax;
    uint32_t prim, cur;           // prim: traversal-order index while the ray is in flight
    PT_DEV bool done() const { return cur == TRAV_DONE; }
    PT_DEV bool atLeaf() const { return cur != TRAV_DONE && (cur & BVH4_LEAF); }
    PT_DEV bool atNode() const { return !(cur & BVH4_LEAF); }
};
PT_DEV void Fast8RayInit(const DevScene &sc, Fast8Ray &fr, const V3 &o, const V3 &d, Float tMax, TravStack8 &st) {
    fr.o = o;
    fr.inv = V3(d.x == 0 ? __builtin_copysignf(1e30f, d.x) : 1 / d.x, d.y == 0 ? __builtin_copysignf(1e30f, d.y) : 1 / d.y,
                d.z == 0 ? __builtin_copysignf(1e30f, d.z) : 1 / d.z);
    fr.shear.init(d);
    fr.tMax = tMax;
    fr.prim = TRAV_MISS;
    st.sp = 0;
    fr.cur = sc.n_nodes8c ? 0u : TRAV_DONE;
}
// pop with culling; LDSONLY: the caller guarantees sp <= PT_LDS_STACK8 (no spill entries to look at)
template <bool LDSONLY>
PT_DEV uint32_t Fast8Pop(TravStack8 &st, Float tMax) {
    if (!LDSONLY) return st.pop(tMax);
    while (st.sp) {
        --st.sp;
        StackEntry8 e = st.lds[st.sp * PT_BLOCK];
        if (__uint_as_float((uint32_t)(e >> 32)) < tMax) return (uint32_t)e;
    }
    return TRAV_DONE;
}
template <bool COUNT>
PT_DEV void Fast8NodeStep(const DevScene &sc, Fast8Ray &fr, TravStack8 &st, TraceCounters *cnt) {
    const char *base = reinterpret_cast<const char *>(sc.nodes8c);
    const uint32_t nb = fr.cur * 80u;
    const uint4 w0 = LdU4(base, nb), w1 = LdU4(base, nb + 16u), w2 = LdU4(base, nb + 32u), w3 = LdU4(base, nb + 48u), w4 = LdU4(base, nb + 64u);
    if (COUNT) ++cnt->nodes;
    const uint32_t wd[20] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w};
    Float t[8];
    const uint32_t mask = Bvh8cStepWords(wd, fr.o.x, fr.o.y, fr.o.z, fr.inv.x, fr.inv.y, fr.inv.z, fr.tMax, t);
    // nearest hit child (lowest slot among equals: pieces of a split leaf are visited in primitive order)
    Float tb = PT_INFINITY;
    uint32_t refBest = TRAV_DONE;
    int best = -1;
    uint32_t refs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        refs[k] = Bvh8cChildRef(wd, k);
        const bool h = (mask >> k) & 1u;
        if (h && (best < 0 || t[k] < tb)) { tb = t[k]; best = k; refBest = refs[k]; }
    }
    const int nh = __builtin_popcount(mask);
    // LDS-only pushes / pop need the stack inside its LDS part before AND after the step: sp + (nh - 1) <= PT_LDS_STACK8 with pushes,
    // sp <= PT_LDS_STACK8 for the pop of a step without a hit child.  A deeper lane sends the wave through the general push / pop (rare).
    if (__any(st.sp + (nh > 0 ? nh : 1) > PT_LDS_STACK8 + 1)) {
#pragma unroll
        for (int k = 7; k >= 0; --k) if (((mask >> k) & 1u) && k != best) st.push(refs[k], t[k]);
        fr.cur = best >= 0 ? refBest : st.pop(fr.tMax);
        return;
    }
#pragma unroll
    for (int k = 7; k >= 0; --k) {   // the other hit children, slot order, with their entry distances (LDS part of the stack only)
        if (((mask >> k) & 1u) && k != best) {
            st.lds[st.sp * PT_BLOCK] = (StackEntry8)refs[k] | ((StackEntry8)__float_as_uint(t[k]) << 32);
            ++st.sp;
        }
    }
    fr.cur = best >= 0 ? refBest : Fast8Pop<true>(st, fr.tMax);
}
// one triangle of the leaf per step (as FastLeafStep), vertices permuted in registers
template <bool ANY, bool COUNT>
PT_DEV void Fast8LeafStep(const DevScene &sc, Fast8Ray &fr, TravStack8 &st, TraceCounters *cnt) {
    const uint32_t first = fr.cur & BVH4_FIRST_MASK, left = (fr.cur >> 27) & 0xfu;
    const float4 *tv = sc.tri_trav + 3 * (size_t)first;
    const float4 a = tv[0], b = tv[1], c = tv[2];
    if (COUNT) ++cnt->tris;
    const uint32_t flags = __float_as_uint(a.w);
    const RayShear &rs = fr.shear;
    V3 p0t = rs.permute(V3(a.x, a.y, a.z) - fr.o), p1t = rs.permute(V3(b.x, b.y, b.z) - fr.o), p2t = rs.permute(V3(c.x, c.y, c.z) - fr.o);
    const Float Sx = rs.Sx, Sy = rs.Sy, Sz = rs.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    const bool edge = e0 == 0.0f || e1 == 0.0f || e2 == 0.0f;
    if (__any(edge)) {
        if (edge) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
    }
    bool ok = !((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0));
    const Float det = e0 + e1 + e2;
    ok = ok && det != 0;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    const Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const Float tmd = fr.tMax * det;
    ok = ok && !(det < 0 && (tScaled >= 0 || tScaled < tmd)) && !(det > 0 && (tScaled <= 0 || tScaled > tmd));
    const Float invDet = 1 / det;
    const Float t = tScaled * invDet;
    const Float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    const Float deltaZ = gamma_n(3) * maxZt;
    const Float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    const Float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    const Float deltaX = gamma_n(5) * (maxXt + maxZt);
    const Float deltaY = gamma_n(5) * (maxYt + maxZt);
    const Float deltaE = 2 * (gamma_n(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    const Float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    const Float deltaT = 3 * (gamma_n(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absf(invDet);
    ok = ok && (t > deltaT) && !(flags & TRI_FLAG_REJECT);
    if (ok) { fr.prim = first; fr.tMax = t; }
    if (ANY && ok) { fr.cur = TRAV_DONE; return; }
    if (left) { fr.cur = BVH4_LEAF | ((left - 1) << 27) | (first + 1); return; }
    if (__any(st.sp > PT_LDS_STACK8)) fr.cur = Fast8Pop<false>(st, fr.tMax);
    else fr.cur = Fast8Pop<true>(st, fr.tMax);
}
