// Measurement aid (NOT product code): issue cost of individual gfx950 instructions at full occupancy, in shader cycles per wave instruction per SIMD.
// valu_probe2 showed (profiles/r04_a_valu_probe2.txt) that one SIMD issues a plain wave64 VALU instruction every ~2.3 cycles, that a scalar instruction
// interleaved with VALU work costs about as much, and that the box test's instruction mix runs at 3.4 cycles per instruction -- so some of its
// instructions are slower than v_fma_f32.  This probe times each candidate alone: 128 copies per loop trip, 8 independent destination registers,
// 8 waves per SIMD (32 per CU), throughput from HOST EVENTS over a ~50 ms launch (the per-wave s_memtime view is distorted by the oldest-first
// arbitration between waves) and the shader clock from s_memtime / s_memrealtime of the same launch.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe/issue_probe.hip -o tools/valu_probe/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct Stamp { unsigned long long c0, c1, r0, r1; };
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// one asm block = 8 instructions writing v-regs %0..%7 (independent), reading %8 %9 %10 (and whatever the op string names)
#define BODY8(op) asm volatile(op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(w), "v"(w4), "v"(w8) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "scc", "v40", "v41", "v42", "v43", "v44", "v45");

#define OP_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define OP_FMA(i)      "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CVT(i)      "v_cvt_f32_u32 %" #i ", %10\n"
#define OP_CVT_SDWA(i) "v_cvt_f32_u32_sdwa %" #i ", %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define OP_CVT_UB(i)   "v_cvt_f32_ubyte1 %" #i ", %10\n"
#define OP_AND(i)      "v_and_b32 %" #i ", 0xffff, %10\n"
#define OP_LSHR(i)     "v_lshrrev_b32 %" #i ", 16, %10\n"
#define OP_CNDM_VCC(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define OP_CNDM_SG(i)  "v_cndmask_b32_e64 %" #i ", %8, %9, s[20:21]\n"
#define OP_CMP_VCC(i)  "v_cmp_le_f32 vcc, %8, %" #i "\n"
#define OP_CMP_SG(i)   "v_cmp_le_f32_e64 s[22:23], %8, %" #i "\n"
#define OP_MAX3(i)     "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN(i)      "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP_PERM(i)     "v_perm_b32 %" #i ", %10, %8, %9\n"
#define OP_FMAMIX(i)   "v_fma_mix_f32 %" #i ", %10, %8, %9 op_sel_hi:[1,0,0]\n"
#define OP_ADDU(i)     "v_add_u32 %" #i ", %" #i ", %10\n"
#define OP_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 2, %10\n"
#define OP_BFE(i)      "v_bfe_u32 %" #i ", %10, 16, 16\n"
#define OP_MULLO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %10\n"
#define OP_MAD24(i)    "v_mad_u32_u24 %" #i ", %" #i ", %10, %10\n"
#define OP_DPP(i)      "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_SAND(i)     "s_and_b64 s[20:21], s[22:23], s[24:25]\n"
#define OP_SNOP(i)     "s_nop 0\n"
#define OP_SAVEEXEC(i) "s_and_saveexec_b64 s[20:21], s[22:23]\n s_or_b64 exec, exec, s[20:21]\n"   /* 2 scalar instructions */
#define OP_BRANCH(i)   "s_cbranch_scc1 1f\n1:\n"   /* (scc = 0 at entry of the block or not: falls through either way) */
#define OP_WAITCNT(i)  "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
#define OP_RCP(i)      "v_rcp_f32 %" #i ", %8\n"
#define OP_MULF64(i)   "v_mul_f64 v[40:41], v[42:43], v[44:45]\n"
#define OP_BCNT(i)     "v_bcnt_u32_b32 %" #i ", %10, 0\n"
#define OP_DSRD(i)     "ds_read_b32 %" #i ", %10\n"
#define OP_DSRD128(i)  "ds_read_b128 v[40:43], %10\n"
#define OP_DSWR(i)     "ds_write_b32 %10, %8\n"
#define OP_BPERM(i)    "ds_bpermute_b32 %" #i ", %10, %8\n"

#define OP_ADDF(i)     "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP_MULF(i)     "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_MAXF_ND(i)  "v_max_f32 %" #i ", %8, %9\n"
#define OP_MINI(i)     "v_min_i32 %" #i ", %" #i ", %10\n"
#define OP_MAXU(i)     "v_max_u32 %" #i ", %" #i ", %10\n"
#define OP_OR(i)       "v_or_b32 %" #i ", %" #i ", %10\n"
#define OP_XOR(i)      "v_xor_b32 %" #i ", %" #i ", %10\n"
#define OP_LSHL(i)     "v_lshlrev_b32 %" #i ", 4, %10\n"
#define OP_SUBU(i)     "v_sub_u32 %" #i ", %" #i ", %10\n"
#define OP_ANDOR(i)    "v_and_or_b32 %" #i ", %10, %8, %9\n"
#define OP_OR3(i)      "v_or3_b32 %" #i ", %10, %8, %9\n"
#define OP_ADD3(i)     "v_add3_u32 %" #i ", %10, %8, %9\n"
#define OP_CMPU(i)     "v_cmp_lt_u32 vcc, %10, %" #i "\n"
#define OP_CMPI_SG(i)  "v_cmp_lt_i32_e64 s[22:23], %10, %" #i "\n"
#define OP_FMA_ND(i)   "v_fma_f32 %" #i ", %8, %9, %10\n"
#define OP_FMAC(i)     "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_MAC_ND(i)   "v_mul_f32 %" #i ", %8, %9\n"
#define OP_CNDM_E64V(i) "v_cndmask_b32_e64 %" #i ", %8, %9, vcc\n"
#define OP_CNDM_VCC2(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define OP_ADDC(i)     "v_addc_co_u32 %" #i ", vcc, %8, %9, vcc\n"
#define OP_PKFMA(i)    "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[40:41]\n"
#define OP_PKMUL(i)    "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n"
#define OP_CVTF16(i)   "v_cvt_f32_f16 %" #i ", %10\n"
#define OP_MED3(i)     "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIN3I(i)    "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
#define OP_DSRD_CF(i)  "ds_read_b32 %" #i ", %11\n"
#define OP_DSWR_CF(i)  "ds_write_b32 %11, %8\n"
#define OP_DSRD64_CF(i) "ds_read_b64 v[40:41], %12\n"
#define OP_BPERM_CF(i) "ds_bpermute_b32 %" #i ", %11, %8\n"
#define OP_READLANE(i) "v_readlane_b32 s20, %8, 3\n"
#define OP_READFIRST(i) "v_readfirstlane_b32 s20, %8\n"
#define OP_SMOV(i)     "s_mov_b32 s20, s21\n"
#define OP_SBCNT(i)    "s_bcnt1_i32_b64 s20, s[22:23]\n"
#define OP_MBCNT(i)    "v_mbcnt_lo_u32_b32 %" #i ", -1, %10\n"
#define OP_CMPX(i)     "v_cmpx_le_f32 %8, %" #i "\n s_mov_b64 exec, s[24:25]\n"

#define OP_MAD64(i)    "v_mad_u64_u32 v[40:41], s[22:23], %10, %10, v[42:43]\n"
#define OP_LSHLADD64(i) "v_lshl_add_u64 v[40:41], v[42:43], 2, v[44:45]\n"
#define OP_LSHL64(i)   "v_lshlrev_b64 v[40:41], 6, v[42:43]\n"
#define OP_MULHI(i)    "v_mul_hi_u32 %" #i ", %" #i ", %10\n"
#define OP_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %10\n"
#define OP_ASHR(i)     "v_ashrrev_i32 %" #i ", 31, %10\n"
#define OP_SUBF(i)     "v_sub_f32 %" #i ", %" #i ", %8\n"
#define OP_MOV64(i)    "v_mov_b64 v[40:41], v[42:43]\n"
#define OP_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %8, %9, %8\n"
#define OP_DIVFMAS(i)  "v_div_fmas_f32 %" #i ", %8, %9, %8\n"
#define OP_DIVFIX(i)   "v_div_fixup_f32 %" #i ", %8, %9, %8\n"
#define OP_SQRT(i)     "v_sqrt_f32 %" #i ", %8\n"
#define OP_CVTF64(i)   "v_cvt_f64_f32 v[40:41], %8\n"
#define OP_ADDF64(i)   "v_add_f64 v[40:41], v[42:43], v[44:45]\n"
#define OP_FMAF64(i)   "v_fma_f64 v[40:41], v[42:43], v[44:45], v[40:41]\n"
#define OP_CVTF32F64(i) "v_cvt_f32_f64 %" #i ", v[42:43]\n"
#define OP_WRLANE(i)   "v_writelane_b32 %" #i ", s20, 5\n"
#define OP_BFI(i)      "v_bfi_b32 %" #i ", %10, %8, %9\n"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %10, %10, 16\n"
#define OP_CVTPK(i)    "v_cvt_pkrtz_f16_f32 %" #i ", %8, %9\n"
#define OP_MAXF16PK(i) "v_pk_max_f16 %" #i ", %8, %9\n"
#define OP_FMA_2DEP(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i "\n"
// round 6, last session: an f16 half of a packed word as the multiplicand of an f32 FMA (the candidate node format of tools/isa_probe/node_step_probe.hip: bvh4h_lane)
#define OP_FMAMIX_LO(i) "v_fma_mix_f32 %" #i ", %10, %8, %9 op_sel_hi:[1,0,0]\n"
#define OP_FMAMIX_HI(i) "v_fma_mix_f32 %" #i ", %10, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define OP_FMAMIX_DEP(i) "v_fma_mix_f32 %" #i ", %10, %8, %" #i " op_sel_hi:[1,0,0]\n"

#define KINDS(X) X(0, OP_MOV, "v_mov_b32") X(1, OP_FMA, "v_fma_f32") X(2, OP_CVT, "v_cvt_f32_u32 (VOP1)") X(3, OP_CVT_SDWA, "v_cvt_f32_u32_sdwa src0_sel:WORD_1") \
    X(4, OP_CVT_UB, "v_cvt_f32_ubyte1") X(5, OP_AND, "v_and_b32 (literal)") X(6, OP_LSHR, "v_lshrrev_b32") X(7, OP_CNDM_VCC, "v_cndmask_b32 (vcc)") \
    X(8, OP_CNDM_SG, "v_cndmask_b32_e64 (SGPR pair)") X(9, OP_CMP_VCC, "v_cmp_le_f32 -> vcc") X(10, OP_CMP_SG, "v_cmp_le_f32_e64 -> SGPR pair") X(11, OP_MAX3, "v_max3_f32") \
    X(12, OP_MIN, "v_min_f32") X(13, OP_PERM, "v_perm_b32") X(14, OP_FMAMIX, "v_fma_mix_f32 (f16 hi half x f32 + f32)") X(15, OP_ADDU, "v_add_u32") X(16, OP_LSHLADD, "v_lshl_add_u32") \
    X(17, OP_BFE, "v_bfe_u32") X(18, OP_MULLO, "v_mul_lo_u32") X(19, OP_MAD24, "v_mad_u32_u24") X(20, OP_DPP, "v_mov_b32_dpp quad_perm") X(21, OP_SAND, "s_and_b64") \
    X(22, OP_SNOP, "s_nop 0") X(23, OP_SAVEEXEC, "s_and_saveexec_b64 + s_or_b64 exec (2 instructions per count)") X(24, OP_BRANCH, "s_cbranch_scc1 (to the next instruction)") \
    X(25, OP_WAITCNT, "s_waitcnt (nothing outstanding)") X(26, OP_RCP, "v_rcp_f32") X(27, OP_MULF64, "v_mul_f64") X(28, OP_BCNT, "v_bcnt_u32_b32") X(29, OP_DSRD, "ds_read_b32 (lane x 16 B: 8-way conflicts)") \
    X(30, OP_DSRD128, "ds_read_b128 (conflict-free)") X(31, OP_DSWR, "ds_write_b32 (lane x 16 B: 8-way conflicts)") X(32, OP_BPERM, "ds_bpermute_b32 (lane x 16)") \
    X(33, OP_ADDF, "v_add_f32") X(34, OP_MULF, "v_mul_f32") X(35, OP_MAXF_ND, "v_max_f32 (no self dependency)") X(36, OP_MINI, "v_min_i32") X(37, OP_MAXU, "v_max_u32") \
    X(38, OP_OR, "v_or_b32") X(39, OP_XOR, "v_xor_b32") X(40, OP_LSHL, "v_lshlrev_b32") X(41, OP_SUBU, "v_sub_u32") X(42, OP_ANDOR, "v_and_or_b32") X(43, OP_OR3, "v_or3_b32") \
    X(44, OP_ADD3, "v_add3_u32") X(45, OP_CMPU, "v_cmp_lt_u32 -> vcc") X(46, OP_CMPI_SG, "v_cmp_lt_i32_e64 -> SGPR pair") X(47, OP_FMA_ND, "v_fma_f32 (no self dependency)") \
    X(48, OP_FMAC, "v_fmac_f32 (VOP2)") X(49, OP_MAC_ND, "v_mul_f32 (no self dependency)") X(50, OP_CNDM_E64V, "v_cndmask_b32_e64 (vcc)") X(51, OP_CNDM_VCC2, "v_cndmask_b32 (vcc), dst = src0") \
    X(52, OP_ADDC, "v_addc_co_u32 (vcc in and out)") X(53, OP_PKFMA, "v_pk_fma_f32") X(54, OP_PKMUL, "v_pk_mul_f32") X(55, OP_CVTF16, "v_cvt_f32_f16") X(56, OP_MED3, "v_med3_f32") \
    X(57, OP_MIN3I, "v_min3_i32") X(58, OP_DSRD_CF, "ds_read_b32, lane x 4 B (conflict-free)") X(59, OP_DSWR_CF, "ds_write_b32, lane x 4 B (conflict-free)") \
    X(60, OP_DSRD64_CF, "ds_read_b64, lane x 8 B (conflict-free)") X(61, OP_BPERM_CF, "ds_bpermute_b32, lane x 4") X(62, OP_READLANE, "v_readlane_b32") \
    X(63, OP_READFIRST, "v_readfirstlane_b32") X(64, OP_SMOV, "s_mov_b32") X(65, OP_SBCNT, "s_bcnt1_i32_b64") X(66, OP_MBCNT, "v_mbcnt_lo_u32_b32") X(67, OP_CMPX, "v_cmpx_le_f32 + s_mov exec (2 per count)") \
    X(68, OP_MAD64, "v_mad_u64_u32") X(69, OP_LSHLADD64, "v_lshl_add_u64") X(70, OP_LSHL64, "v_lshlrev_b64") X(71, OP_MULHI, "v_mul_hi_u32") X(72, OP_ADDCO, "v_add_co_u32 (-> vcc)") \
    X(73, OP_ASHR, "v_ashrrev_i32") X(74, OP_SUBF, "v_sub_f32") X(75, OP_MOV64, "v_mov_b64") X(76, OP_DIVSCALE, "v_div_scale_f32") X(77, OP_DIVFMAS, "v_div_fmas_f32") X(78, OP_DIVFIX, "v_div_fixup_f32") \
    X(79, OP_SQRT, "v_sqrt_f32") X(80, OP_CVTF64, "v_cvt_f64_f32") X(81, OP_ADDF64, "v_add_f64") X(82, OP_FMAF64, "v_fma_f64") X(83, OP_CVTF32F64, "v_cvt_f32_f64") X(84, OP_WRLANE, "v_writelane_b32") \
    X(85, OP_BFI, "v_bfi_b32") X(86, OP_ALIGNBIT, "v_alignbit_b32") X(87, OP_CVTPK, "v_cvt_pkrtz_f16_f32") X(88, OP_MAXF16PK, "v_pk_max_f16") X(89, OP_FMA_2DEP, "v_fma_f32 (dst = src0 = src2)") \
    X(90, OP_FMAMIX_LO, "v_fma_mix_f32 (f16 low half x f32 + f32, no self dependency)") X(91, OP_FMAMIX_HI, "v_fma_mix_f32 (f16 high half)") X(92, OP_FMAMIX_DEP, "v_fma_mix_f32 (dst = src2)")
#define NKINDS 33

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(float *out, Stamp *st, int iters) {
    __shared__ float lds[4096];
    const unsigned lane = threadIdx.x & 63u;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f + a0, c = 0.5f;
    uint32_t w = (threadIdx.x * 16u) & 16383u;   // also a conflict-free, 16-byte aligned LDS byte address
    const uint32_t w4 = (threadIdx.x & 63u) * 4u, w8 = (threadIdx.x & 63u) * 8u;
    lds[threadIdx.x] = a0;
    __syncthreads();
    asm volatile("s_mov_b64 s[22:23], exec\n s_mov_b64 s[24:25], exec\n s_mov_b64 s[20:21], exec\n v_cmp_eq_u32 vcc, %0, %0\n" :: "v"(w) : "s20", "s21", "s22", "s23", "s24", "s25", "vcc");
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#define X(k, OP, name) if (KIND == k) { REP16(BODY8(OP)) }
        KINDS(X)
#undef X
        if ((KIND >= 29 && KIND <= 32) || (KIND >= 58 && KIND <= 61)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[(threadIdx.x * 7) & 4095];
    if (lane == 0) { Stamp s = {c0, c1, r0, r1}; st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s; }
}

template <int KIND> static void run(const char *name, int cus, float *out, Stamp *st, int wallKHz, double base) {
    const int wavesPerSimd = 8, blocks = cus * wavesPerSimd, iters = 25000;   // 25000 x 128 x 8 = 25.6 M instructions per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<Stamp> h(blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (auto &s : h) if (s.c1 > s.c0 && s.r1 > s.r0) clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * wallKHz * 1e-6);
    std::sort(clk.begin(), clk.end());
    const double ghz = clk.empty() ? 0 : clk[clk.size() / 2], n = (double)iters * 128 * wavesPerSimd;
    const double cyc = ms * 1e6 * ghz / n;
    printf("%2d  %-62s %8.2f ms  clock %.3f GHz  %.3f cycles per instruction per SIMD\n", KIND, name, ms, ghz, cyc);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int wallRate = 0;
    if (hipDeviceGetAttribute(&wallRate, hipDeviceAttributeWallClockRate, 0) != hipSuccess || wallRate <= 0) wallRate = 100000;
    float *out; Stamp *st;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    (void)hipMalloc(&st, (size_t)cus * 8 * 4 * sizeof(Stamp));
    printf("%s: %d CUs; 8 waves per SIMD, 128 copies of the instruction per loop trip, throughput from host events\n", p.name, cus);
#define X(k, OP, name) if (k >= first) run<k>(name, cus, out, st, wallRate, 0);
    KINDS(X)
#undef X
    return 0;
}
