// Measurement aid (NOT product code): the VALU issue ceiling of one gfx950 SIMD, in SHADER CYCLES, with the shader clock measured in the same
// launch.  Round 3's probe (valu_probe.hip) timed 2-6 ms launches of DEPENDENT v_fma_f32 chains with host events and turned the time into
// cycles at an ASSUMED 2.4 GHz; VERDICT r3 item 3 asks for (i) independent instructions, (ii) the measured clock, (iii) the traversal's
// instruction mix.  Here every wave stamps s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz) around its loop:
//     cycles per wave instruction per SIMD = (s_memtime ticks of the loop) / (instructions per wave x waves per SIMD)
//     shader clock                         = s_memtime ticks / s_memrealtime ticks x 100 MHz
// and the host times the same launch with events (third opinion).  Launches are sized to ~40 ms and repeated so that DVFS has settled.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe/valu_probe2.hip -o tools/valu_probe/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// Each body is 64 VALU instructions per trip; operands chosen so that the 8 / 16 destination registers are independent (INDEP) or one chain (DEP).
template <int KIND>
__global__ void __launch_bounds__(256) k_probe(float *out, Stamp *st, int iters, unsigned long long mask) {
    const unsigned lane = threadIdx.x & 63u;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    uint32_t w0 = threadIdx.x * 2654435761u, w1 = w0 ^ 0x9e3779b9u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    const bool on = (mask >> lane) & 1ull;
    __syncthreads();
    if (on) {
        c0 = __builtin_readcyclecounter();   // s_memtime
        r0 = wall_clock64();                 // s_memrealtime
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {   // 8 independent v_fma_f32 chains
                REP4(REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
                          asm volatile("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
                     )   // 16 x 8 = 128 instructions per trip
            } else if (KIND == 1) {   // one dependent chain
                REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));)
            } else if (KIND == 2) {   // 4 independent v_pk_fma_f32 chains (8 FMAs per 4 instructions)
                REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
            } else if (KIND == 3) {   // the box test's mix: per child pair 2 x (v_cndmask (SGPR mask), 2 x v_cvt_f32_u32 sdwa, 2 x v_fma, then v_max3 / v_min3 / v_cmp) -- independent
                REP16(asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                   "v_cvt_f32_u32_sdwa %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_fma_f32 %3, %1, %10, %11\n"
                                   "v_fma_f32 %4, %2, %10, %11\n v_max3_f32 %5, %3, %4, %10\n v_min3_f32 %6, %3, %4, %11\n v_cmp_le_f32 vcc, %5, %6\n"
                                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(w0), "v"(w1), "v"(b), "v"(c) : "vcc");)
            } else if (KIND == 4) {   // v_mov_b32 (the cheapest VALU)
                REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(b));)
            } else if (KIND == 5) {   // VALU + SALU interleaved 1:1 (does a scalar instruction take a VALU issue slot of its wave's SIMD?)
                REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n"
                                   "v_fma_f32 %3, %3, %8, %9\n s_add_u32 s23, s23, 1\n v_fma_f32 %4, %4, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s21, s21, 1\n"
                                   "v_fma_f32 %6, %6, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s23, s23, 1\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "scc");)
            }
        }
        c1 = __builtin_readcyclecounter();
        r1 = wall_clock64();
    }
    a0 += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0) { Stamp s = {c0, c1, r0, r1}; st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s; }
}

static const int N_PER_TRIP[] = {128, 128, 128, 128, 128, 128};   // VALU instructions per loop trip of each KIND (KIND 5: + 128 SALU)
static const char *KIND_NAME[] = {"v_fma_f32, 8 independent chains", "v_fma_f32, one dependent chain", "v_pk_fma_f32, 4 independent chains (2 FMAs per lane per instruction)",
                                  "box-test mix (cndmask/cvt_sdwa/fma/max3/min3/cmp), independent", "v_mov_b32", "v_fma_f32 + s_add_u32 interleaved 1:1 (VALU count only)"};

template <int KIND> static void run(int cus, int wavesPerSimd, unsigned long long mask, const char *maskName, float *out, Stamp *st, int wallRateKHz) {
    const int blocks = cus * wavesPerSimd;   // 256-thread blocks = one wave per SIMD of a CU each; all co-resident
    const int iters = 400000 / wavesPerSimd;   // ~50 M instructions per SIMD per launch: 40-60 ms
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Stamp> h(blocks * 4);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {   // the first launch warms the clocks up; the second is reported
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, st, iters, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto &s : h) {
        if (s.c1 <= s.c0 || s.r1 <= s.r0) continue;
        cyc.push_back((double)(s.c1 - s.c0));
        clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * wallRateKHz * 1e-6);   // GHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double instrPerWave = (double)iters * N_PER_TRIP[KIND];
    const double medCyc = cyc.empty() ? 0 : cyc[cyc.size() / 2], medClk = clk.empty() ? 0 : clk[clk.size() / 2];
    printf("kind %d  waves/SIMD %d  %-10s  s_memtime: %.3f cycles per wave instruction per SIMD (= %.3f per wave)   clock %.3f GHz (s_memtime / s_memrealtime)   host events: %.2f ms -> %.3f ns per instruction per SIMD\n",
           KIND, wavesPerSimd, maskName, medCyc / (instrPerWave * wavesPerSimd), medCyc / instrPerWave, medClk, ms, ms * 1e6 / (instrPerWave * wavesPerSimd));
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int wallRate = 0;
    hipDeviceGetAttribute(&wallRate, hipDeviceAttributeWallClockRate, 0);
    if (wallRate <= 0) wallRate = 100000;
    printf("%s: %d CUs, clockRate attribute %d kHz, wall clock (s_memrealtime) %d kHz\n", p.name, cus, p.clockRate, wallRate);
    float *out; Stamp *st;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipMalloc(&st, (size_t)cus * 8 * 4 * sizeof(Stamp));
    const unsigned long long full = ~0ull, half = 0xFFFFFFFFull, alt = 0x5555555555555555ull;
    for (int k = 0; k < 6; ++k) printf("kind %d = %s\n", k, KIND_NAME[k]);
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>(cus, w, full, "64 lanes", out, st, wallRate);
        run<1>(cus, w, full, "64 lanes", out, st, wallRate);
        run<2>(cus, w, full, "64 lanes", out, st, wallRate);
        run<3>(cus, w, full, "64 lanes", out, st, wallRate);
        run<4>(cus, w, full, "64 lanes", out, st, wallRate);
        run<5>(cus, w, full, "64 lanes", out, st, wallRate);
    }
    for (int w : {6}) {   // partially filled waves at the traversal's occupancy
        run<0>(cus, w, half, "lower 32", out, st, wallRate);
        run<0>(cus, w, alt, "every 2nd", out, st, wallRate);
        run<3>(cus, w, half, "lower 32", out, st, wallRate);
        run<3>(cus, w, alt, "every 2nd", out, st, wallRate);
    }
    return 0;
}
