// Measurement aid (NOT product code): how fast does a CU issue VALU instructions when only part of a wave's lanes are active?
// The traversal kernels run with 47 % of the lanes active per VALU instruction (profiles/r03_c_*) and their SQ counters add up to more VALU
// quad-cycles than the launch has -- which is only possible if a wave64 instruction whose upper / lower lanes are all inactive takes fewer than
// four cycles.  This probe times a long dependent chain of v_fma_f32 per wave at 1..8 waves per SIMD with an EXEC mask of 64, 32 (one half), 16
// (one quarter) and 32 scattered (every other lane) active lanes.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe/valu_probe.hip -o tools/valu_probe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_chain(float *out, int iters, unsigned long long mask) {
    const unsigned lane = threadIdx.x & 63u;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, a); }   // two dependent chains: 32 VALU per trip
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(float));
    const int iters = 20000;
    struct M { const char *name; unsigned long long m; } masks[] = {{"64 lanes", ~0ull}, {"lower 32", 0xFFFFFFFFull}, {"lower 16", 0xFFFFull}, {"every other lane (32)", 0x5555555555555555ull},
                                                                     {"lanes 0-15 + 32-47 (32)", 0x0000FFFF0000FFFFull}, {"one lane", 1ull}};
    printf("%d CUs, clock %d kHz; %d x 32 dependent v_fma_f32 per wave\n", cus, p.clockRate, iters);
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        for (auto &mk : masks) {
            const int blocks = cus * wavesPerSimd;   // 256-thread blocks = 4 waves = one wave per SIMD of a CU per block
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, out, 100, mk.m);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, out, iters, mk.m);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instrPerSimd = (double)iters * 32 * wavesPerSimd;
            printf("waves/SIMD %d  %-26s %8.3f ms  -> %.2f cycles per wave instruction per SIMD at %.2f GHz\n", wavesPerSimd, mk.name, ms, ms * 1e-3 * p.clockRate * 1e3 / instrPerSimd, p.clockRate * 1e-6);
        }
    }
    return 0;
}
