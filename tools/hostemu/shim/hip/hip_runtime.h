// DEVELOPER TOOL, NOT PART OF THE PRODUCT: a minimal stand-in for <hip/hip_runtime.h> that lets the device sources of
// pbrt-v3-distributed_amd/csrc be compiled for the x86 host, so that kernel LOGIC (queue bookkeeping, sampler dimension order,
// medium / BSSRDF state machines) can be stepped through under gdb / sanitizers when no GPU is at hand.  One-lane "waves",
// one-thread blocks (PT_BLOCK is rewritten to 1 by build.sh), blocks run on host threads.  Nothing under pbrt-v3-distributed_amd/,
// tests/, bench.py or __graft_entry__.py builds, loads or refers to this; see tools/hostemu/README.md.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __constant__
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) x

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct int2 { int x, y; };
struct float3 { float x, y, z; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return {x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };
struct emu_idx { uint32_t x = 0, y = 0, z = 0; };
extern thread_local emu_idx threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
extern thread_local void *emu_dyn_lds;

// ---- wave intrinsics of a one-lane wave
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline bool __any(bool p) { return p; }
static inline bool __all(bool p) { return p; }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_down(T, int) { return T(); }   // the source lane is outside the wave: contributes nothing to a reduction
template <class T> static inline T __shfl_xor(T v, int) { return v; }
static inline uint32_t __lane_id() { return 0; }
static inline void __syncthreads() {}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline uint32_t __brev(uint32_t v) { v = (v >> 16) | (v << 16); v = ((v & 0xff00ff00u) >> 8) | ((v & 0x00ff00ffu) << 8); v = ((v & 0xf0f0f0f0u) >> 4) | ((v & 0x0f0f0f0fu) << 4); v = ((v & 0xccccccccu) >> 2) | ((v & 0x33333333u) << 2); return ((v & 0xaaaaaaaau) >> 1) | ((v & 0x55555555u) << 1); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }   // one-lane waves

#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_global_load_lds(src, dst, size, off, aux) ((void)(src), (void)(dst))   /* cache-warming loads: nothing to emulate */
#define __builtin_amdgcn_s_waitcnt(v) ((void)0)

template <class T> static inline T emu_atomic_add_int(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return emu_atomic_add_int(p, v); }
static inline int atomicAdd(int *p, int v) { return emu_atomic_add_int(p, v); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return emu_atomic_add_int(p, v); }
static inline float atomicAdd(float *p, float v) {
    uint32_t *u = reinterpret_cast<uint32_t *>(p), old = __atomic_load_n(u, __ATOMIC_RELAXED);
    while (true) { float nf = __uint_as_float(old) + v; uint32_t nu = __float_as_uint(nf); if (__atomic_compare_exchange_n(u, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return __uint_as_float(old); }
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) { uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return old; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// k_shade's three queue appends use lanes 0..2 of the wave for the three atomics; a one-lane wave issues them one after the other
static inline void emu_wave_append3(uint32_t *c0, uint32_t *c1, uint32_t *c2, bool a0, bool a1, bool a2, uint32_t *p0, uint32_t *p1, uint32_t *p2) {
    *p0 = a0 ? atomicAdd(c0, 1u) : 0u; *p1 = a1 ? atomicAdd(c1, 1u) : 0u; *p2 = a2 ? atomicAdd(c2, 1u) : 0u;
}

// ---- runtime API
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct emu_stream *hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; size_t totalGlobalMem; };
static inline const char *hipGetErrorString(hipError_t) { return "host emulation error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
enum { hipDeviceAttributeWallClockRate = 10017 };
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 100000; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    const char *e = std::getenv("PBRT_EMU_CUS");
    p->multiProcessorCount = e ? std::atoi(e) : 2; std::snprintf(p->name, sizeof p->name, "host emulation"); p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = nullptr; return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorUnknown; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)2 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbolAsync(T &sym, const void *s, size_t n, size_t off, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy((char *)&sym + off, s, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { std::memcpy((char *)&sym + off, s, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches are synchronous here
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

// ---- kernel launch: blocks spread over host threads, the threads of a block run one after the other (kernels that
// synchronise inside a block are only correct with one-thread blocks, which is what build.sh configures)
template <class F> static inline void emu_launch(dim3 grid, dim3 block, size_t shmem, F &&body) {
    static int nthreads = [] { const char *e = std::getenv("PBRT_EMU_THREADS"); int n = e ? std::atoi(e) : 8; return n < 1 ? 1 : n; }();
    std::atomic<uint32_t> next(0);
    const uint32_t nblocks = grid.x * grid.y * grid.z;
    auto worker = [&]() {
        std::vector<char> lds(shmem + 16);
        emu_dyn_lds = lds.data();
        gridDim = grid; blockDim = block;
        for (uint32_t b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1)) {
            blockIdx.x = b % grid.x; blockIdx.y = (b / grid.x) % grid.y; blockIdx.z = b / (grid.x * grid.y);
            for (uint32_t t = 0; t < block.x * block.y * block.z; ++t) {
                threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
                body();
            }
        }
    };
    int nt = (int)std::min<uint32_t>((uint32_t)nthreads, nblocks);
    if (nt <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; ++i) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
}
#define hipLaunchKernelGGL(K, G, B, S, ST, ...) emu_launch(dim3(G), dim3(B), (size_t)(S), [&]() { K(__VA_ARGS__); })
