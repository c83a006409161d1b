// platform.h (CPU EMULATOR) -- TEST INFRASTRUCTURE ONLY.
//
// Same vocabulary as gs2mesh_amd/csrc/hip/platform.h, implemented on CPU fibers, so that the
// kernel SOURCE under gs2mesh_amd/csrc can be executed in the GPU-less build container and
// compared with the oracle before any GPU minute is spent.  Each workgroup runs as a set of
// ucontext fibers on one OS thread (deterministic, lock-step at barriers / wave collectives);
// workgroups are spread over OS threads with OpenMP.  Wavefront = 64 consecutive threads.
//
// The result, tests/emu/_build/libgs2mesh_emu.so, is loaded ONLY by tests (tests/emu_lib.py).
// The product package never loads it and has no CPU fallback.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <functional>

#define GS2M_KERNEL
#define GS2M_WAVES_PER_SIMD(n)
#define GS2M_DEVICE static inline
#define GS2M_PLATFORM_NAME "cpu-emulator"
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__
#define __forceinline__ inline
#define __host__
#define __device__

// ---- vector types -----------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
    unsigned x, y, z;
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(8) float2a { float x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct double3 { double x, y, z; };
struct uchar3 { unsigned char x, y, z; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

// ---- runtime (tests/emu/emu_runtime.cpp) -----------------------------------------------------
namespace emu {
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
void sync_block();
int sync_count(int pred);
unsigned long long ballot(int pred);
unsigned long long shfl_bits(unsigned long long bits, int src_lane);  // absolute lane in the wave
int lane();
void* dyn_lds();
// every live lane of the wave deposits `bytes` at `mine`; returns with all[64][bytes] filled (exited lanes: zeros)
void wave_gather(const void* mine, size_t bytes, void* all);
}  // namespace emu

#define GS2M_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(::emu::dyn_lds())
#define GS2M_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    ::emu::launch(grid, block, lds_bytes, [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { ::emu::sync_block(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() {}
static inline int gs2m_syncthreads_count(int pred) { return ::emu::sync_count(pred); }
static inline unsigned long long gs2m_ballot(int pred) { return ::emu::ballot(pred); }
static inline unsigned long long gs2m_ballot_b(bool pred) { return ::emu::ballot(pred ? 1 : 0); }
static inline bool gs2m_any_active_lane(bool) { return true; }   // divergent code: not a collective here (see hip/platform.h)
static inline bool gs2m_lanes(unsigned long long mask) { return (mask >> ::emu::lane()) & 1ull; }
static inline void gs2m_wave_sync() { (void)::emu::ballot(0); }
static inline int gs2m_uniform(int v) { return v; }
static inline unsigned gs2m_load_agent(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
static inline unsigned long long gs2m_load_agent(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
static inline unsigned gs2m_load_u32_unaligned(const unsigned char* p) { unsigned v; memcpy(&v, p, 4); return v; }
static inline int gs2m_lane() { return ::emu::lane(); }
static inline int gs2m_popc64(unsigned long long m) { return __builtin_popcountll(m); }
template <typename T>
static inline T gs2m_shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl of <= 8 bytes");
    unsigned long long b = 0;
    memcpy(&b, &v, sizeof(T));
    b = ::emu::shfl_bits(b, src_lane & 63);
    T r;
    memcpy(&r, &b, sizeof(T));
    return r;
}
template <typename T>
static inline T gs2m_shfl_xor(T v, int mask) { return gs2m_shfl(v, ::emu::lane() ^ mask); }
template <typename T>
static inline T gs2m_shfl_up(T v, int d) {
    int l = ::emu::lane();
    return gs2m_shfl(v, l - d >= 0 ? l - d : l);
}
static inline float gs2m_add_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float gs2m_fast_exp(float x) { return expf(x); }
static inline float gs2m_fast_log(float x) { return logf(x); }
static inline float gs2m_fast_rcp(float x) { return 1.0f / x; }
static inline float gs2m_fast_sqrt(float x) { return sqrtf(x); }
static inline float gs2m_fast_exp2(float x) { return exp2f(x); }
static inline float gs2m_fast_log2(float x) { return log2f(x); }
#define GS2M_KEEP_F32(x) ((void)0)
#define GS2M_NO_IF_CONVERT() ((void)0)
#define GS2M_OPAQUE_SGPR(x) ((void)0)
#define GS2M_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GS2M_SCHED_BARRIER() ((void)0)
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
static inline int __ffsll(unsigned long long m) { return __builtin_ffsll((long long)m); }
static inline int __clzll(long long m) { return m == 0 ? 64 : __builtin_clzll((unsigned long long)m); }
static inline int __popc(unsigned m) { return __builtin_popcount(m); }
static inline int __ffs(int m) { return __builtin_ffs(m); }

static inline void gs2m_global_load_lds16(const void* g, void* lds_base) { memcpy((char*)lds_base + 16 * ::emu::lane(), g, 16); }
static inline void gs2m_wait_dma() {}
static inline unsigned long long gs2m_clock() { return 0ull; }

// ---- bit casts / math ----------------------------------------------------------------------
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- atomics (real: workgroups run on different OS threads) -------------------------------
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMin(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMin(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}
static inline int atomicCAS(int* p, int cmp, int v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return cmp;
}

// ---- host API subset ------------------------------------------------------------------------
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
typedef void* hipStream_t;
struct emu_event {
    std::chrono::steady_clock::time_point t;
};
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

static inline hipError_t hipMalloc(void** p, size_t n) {
    size_t r = (n + 255) / 256 * 256;
    if (r == 0) r = 256;
    *p = aligned_alloc(256, r);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F>
static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
