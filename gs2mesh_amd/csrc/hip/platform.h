// platform.h (product build) -- the only platform the shipped library is built for:
// HIP on gfx950 (MI355X, CDNA4, wave64).  Kernels include "platform.h" and use the small
// vocabulary below; tests/emu/platform.h provides the same vocabulary on CPU fibers so the
// kernel SOURCE can be executed in the GPU-less build container (test infrastructure only,
// never linked into libgs2mesh_amd.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS2M_KERNEL __global__
#define GS2M_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // occupancy of a kernel by construction (register budget = 512 / n)
#define GS2M_DEVICE __device__ __forceinline__
#define GS2M_PLATFORM_NAME "hip-gfx950"

// Dynamic LDS, 16-byte aligned (guide G17: keep the dynamic-LDS base 16-B aligned).
#define GS2M_DYN_LDS(type, name)                                              \
    extern __shared__ __attribute__((aligned(16))) unsigned char gs2m_dyn_lds_raw[]; \
    type* name = reinterpret_cast<type*>(gs2m_dyn_lds_raw)

#define GS2M_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, (hipStream_t)(stream), __VA_ARGS__)

// wave64 vocabulary
GS2M_DEVICE unsigned long long gs2m_ballot(int pred) { return __ballot(pred); }
// ballot of a predicate that is already a lane mask (no 0/1 materialisation + v_cmp_ne as __ballot(int) costs), and its
// inverse: a wave-uniform 64-bit mask used as a per-lane predicate (no instruction: the mask IS the v_cndmask operand)
GS2M_DEVICE unsigned long long gs2m_ballot_b(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
GS2M_DEVICE bool gs2m_lanes(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// "does any ACTIVE lane satisfy pred?" inside divergent code (ballot under the execution mask).  Only ever used to skip
// work no lane needs: taking the branch when the answer is no must be harmless (the CPU emulator always answers yes).
GS2M_DEVICE bool gs2m_any_active_lane(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }
GS2M_DEVICE int gs2m_lane() { return (int)(threadIdx.x & 63u); }
GS2M_DEVICE int gs2m_popc64(unsigned long long m) { return __popcll(m); }
template <typename T>
GS2M_DEVICE T gs2m_shfl(T v, int src_lane) { return __shfl(v, src_lane, 64); }
template <typename T>
GS2M_DEVICE T gs2m_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
template <typename T>
GS2M_DEVICE T gs2m_shfl_up(T v, int d) { return __shfl_up(v, d, 64); }

// Orders LDS traffic of ONE wave (producer lanes -> consumer lanes of the same wave): DS ops of a wave
// execute in order, so only the compiler must be fenced.
GS2M_DEVICE void gs2m_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Loads that bypass the (non-coherent) per-CU L1: see the latest value another CU published with an
// atomic (guide: relaxed agent-scope load = global_load sc1).
GS2M_DEVICE unsigned gs2m_load_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GS2M_DEVICE unsigned long long gs2m_load_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 4 bytes at an arbitrary byte address (gfx9+ global loads need no alignment)
GS2M_DEVICE unsigned gs2m_load_u32_unaligned(const unsigned char* p) {
    typedef unsigned __attribute__((aligned(1))) u32_u;
    return *reinterpret_cast<const u32_u*>(p);
}
// value known to be identical in all lanes -> SGPR (lets the compiler branch on the scalar unit)
GS2M_DEVICE int gs2m_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
GS2M_DEVICE int gs2m_syncthreads_count(int pred) { return __syncthreads_count(pred); }

// fast exp for the blend kernel: v_exp_f32(x * log2e) (documented tolerance in DESIGN.md)
GS2M_DEVICE float gs2m_fast_exp(float x) { return __expf(x); }
GS2M_DEVICE float gs2m_fast_log(float x) { return __logf(x); }
GS2M_DEVICE float gs2m_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // v_rcp_f32 (1 ulp)
GS2M_DEVICE float gs2m_fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32 (1 ulp)
GS2M_DEVICE float gs2m_fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (flushes denormal results)
GS2M_DEVICE float gs2m_fast_log2(float x) { return __log2f(x); }
// a + b rounded on its own: never contracted into an FMA with a neighbouring product (two code shapes that must round alike)
// (HIP's __fadd_rn is a plain `a + b`, which -ffp-contract=fast may fuse with a product feeding it: one v_add_f32, by hand)
GS2M_DEVICE float gs2m_add_rn(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// keep a loop-invariant float in its VGPR (stops the compiler re-materialising int->float converts in hot loops)
#define GS2M_KEEP_F32(x) asm volatile("" : "+v"(x))
// a wave-uniform int the compiler must treat as an opaque scalar register (stops re-association of mask tests)
#define GS2M_OPAQUE_SGPR(x) asm volatile("" : "+s"(x))
#define GS2M_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))   // a per-lane value the compiler must re-read (stops hoisting of loads addressed by it)
// instruction-scheduling fence (nothing moves across it)
// keeps a rarely taken, wave-uniform branch a BRANCH (stops the compiler turning it into selects executed every time)
#define GS2M_NO_IF_CONVERT() asm volatile("" ::: "memory")
#define GS2M_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)

// ---- LDS-DMA gather: lane l's 16 bytes at `g` land at lds_base[l] without passing through a VGPR
// (global_load_lds_dwordx4; completion is tracked by vmcnt: gs2m_wait_dma before the LDS is read)
GS2M_DEVICE void gs2m_global_load_lds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// shader clock (s_memtime): phase stamps of the profile builds
// One asm statement: the value is complete when the statement ends (the scalar-memory return is asynchronous -- a
// destination the compiler believes free would be overwritten later), and it waits for the scalar / LDS counter only, never
// for vector memory (the builtin made the compiler drain the record DMA at every stamp: 7.5x slower kernel).
GS2M_DEVICE unsigned long long gs2m_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
GS2M_DEVICE void gs2m_wait_dma() { __builtin_amdgcn_s_waitcnt(0x0f70); }  // s_waitcnt vmcnt(0)
