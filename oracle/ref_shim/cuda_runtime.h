/* cuda_runtime.h -- CUDA-on-CPU execution shim for oracle/_ref (TEST INFRASTRUCTURE ONLY).
 *
 * Lets the REFERENCE's own device code (diff-gaussian-rasterization cuda_rasterizer/forward.cu, auxiliary.h and
 * the three small kernels of rasterizer_impl.cu) compile with g++ from the sources where they lie under
 * /root/reference, so that the restated oracle (oracle/raster_oracle.c) can be checked against it.  Nothing of
 * the reference is copied into this repository: oracle/build_ref.py cuts the host launchers (the `<<< >>>`
 * syntax) out of a temporary copy under oracle/_ref/ (git-ignored) at build time.
 *
 * Execution model: a kernel "launch" runs the grid block after block; the threads of a block are ucontext
 * fibers on one OS thread, __syncthreads / block.sync / __syncthreads_count switch back to a round-robin
 * scheduler (oracle/ref_driver.cpp).  __shared__ is static thread_local: one instance per OS thread = per block
 * in flight. */
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define CUDA_VERSION 11080

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

/* CUDA's global min/max overload set (the mixed signed/unsigned ones convert to unsigned, as CUDA does) */
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
static inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
static inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
static inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

namespace refshim {
struct ThreadCoords {
    uint3 tid, bid;
    dim3 bdim, gdim;
};
extern thread_local ThreadCoords tc;
int barrier(int pred); /* yields to the block scheduler; returns the number of threads whose pred != 0 */
}  // namespace refshim
#define threadIdx (refshim::tc.tid)
#define blockIdx (refshim::tc.bid)
#define blockDim (refshim::tc.bdim)
#define gridDim (refshim::tc.gdim)
static inline void __syncthreads() { (void)refshim::barrier(0); }
static inline int __syncthreads_count(int pred) { return refshim::barrier(pred); }
static inline void __trap() { abort(); }
