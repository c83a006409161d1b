/* cooperative_groups.h shim: the subset the reference kernels use (see cuda_runtime.h) */
#pragma once
#include "cuda_runtime.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const {
        const unsigned long long b = (unsigned long long)blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;
        const unsigned long long t = threadIdx.x + blockDim.x * threadIdx.y;
        return b * ((unsigned long long)blockDim.x * blockDim.y) + t;
    }
};
struct thread_block {
    dim3 group_index() const { return dim3(blockIdx.x, blockIdx.y, blockIdx.z); }
    dim3 thread_index() const { return dim3(threadIdx.x, threadIdx.y, threadIdx.z); }
    unsigned thread_rank() const { return threadIdx.x + blockDim.x * threadIdx.y; }
    void sync() const { __syncthreads(); }
};
static inline grid_group this_grid() { return grid_group(); }
static inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
