// Micro-benchmark (round 5): could k_scatter take its key positions from GLOBAL atomic cursors (one per
// tile, or one per (XCD, tile)) instead of workgroup-private LDS cursors over per-workgroup segments?  That layout would let an
// XCD's keys of a tile land densely in arrival order (whole lines in its L2) instead of as 32 one-key segments.
// Shape of a C2 two-pair launch: 4.45 M keys over 4 views x 3800 tiles, 256 workgroups of 1024 threads, ~17 keys per thread.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/atomic_cursors.hip -o tools/ubench/atomic_cursors && tools/ubench/atomic_cursors
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// MODE 0: one global cursor per tile; 1: one per (XCD, tile) (blockIdx % 8 = the XCD a workgroup runs on);
// 2: LDS cursors (today's scheme: positions inside the workgroup's own segment), the same store pattern as today
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned* cursors, unsigned long long* keys, unsigned n_tiles, unsigned per_thread, unsigned cap,
                                          unsigned seg /* keys per (workgroup, tile) segment, MODE 2 */) {
    extern __shared__ unsigned lcur[];
    if (MODE == 2) {
        for (unsigned i = threadIdx.x; i < n_tiles; i += 1024) lcur[i] = 0;
        __syncthreads();
    }
    const unsigned xcd = blockIdx.x % 8u;
    unsigned s = rnd(blockIdx.x * 1024u + threadIdx.x + 1u);
    for (unsigned j = 0; j < per_thread; ++j) {
        s = rnd(s + j);
        const unsigned t = s % n_tiles;
        unsigned pos;
        if (MODE == 0) pos = atomicAdd(&cursors[t], 1u) + t * (cap / n_tiles);
        else if (MODE == 1) pos = atomicAdd(&cursors[xcd * n_tiles + t], 1u) + t * (cap / n_tiles) + xcd * (cap / n_tiles / 8u);
        else pos = atomicAdd(&lcur[t], 1u) + t * (cap / n_tiles) + (blockIdx.x % 8u * 32u + blockIdx.x / 8u) * seg;
        if (pos < cap) keys[pos] = ((unsigned long long)s << 32) | j;
    }
}

int main() {
    const unsigned n_tiles = 4 * 3800, per_thread = 17, grid = 256;
    const unsigned total = grid * 1024 * per_thread;                 // 4.46 M keys
    const unsigned cap = n_tiles * 512;                              // 512 slots per tile (293 used on average)
    unsigned* cur;
    unsigned long long* keys;
    CHK(hipMalloc(&cur, sizeof(unsigned) * 8 * n_tiles));
    CHK(hipMalloc(&keys, sizeof(unsigned long long) * cap));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipMemset(cur, 0, sizeof(unsigned) * 8 * n_tiles));
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            if (mode == 0) k<0><<<grid, 1024>>>(cur, keys, n_tiles, per_thread, cap, 0);
            else if (mode == 1) k<1><<<grid, 1024>>>(cur, keys, n_tiles, per_thread, cap, 0);
            else k<2><<<grid, 1024, n_tiles * 4>>>(cur, keys, n_tiles, per_thread, cap, 2);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-58s %8.1f us for %.2f M keys\n",
               mode == 0 ? "global cursor per tile (atomicAdd with return)" : mode == 1 ? "global cursor per (XCD, tile)" : "LDS cursors, per-workgroup segments (today's pattern)",
               best * 1e3, total / 1e6);
    }
    return 0;
}
