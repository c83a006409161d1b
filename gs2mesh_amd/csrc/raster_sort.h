// raster_sort.h -- tile offsets and per-tile depth sort.
//
// Replaces cub::DeviceScan::InclusiveSum + the 4-byte D2H sync + cub::DeviceRadixSort::SortPairs
// on 64-bit keys + identifyTileRanges (rasterizer_impl.cu:277-314).  After the counting sort by
// tile (raster_project.h) each tile's instances are contiguous; a workgroup sorts them by the
// unique 64-bit key depth_bits<<32 | id with a bitonic network in LDS.  Order = ascending depth,
// ties by ascending Gaussian id = the reference's stable radix-sort order.
#pragma once
#include "raster_common.h"

// hist[v][wg][t]: per-workgroup counts -> exclusive prefix over workgroups (in place);
// tile_count[v][t] = column total.  A 256-thread workgroup owns 64 consecutive tiles (lane) x 4
// segments of the workgroup axis (wave): pass 1 sums each segment with back-to-back independent
// loads (coalesced: consecutive lanes = consecutive tiles), the 4 segment sums are exchanged through
// LDS, pass 2 re-reads the segment (L2-resident) and writes the running prefix.
GS2M_KERNEL void __launch_bounds__(256)
k_hist_colscan(unsigned* __restrict__ hist, int n_wg, int tiles, unsigned* __restrict__ tile_count) {
    __shared__ unsigned seg_sum[4][64];
    const int lane = (int)(threadIdx.x & 63u), seg = (int)(threadIdx.x >> 6);
    const int t = (int)blockIdx.x * 64 + lane;
    const int v = (int)blockIdx.y;
    const int per = (n_wg + 3) / 4;
    const int w0 = seg * per;
    const int w1 = w0 + per < n_wg ? w0 + per : n_wg;
    unsigned* col = hist + (size_t)v * n_wg * tiles + t;
    unsigned s = 0;
    if (t < tiles) {
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            unsigned x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = col[(size_t)(w + u) * tiles];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        for (; w < w1; ++w) s += col[(size_t)w * tiles];
    }
    seg_sum[seg][lane] = s;
    __syncthreads();
    if (t < tiles) {
        unsigned run = 0;
        for (int k = 0; k < seg; ++k) run += seg_sum[k][lane];
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            unsigned x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = col[(size_t)(w + u) * tiles];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                col[(size_t)(w + u) * tiles] = run;
                run += x[u];
            }
        }
        for (; w < w1; ++w) {
            const unsigned x = col[(size_t)w * tiles];
            col[(size_t)w * tiles] = run;
            run += x;
        }
        if (seg == 3) tile_count[(size_t)v * tiles + t] = run;
    }
}

// Exclusive scan of tile_count over tiles: one 1024-thread workgroup per view.
// tile_start[v][0..tiles]; status[v] = {N, N > cap}.
GS2M_KERNEL void __launch_bounds__(1024)
k_tile_scan(const unsigned* __restrict__ tile_count, unsigned* __restrict__ tile_start, int tiles,
            ViewStatus* __restrict__ status, unsigned cap) {
    __shared__ unsigned part[1024];
    const int tid = (int)threadIdx.x;
    const int v = (int)blockIdx.x;
    const unsigned* cnt = tile_count + (size_t)v * tiles;
    unsigned* start = tile_start + (size_t)v * (tiles + 1);
    const int per = (tiles + 1023) / 1024;
    const int lo = tid * per;
    const int hi = lo + per < tiles ? lo + per : tiles;
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned add = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    unsigned run = part[tid] - s;  // exclusive
    for (int i = lo; i < hi; ++i) {
        start[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) {
        const unsigned total = part[1023];
        start[tiles] = total;
        status[v].num_rendered = total;
        status[v].overflow = total > cap ? 1u : 0u;
    }
}

// Bitonic network over s[0..NP2) in LDS, 256 threads.  Thread t owns compare-exchange pairs
// p = t, t+256, ...; pair p of a stride-j stage is (i, i|j) with i = p with a 0 inserted at bit log2(j),
// so every thread is busy in every stage.  A wave's 64 consecutive pairs live in one aligned block
// of 128 elements whenever j <= 64: those stages (the first 28 of any sort, and the last 7 of every
// merge phase) only need wave-level ordering; workgroup barriers remain for the j >= 128 stages only
// (10 instead of 66 for 2048 keys).  NP2 is a template parameter: the pair loop unrolls and the LDS
// reads of a stage are issued back to back.
template <int NP2>
GS2M_DEVICE void bitonic_lds(unsigned long long* s, int tid) {
    constexpr int PAIRS = (NP2 / 2 + 255) / 256;
    bool prev_block_level = true;  // the load that filled s[] was a workgroup-level step
#pragma unroll 1
    for (int k = 2; k <= NP2; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool block_level = j >= 128;
            if (block_level || prev_block_level) __syncthreads();
            else gs2m_wave_sync();
            prev_block_level = block_level;
            unsigned long long a[PAIRS], b[PAIRS];
            int ia[PAIRS], ib[PAIRS];
#pragma unroll
            for (int m = 0; m < PAIRS; ++m) {
                const int p = tid + 256 * m;
                ia[m] = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                ib[m] = ia[m] | j;
                if (p < NP2 / 2) {
                    a[m] = s[ia[m]];
                    b[m] = s[ib[m]];
                }
            }
#pragma unroll
            for (int m = 0; m < PAIRS; ++m) {
                const int p = tid + 256 * m;
                if (p < NP2 / 2) {
                    const bool up = (ia[m] & k) == 0;
                    if ((a[m] > b[m]) == up) {
                        s[ia[m]] = b[m];
                        s[ib[m]] = a[m];
                    }
                }
            }
        }
    }
    __syncthreads();
}
GS2M_DEVICE void bitonic_lds_dispatch(unsigned long long* s, int np2, int tid) {
    switch (np2) {
        case 2: bitonic_lds<2>(s, tid); break;
        case 4: bitonic_lds<4>(s, tid); break;
        case 8: bitonic_lds<8>(s, tid); break;
        case 16: bitonic_lds<16>(s, tid); break;
        case 32: bitonic_lds<32>(s, tid); break;
        case 64: bitonic_lds<64>(s, tid); break;
        case 128: bitonic_lds<128>(s, tid); break;
        case 256: bitonic_lds<256>(s, tid); break;
        case 512: bitonic_lds<512>(s, tid); break;
        case 1024: bitonic_lds<1024>(s, tid); break;
        case 2048: bitonic_lds<2048>(s, tid); break;
        default: bitonic_lds<4096>(s, tid); break;
    }
}

// Small tiles (2 <= n <= GS2M_SORT_WAVE): ONE WAVE per (tile, view), 64-thread workgroups, 4 KiB of
// LDS -> up to 32 resident waves per CU, and no workgroup barrier anywhere: the whole array belongs
// to the wave.  NP2 is a template parameter so the pair loop unrolls and the LDS accesses of one
// stage are issued back to back.
#define GS2M_SORT_WAVE 512
template <int NP2>
GS2M_DEVICE void bitonic_wave(unsigned long long* s, int lane) {
#pragma unroll 1
    for (int k = 2; k <= NP2; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            gs2m_wave_sync();
            constexpr int PAIRS = (NP2 / 2 + 63) / 64;
            unsigned long long a[PAIRS], b[PAIRS];
            int ia[PAIRS], ib[PAIRS];
#pragma unroll
            for (int m = 0; m < PAIRS; ++m) {
                const int p = lane + 64 * m;
                ia[m] = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                ib[m] = ia[m] | j;
                if (p < NP2 / 2) {
                    a[m] = s[ia[m]];
                    b[m] = s[ib[m]];
                }
            }
#pragma unroll
            for (int m = 0; m < PAIRS; ++m) {
                const int p = lane + 64 * m;
                if (p < NP2 / 2) {
                    const bool up = (ia[m] & k) == 0;
                    if ((a[m] > b[m]) == up) {
                        s[ia[m]] = b[m];
                        s[ib[m]] = a[m];
                    }
                }
            }
        }
    }
    gs2m_wave_sync();
}

GS2M_KERNEL void __launch_bounds__(64)
k_sort_tiles_small(unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start, int tiles,
                   unsigned cap) {
    __shared__ unsigned long long s[GS2M_SORT_WAVE];
    const int lane = (int)threadIdx.x;
    const int t = (int)blockIdx.x, v = (int)blockIdx.y;
    unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
    unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
    if (b > cap) b = cap;
    if (e > cap) e = cap;
    const int n = (int)(e - b);
    if (n <= 1 || n > GS2M_SORT_WAVE) return;  // larger tiles: k_sort_tiles
    unsigned long long* kv = keys + (size_t)v * cap + b;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = lane; i < np2; i += 64) s[i] = i < n ? kv[i] : ~0ull;
    switch (np2) {
        case 2: bitonic_wave<2>(s, lane); break;
        case 4: bitonic_wave<4>(s, lane); break;
        case 8: bitonic_wave<8>(s, lane); break;
        case 16: bitonic_wave<16>(s, lane); break;
        case 32: bitonic_wave<32>(s, lane); break;
        case 64: bitonic_wave<64>(s, lane); break;
        case 128: bitonic_wave<128>(s, lane); break;
        case 256: bitonic_wave<256>(s, lane); break;
        default: bitonic_wave<512>(s, lane); break;
    }
    for (int i = lane; i < n; i += 64) kv[i] = s[i];
}

// One workgroup per (tile, view).  n <= GS2M_SORT_LDS: bitonic in LDS.  Larger tiles: LDS-sorted
// runs of GS2M_SORT_LDS, then rank-based merge passes through HBM (keys <-> tmp) by the same
// workgroup (keys are unique, so rank = index in own run + lower_bound in the sibling run).
GS2M_KERNEL void __launch_bounds__(256)
k_sort_tiles(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ tmp,
             const unsigned* __restrict__ tile_start, int tiles, unsigned cap) {
    __shared__ unsigned long long s[GS2M_SORT_LDS];
    const int tid = (int)threadIdx.x;
    const int t = (int)blockIdx.x, v = (int)blockIdx.y;
    unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
    unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
    if (b > cap) b = cap;
    if (e > cap) e = cap;
    const int n = (int)(e - b);
    if (n <= GS2M_SORT_WAVE) return;  // uniform across the workgroup; small tiles: k_sort_tiles_small
    unsigned long long* kv = keys + (size_t)v * cap + b;
    unsigned long long* tv = tmp + (size_t)v * cap + b;
    const int nruns = (n + GS2M_SORT_LDS - 1) / GS2M_SORT_LDS;
    for (int run = 0; run < nruns; ++run) {
        const int r0 = run * GS2M_SORT_LDS;
        const int rn = n - r0 < GS2M_SORT_LDS ? n - r0 : GS2M_SORT_LDS;
        int np2 = 2;
        while (np2 < rn) np2 <<= 1;
        for (int i = tid; i < np2; i += 256) s[i] = i < rn ? kv[r0 + i] : ~0ull;
        bitonic_lds_dispatch(s, np2, tid);
        for (int i = tid; i < rn; i += 256) kv[r0 + i] = s[i];
        __syncthreads();
    }
    if (nruns == 1) return;
    unsigned long long* src = kv;
    unsigned long long* dst = tv;
    for (int w = GS2M_SORT_LDS; w < n; w <<= 1) {
        for (int i = tid; i < n; i += 256) {
            const int blk = i / (2 * w);
            const int a0 = blk * 2 * w;
            const int a1 = a0 + w < n ? a0 + w : n;            // A = [a0,a1)
            const int b1 = a0 + 2 * w < n ? a0 + 2 * w : n;    // B = [a1,b1)
            const unsigned long long key = src[i];
            int lo, hi, base;
            if (i < a1) {  // element of A: count of B elements smaller than key
                lo = a1;
                hi = b1;
                base = i - a0;
            } else {
                lo = a0;
                hi = a1;
                base = i - a1;
            }
            const int lo0 = lo;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (src[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            dst[a0 + base + (lo - lo0)] = key;
        }
        __syncthreads();
        unsigned long long* x = src;
        src = dst;
        dst = x;
    }
    if (src != kv) {
        for (int i = tid; i < n; i += 256) kv[i] = src[i];
    }
}
