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
// tile_start[v][0..tiles]; status[v] = {N, N > cap}; sticky = {max N, any N > cap} since the last status query.
GS2M_KERNEL void __launch_bounds__(1024)
k_tile_scan(const unsigned* __restrict__ tile_count, unsigned* __restrict__ tile_start, int tiles,
            ViewStatus* __restrict__ status, ViewStatus* __restrict__ sticky, unsigned cap) {
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
        // sticky word of the handle: survives later calls until gs2m_raster_status consumes it
        if (total > cap) atomicOr(&sticky->overflow, 1u);
        atomicMax(&sticky->num_rendered, total);
    }
}

// ---- register-blocked bitonic sort -----------------------------------------------------------
// Thread t holds E consecutive elements v[0..E) = elements t*E .. t*E+E-1 of the network.  A stage with
// partner distance j is
//   j <  E          : a compare-exchange between two registers of the same lane (no data movement);
//   E <= j < 64 E   : the partner element sits in lane ^ (j/E) at the same register slot: one cross-lane
//                     exchange per register (no LDS banks involved);
//   j >= 64 E       : (workgroup sort only) the partner is in another wave: through LDS, transposed layout
//                     s[r * THREADS + t] (conflict-free), 3 such stages for any size up to 4096.
// For 256 keys on one wave (E = 4) 15 of the 36 stages are register-only and none touches LDS memory
// (the all-LDS network it replaces: 36 stages x 4 ds_read_b64 + up to 4 ds_write_b64 per lane, 39 % of the
// LDS cycles lost to bank conflicts -- PMC, profiles/).  Direction of element i in phase k: ascending iff
// (i & k) == 0; the element keeps the minimum iff (it is the lower index of its pair) == ascending.
GS2M_DEVICE void cmpx_keep(unsigned long long& a, unsigned long long p, bool keep_min) {
    const bool p_less = p < a;
    a = (p_less == keep_min) ? p : a;
}

// stage (K, J) and, recursively, the smaller strides of phase K on the elements of one wave (compile-time
// recursion: every register index is a constant); gtid = index of the thread in the sort
template <int E, int K, int J>
GS2M_DEVICE void bitonic_stages_wave(unsigned long long (&v)[E], int gtid, int lane) {
    if constexpr (J >= 1) {
        if constexpr (J >= E) {
            constexpr int d = J / E;
            const bool up = ((gtid * E) & K) == 0;
            const bool keep_min = ((lane & d) == 0) == up;
#pragma unroll
            for (int r = 0; r < E; ++r) cmpx_keep(v[r], gs2m_shfl_xor(v[r], d), keep_min);
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if ((r & J) == 0) {
                    const bool up = K < E ? ((r & K) == 0) : (((gtid * E) & K) == 0);
                    const unsigned long long a = v[r], b = v[r | J];
                    const bool sw = (b < a) == up;
                    v[r] = sw ? b : a;
                    v[r | J] = sw ? a : b;
                }
            }
        }
        bitonic_stages_wave<E, K, J / 2>(v, gtid, lane);
    }
}
// phase K restricted to the strides that stay inside a wave (<= 32 E)
template <int E, int K>
GS2M_DEVICE void bitonic_phase_wave(unsigned long long (&v)[E], int gtid, int lane) {
    bitonic_stages_wave<E, K, (K / 2 < 32 * E ? K / 2 : 32 * E)>(v, gtid, lane);
}

// whole sort of 64*E keys held by one wave
template <int E>
GS2M_DEVICE void bitonic_wave_regs(unsigned long long (&v)[E], int lane) {
    bitonic_phase_wave<E, 2>(v, lane, lane);
    if constexpr (E * 64 >= 4) bitonic_phase_wave<E, 4>(v, lane, lane);
    if constexpr (E * 64 >= 8) bitonic_phase_wave<E, 8>(v, lane, lane);
    if constexpr (E * 64 >= 16) bitonic_phase_wave<E, 16>(v, lane, lane);
    if constexpr (E * 64 >= 32) bitonic_phase_wave<E, 32>(v, lane, lane);
    if constexpr (E * 64 >= 64) bitonic_phase_wave<E, 64>(v, lane, lane);
    if constexpr (E * 64 >= 128) bitonic_phase_wave<E, 128>(v, lane, lane);
    if constexpr (E * 64 >= 256) bitonic_phase_wave<E, 256>(v, lane, lane);
    if constexpr (E * 64 >= 512) bitonic_phase_wave<E, 512>(v, lane, lane);
    if constexpr (E * 64 >= 1024) bitonic_phase_wave<E, 1024>(v, lane, lane);
}

// one cross-wave stage (stride j >= 64 E) of phase K through LDS; 256 threads, s holds 256 * E keys
template <int E>
GS2M_DEVICE void bitonic_stage_block(unsigned long long (&v)[E], unsigned long long* s, int tid, int K, int j) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) s[r * 256 + tid] = v[r];
    __syncthreads();
    const int dt = j / E;
    const bool up = ((tid * E) & K) == 0;
    const bool keep_min = ((tid & dt) == 0) == up;
#pragma unroll
    for (int r = 0; r < E; ++r) cmpx_keep(v[r], s[r * 256 + (tid ^ dt)], keep_min);
}

// whole sort of 256*E keys held by a 256-thread workgroup (E = 4, 8, 16 -> 1024, 2048, 4096 keys)
template <int E>
GS2M_DEVICE void bitonic_block_regs(unsigned long long (&v)[E], unsigned long long* s, int tid) {
    const int lane = tid & 63;
    bitonic_phase_wave<E, 2>(v, tid, lane);
    bitonic_phase_wave<E, 4>(v, tid, lane);
    bitonic_phase_wave<E, 8>(v, tid, lane);
    bitonic_phase_wave<E, 16>(v, tid, lane);
    bitonic_phase_wave<E, 32>(v, tid, lane);
    bitonic_phase_wave<E, 64>(v, tid, lane);
    bitonic_phase_wave<E, 128>(v, tid, lane);
    bitonic_phase_wave<E, 256>(v, tid, lane);
    if constexpr (64 * E >= 512) bitonic_phase_wave<E, 512>(v, tid, lane);
    if constexpr (64 * E >= 1024) bitonic_phase_wave<E, 1024>(v, tid, lane);
    // phases 128 E and 256 E: strides 64 E (and 128 E) cross waves
    bitonic_stage_block<E>(v, s, tid, 128 * E, 64 * E);
    bitonic_phase_wave<E, 128 * E>(v, tid, lane);
    bitonic_stage_block<E>(v, s, tid, 256 * E, 128 * E);
    bitonic_stage_block<E>(v, s, tid, 256 * E, 64 * E);
    bitonic_phase_wave<E, 256 * E>(v, tid, lane);
}

// Small tiles (2 <= n <= GS2M_SORT_WAVE): ONE WAVE per (tile, view), 64-thread workgroups, no LDS and no
// barrier: E = 1, 2, 4, 8 keys per lane for n <= 64, 128, 256, 512.
#define GS2M_SORT_WAVE 512
template <int E>
GS2M_DEVICE void sort_wave_regs(unsigned long long* __restrict__ kv, int n, int lane) {
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = lane * E + r;
        v[r] = i < n ? kv[i] : ~0ull;
    }
    bitonic_wave_regs<E>(v, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = lane * E + r;
        if (i < n) kv[i] = v[r];
    }
}

GS2M_KERNEL void __launch_bounds__(64)
k_sort_tiles_small(unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start, int tiles,
                   unsigned cap) {
    const int lane = (int)threadIdx.x;
    const int t = (int)blockIdx.x, v = (int)blockIdx.y;
    unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
    unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
    if (b > cap) b = cap;
    if (e > cap) e = cap;
    const int n = (int)(e - b);
    if (n <= 1 || n > GS2M_SORT_WAVE) return;  // larger tiles: k_sort_tiles
    unsigned long long* kv = keys + (size_t)v * cap + b;
    if (n <= 64) sort_wave_regs<1>(kv, n, lane);
    else if (n <= 128) sort_wave_regs<2>(kv, n, lane);
    else if (n <= 256) sort_wave_regs<4>(kv, n, lane);
    else sort_wave_regs<8>(kv, n, lane);
}

template <int E>
GS2M_DEVICE void sort_block_regs(unsigned long long* __restrict__ kv, int n, unsigned long long* s, int tid) {
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = tid * E + r;
        v[r] = i < n ? kv[i] : ~0ull;
    }
    bitonic_block_regs<E>(v, s, tid);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = tid * E + r;
        if (i < n) kv[i] = v[r];
    }
}

// One workgroup per (tile, view).  n <= GS2M_SORT_LDS: register-blocked bitonic (3 stages through LDS).  Larger tiles: LDS-sorted
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
        if (rn <= 1024) sort_block_regs<4>(kv + r0, rn, s, tid);
        else if (rn <= 2048) sort_block_regs<8>(kv + r0, rn, s, tid);
        else sort_block_regs<16>(kv + r0, rn, s, tid);
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
