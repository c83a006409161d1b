// raster_sort.h -- tile offsets and per-tile depth sort.
//
// Replaces cub::DeviceScan::InclusiveSum + the 4-byte D2H sync + cub::DeviceRadixSort::SortPairs
// on 64-bit keys + identifyTileRanges (rasterizer_impl.cu:277-314).  After the counting sort by
// tile (raster_project.h) each tile's instances are contiguous; a workgroup sorts them by the
// unique 64-bit key depth_bits<<32 | id with a bitonic network in LDS.  Order = ascending depth,
// ties by ascending Gaussian id = the reference's stable radix-sort order.
#pragma once
#include <type_traits>
#include "raster_common.h"

#define GS2M_SORT_WAVE 512           // lists up to here: one wave, register bitonic (k_sort_tiles_small)
#define GS2M_SORT_CLASSES 3          // work lists of the larger ones: (512, 4096], (4096, 8192], > 8192 instances
#define GS2M_SORT_BUCKET_CAP 8192

// hist[v][wg][t]: per-workgroup counts -> exclusive prefix over workgroups (in place);
// tile_count[v][t] = column total.  A 1024-thread workgroup owns 64 consecutive tiles (lane) x GS2M_COLSCAN_SEGS
// segments of the workgroup axis (wave): pass 1 sums each segment with back-to-back independent loads (coalesced:
// consecutive lanes = consecutive tiles), the segment sums are exchanged through LDS, pass 2 re-reads the segment
// (L2-resident) and writes the running prefix.  16 segments: a thread's dependent chain is ~15 rows for the usual
// ~256-row matrix (4 segments of 256-thread workgroups left the kernel latency-bound at 120 workgroups).
#define GS2M_COLSCAN_SEGS 16
// Exclusive scan of tile_count over tiles: one 1024-thread workgroup per view.
// tile_start[v][0..tiles]; status[v] = {N, N > cap}; sticky = {max N, any N > cap} since the last status query.
// (Round 4 tried to run this in the LAST workgroup of k_hist_colscan to finish a view -- ticket counter + __threadfence -- to
// save the launch: the agent-scope release every one of the ~470 column-scan workgroups then executes writes back its XCD's
// whole L2 (the histogram rows the counting kernel just wrote are dirty there): k_hist_colscan 21 -> 129 us per launch, step
// 0.304 -> 0.392 ms.  Reverted; profiles/r4_experiments.txt.)
GS2M_KERNEL void __launch_bounds__(1024)
k_tile_scan(const unsigned* __restrict__ tile_count, unsigned* __restrict__ tile_start, int tiles, int gx,
            ViewStatus* __restrict__ status_all, ViewStatus* __restrict__ sticky, unsigned cap, unsigned* __restrict__ sort_lists) {
    const int v = (int)blockIdx.x, n_views = (int)gridDim.x;
    ViewStatus* status = status_all + v;
    __shared__ unsigned part[16];    // wave totals of the scan
    __shared__ unsigned n_class[GS2M_SORT_CLASSES];
    __shared__ unsigned w_pos[64];   // compositing schedule: 64 weight buckets of the list chunks
    __shared__ unsigned w_max;       // heaviest chunk of the view
    const int tid = (int)threadIdx.x;
    const unsigned* cnt = tile_count + (size_t)v * tiles;
    unsigned* start = tile_start + (size_t)v * (tiles + 1);
    const int per = (tiles + 1023) / 1024;
    const int lo = tid * per;
    const int hi = lo + per < tiles ? lo + per : tiles;
    // work lists of the per-tile sort: tiles with more than GS2M_SORT_WAVE instances, by size class (k_sort_tiles_*
    // walk them with a small grid instead of launching a workgroup per tile that mostly has nothing to do)
    unsigned* lists = sort_lists + (size_t)v * GS2M_SORT_CLASSES * (tiles + 1);
    if (tid < GS2M_SORT_CLASSES) n_class[tid] = 0u;
    if (tid < 64) w_pos[tid] = 0u;
    if (tid == 64) w_max = 0u;
    __syncthreads();
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) {
        const unsigned c = cnt[i];
        s += c;
        if (c > GS2M_SORT_WAVE) {
            const int cls = c <= 4096u ? 0 : (c <= GS2M_SORT_BUCKET_CAP ? 1 : 2);
            lists[cls * (tiles + 1) + 1 + atomicAdd(&n_class[cls], 1u)] = (unsigned)i;
        }
    }
    // inclusive scan over the 1024 partials: a shuffle scan inside every wave, then the 16 wave totals
    const int lane = tid & 63, wave = tid >> 6;
    unsigned incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = gs2m_shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned before = 0u, total_all = 0u;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned x = part[w];
        before += w < wave ? x : 0u;
        total_all += x;
    }
    unsigned run = before + incl - s;  // exclusive
    for (int i = lo; i < hi; ++i) {
        start[i] = run;
        run += cnt[i];
    }
    // Compositing schedule.  A compositing wave lives ~100 us and a launch has only ~2 generations of them: dispatched in
    // image order the kernel ends on whatever long lists happen to come last, and an XCD that owns a dense part of the image
    // finishes long after the others.  The lists are grouped into CHUNKS of GS2M_SCHED_CW x GS2M_SCHED_CH neighbours (they
    // share Gaussians: one L2), the chunks are ranked by descending weight (64 buckets up to the heaviest) and dealt to the
    // XCDs round-robin: rank p goes to XCD p % 8 as its (p / 8)-th chunk (block b runs on XCD b % 8 and blocks are
    // dispatched in order) -- every XCD gets the same share of heavy chunks and ends on its lightest ones.
    unsigned* order = sort_lists + (size_t)n_views * GS2M_SORT_CLASSES * (tiles + 1) + (size_t)v * tiles;
    const int lrows = tiles / gx;                                               // rows of lists
    const int cpr = (gx + GS2M_SCHED_CW - 1) / GS2M_SCHED_CW;                   // chunks per chunk row
    const int nch = cpr * ((lrows + GS2M_SCHED_CH - 1) / GS2M_SCHED_CH);       // chunks
    auto chunk_weight = [&](int c) -> unsigned {
        const int crow = c / cpr, x0 = (c - crow * cpr) * GS2M_SCHED_CW, y0 = crow * GS2M_SCHED_CH;
        unsigned w = 0u;
        for (int k = 0; k < GS2M_SCHED_CHUNK; ++k) {
            const int lx = x0 + k % GS2M_SCHED_CW, ly = y0 + k / GS2M_SCHED_CW;
            if (lx < gx && ly < lrows) w += cnt[ly * gx + lx];
        }
        return w;
    };
    unsigned cw[2] = {0u, 0u};   // a thread owns chunks tid and tid + 1024 (nch <= 2048: larger grids fall back to rank = id)
    const bool sched = nch <= 2048;
    if (sched) {
        for (int k = 0; k < 2; ++k)
            if (tid + 1024 * k < nch) {
                cw[k] = chunk_weight(tid + 1024 * k);
                atomicMax(&w_max, cw[k]);
            }
    }
    __syncthreads();
    int wsh = 0;                                                  // bucket width 2^wsh: the heaviest chunk falls into the top bucket
    while ((w_max >> wsh) > 63u) ++wsh;
    if (sched)
        for (int k = 0; k < 2; ++k)
            if (tid + 1024 * k < nch) atomicAdd(&w_pos[63u - (cw[k] >> wsh)], 1u);
    __syncthreads();
    if (tid < 64) {   // exclusive prefix over the 64 buckets: one wave, six shuffle steps
        const unsigned c = w_pos[tid];
        unsigned incl2 = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = gs2m_shfl_up(incl2, d);
            if (tid >= d) incl2 += y;
        }
        w_pos[tid] = incl2 - c;
    }
    __syncthreads();
    for (int k = 0; k < 2; ++k)
        if (tid + 1024 * k < nch) order[sched ? atomicAdd(&w_pos[63u - (cw[k] >> wsh)], 1u) : (unsigned)(tid + 1024 * k)] = (unsigned)(tid + 1024 * k);
    if (!sched)
        for (int c = tid + 2048; c < nch; c += 1024) order[c] = (unsigned)c;
    if (tid < GS2M_SORT_CLASSES) {   // complete: every thread passed the scan's barriers
        lists[tid * (tiles + 1)] = n_class[tid];
        status->n_class[tid] = n_class[tid];   // read back with the status: sizes the next call's class grids (a hint)
    }
    if (tid == 1023) {
        const unsigned total = total_all;
        start[tiles] = total;
        status->num_rendered = total;
        status->overflow = total > cap ? 1u : 0u;
        // sticky word of the handle: survives later calls until gs2m_raster_status consumes it
        if (total > cap) atomicOr(&sticky->overflow, 1u);
        atomicMax(&sticky->num_rendered, total);
    }
}

GS2M_KERNEL void __launch_bounds__(64 * GS2M_COLSCAN_SEGS)
k_hist_colscan(unsigned* __restrict__ hist, int n_wg, int tiles, unsigned* __restrict__ tile_count) {
    __shared__ unsigned seg_sum[GS2M_COLSCAN_SEGS][64];
    const int lane = (int)(threadIdx.x & 63u), seg = (int)(threadIdx.x >> 6);
    const int t = (int)blockIdx.x * 64 + lane;
    const int v = (int)blockIdx.y;
    const int per = (n_wg + GS2M_COLSCAN_SEGS - 1) / GS2M_COLSCAN_SEGS;
    const int w0 = seg * per < n_wg ? seg * per : n_wg;
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
        unsigned run = 0, total = 0;
#pragma unroll
        for (int k = 0; k < GS2M_COLSCAN_SEGS; ++k) {
            const unsigned x = seg_sum[k][lane];
            run += k < seg ? x : 0u;
            total += x;
        }
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
        if (seg == 0) tile_count[(size_t)v * tiles + t] = total;
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

// Bucket + rank sort of one list by ONE WAVE (64 < n <= 64 E <= 512 keys; round 3): the same idea as sort_list_bucket below,
// without a workgroup barrier.  The depths of a tile's instances spread over its frustum, so a monotone map of the depth onto
// 64 E equal-width buckets leaves 0-3 keys in almost every bucket: one LDS counting pass gives every key its bucket and its
// arrival slot, a wave scan turns the counts into starts, and the exact position inside the bucket is the number of smaller
// keys there (keys are unique, so the ranks are a permutation: the same total order as any comparison sort -- ascending
// depth, ties by id).  ~60 instructions per key-lane instead of the ~45 compare-exchange stages x E registers of the bitonic
// network (C2: k_sort_tiles_small 31.6 -> see profiles/r3_experiments.txt).  Returns false (nothing written) when a bucket holds
// more than GS2M_WAVE_BUCKET_MAX keys (clustered or equal depths): the caller then runs the bitonic network.
#define GS2M_WAVE_BUCKET_MAX 12
// Round 6: depths that CLUSTER (a trained splat's Gaussians sit on surfaces: a tile's frustum crosses a shell twice, and a third
// of its keys fall into 2 % of its depth range) overflowed the equal-width buckets -- 43 % of the wave-sorted lists of the
// C2-sized `synthetic.trained_like` scene (72 % of their keys) fell back to the bitonic network (sort 12.8 -> 35.7 us per pair
// against `synth_v1`).  A list whose fullest bucket holds more than GS2M_WAVE_BUCKET_REFINE keys is REFINED instead: bucket b with
// c_b keys is split into c_b equal-width sub-buckets (fine bucket = start_b + floor(frac_b * c_b), frac_b = position inside b:
// monotone in the depth, n fine buckets in all), a second counting pass places the keys, and the rank loop runs over the fine
// buckets (measured on that scene by CPU replay: no list overflows any more, fullest fine bucket 3 .. 8 keys).  Same result:
// the ranks are still a permutation in key order.
#define GS2M_WAVE_BUCKET_REFINE 6
template <int E>
GS2M_DEVICE bool sort_wave_bucket(unsigned long long* __restrict__ kv, const int n, const int lane, unsigned long long* s_key,
                                  unsigned* s_cnt, unsigned* s_fine) {
    static_assert(E >= 2 && (E & 1) == 0, "two 16-bit counters per word, E / 2 words per lane");
    constexpr int NB = 64 * E;      // buckets (>= n)
    unsigned long long k[E];
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = r * 64 + lane;
        k[r] = i < n ? kv[i] : ~0ull;
        if (i < n) {
            const unsigned d = (unsigned)(k[r] >> 32);
            dmin = d < dmin ? d : dmin;
            dmax = d > dmax ? d : dmax;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const unsigned a = gs2m_shfl_xor(dmin, m), c = gs2m_shfl_xor(dmax, m);
        dmin = a < dmin ? a : dmin;
        dmax = c > dmax ? c : dmax;
    }
    gs2m_wave_sync();   // the previous list's readers of s_cnt / s_key are done (a wave may sort several lists)
#pragma unroll
    for (int w = 0; w < E / 2; ++w) s_cnt[w * 64 + lane] = 0u;
    gs2m_wave_sync();
    // monotone map depth -> bucket (float conversion, multiplication by a positive constant and truncation are non-decreasing)
    const float scale = (float)NB / ((float)(dmax - dmin) + 1.0f) * 0.99999f;
    auto bucket_of = [&](unsigned long long key) -> unsigned {
        const unsigned b = (unsigned)((float)((unsigned)(key >> 32) - dmin) * scale);
        return b < (unsigned)NB ? b : (unsigned)NB - 1u;
    };
    // counting pass over `cnt` (two 16-bit counters per word): the returned old value is the key's arrival slot in its bucket;
    // then lane l owns buckets [l E, (l + 1) E) = words [l E / 2, (l + 1) E / 2): exclusive scan -> starts in place, the
    // fullest bucket of the list as the return value (wave-uniform)
    unsigned short slot[E];
    unsigned bk[E];     // bucket of k[r] in the map in force (coarse, then fine)
    auto count_and_scan = [&](unsigned* cnt) __attribute__((always_inline)) -> unsigned {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            slot[r] = 0;
            if (r * 64 + lane < n) {
                const unsigned b = bk[r];
                const unsigned old = atomicAdd(&cnt[b >> 1], 1u << ((b & 1u) << 4));
                slot[r] = (unsigned short)((old >> ((b & 1u) << 4)) & 0xffffu);
            }
        }
        gs2m_wave_sync();
        unsigned cw[E / 2];
        unsigned sum = 0u, mx = 0u;
#pragma unroll
        for (int w = 0; w < E / 2; ++w) {
            cw[w] = cnt[lane * (E / 2) + w];
            const unsigned c0 = cw[w] & 0xffffu, c1 = cw[w] >> 16;
            sum += c0 + c1;
            mx = c0 > mx ? c0 : mx;
            mx = c1 > mx ? c1 : mx;
        }
        unsigned incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = gs2m_shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const unsigned o = gs2m_shfl_xor(mx, m);
            mx = o > mx ? o : mx;
        }
        unsigned run = incl - sum;
#pragma unroll
        for (int w = 0; w < E / 2; ++w) {
            const unsigned c0 = cw[w] & 0xffffu, c1 = cw[w] >> 16;
            cnt[lane * (E / 2) + w] = run | ((run + c0) << 16);   // starts (<= 512)
            run += c0 + c1;
        }
        if (lane == 63) cnt[NB / 2] = (unsigned)n;                 // sentinel: start of bucket NB
        gs2m_wave_sync();
        return (unsigned)gs2m_uniform((int)mx);
    };
    auto start_in = [&](const unsigned* cnt, unsigned b) -> unsigned { return (cnt[b >> 1] >> ((b & 1u) << 4)) & 0xffffu; };
#pragma unroll
    for (int r = 0; r < E; ++r) bk[r] = r * 64 + lane < n ? bucket_of(k[r]) : 0u;
    unsigned mx = count_and_scan(s_cnt);
    // fine bucket of a key: its coarse bucket b split into c_b equal parts (reads the coarse starts: s_cnt stays as it is)
    auto fine_of = [&](unsigned long long key) -> unsigned {
        const float x = (float)((unsigned)(key >> 32) - dmin) * scale;
        unsigned b = (unsigned)x;
        b = b < (unsigned)NB ? b : (unsigned)NB - 1u;
        const unsigned st = start_in(s_cnt, b), c = start_in(s_cnt, b + 1u) - st;
        float frac = x - (float)b;                              // exact: both < 2^24, x - floor(x) is representable
        frac = frac < 0.99999f ? frac : 0.99999f;               // (the clamped last bucket)
        const unsigned sub = (unsigned)(frac * (float)c);
        return st + (sub < c ? sub : c - 1u);                   // c >= 1 for the bucket of an existing key
    };
    const bool refined = mx > GS2M_WAVE_BUCKET_REFINE;        // wave-uniform
    const unsigned* starts = s_cnt;
    if (refined) {
#pragma unroll
        for (int w = 0; w < E / 2; ++w) s_fine[w * 64 + lane] = 0u;
#pragma unroll
        for (int r = 0; r < E; ++r) bk[r] = r * 64 + lane < n ? fine_of(k[r]) : 0u;
        gs2m_wave_sync();
        mx = count_and_scan(s_fine);
        starts = s_fine;
    }
    if (mx > GS2M_WAVE_BUCKET_MAX) return false;   // wave-uniform (equal depths in bulk: the bitonic network)
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const unsigned st = start_in(starts, bk[r]);   // unconditional read: E reads in flight (lanes without a key read bucket 0)
        if (r * 64 + lane < n) s_key[st + slot[r]] = k[r];
    }
    gs2m_wave_sync();
    // rank inside the bucket -> final position.  A lane first reads its E keys and the bounds of their buckets, then all E
    // buckets are walked in lock step for `mx` rounds (the largest bucket of the list: wave-uniform, <= GS2M_WAVE_BUCKET_MAX,
    // typically 3-5): E independent LDS reads per round instead of E dependent walks one after the other.
    // All LDS reads are unconditional (every index is inside the arrays; lanes without a key get len = 0): a read under a
    // lane condition compiles to its own branch + wait, which would serialise the E reads again.
    unsigned long long key[E];
    unsigned lo[E], len[E], rank[E];
#pragma unroll
    for (int r = 0; r < E; ++r) key[r] = s_key[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const bool in = r * 64 + lane < n;
        const unsigned b = in ? (refined ? fine_of(key[r]) : bucket_of(key[r])) : 0u;
        const unsigned st = start_in(starts, b), en = start_in(starts, b + 1u);
        lo[r] = st;
        len[r] = in ? en - st : 0u;
        rank[r] = 0u;
    }
    const unsigned rounds = mx;
    for (unsigned q = 0; q < rounds; ++q) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const bool more = q < len[r];
            const unsigned long long other = s_key[more ? lo[r] + q : lo[r]];
            rank[r] += (more && other < key[r]) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < E; ++r)
        if (r * 64 + lane < n) kv[lo[r] + rank[r]] = key[r];
    return true;
}

// LDS-free stand-in for the three size-class kernels, launched INSTEAD of them when the previous call on the handle found
// every class empty (the usual case at 16 x 32 tiles of a 300 k-Gaussian model: no list above 512 instances).  The class
// kernels carry 40-80 KiB of static LDS per workgroup; under the pipelined load (a compositing grid of another stream holds
// 157 of the 160 KiB of every CU) even their EMPTY launches waited 30-40 us each for a CU with room -- three times per view
// on the critical chain of its slot (kernel trace, round 3).  This kernel needs no LDS, so its 256-thread workgroups are
// placed at once.  It is not a hint-dependent shortcut: the hint may be stale (scene or camera change, status copy not yet
// landed), so if the view DOES have larger lists it sorts them, exactly and in bounded time -- runs of 256 keys by rank
// (keys are unique: rank = number of smaller keys of the run = position in the run), then rank-merge passes between `keys`
// and `tmp` as in k_sort_tiles: O(256 n + n log^2 n) instead of the O(n^2) of a rank sort over the whole list (a stale hint
// on a C3-like view used to cost milliseconds).  The lists are L1 / L2 resident and a workgroup only re-reads what its own
// threads wrote, visible after the workgroup barrier.
// Round 4: no launch of its own any more -- the first GS2M_SORT_RANK_BLOCKS workgroups of k_sort_tiles_small run it after their
// own lists (one launch less on the chain of every pass).
#define GS2M_SORT_RANK_BLOCKS 32
GS2M_DEVICE void sort_class_lists_rank(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ tmp,
                                       const unsigned* __restrict__ tile_start, int tiles, unsigned cap,
                                       const unsigned* __restrict__ sort_lists, const int v, const unsigned first, const unsigned stride,
                                       const int tid) {
    for (int cls = 0; cls < GS2M_SORT_CLASSES; ++cls) {
        const unsigned* list = sort_lists + ((size_t)v * GS2M_SORT_CLASSES + cls) * (tiles + 1);
        const unsigned count = list[0];
        for (unsigned li = first; li < count; li += stride) {
            const int t = (int)list[1 + li];
            unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
            unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
            if (b > cap) b = cap;
            if (e > cap) e = cap;
            const int n = (int)(e - b);
            unsigned long long* kv = keys + (size_t)v * cap + b;
            unsigned long long* tv = tmp + (size_t)v * cap + b;
            // runs of 256 keys, each sorted by rank into tmp
            for (int r0 = 0; r0 < n; r0 += 256) {
                const int rn = n - r0 < 256 ? n - r0 : 256;
                if (tid < rn) {
                    const unsigned long long key = kv[r0 + tid];
                    int rank = 0;
                    for (int j = 0; j < rn; ++j) rank += kv[r0 + j] < key ? 1 : 0;
                    tv[r0 + rank] = key;
                }
            }
            __syncthreads();
            unsigned long long* src = tv;
            unsigned long long* dst = kv;
            for (int w = 256; w < n; w <<= 1) {
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
            __syncthreads();
        }
    }
}

// WPB waves per workgroup, one list per wave (no workgroup barrier: a wave leaves as soon as its list is done).  C2 launches
// 7600 lists per pair: as single-wave workgroups their dispatch alone took a third of the kernel.
template <int WPB>
GS2M_KERNEL void __launch_bounds__(64 * WPB)
k_sort_tiles_small(unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start, int tiles,
                   unsigned cap, unsigned long long* __restrict__ tmp, const unsigned* __restrict__ sort_lists, int fold_rank) {
    __shared__ unsigned long long s_key_all[WPB][GS2M_SORT_WAVE];
    __shared__ unsigned s_cnt_all[WPB][GS2M_SORT_WAVE / 2 + 2];
    __shared__ unsigned s_fine_all[WPB][GS2M_SORT_WAVE / 2 + 2];    // counters / starts of the refined (fine) buckets
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    unsigned long long* s_key = s_key_all[wave];
    unsigned* s_cnt = s_cnt_all[wave];
    unsigned* s_fine = s_fine_all[wave];
    const int t = gs2m_uniform((int)blockIdx.x * WPB + wave), v = (int)blockIdx.y;
    if (t < tiles) {
        unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
        unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
        if (b > cap) b = cap;
        if (e > cap) e = cap;
        const int n = (int)(e - b);
        unsigned long long* kv = keys + (size_t)v * cap + b;
        if (n <= 1 || n > GS2M_SORT_WAVE) {
            // nothing to do / a larger list: the size-class kernels, or the folded stand-in below
        } else if (n <= 64) {
            sort_wave_regs<1>(kv, n, lane);
        } else if (n <= 128) {
            if (!sort_wave_bucket<2>(kv, n, lane, s_key, s_cnt, s_fine)) sort_wave_regs<2>(kv, n, lane);
        } else if (n <= 256) {
            if (!sort_wave_bucket<4>(kv, n, lane, s_key, s_cnt, s_fine)) sort_wave_regs<4>(kv, n, lane);
        } else {
            if (!sort_wave_bucket<8>(kv, n, lane, s_key, s_cnt, s_fine)) sort_wave_regs<8>(kv, n, lane);
        }
    }
    // fold_rank: the previous call on the handle found every size class empty, so the class kernels are not launched; the first
    // workgroups sort whatever larger lists there are after all (sort_class_lists_rank: exact, bounded; needs 256 threads)
    if (WPB == 4 && fold_rank && blockIdx.x < GS2M_SORT_RANK_BLOCKS)
        sort_class_lists_rank(keys, tmp, tile_start, tiles, cap, sort_lists, v, blockIdx.x,
                              gridDim.x < GS2M_SORT_RANK_BLOCKS ? gridDim.x : GS2M_SORT_RANK_BLOCKS, (int)threadIdx.x);
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

// ---- bucket + rank sort of one tile list (GS2M_SORT_WAVE < n <= GS2M_SORT_LDS keys) ---------------------------------
// The keys of a tile are depth_bits << 32 | id with depths spread over the tile's frustum, so a monotone map of the
// depth onto ~n equal-width buckets leaves 0-3 keys in almost every bucket: one LDS counting pass places every key in
// its bucket, and the exact position inside the bucket is the number of smaller keys there (keys are unique, so the
// ranks are a permutation and the result is the same total order as any comparison sort: ascending depth, ties by id).
// O(n) LDS traffic instead of the O(n log^2 n) compare-exchanges of the bitonic network; replaces
// cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:303-308) for the mid-size lists that dominate a 2 M-Gaussian
// scene (C3: 7.5 M keys in 4 240 lists, 275 us -> see profiles/r2*).  A list whose depths are so clustered that a
// bucket holds more than GS2M_BUCKET_MAX keys falls back to the bitonic network (same result).
// CAP = keys per list this instantiation takes (> CAP / 2 is left to it, <= CAP / 2 to the smaller one), THREADS = CAP / 16
// or CAP / 8: <4096, .> (40 KiB of LDS: 4 workgroups per CU) for the bulk, <8192, .> (80 KiB: 2 per CU) for the densest
// tiles.  The sort of a list is a chain of short LDS phases between barriers: with CAP / 8 threads a list gets twice the
// waves for the same LDS, i.e. 8 instead of 4 waves per SIMD to cover the LDS latency.
#define GS2M_BUCKET_MAX 48
template <int CAP, int THREADS>
GS2M_DEVICE void sort_list_bucket(unsigned long long* __restrict__ kv, const int n, unsigned long long* s_key, unsigned* s_cnt,
                                  unsigned* s_red, const int tid) {
    constexpr int NW = THREADS / 64, E = CAP / THREADS;
    static_assert(E == 16 || E == 8, "8 or 16 keys per thread");
    const int lane = tid & 63, wave = tid >> 6;
    // Round 6: a list of n keys only fills the first ER = ceil(n / THREADS) of a thread's E key slots (a 600-key list on the 512-thread
    // class: 2 of 8; C3's 4 900-key lists on the 1024-thread class: 5 of 8).  The slots beyond are skipped by wave-uniform tests
    // instead of being executed under an empty execution mask (the sort is ~775 vector instructions per wave and list, PMC).
    // (the <8192 x 8> class keeps all E slots: its lists fill 5+ of 8, and the tests + the per-count copies of the rank loop cost it
    // registers -- C3 55.1 -> 56.0, C4 65.9 -> 68.4 us with the skip everywhere; trained-like 26.8 -> 24.6, C5 33.2 -> 29.4)
    constexpr bool SKIP_DEAD = CAP <= 4096;
    const int ER = SKIP_DEAD ? (n + THREADS - 1) / THREADS : E;
    // ---- the keys (16 per thread, registers) and the depth range of the list
    unsigned long long k[E];
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = r * THREADS + tid;
        k[r] = ~0ull;
        if (r < ER) k[r] = i < n ? kv[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r < ER && r * THREADS + tid < n) {
            const unsigned d = (unsigned)(k[r] >> 32);
            dmin = d < dmin ? d : dmin;
            dmax = d > dmax ? d : dmax;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const unsigned a = gs2m_shfl_xor(dmin, m), c = gs2m_shfl_xor(dmax, m);
        dmin = a < dmin ? a : dmin;
        dmax = c > dmax ? c : dmax;
    }
    if (lane == 0) {
        s_red[wave] = dmin;
        s_red[NW + wave] = dmax;
    }
    // buckets: a power of two >= n when that fits, at most CAP / 2 (>= 4 per thread for the scan).  The cap keeps the
    // counters at CAP / 4 words: 36.1 / 72.2 KiB per workgroup = 4 / 2 workgroups per CU (with CAP buckets the arrays were
    // 144 bytes over the half / quarter of the 160 KiB and the kernels ran at 1 / 3 workgroups per CU); a list of n > CAP / 2
    // keys then averages up to 2 keys per bucket.
    int nb = THREADS * 4;
    while (nb < n && nb < CAP / 2) nb <<= 1;
    for (int i = tid; i < nb / 2 + 1; i += THREADS) s_cnt[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        dmin = s_red[w] < dmin ? s_red[w] : dmin;
        dmax = s_red[NW + w] > dmax ? s_red[NW + w] : dmax;
    }
    // monotone map depth -> bucket: float conversion, multiplication by a positive constant and truncation are all
    // non-decreasing, so a key with a larger depth never lands in a smaller bucket (equal depths share a bucket)
    const float scale = (float)nb / ((float)(dmax - dmin) + 1.0f) * 0.99999f;
    auto bucket_of = [&](unsigned long long key) -> unsigned {
        const unsigned b = (unsigned)((float)((unsigned)(key >> 32) - dmin) * scale);
        return b < (unsigned)nb ? b : (unsigned)nb - 1u;
    };
    // ---- count (the returned old value is the key's arrival slot in its bucket), exclusive scan, place
    unsigned short slot[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = r * THREADS + tid;
        slot[r] = 0;
        if (r < ER && i < n) {
            const unsigned b = bucket_of(k[r]);
            const unsigned old = atomicAdd(&s_cnt[b >> 1], 1u << ((b & 1u) << 4));
            slot[r] = (unsigned short)((old >> ((b & 1u) << 4)) & 0xffffu);
        }
    }
    __syncthreads();
    // thread t owns buckets [t * per, (t + 1) * per): per >= 4, whole words
    const int per = nb / THREADS;
    unsigned sum = 0u, mx = 0u;
    for (int w = 0; w < per / 2; ++w) {
        const unsigned c = s_cnt[tid * (per / 2) + w];
        const unsigned c0 = c & 0xffffu, c1 = c >> 16;
        sum += c0 + c1;
        mx = c0 > mx ? c0 : mx;
        mx = c1 > mx ? c1 : mx;
    }
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = gs2m_shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const unsigned o = gs2m_shfl_xor(mx, m);
        mx = o > mx ? o : mx;
    }
    if (lane == 63) s_red[2 * NW + wave] = incl;
    if (lane == 0) s_red[3 * NW + wave] = mx;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_red[2 * NW + w];
    unsigned wg_max = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) wg_max = s_red[3 * NW + w] > wg_max ? s_red[3 * NW + w] : wg_max;
    wg_max = (unsigned)gs2m_uniform((int)wg_max);
    if (wg_max > GS2M_BUCKET_MAX) {
        // clustered depths: LDS bitonic network over the whole list (same result; O(n log^2 n))
        __syncthreads();
        int m = 1024;
        while (m < n) m <<= 1;
        for (int i = tid; i < m; i += THREADS) s_key[i] = i < n ? kv[i] : ~0ull;
        __syncthreads();
        for (int kk = 2; kk <= m; kk <<= 1) {
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                for (int i = tid; i < m / 2; i += THREADS) {
                    const int lo_i = ((i & ~(jj - 1)) << 1) | (i & (jj - 1)), hi_i = lo_i | jj;   // pair (lo_i, lo_i + jj)
                    const bool up = (lo_i & kk) == 0;
                    const unsigned long long x = s_key[lo_i], y = s_key[hi_i];
                    if ((y < x) == up) {
                        s_key[lo_i] = y;
                        s_key[hi_i] = x;
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < n; i += THREADS) kv[i] = s_key[i];
        return;
    }
    for (int w = 0; w < per / 2; ++w) {
        const unsigned c = s_cnt[tid * (per / 2) + w];
        const unsigned c0 = c & 0xffffu, c1 = c >> 16;
        s_cnt[tid * (per / 2) + w] = run | ((run + c0) << 16);   // starts (<= 4096 each)
        run += c0 + c1;
    }
    if (tid == THREADS - 1) s_cnt[nb / 2] = (unsigned)n;                 // sentinel: start of bucket nb
    __syncthreads();
    auto start_of = [&](unsigned b) -> unsigned { return (s_cnt[b >> 1] >> ((b & 1u) << 4)) & 0xffffu; };
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r < ER) {   // wave-uniform
            const int i = r * THREADS + tid;
            const unsigned st = start_of(i < n ? bucket_of(k[r]) : 0u);   // unconditional read: the live reads in flight together
            if (i < n) s_key[st + slot[r]] = k[r];
        }
    }
    __syncthreads();
    // ---- rank inside the bucket -> final position; slot j of the bucket-ordered array goes to lo + rank
    // (as in sort_wave_bucket: keys and bucket bounds first, then the buckets of a thread's keys walked in lock step for
    // `wg_max` rounds -- 8 independent LDS reads per round; two passes of 8 keys when a thread holds 16)
    constexpr int RC = 8;
    // one chunk of up to RC key slots per thread with LIVE of them in use (compile-time: the rounds loop stays branch-free)
    auto rank_chunk = [&](const int c, auto live_tag) __attribute__((always_inline)) {
        constexpr int LIVE = (int)decltype(live_tag)::value;
        unsigned long long key[LIVE];
        unsigned lo[LIVE], len[LIVE], rank[LIVE];
        // unconditional LDS reads (indices inside the arrays: j < CAP, buckets <= nb), lanes without a key get len = 0
#pragma unroll
        for (int r = 0; r < LIVE; ++r) key[r] = s_key[(c * RC + r) * THREADS + tid];
#pragma unroll
        for (int r = 0; r < LIVE; ++r) {
            const bool in = (c * RC + r) * THREADS + tid < n;
            const unsigned b = in ? bucket_of(key[r]) : 0u;
            const unsigned st = start_of(b), en = start_of(b + 1u);
            lo[r] = st;
            len[r] = in ? en - st : 0u;
            rank[r] = 0u;
        }
        for (unsigned q = 0; q < wg_max; ++q) {
#pragma unroll
            for (int r = 0; r < LIVE; ++r) {
                const bool more = q < len[r];
                const unsigned long long other = s_key[more ? lo[r] + q : lo[r]];
                rank[r] += (more && other < key[r]) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < LIVE; ++r)
            if ((c * RC + r) * THREADS + tid < n) kv[lo[r] + rank[r]] = key[r];
    };
#pragma unroll
    for (int c = 0; c < E / RC; ++c) {
        const int live = ER - c * RC;   // wave-uniform
        if (live <= 0) break;
        switch (live < RC ? live : RC) {
            case 1: rank_chunk(c, std::integral_constant<int, 1>{}); break;
            case 2: rank_chunk(c, std::integral_constant<int, 2>{}); break;
            case 3: rank_chunk(c, std::integral_constant<int, 3>{}); break;
            case 4: rank_chunk(c, std::integral_constant<int, 4>{}); break;
            case 5: rank_chunk(c, std::integral_constant<int, 5>{}); break;
            case 6: rank_chunk(c, std::integral_constant<int, 6>{}); break;
            case 7: rank_chunk(c, std::integral_constant<int, 7>{}); break;
            default: rank_chunk(c, std::integral_constant<int, 8>{}); break;
        }
    }
}

// list = sort_lists[v][cls]: word 0 = number of tiles of the class, then their ids (k_tile_scan); a small grid walks it
template <int CAP, int THREADS, int CLS>
GS2M_DEVICE void sort_tiles_bucket_body(unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start, int tiles,
                                        unsigned cap, const unsigned* __restrict__ sort_lists) {
    __shared__ unsigned long long s_key[CAP];   // keys grouped by bucket
    __shared__ unsigned s_cnt[CAP / 4 + 2];     // two 16-bit counters / start offsets per word: <= CAP / 2 buckets
    __shared__ unsigned s_red[4 * (THREADS / 64)];
    const int tid = (int)threadIdx.x, v = (int)blockIdx.y;
    const unsigned* list = sort_lists + ((size_t)v * GS2M_SORT_CLASSES + CLS) * (tiles + 1);
    const unsigned count = list[0];
    for (unsigned li = blockIdx.x; li < count; li += gridDim.x) {
        const int t = (int)list[1 + li];
        unsigned b0 = tile_start[(size_t)v * (tiles + 1) + t];
        unsigned e0 = tile_start[(size_t)v * (tiles + 1) + t + 1];
        if (b0 > cap) b0 = cap;
        if (e0 > cap) e0 = cap;
        const int n = (int)(e0 - b0);
        if (n > 1) {
            if (n <= CAP) sort_list_bucket<CAP, THREADS>(keys + (size_t)v * cap + b0, n, s_key, s_cnt, s_red, tid);
        }
        __syncthreads();   // the LDS arrays are reused by the next list
    }
}
// The occupancy attribute (a literal: it takes no template argument) makes the residency the LDS allows real: 4 workgroups
// of the <4096> class, 2 of the <8192> class per CU = 8 waves per SIMD with 8 keys per thread (64 registers; 16 keys per
// thread = 4 waves per SIMD measured slower in round 3 and was removed).
#define GS2M_SORT_BUCKET_KERNEL(NAME, CAP, THREADS, CLS, WAVES)                                                           \
    GS2M_KERNEL void __launch_bounds__(THREADS) GS2M_WAVES_PER_SIMD(WAVES)                                                      \
    NAME(unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start, int tiles, unsigned cap,        \
         const unsigned* __restrict__ sort_lists) {                                                                      \
        sort_tiles_bucket_body<CAP, THREADS, CLS>(keys, tile_start, tiles, cap, sort_lists);                              \
    }
GS2M_SORT_BUCKET_KERNEL(k_sort_tiles_bucket_4096x8, 4096, 512, 0, 8)
GS2M_SORT_BUCKET_KERNEL(k_sort_tiles_bucket_8192x8, 8192, 1024, 1, 8)

// Lists of more than GS2M_SORT_BUCKET_CAP instances (rare: a tile stack of a very dense scene): LDS-sorted runs of
// GS2M_SORT_LDS keys (register-blocked bitonic, 3 stages through LDS), then rank-based merge passes through HBM
// (keys <-> tmp) by the same workgroup (keys are unique, so rank = index in own run + lower_bound in the sibling run).
// The merge passes read keys other threads of the workgroup wrote to global memory: same CU, same L1 -> visible after
// the workgroup barrier.
GS2M_KERNEL void __launch_bounds__(256)
k_sort_tiles(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ tmp,
             const unsigned* __restrict__ tile_start, int tiles, unsigned cap, const unsigned* __restrict__ sort_lists) {
    __shared__ unsigned long long s[GS2M_SORT_LDS];
    const int tid = (int)threadIdx.x;
    const int v = (int)blockIdx.y;
    const unsigned* list = sort_lists + ((size_t)v * GS2M_SORT_CLASSES + 2) * (tiles + 1);   // lists of > 8192 instances
    const unsigned count = list[0];
    for (unsigned li = blockIdx.x; li < count; li += gridDim.x) {
        const int t = (int)list[1 + li];
        unsigned b = tile_start[(size_t)v * (tiles + 1) + t];
        unsigned e = tile_start[(size_t)v * (tiles + 1) + t + 1];
        if (b > cap) b = cap;
        if (e > cap) e = cap;
        const int n = (int)(e - b);
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
        __syncthreads();
    }
}

