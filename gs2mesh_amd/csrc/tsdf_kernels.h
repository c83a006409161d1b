// tsdf_kernels.h -- per-frame TSDF integration kernels (compiled with -ffp-contract=off).
//
// Reference path: volume.integrate(rgbd, intrinsic, inv(extrinsic)) (gs2mesh_utils/tsdf_utils.py:107)
// = Open3D 0.17 ScalableTSDFVolume::Integrate (restated in oracle/tsdf_oracle.cpp):
//   (i)  back-project every `stride`-th valid depth pixel to world space in fp64, mark every
//        16^3 block overlapping the +-sdf_trunc box of the point (first touch in this frame);
//   (ii) for each marked block sweep its 4096 voxels: project the voxel centre, fetch depth,
//        sdf = (d - z) * ||ray||, and where sdf > -trunc update the running means.
// Here: (i) = k_tsdf_touch (thread per strided pixel, wave-deduplicated block boxes, blocks
// found-or-inserted in a device hash table and stamped with the frame id) + k_tsdf_compact (stamped
// cells -> the frame's block list); (ii) = k_tsdf_integrate, one 256-thread workgroup per touched
// block (grid-stride over the device-side list: no host sync between the phases), sweeping the
// block's 4x4x4 micro-blocks with the same fp32 camera-space z chain as upstream (replayed from the
// column start, so every rounding matches).
// The RGBD conversion (depth/scale, >= trunc -> 0; RGBDImage.create_from_color_and_depth,
// tsdf_utils.py:88-93) and TSDF.run's mask / min-depth preprocessing (tsdf_utils.py:68-83) are
// fused into the depth fetch.
#pragma once
#include "tsdf_common.h"

// depth as the integrator sees it (tsdf_utils.py:68-93 + Image::ConvertDepthToFloatImage)
GS2M_DEVICE float tsdf_fetch_depth(const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                                   const TsdfFrame& f, int u, int v) {
    const size_t p = (size_t)v * f.W + u;
    float d = depth[p];
    if (f.use_mask && mask[p] == 0) d = d * 0.0f;           // depth = depth * mask
    if (f.use_min && d < f.min_depth_f) d = 0.0f;            // depth[depth < min] = 0
    if (f.depth_scale_f != 1.0f) d /= f.depth_scale_f;       // *p /= (float)depth_scale (x / 1 == x: skipped, uniform)
    if (d >= f.depth_trunc_up_f) d = 0.0f;                   // if (*p >= depth_trunc) *p = 0  (float vs double threshold, exactly)
    return d;
}

// find-or-insert `key`; returns the hash index or 0xffffffff on failure (table full)
GS2M_DEVICE unsigned tsdf_find_or_insert(const TsdfVolume& V, unsigned long long key, int bx, int by, int bz) {
    const unsigned mask = V.hash_cap - 1u;
    unsigned h = tsdf_hash(key) & mask;
    for (unsigned probe = 0; probe < V.hash_cap; ++probe, h = (h + 1u) & mask) {
        unsigned long long k = V.hash_keys[h];
        if (k == key) return h;
        if (k == GS2M_TSDF_EMPTY) {
            const unsigned long long prev = atomicCAS(&V.hash_keys[h], GS2M_TSDF_EMPTY, key);
            if (prev == GS2M_TSDF_EMPTY) {
                // we inserted: take a pool slot (the pool is zero-initialised = a fresh Open3D unit)
                const unsigned slot = atomicAdd(&V.counters[0], 1u);
                if (slot < V.max_blocks) {
                    V.block_keys[3 * (size_t)slot] = bx;
                    V.block_keys[3 * (size_t)slot + 1] = by;
                    V.block_keys[3 * (size_t)slot + 2] = bz;
                    V.hash_vals[h] = (int)slot;
                } else {
                    atomicOr(&V.counters[2], 1u);  // pool exhausted
                }
                return h;
            }
            if (prev == key) return h;
            // somebody else took the cell for another key: keep probing
        }
    }
    atomicOr(&V.counters[2], 2u);  // hash table full
    return 0xffffffffu;
}

// read-only lookup: slot or -1
GS2M_DEVICE int tsdf_lookup(const TsdfVolume& V, unsigned long long key) {
    const unsigned mask = V.hash_cap - 1u;
    unsigned h = tsdf_hash(key) & mask;
    for (unsigned probe = 0; probe < V.hash_cap; ++probe, h = (h + 1u) & mask) {
        const unsigned long long k = V.hash_keys[h];
        if (k == key) return V.hash_vals[h];
        if (k == GS2M_TSDF_EMPTY) return -1;
    }
    return -1;
}

// k_tsdf_touch: one thread per strided pixel back-projects its point (fp64, as upstream) and gets the
// block box [lo, hi] of its +-trunc cube (up to 27 blocks with the reference defaults).  Neighbouring
// pixels share almost all of those blocks -- a frame marks ~2 k distinct blocks out of ~3 M
// (point, block) pairs -- and every hash probe is a random L2 request, so the pairs are deduplicated
// per wave first: the 64 boxes go to LDS, the wave takes the union box, lane l tests candidate
// block base + l of the union against the 64 boxes (LDS broadcast reads) and only then probes the
// hash table.  ~50x fewer probes / stamp atomics than one probe per (point, block).
struct TouchStage {
    int lo[3][64];
    int hi[3][64];
};

GS2M_DEVICE int wave_min_i(int v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const int o = gs2m_shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}
GS2M_DEVICE int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const int o = gs2m_shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}

// BATCH_BIT < 0: per-frame mode, the touched hash cell is stamped with the frame id; >= 0: batch mode, bit BATCH_BIT of
// the cell's frame mask is set.
GS2M_DEVICE void tsdf_touch_body(const TsdfVolume& V, const TsdfFrame& f, const float* __restrict__ depth,
                                 const unsigned char* __restrict__ mask, TouchStage* stage_all, int batch_bit) {
    const int lane = (int)(threadIdx.x & 63u);
    TouchStage* stg = &stage_all[threadIdx.x >> 6];
    const int idx = (int)(blockIdx.x * 256u + threadIdx.x);
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    bool valid = false;
    if (idx < f.nx * f.ny) {
        const int i = (idx / f.nx) * f.stride;  // row
        const int j = (idx % f.nx) * f.stride;  // column
        const float p = tsdf_fetch_depth(depth, mask, f, j, i);
        if (p > 0.0f) {
            valid = true;
            // PointCloudFactory.cpp CreatePointCloudFromFloatDepthImage (fp64)
            const double z = (double)p;
            const double x = (j - f.cx) * z / f.fx;
            const double y = (i - f.cy) * z / f.fy;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double pw = f.pose[4 * r + 0] * x + f.pose[4 * r + 1] * y + f.pose[4 * r + 2] * z + f.pose[4 * r + 3] * 1.0;
                // ScalableTSDFVolume::LocateVolumeUnit: floor(p / volume_unit_length)
                lo[r] = (int)floor((pw - f.sdf_trunc) / f.unit_length);
                hi[r] = (int)floor((pw + f.sdf_trunc) / f.unit_length);
            }
        }
    }
    if (gs2m_ballot(valid ? 1 : 0) == 0ull) return;  // wave-uniform
    int wlo[3], whi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        wlo[r] = wave_min_i(lo[r]);
        whi[r] = wave_max_i(hi[r]);
    }
    // Neighbouring pixels mostly have the SAME block box (a +-trunc cube is ~1.3 blocks wide, 64 strided pixels span ~5): only the
    // boxes that differ from the previous lane's are staged (round 4: typically 5-15 of 64), so the containment loop below runs
    // over the distinct boxes instead of all 64 lanes.  The union of the kept boxes is the union of all boxes: same block set.
    bool same = lane > 0 && valid;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int plo = gs2m_shfl_up(lo[r], 1), phi = gs2m_shfl_up(hi[r], 1);
        same = same && plo == lo[r] && phi == hi[r];
    }
    const bool pvalid = gs2m_shfl_up(valid ? 1 : 0, 1) != 0;
    const bool keep = valid && !(same && pvalid);
    const unsigned long long kept = gs2m_ballot(keep ? 1 : 0);
    const int n_keep = gs2m_popc64(kept);
    if (keep) {
        const int slot = gs2m_popc64(kept & ((1ull << lane) - 1ull));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            stg->lo[r][slot] = lo[r];
            stg->hi[r][slot] = hi[r];
        }
    }
    gs2m_wave_sync();
    const long long total_ll = ((long long)whi[0] - wlo[0] + 1) * ((long long)whi[1] - wlo[1] + 1) *
                               ((long long)whi[2] - wlo[2] + 1);
    if (total_ll > (1ll << 24)) {  // a wave spanning > 16 M blocks: not a depth map of this volume
        if (lane == 0) atomicOr(&V.counters[2], 4u);
        return;
    }
    const int ny = whi[1] - wlo[1] + 1, nz = whi[2] - wlo[2] + 1;
    const int nyz = ny * nz;
    const int total = (int)total_ll;
    for (int base = 0; base < total; base += 64) {
        const int c = base + lane;
        if (c >= total) continue;
        const int cx = c / nyz, rem = c - cx * nyz;
        const int cy = rem / nz;
        const int bx = wlo[0] + cx, by = wlo[1] + cy, bz = wlo[2] + (rem - cy * nz);
        // containment in ANY staged box: branch-free so the broadcast LDS reads pipeline
        int in_any = 0;
#pragma unroll 4
        for (int k = 0; k < n_keep; ++k)
            in_any |= (int)(bx >= stg->lo[0][k]) & (int)(bx <= stg->hi[0][k]) & (int)(by >= stg->lo[1][k]) &
                      (int)(by <= stg->hi[1][k]) & (int)(bz >= stg->lo[2][k]) & (int)(bz <= stg->hi[2][k]);
        const bool inside = in_any != 0;
        if (!inside) continue;
        if (!tsdf_key_in_range(bx, by, bz)) {
            atomicOr(&V.counters[2], 4u);
            continue;
        }
        const unsigned h = tsdf_find_or_insert(V, tsdf_pack_key(bx, by, bz), bx, by, bz);
        if (h == 0xffffffffu) continue;
        // mark "touched in this frame" (touched_volume_units_ of upstream).  No list append here: ~2 k
        // returning atomics on ONE counter serialise at ~12 ns each (guide: fanin); k_tsdf_compact builds
        // the list with one atomic per 1024 hash cells instead.
        if (batch_bit < 0) {
            if (V.stamp[h] != f.frame_id) V.stamp[h] = f.frame_id;
        } else {
            const unsigned long long bit = 1ull << batch_bit;
            if (!(V.fmask[h] & bit)) atomicOr(&V.fmask[h], bit);
        }
    }
}

GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_touch(TsdfVolume V, TsdfFrame f, const float* __restrict__ depth, const unsigned char* __restrict__ mask) {
    __shared__ TouchStage stage_all[4];
    tsdf_touch_body(V, f, depth, mask, stage_all, -1);
}

// batch mode: blockIdx.y = frame of the batch
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_touch_batch(TsdfVolume V, const TsdfBatchFrame* __restrict__ frames) {
    __shared__ TouchStage stage_all[4];
    const TsdfBatchFrame& bf = frames[blockIdx.y];
    tsdf_touch_body(V, bf.f, bf.depth, bf.mask, stage_all, (int)blockIdx.y);
}


// Compacts the hash cells stamped in this frame into V.touched (order = hash order).
// frame_id == 0: batch mode -- cells whose frame mask is non-zero.
GS2M_KERNEL void __launch_bounds__(1024)
k_tsdf_compact(TsdfVolume V, unsigned frame_id) {
    __shared__ unsigned wave_cnt[16];
    __shared__ unsigned wg_base;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned h = blockIdx.x * 1024u + (unsigned)tid;
    const bool t = h < V.hash_cap && (frame_id ? V.stamp[h] == frame_id : V.fmask[h] != 0ull);
    const unsigned long long m = gs2m_ballot(t ? 1 : 0);
    if (lane == 0) wave_cnt[wave] = (unsigned)gs2m_popc64(m);
    __syncthreads();
    if (tid == 0) {
        unsigned s = 0;
        for (int k = 0; k < 16; ++k) {
            const unsigned c = wave_cnt[k];
            wave_cnt[k] = s;
            s += c;
        }
        wg_base = s ? atomicAdd(&V.counters[1], s) : 0u;
    }
    __syncthreads();
    if (t) {
        const unsigned rank = (unsigned)gs2m_popc64(m & ((1ull << lane) - 1ull));
        V.touched[wg_base + wave_cnt[wave] + rank] = h;
    }
}

// k_tsdf_integrate: one 256-thread workgroup per touched block (grid-stride over the device-side
// list).  Wave w, lane l sweep the block's 64 micro-blocks of 4x4x4 voxels (tsdf_common.h), four
// micro-blocks in flight per wave: all projections first (pure ALU), then all depth gathers, then all
// state loads + colour gathers, then the stores = ~4 dependent memory round trips per 4 micro-blocks.
// The camera-space point of a voxel is advanced along z by repeated fp32 addition exactly as upstream
// does (rounding accumulates along z), so a lane replays the z steps below its voxel.
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_integrate(TsdfVolume V, TsdfFrame f, const float* __restrict__ depth, const unsigned char* __restrict__ color,
                 const unsigned char* __restrict__ mask) {
    __shared__ float s_p0[16], s_p1[16];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lz = lane >> 4, lx = (lane >> 2) & 3, ly = lane & 3;
    const unsigned n_touched = V.counters[1];
    const int last_pix = f.W * f.H - 1;
    for (unsigned it = blockIdx.x; it < n_touched; it += gridDim.x) {
        const unsigned h = V.touched[it];
        // block index from the hash key itself (no dependent block_keys load); slot in parallel
        const unsigned long long key = V.hash_keys[h];
        const int slot = V.hash_vals[h];
        if (slot < 0) continue;  // pool overflow (flagged); uniform across the workgroup
        const int bx = (int)((key >> 42) & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        const int by = (int)((key >> 21) & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        const int bz = (int)(key & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        // OpenVolumeUnit: origin = index.cast<double>() * volume_unit_length
        const double ox = (double)bx * f.unit_length, oy = (double)by * f.unit_length, oz = (double)bz * f.unit_length;
        const float p2 = (float)(f.half_voxel_length_f + oz);
        // voxel-centre coordinates of the 16 x / y indices, once per block (fp64 add + casts as upstream:
        // float(half + vl*i + origin)), shared through LDS
        __syncthreads();  // previous block's readers are done
        if (tid < 16) s_p0[tid] = (float)(f.half_voxel_length_f + f.voxel_length_f * tid + ox);
        else if (tid < 32) s_p1[tid - 16] = (float)(f.half_voxel_length_f + f.voxel_length_f * (tid - 16) + oy);
        __syncthreads();
        float* bt = V.tsdf + (size_t)slot * GS2M_TSDF_VOX;
        float* bw = V.weight + (size_t)slot * GS2M_TSDF_VOX;
        unsigned* bc = V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            int pix[4], vi[4];
            float zc[4], mult[4], d[4], tnew[4];
            // (a) projections.  g is the micro-block z index for all four k: mb = g*16 + wave*4 + k
            const int z0 = g * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int mb = g * 16 + wave * 4 + k;  // micro-block (mz = g, mx = wave, my = k)
                const int x = wave * 4 + lx, y = k * 4 + ly;
                vi[k] = mb * 64 + lane;
                // UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier
                const float p0 = s_p0[x];
                const float p1 = s_p1[y];
                float pc0 = f.E[0] * p0 + f.E[1] * p1 + f.E[2] * p2 + f.E[3] * 1.f;
                float pc1 = f.E[4] * p0 + f.E[5] * p1 + f.E[6] * p2 + f.E[7] * 1.f;
                float pc2 = f.E[8] * p0 + f.E[9] * p1 + f.E[10] * p2 + f.E[11] * 1.f;
                // replay the z steps below this voxel: z0 wave-uniform steps, then lz (0..3) of its own
                for (int s = 0; s < z0; ++s) {
                    pc0 += f.Es02;
                    pc1 += f.Es12;
                    pc2 += f.Es22;
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (s < lz) {
                        pc0 += f.Es02;
                        pc1 += f.Es12;
                        pc2 += f.Es22;
                    }
                }
                pix[k] = -1;
                zc[k] = pc2;
                mult[k] = 0.0f;
                if (pc2 <= 0) continue;
                const float u_f = pc0 * f.fx_f / pc2 + f.cx_f + 0.5f;
                const float v_f = pc1 * f.fy_f / pc2 + f.cy_f + 0.5f;
                if (!(u_f >= 0.0001f && u_f < f.safe_w && v_f >= 0.0001f && v_f < f.safe_h)) continue;
                const int u = (int)u_f;
                const int v = (int)v_f;
                pix[k] = v * f.W + u;
                // Image::CreateDepthToCameraDistanceMultiplierFloatImage, evaluated on the fly
                const float xx = (u - f.cx_f) * f.fx_inv_f;
                const float yy = (v - f.cy_f) * f.fy_inv_f;
                mult[k] = sqrtf(xx * xx + yy * yy + 1.0f);
            }
            // (b) depth gathers, back to back
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = 0.0f;
                if (pix[k] >= 0) {
                    float dd = depth[pix[k]];
                    if (f.use_mask && mask[pix[k]] == 0) dd = dd * 0.0f;
                    if (f.use_min && dd < f.min_depth_f) dd = 0.0f;
                    if (f.depth_scale_f != 1.0f) dd /= f.depth_scale_f;  // x / 1 == x exactly: uniform skip
                    if (dd >= f.depth_trunc_up_f) dd = 0.0f;
                    d[k] = dd;
                }
            }
            // (c) decide
            bool upd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sdf = (d[k] - zc[k]) * mult[k];
                upd[k] = pix[k] >= 0 && d[k] > 0.0f && sdf > -f.sdf_trunc_f;
                tnew[k] = fminf(1.0f, sdf * f.sdf_trunc_inv_f);
            }
            // (d) state loads + colour gathers
            float w[4], t[4];
            unsigned c0[4], c1[4], c2[4], rgbp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (upd[k]) {
                    w[k] = bw[vi[k]];
                    t[k] = bt[vi[k]];
                    if (V.has_color) {
                        c0[k] = bc[vi[k]];
                        c1[k] = bc[GS2M_TSDF_VOX + vi[k]];
                        c2[k] = bc[2 * GS2M_TSDF_VOX + vi[k]];
                        // r | g<<8 | b<<16 in ONE gather (the 4th byte belongs to the next pixel; the last pixel of
                        // the image is read bytewise so that nothing past the buffer is touched)
                        const unsigned char* c = color + 3 * (size_t)pix[k];
                        rgbp[k] = pix[k] < last_pix ? gs2m_load_u32_unaligned(c)
                                                    : ((unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16));
                    }
                }
            }
            // (e) update + stores
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (upd[k]) {
                    bt[vi[k]] = (t[k] * w[k] + tnew[k]) / (w[k] + 1.0f);
                    if (V.has_color) {
                        bc[vi[k]] = c0[k] + (rgbp[k] & 0xffu);
                        bc[GS2M_TSDF_VOX + vi[k]] = c1[k] + ((rgbp[k] >> 8) & 0xffu);
                        bc[2 * GS2M_TSDF_VOX + vi[k]] = c2[k] + ((rgbp[k] >> 16) & 0xffu);
                    }
                    bw[vi[k]] = w[k] + 1.0f;
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) V.totals[0] += n_touched;
}

// k_tsdf_integrate_batch: the frames of a batch in ONE sweep over the touched blocks, voxel-stationary (SURVEY.md 7,
// step 6 iii).  A 256-thread workgroup owns a z-quarter zq of a block; thread (x, y) keeps the state of the 4 voxels
// (x, y, 4 zq .. 4 zq + 3) in registers (20 words), walks the frames that touched the block IN FRAME ORDER (bit f of the
// block's frame mask; Open3D integrates a block for a frame only if that frame's points touch it) and applies exactly the
// per-frame update (same fp32 sequence, same incremental z chain: a thread replays the 4 zq steps below its run once per
// frame, then steps along it) -- results are bit-identical to calling k_tsdf_integrate frame by frame.  Compared with the
// per-frame kernel: the voxel state is read and written once per batch instead of once per frame (40 B x updated voxels
// x frames -> 20 B x voxels of touched blocks), 4.5 instead of ~22 replayed additions per voxel and frame, the distance
// multiplier (a correctly rounded sqrt) only where a depth sample exists, frame uniforms by scalar loads.
// One 256-thread workgroup per (block, z-quarter) (round 3; the round-2 form, one 1024-thread workgroup per block at 94 VGPRs =
// 4 waves per SIMD, was removed in round 4): 88 VGPRs admit 5 such workgroups per CU, the work items are four times finer for
// the dynamic hand-out, zq (and with it the replay loop of the z chain) is wave-uniform, and the block's frame mask is cleared
// by k_tsdf_clear_fmask afterwards (the four quarters of a block run on different workgroups).  C2: 39.2 -> 32.7 us per frame
// in sweeps of 10, 36.9 -> 29.7 in sweeps of 24; forcing 6 / 7 waves per SIMD (80 / 72 VGPRs, spills) gains nothing (32.9 /
// 35.7).  Same arithmetic per voxel: bit-identical.
// Round 4, measured and not kept (profiles/r4_experiments.txt): the frames of a block probed four at a time by the four waves
// of the workgroup ("frame lanes": LDS exchange + barrier per round, one voxel of state per thread) 30 -> 80 us per frame --
// the dependent round trips only move from the frame loop to a loop over x-quarters; the colour gathered together with the
// depth (one round trip per frame instead of two dependent ones, but for every probed voxel) 30.5 -> 35.2: the sweep is bound
// by the number of gathers and its ~12 M vector instructions per frame, not by exposed latency; 6 waves per SIMD (80 VGPRs,
// 5 spilled) 30.5 -> 30.5 on C2, 76.6 -> 72.1 on C4 (not worth a second kernel).
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_integrate_batch(TsdfVolume V, const TsdfBatchFrame* __restrict__ frames) {
    __shared__ float s_p0[16], s_p1[16];
    const int tid = (int)threadIdx.x;
    const int x = (tid >> 4) & 15, y = tid & 15;
    __shared__ unsigned s_it;
    const unsigned n_touched = V.counters[1] * 4u;   // work items: (block, z-quarter)
    const TsdfFrame& f0 = frames[0].f;   // volume constants (voxel length, truncation) are the same in every frame
    for (;;) {
        // blocks are handed out dynamically (counters[3], zeroed before the launch): a block costs as many frame passes as
        // frames touched it (1 ... the batch size), so a static round-robin leaves the workgroups 1.5x apart at the end
        __syncthreads();  // previous block's readers of s_p0 / s_p1 / s_it (and of fmask) are done
        if (tid == 0) s_it = atomicAdd(&V.counters[3], 1u);
        __syncthreads();
        const unsigned it = s_it;
        if (it >= n_touched) break;
        const int zq = (int)(it & 3u);
        const unsigned h = V.touched[it >> 2];
        const unsigned long long key = V.hash_keys[h];
        const int slot = V.hash_vals[h];
        const unsigned long long fm = V.fmask[h];
        if (slot < 0) continue;   // pool overflow (flagged); uniform across the workgroup
        const int bx = (int)((key >> 42) & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        const int by = (int)((key >> 21) & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        const int bz = (int)(key & 0x1fffffull) - GS2M_TSDF_KEY_BIAS;
        const double ox = (double)bx * f0.unit_length, oy = (double)by * f0.unit_length, oz = (double)bz * f0.unit_length;
        const float p2 = (float)(f0.half_voxel_length_f + oz);
        if (tid < 16) s_p0[tid] = (float)(f0.half_voxel_length_f + f0.voxel_length_f * tid + ox);
        else if (tid < 32) s_p1[tid - 16] = (float)(f0.half_voxel_length_f + f0.voxel_length_f * (tid - 16) + oy);
        __syncthreads();
        const float p0 = s_p0[x], p1 = s_p1[y];
        float* bt = V.tsdf + (size_t)slot * GS2M_TSDF_VOX;
        float* bw = V.weight + (size_t)slot * GS2M_TSDF_VOX;
        unsigned* bc = V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX;
        const int vi0 = GS2M_TSDF_VINDEX(x, y, 4 * zq);   // the run z = 4 zq .. 4 zq + 3 sits at vi0 + 16 j (one micro-block)
        float w[4], t[4];
        unsigned c0[4], c1[4], c2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = bw[vi0 + 16 * j];
            t[j] = bt[vi0 + 16 * j];
            c0[j] = c1[j] = c2[j] = 0u;
            if (V.has_color) {
                c0[j] = bc[vi0 + 16 * j];
                c1[j] = bc[GS2M_TSDF_VOX + vi0 + 16 * j];
                c2[j] = bc[2 * GS2M_TSDF_VOX + vi0 + 16 * j];
            }
        }
        // Software pipeline over the frames of the block: the projection of frame n + 1 and its depth gathers are issued
        // before frame n is applied, so a wave always has the next frame's memory round trip in flight (the kernel is bound
        // by gather latency: 63 % of the wave cycles wait at 4 waves per SIMD, PMC).  Frames are still APPLIED in ascending
        // order with the same arithmetic: bit-identical to the frame-by-frame path.
        struct Probe {
            int fi;
            int pix[4];
            unsigned uv[4];        // u | v << 16 (image sizes <= 65535, checked by the host side)
            float zc[4], d[4];
        };
        auto probe = [&](const int fi, Probe& P) __attribute__((always_inline)) {
            const TsdfBatchFrame& bf = frames[fi];
            const TsdfFrame& f = bf.f;
            const float* __restrict__ depth = bf.depth;
            const unsigned char* __restrict__ mask = bf.mask;
            P.fi = fi;
            // UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier: camera-space centre of voxel (x, y, 0) ...
            float pc0 = f.E[0] * p0 + f.E[1] * p1 + f.E[2] * p2 + f.E[3] * 1.f;
            float pc1 = f.E[4] * p0 + f.E[5] * p1 + f.E[6] * p2 + f.E[7] * 1.f;
            float pc2 = f.E[8] * p0 + f.E[9] * p1 + f.E[10] * p2 + f.E[11] * 1.f;
            // ... advanced along z by repeated addition, as upstream does (rounding accumulates along z)
            for (int s = 0; s < 4 * zq; ++s) {
                pc0 += f.Es02;
                pc1 += f.Es12;
                pc2 += f.Es22;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                P.pix[j] = -1;
                P.uv[j] = 0u;
                P.zc[j] = pc2;
                if (!(pc2 <= 0)) {
                    const float u_f = pc0 * f.fx_f / pc2 + f.cx_f + 0.5f;
                    const float v_f = pc1 * f.fy_f / pc2 + f.cy_f + 0.5f;
                    if (u_f >= 0.0001f && u_f < f.safe_w && v_f >= 0.0001f && v_f < f.safe_h) {
                        const int uu = (int)u_f, vv = (int)v_f;
                        P.uv[j] = (unsigned)uu | ((unsigned)vv << 16);
                        P.pix[j] = vv * f.W + uu;
                    }
                }
                pc0 += f.Es02;
                pc1 += f.Es12;
                pc2 += f.Es22;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // depth gathers, back to back (consumed by apply() one frame later)
                P.d[j] = 0.0f;
                if (P.pix[j] >= 0) {
                    float dd = depth[P.pix[j]];
                    if (f.use_mask && mask[P.pix[j]] == 0) dd = dd * 0.0f;
                    P.d[j] = dd;
                }
            }
        };
        auto apply = [&](Probe& P) __attribute__((always_inline)) {
            const TsdfBatchFrame& bf = frames[P.fi];
            const TsdfFrame& f = bf.f;
            const unsigned char* __restrict__ color = bf.color;
            const int last_pix = f.W * f.H - 1;
            unsigned rgbp[4];
            bool upd[4];
            float tnew[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                upd[j] = false;
                tnew[j] = 0.0f;
                rgbp[j] = 0u;
                float dd = P.d[j];
                if (P.pix[j] >= 0) {
                    if (f.use_min && dd < f.min_depth_f) dd = 0.0f;
                    if (f.depth_scale_f != 1.0f) dd /= f.depth_scale_f;
                    if (dd >= f.depth_trunc_up_f) dd = 0.0f;
                }
                if (P.pix[j] >= 0 && dd > 0.0f) {
                    // Image::CreateDepthToCameraDistanceMultiplierFloatImage, evaluated on the fly and only here
                    const float xx = ((int)(P.uv[j] & 0xffffu) - f.cx_f) * f.fx_inv_f;
                    const float yy = ((int)(P.uv[j] >> 16) - f.cy_f) * f.fy_inv_f;
                    const float mult = sqrtf(xx * xx + yy * yy + 1.0f);
                    const float sdf = (dd - P.zc[j]) * mult;
                    if (sdf > -f.sdf_trunc_f) {
                        upd[j] = true;
                        tnew[j] = fminf(1.0f, sdf * f.sdf_trunc_inv_f);
                        if (V.has_color) {
                            const unsigned char* c = color + 3 * (size_t)P.pix[j];
                            rgbp[j] = P.pix[j] < last_pix ? gs2m_load_u32_unaligned(c)
                                                          : ((unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16));
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (upd[j]) {
                    t[j] = (t[j] * w[j] + tnew[j]) / (w[j] + 1.0f);
                    c0[j] += rgbp[j] & 0xffu;
                    c1[j] += (rgbp[j] >> 8) & 0xffu;
                    c2[j] += (rgbp[j] >> 16) & 0xffu;
                    w[j] = w[j] + 1.0f;
                }
            }
        };
        unsigned long long todo = fm;
        Probe A, B;
        if (todo != 0ull) {
            probe(__ffsll(todo) - 1, A);   // frames in ascending order
            todo &= todo - 1ull;
            while (todo != 0ull) {
                probe(__ffsll(todo) - 1, B);
                todo &= todo - 1ull;
                apply(A);
                if (todo == 0ull) {
                    A = B;
                    break;
                }
                probe(__ffsll(todo) - 1, A);
                todo &= todo - 1ull;
                apply(B);
                if (todo == 0ull) break;
            }
            apply(A);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bw[vi0 + 16 * j] = w[j];
            bt[vi0 + 16 * j] = t[j];
            if (V.has_color) {
                bc[vi0 + 16 * j] = c0[j];
                bc[GS2M_TSDF_VOX + vi0 + 16 * j] = c1[j];
                bc[2 * GS2M_TSDF_VOX + vi0 + 16 * j] = c2[j];
            }
        }
        if (tid == 0 && zq == 0) atomicAdd(&V.totals[0], (unsigned long long)gs2m_popc64(fm));   // block updates = sum over frames of touched blocks
    }
}

// after k_tsdf_integrate_batch<1>: frame masks of the touched blocks back to 0 for the next batch
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_clear_fmask(TsdfVolume V) {
    const unsigned n = V.counters[1];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) V.fmask[V.touched[i]] = 0ull;
}

// Reset without touching the unused part of the pool: blocks are handed out in slot order, so only slots
// [0, counters[0]) can hold state (a dense 512^3 pool is 2.7 GB; a scan touches a few hundred MB of it).
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_clear_used(TsdfVolume V) {
    unsigned n = V.counters[0];
    if (n > V.max_blocks) n = V.max_blocks;
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (unsigned slot = blockIdx.x; slot < n; slot += gridDim.x) {
        float4* t4 = reinterpret_cast<float4*>(V.tsdf + (size_t)slot * GS2M_TSDF_VOX);
        float4* w4 = reinterpret_cast<float4*>(V.weight + (size_t)slot * GS2M_TSDF_VOX);
        for (int i = (int)threadIdx.x; i < GS2M_TSDF_VOX / 4; i += 256) {
            t4[i] = z;
            w4[i] = z;
        }
        if (V.has_color) {
            float4* c4 = reinterpret_cast<float4*>(V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX);
            for (int i = (int)threadIdx.x; i < 3 * GS2M_TSDF_VOX / 4; i += 256) c4[i] = z;
        }
        if (threadIdx.x == 0) V.halo[slot] = 0;
    }
}

// gs2m_tsdf_replace: the slots [first, counters[0]) back to the clean state (the blocks about to be unpacked overwrite slots
// [0, first) completely: clearing those as well -- what a reset does -- would write the reduced volume's bytes twice)
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_clear_from(TsdfVolume V, unsigned first) {
    unsigned n = V.counters[0];
    if (n > V.max_blocks) n = V.max_blocks;
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (unsigned slot = first + blockIdx.x; slot < n; slot += gridDim.x) {
        float4* t4 = reinterpret_cast<float4*>(V.tsdf + (size_t)slot * GS2M_TSDF_VOX);
        float4* w4 = reinterpret_cast<float4*>(V.weight + (size_t)slot * GS2M_TSDF_VOX);
        for (int i = (int)threadIdx.x; i < GS2M_TSDF_VOX / 4; i += 256) {
            t4[i] = z;
            w4[i] = z;
        }
        if (V.has_color) {
            float4* c4 = reinterpret_cast<float4*>(V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX);
            for (int i = (int)threadIdx.x; i < 3 * GS2M_TSDF_VOX / 4; i += 256) c4[i] = z;
        }
        if (threadIdx.x == 0) V.halo[slot] = 0;
    }
}

// gs2m_tsdf_replace, after its unpack: the unpack hands out slots [0, counters[0]) -- exactly [0, upto) for `upto` distinct in-range
// keys, fewer when a key repeats, lies outside the key range or the hash table is full.  The slots it did NOT hand out,
// [counters[0], upto), were skipped by k_tsdf_clear_from and still hold the previous volume's state: cleared here, on the device,
// so that stale voxels never reappear in a block allocated later.  No work in the regular case (counters[0] == upto).
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_clear_gap(TsdfVolume V, unsigned upto) {
    unsigned first = V.counters[0];
    if (upto > V.max_blocks) upto = V.max_blocks;
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (unsigned slot = first + blockIdx.x; slot < upto; slot += gridDim.x) {
        float4* t4 = reinterpret_cast<float4*>(V.tsdf + (size_t)slot * GS2M_TSDF_VOX);
        float4* w4 = reinterpret_cast<float4*>(V.weight + (size_t)slot * GS2M_TSDF_VOX);
        for (int i = (int)threadIdx.x; i < GS2M_TSDF_VOX / 4; i += 256) {
            t4[i] = z;
            w4[i] = z;
        }
        if (V.has_color) {
            float4* c4 = reinterpret_cast<float4*>(V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX);
            for (int i = (int)threadIdx.x; i < 3 * GS2M_TSDF_VOX / 4; i += 256) c4[i] = z;
        }
        if (threadIdx.x == 0) V.halo[slot] = 0;
    }
}

// ---- multi-GPU exchange -----------------------------------------------------------------------
// Exchange forms (GS2M_XFORM_*, include/gs2mesh_amd.h):
//   0 SUM_F32     one fp32 buffer [n][5][4096]: planes wsum = tsdf * weight, weight, sum r, sum g, sum b -- counts and colour
//                 sums are integers < 2^24, exact in fp32 and independent of the reduction order (one fp32 SUM collective);
//   1 RAW_F32     the same buffer with the planes VERBATIM (tsdf, weight, sums): for blocks that are already reduced
//                 (halo copies: fl(fl(t * w) / w) is not always t);
//   2 SUM_PACKED  wsum as fp32 [n][4096] + ONE int64 per voxel  w | sum r << 10 | sum g << 28 | sum b << 46  (valid while the
//                 volume has seen <= 1023 frames: w < 2^10, colour sums <= 255 * 1023 < 2^18, no carry between the fields
//                 under an integer SUM): 12 instead of 20 bytes per voxel on the wire.
// pack: one workgroup per canonical key (zeros where the block is not allocated here or only a halo copy).
#define GS2M_XF_SUM_F32 0
#define GS2M_XF_RAW_F32 1
#define GS2M_XF_SUM_PACKED 2
template <int FORM>
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_pack(TsdfVolume V, const int* __restrict__ keys, float* __restrict__ buf, long long* __restrict__ ibuf) {
    const int tid = (int)threadIdx.x;
    const size_t b = blockIdx.x;
    const int bx = keys[3 * b], by = keys[3 * b + 1], bz = keys[3 * b + 2];
    int slot = tsdf_key_in_range(bx, by, bz) ? tsdf_lookup(V, tsdf_pack_key(bx, by, bz)) : -1;
    if (slot >= 0 && FORM != GS2M_XF_RAW_F32 && V.halo[slot]) slot = -1;   // a halo copy is another rank's block: never summed twice
    float* o = buf + b * (FORM == GS2M_XF_SUM_PACKED ? 1 : 5) * GS2M_TSDF_VOX;
    for (int i = tid; i < GS2M_TSDF_VOX; i += 256) {
        float w = 0.f, t = 0.f;
        unsigned c0 = 0u, c1 = 0u, c2 = 0u;
        if (slot >= 0) {
            w = V.weight[(size_t)slot * GS2M_TSDF_VOX + i];
            t = V.tsdf[(size_t)slot * GS2M_TSDF_VOX + i];
            if (FORM != GS2M_XF_RAW_F32) t = t * w;
            if (V.has_color) {
                c0 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + i];
                c1 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i];
                c2 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i];
            }
        }
        o[i] = t;
        if (FORM == GS2M_XF_SUM_PACKED) {
            // the fields only stay apart under an integer SUM while w < 2^10 and the colour sums < 2^18 IN THE RESULT; a
            // local value that already exceeds its field is flagged (status bit 8): the caller's frame bound was wrong
            // (state injected through unpack, C-API users) -- checked by gs2mesh_amd.parallel after the exchange
            if (w > 1023.0f || ((c0 | c1 | c2) >> 18) != 0u) atomicOr(&V.counters[2], 8u);
            ibuf[b * GS2M_TSDF_VOX + i] = (long long)((unsigned long long)(unsigned)w | ((unsigned long long)c0 << 10) |
                                                      ((unsigned long long)c1 << 28) | ((unsigned long long)c2 << 46));
        } else {
            o[GS2M_TSDF_VOX + i] = w;
            o[2 * GS2M_TSDF_VOX + i] = (float)c0;
            o[3 * GS2M_TSDF_VOX + i] = (float)c1;
            o[4 * GS2M_TSDF_VOX + i] = (float)c2;
        }
    }
}

// unpack: replaces the state of the listed blocks (allocating as needed): tsdf = wsum / weight (SUM forms) or verbatim
// (RAW); `halo` marks them as neighbour-only blocks of another rank's part of the volume (mesh extraction reads them but
// starts no cube there; pack / block_keys skip them).
template <int FORM>
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_unpack(TsdfVolume V, const int* __restrict__ keys, const float* __restrict__ buf, const long long* __restrict__ ibuf,
              int halo) {
    __shared__ int s_slot;
    const int tid = (int)threadIdx.x;
    const size_t b = blockIdx.x;
    if (tid == 0) {
        const int bx = keys[3 * b], by = keys[3 * b + 1], bz = keys[3 * b + 2];
        int slot = -1;
        if (tsdf_key_in_range(bx, by, bz)) {
            const unsigned h = tsdf_find_or_insert(V, tsdf_pack_key(bx, by, bz), bx, by, bz);
            if (h != 0xffffffffu) slot = V.hash_vals[h];
        } else {
            atomicOr(&V.counters[2], 4u);
        }
        if (slot >= 0) V.halo[slot] = (unsigned char)(halo ? 1 : 0);
        s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) return;
    const float* in = buf + b * (FORM == GS2M_XF_SUM_PACKED ? 1 : 5) * GS2M_TSDF_VOX;
    for (int i = tid; i < GS2M_TSDF_VOX; i += 256) {
        float w;
        unsigned c0, c1, c2;
        if (FORM == GS2M_XF_SUM_PACKED) {
            const unsigned long long p = (unsigned long long)ibuf[b * GS2M_TSDF_VOX + i];
            w = (float)(unsigned)(p & 0x3ffull);
            c0 = (unsigned)((p >> 10) & 0x3ffffull);
            c1 = (unsigned)((p >> 28) & 0x3ffffull);
            c2 = (unsigned)(p >> 46);
        } else {
            w = in[GS2M_TSDF_VOX + i];
            c0 = (unsigned)in[2 * GS2M_TSDF_VOX + i];
            c1 = (unsigned)in[3 * GS2M_TSDF_VOX + i];
            c2 = (unsigned)in[4 * GS2M_TSDF_VOX + i];
        }
        V.weight[(size_t)slot * GS2M_TSDF_VOX + i] = w;
        V.tsdf[(size_t)slot * GS2M_TSDF_VOX + i] = FORM == GS2M_XF_RAW_F32 ? in[i] : (w > 0.f ? in[i] / w : 0.f);
        if (V.has_color) {
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + i] = c0;
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i] = c1;
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i] = c2;
        }
    }
}

// block keys of the first n slots; halo copies (another rank's blocks) are reported as the out-of-range sentinel key
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_owned_keys(TsdfVolume V, unsigned n, int* __restrict__ keys) {
    const unsigned s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n) return;
    const bool h = V.halo[s] != 0;
    const int sentinel = GS2M_TSDF_KEY_BIAS - 1;
    keys[3 * (size_t)s] = h ? sentinel : V.block_keys[3 * (size_t)s];
    keys[3 * (size_t)s + 1] = h ? sentinel : V.block_keys[3 * (size_t)s + 1];
    keys[3 * (size_t)s + 2] = h ? sentinel : V.block_keys[3 * (size_t)s + 2];
}

// ---- block-map key exchange (round 5) -----------------------------------------------------------
// The union of the ranks' block sets is formed as a dense MAP over a window of block indices: every rank marks its own blocks,
// ONE all_reduce merges the ranks, and the canonical key list is the marked cells in cell order -- identical on every rank, no
// gather, no sort, no host pass over keys.  SURVEY 8e asks for a bitmap under a bitwise OR; RCCL (like NCCL) has no bitwise
// reduction, so the map holds one BYTE per block and the collective is MAX over uint8 (32 KiB for a 32^3-block window, 256 KiB
// for the default 64^3).  Window: lo[k] <= b[k] < lo[k] + dim[k];
//   cell = ((bx - lo.x) * dim.y + (by - lo.y)) * dim.z + (bz - lo.z)     (ascending cell = ascending (x, y, z) = packed-key order)
// The header travels behind the map in the same buffer (GS2M_TSDF_MAP_HEADER bytes + 8 per rank), arranged so that a bytewise MAX
// reduces it: flags are 0 / 1 bytes; values that must agree are sent as the bytes of x and of ~x (byte + complement byte == 255
// after the MAX <=> all ranks sent the same byte); every rank writes its frame counts into its OWN slot (the others send zeros).
//   [0] a block outside the window (every rank then takes the gather path)     [1..4] overflow flags 1, 2, 4, 8 of V.counters[2]
//   [5] replicated   [6] holds halo copies   [8..11] / [12..15] max_blocks / ~max_blocks   [16..19] / [20..23] window hash / ~hash
//   [24..27] number of marked cells (written by k_tsdf_map_keys on the reduced buffer)
//   [32 + 8 r .. +3] frames_local, [36 + 8 r .. +3] frames_base of rank r (little endian)
#define GS2M_TSDF_MAP_HEADER 32
GS2M_DEVICE void tsdf_put_u32(unsigned char* p, unsigned v) {
    p[0] = (unsigned char)(v & 255u);
    p[1] = (unsigned char)((v >> 8) & 255u);
    p[2] = (unsigned char)((v >> 16) & 255u);
    p[3] = (unsigned char)(v >> 24);
}
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_block_map(TsdfVolume V, int lox, int loy, int loz, int dx, int dy, int dz, unsigned char* __restrict__ cells, unsigned n_cells,
                 unsigned flags, unsigned win_hash, int rank, unsigned frames_local, unsigned frames_base) {
    unsigned n = V.counters[0];
    if (n > V.max_blocks) n = V.max_blocks;
    unsigned char* hdr = cells + n_cells;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned ov = V.counters[2];
        for (int b = 0; b < 4; ++b) hdr[1 + b] = (unsigned char)((ov >> b) & 1u);
        hdr[5] = (unsigned char)(flags & 1u);
        hdr[6] = (unsigned char)((flags >> 1) & 1u);
        tsdf_put_u32(hdr + 8, V.max_blocks);
        tsdf_put_u32(hdr + 12, ~V.max_blocks);
        tsdf_put_u32(hdr + 16, win_hash);
        tsdf_put_u32(hdr + 20, ~win_hash);
        tsdf_put_u32(hdr + GS2M_TSDF_MAP_HEADER + 8 * rank, frames_local);
        tsdf_put_u32(hdr + GS2M_TSDF_MAP_HEADER + 8 * rank + 4, frames_base);
    }
    for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < n; s += gridDim.x * 256u) {
        if (V.halo[s]) continue;     // a halo copy is another rank's block
        const int bx = V.block_keys[3 * (size_t)s] - lox, by = V.block_keys[3 * (size_t)s + 1] - loy, bz = V.block_keys[3 * (size_t)s + 2] - loz;
        if (bx < 0 || by < 0 || bz < 0 || bx >= dx || by >= dy || bz >= dz) {
            hdr[0] = 1;
            continue;
        }
        cells[((size_t)bx * (unsigned)dy + (unsigned)by) * (unsigned)dz + (unsigned)bz] = 1;
    }
}

// Marked cells of the (reduced) map -> keys [n][3] in cell order; header bytes [24..27] = n.  One 1024-thread workgroup: thread t
// owns the consecutive cells [t * per, (t + 1) * per), per a multiple of 16 (16-byte loads): count, workgroup exclusive scan,
// emit.  Keys beyond max_keys are counted, not written (the caller compares n with its buffer).
GS2M_KERNEL void __launch_bounds__(1024)
k_tsdf_map_keys(int lox, int loy, int loz, int dy, int dz, unsigned char* __restrict__ cells, unsigned n_cells, int* __restrict__ keys,
                unsigned max_keys) {
    __shared__ unsigned wave_sum[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned per = ((n_cells + 1023u) / 1024u + 15u) & ~15u;
    const unsigned c0 = (unsigned)tid * per < n_cells ? (unsigned)tid * per : n_cells;
    const unsigned c1 = c0 + per < n_cells ? c0 + per : n_cells;
    unsigned cnt = 0;
    for (unsigned c = c0; c < c1; c += 16u) {
        if (c + 16u <= c1) {
            const uint4 v = *reinterpret_cast<const uint4*>(cells + c);      // the buffer is 16-byte aligned, c0 a multiple of 16
            // cells are 0 / 1 bytes: the byte sum of a word = its number of marked cells
            const unsigned t = (v.x & 0x01010101u) + (v.y & 0x01010101u) + (v.z & 0x01010101u) + (v.w & 0x01010101u);
            cnt += (t & 255u) + ((t >> 8) & 255u) + ((t >> 16) & 255u) + (t >> 24);
        } else {
            for (unsigned k = c; k < c1; ++k) cnt += cells[k] ? 1u : 0u;
        }
    }
    unsigned inc = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = gs2m_shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < wave; ++k) base += wave_sum[k];
    unsigned pos = base + inc - cnt;
    if (tid == 1023) tsdf_put_u32(cells + n_cells + 24, base + inc);
    if (cnt == 0u) return;
    for (unsigned c = c0; c < c1; ++c) {
        if (!cells[c]) continue;
        if (pos < max_keys) {
            const unsigned z = c % (unsigned)dz, xy = c / (unsigned)dz;
            keys[3 * (size_t)pos] = (int)(xy / (unsigned)dy) + lox;
            keys[3 * (size_t)pos + 1] = (int)(xy % (unsigned)dy) + loy;
            keys[3 * (size_t)pos + 2] = (int)z + loz;
        }
        ++pos;
    }
}
