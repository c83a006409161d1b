// tsdf_kernels.h -- per-frame TSDF integration kernels (compiled with -ffp-contract=off).
//
// Reference path: volume.integrate(rgbd, intrinsic, inv(extrinsic)) (gs2mesh_utils/tsdf_utils.py:107)
// = Open3D 0.17 ScalableTSDFVolume::Integrate (restated in oracle/tsdf_oracle.cpp):
//   (i)  back-project every `stride`-th valid depth pixel to world space in fp64, mark every
//        16^3 block overlapping the +-sdf_trunc box of the point (first touch in this frame);
//   (ii) for each marked block sweep its 4096 voxels: project the voxel centre, fetch depth,
//        sdf = (d - z) * ||ray||, and where sdf > -trunc update the running means.
// Here: (i) = k_tsdf_touch, one thread per strided pixel, blocks found-or-inserted in a device
// hash table and appended once per frame to a touched list (atomic frame stamp); (ii) =
// k_tsdf_integrate, one 256-thread workgroup per touched block (persistent grid-stride loop over
// the device-side list: no host sync between the phases), thread (x,y) walks z with the same
// incremental fp32 camera-space step as upstream so every rounding matches.
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
    d /= f.depth_scale_f;                                    // *p /= (float)depth_scale
    if ((double)d >= f.depth_trunc) d = 0.0f;                // if (*p >= depth_trunc) *p = 0
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

// One thread per (strided pixel, block offset): the +-trunc box of a point spans at most
// `span` = floor(2 trunc / L) + 2 blocks per axis, so thread idx handles point idx / span^3 and the
// block lo + offset(idx % span^3) if it lies inside the point's [lo, hi] box.  Every thread runs one
// short hash probe (2-3 dependent loads) instead of one thread walking 27 of them back to back.
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_touch(TsdfVolume V, TsdfFrame f, const float* __restrict__ depth, const unsigned char* __restrict__ mask,
             int span) {
    const int span3 = span * span * span;
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int idx = (int)(gidx / span3);
    if (idx >= f.nx * f.ny) return;
    const int off = (int)(gidx - (long long)idx * span3);
    const int i = (idx / f.nx) * f.stride;  // row
    const int j = (idx % f.nx) * f.stride;  // column
    const float p = tsdf_fetch_depth(depth, mask, f, j, i);
    if (!(p > 0.0f)) return;
    // PointCloudFactory.cpp CreatePointCloudFromFloatDepthImage (fp64)
    const double z = (double)p;
    const double x = (j - f.cx) * z / f.fx;
    const double y = (i - f.cy) * z / f.fy;
    int lo[3], hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double pw = f.pose[4 * r + 0] * x + f.pose[4 * r + 1] * y + f.pose[4 * r + 2] * z + f.pose[4 * r + 3] * 1.0;
        // ScalableTSDFVolume::LocateVolumeUnit: floor(p / volume_unit_length)
        lo[r] = (int)floor((pw - f.sdf_trunc) / f.unit_length);
        hi[r] = (int)floor((pw + f.sdf_trunc) / f.unit_length);
    }
    const int bx = lo[0] + off / (span * span), by = lo[1] + (off / span) % span, bz = lo[2] + off % span;
    if (bx > hi[0] || by > hi[1] || bz > hi[2]) return;
    if (!tsdf_key_in_range(bx, by, bz)) {
        atomicOr(&V.counters[2], 4u);
        return;
    }
    const unsigned h = tsdf_find_or_insert(V, tsdf_pack_key(bx, by, bz), bx, by, bz);
    if (h == 0xffffffffu) return;
    // first touch in this frame? (touched_volume_units_ of upstream)
    if (V.stamp[h] == f.frame_id) return;  // plain read: a stale miss only costs an atomic
    if (atomicExch(&V.stamp[h], f.frame_id) != f.frame_id) {
        const unsigned t = atomicAdd(&V.counters[1], 1u);
        V.touched[t] = h;
    }
}

// One 256-thread workgroup per touched block; thread (x,y) walks z.  The 16 z steps are processed in
// two groups of 8 with all projections first, then all depth gathers, then all state loads, then the
// stores: ~4 dependent memory round trips per group instead of 2 per voxel.
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_integrate(TsdfVolume V, TsdfFrame f, const float* __restrict__ depth, const unsigned char* __restrict__ color,
                 const unsigned char* __restrict__ mask) {
    const int tid = (int)threadIdx.x;
    const int x = tid >> 4, y = tid & 15;
    const unsigned n_touched = V.counters[1];
    for (unsigned it = blockIdx.x; it < n_touched; it += gridDim.x) {
        const unsigned h = V.touched[it];
        const int slot = V.hash_vals[h];
        if (slot < 0) continue;  // pool overflow (flagged)
        const int bx = V.block_keys[3 * (size_t)slot], by = V.block_keys[3 * (size_t)slot + 1],
                  bz = V.block_keys[3 * (size_t)slot + 2];
        // OpenVolumeUnit: origin = index.cast<double>() * volume_unit_length
        const double ox = (double)bx * f.unit_length, oy = (double)by * f.unit_length, oz = (double)bz * f.unit_length;
        // UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier
        const float p0 = (float)(f.half_voxel_length_f + f.voxel_length_f * x + ox);
        const float p1 = (float)(f.half_voxel_length_f + f.voxel_length_f * y + oy);
        const float p2 = (float)(f.half_voxel_length_f + oz);
        float pc0 = f.E[0] * p0 + f.E[1] * p1 + f.E[2] * p2 + f.E[3] * 1.f;
        float pc1 = f.E[4] * p0 + f.E[5] * p1 + f.E[6] * p2 + f.E[7] * 1.f;
        float pc2 = f.E[8] * p0 + f.E[9] * p1 + f.E[10] * p2 + f.E[11] * 1.f;
        float* bt = V.tsdf + (size_t)slot * GS2M_TSDF_VOX;
        float* bw = V.weight + (size_t)slot * GS2M_TSDF_VOX;
        unsigned* bc = V.rgb + (size_t)slot * 3 * GS2M_TSDF_VOX;
#pragma unroll 1
        for (int zg = 0; zg < GS2M_TSDF_RES; zg += 8) {
            int pix[8];      // pixel index or -1
            float zc[8], mult[8], d[8], tnew[8];
            // (a) projections: pure ALU, same incremental fp32 chain as upstream
#pragma unroll
            for (int k = 0; k < 8; ++k, pc0 += f.Es02, pc1 += f.Es12, pc2 += f.Es22) {
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
            for (int k = 0; k < 8; ++k) {
                d[k] = 0.0f;
                if (pix[k] >= 0) {
                    float dd = depth[pix[k]];
                    if (f.use_mask && mask[pix[k]] == 0) dd = dd * 0.0f;
                    if (f.use_min && dd < f.min_depth_f) dd = 0.0f;
                    dd /= f.depth_scale_f;
                    if ((double)dd >= f.depth_trunc) dd = 0.0f;
                    d[k] = dd;
                }
            }
            // (c) decide
            bool upd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float sdf = (d[k] - zc[k]) * mult[k];
                upd[k] = pix[k] >= 0 && d[k] > 0.0f && sdf > -f.sdf_trunc_f;
                tnew[k] = fminf(1.0f, sdf * f.sdf_trunc_inv_f);
            }
            // (d) state loads
            float w[8], t[8];
            unsigned c0[8], c1[8], c2[8], r[8], g[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (upd[k]) {
                    const int vi = (zg + k) * 256 + tid;
                    w[k] = bw[vi];
                    t[k] = bt[vi];
                    if (V.has_color) {
                        c0[k] = bc[vi];
                        c1[k] = bc[GS2M_TSDF_VOX + vi];
                        c2[k] = bc[2 * GS2M_TSDF_VOX + vi];
                        const unsigned char* c = color + 3 * (size_t)pix[k];
                        r[k] = c[0];
                        g[k] = c[1];
                        b[k] = c[2];
                    }
                }
            }
            // (e) update + stores
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (upd[k]) {
                    const int vi = (zg + k) * 256 + tid;
                    bt[vi] = (t[k] * w[k] + tnew[k]) / (w[k] + 1.0f);
                    if (V.has_color) {
                        bc[vi] = c0[k] + r[k];
                        bc[GS2M_TSDF_VOX + vi] = c1[k] + g[k];
                        bc[2 * GS2M_TSDF_VOX + vi] = c2[k] + b[k];
                    }
                    bw[vi] = w[k] + 1.0f;
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) V.totals[0] += n_touched;
}

// ---- multi-GPU exchange -----------------------------------------------------------------------
// pack: one workgroup per canonical key; sum form (wsum = tsdf*weight) so that the host can
// all-reduce / reduce-scatter with RCCL.
GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_pack(TsdfVolume V, const int* __restrict__ keys, float* __restrict__ wsum, float* __restrict__ weight,
            unsigned* __restrict__ rgb) {
    const int tid = (int)threadIdx.x;
    const size_t b = blockIdx.x;
    const int bx = keys[3 * b], by = keys[3 * b + 1], bz = keys[3 * b + 2];
    const int slot = tsdf_key_in_range(bx, by, bz) ? tsdf_lookup(V, tsdf_pack_key(bx, by, bz)) : -1;
    for (int i = tid; i < GS2M_TSDF_VOX; i += 256) {
        float w = 0.f, t = 0.f;
        unsigned c0 = 0, c1 = 0, c2 = 0;
        if (slot >= 0) {
            w = V.weight[(size_t)slot * GS2M_TSDF_VOX + i];
            t = V.tsdf[(size_t)slot * GS2M_TSDF_VOX + i] * w;
            if (V.has_color) {
                c0 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + i];
                c1 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i];
                c2 = V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i];
            }
        }
        wsum[b * GS2M_TSDF_VOX + i] = t;
        weight[b * GS2M_TSDF_VOX + i] = w;
        if (rgb) {
            rgb[b * 3 * GS2M_TSDF_VOX + i] = c0;
            rgb[b * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i] = c1;
            rgb[b * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i] = c2;
        }
    }
}

GS2M_KERNEL void __launch_bounds__(256)
k_tsdf_unpack(TsdfVolume V, const int* __restrict__ keys, const float* __restrict__ wsum,
              const float* __restrict__ weight, const unsigned* __restrict__ rgb) {
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
        s_slot = slot;
    }
    __syncthreads();
    const int slot = s_slot;
    if (slot < 0) return;
    for (int i = tid; i < GS2M_TSDF_VOX; i += 256) {
        const float w = weight[b * GS2M_TSDF_VOX + i];
        V.weight[(size_t)slot * GS2M_TSDF_VOX + i] = w;
        V.tsdf[(size_t)slot * GS2M_TSDF_VOX + i] = w > 0.f ? wsum[b * GS2M_TSDF_VOX + i] / w : 0.f;
        if (V.has_color && rgb) {
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + i] = rgb[b * 3 * GS2M_TSDF_VOX + i];
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i] = rgb[b * 3 * GS2M_TSDF_VOX + GS2M_TSDF_VOX + i];
            V.rgb[(size_t)slot * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i] =
                rgb[b * 3 * GS2M_TSDF_VOX + 2 * GS2M_TSDF_VOX + i];
        }
    }
}
