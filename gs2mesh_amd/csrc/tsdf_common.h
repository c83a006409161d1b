// tsdf_common.h -- block-sparse TSDF volume in HBM (see DESIGN.md "TSDF data layout").
//
// Replaces Open3D 0.17 ScalableTSDFVolume (hash map of UniformTSDFVolume units on the host,
// integrated serially with OpenMP over 16 x-slices) as used by gs2mesh_utils/tsdf_utils.py:53-56,107.
//   * blocks of 16^3 voxels live in a pre-allocated pool, addressed through an open-addressing
//     hash table keyed by the block index (unbounded volume, like upstream);
//   * per voxel: tsdf f32 (running mean, same fp32 update as upstream), weight f32, and the colour
//     as an exact integer SUM of the u8 samples (3 x u32; upstream keeps a float64 running mean --
//     mean = sum / weight);
//   * SoA, block-major; inside a block the 4096 voxels are stored as 64 micro-blocks of 4x4x4:
//     voxel (x,y,z) at ((z>>2)*16 + (x>>2)*4 + (y>>2))*64 + (z&3)*16 + (x&3)*4 + (y&3).  A wave owns
//     one micro-block per step: its 64 state words are 256 contiguous bytes (one coalesced access per
//     array) AND its 64 voxel centres project into a ~14x14-pixel patch, so the depth / colour gathers
//     touch ~4x fewer cache lines than a 4x16 slice of a z-plane would.
#pragma once
#include "platform.h"

#define GS2M_TSDF_RES 16
#define GS2M_TSDF_VOX 4096
#define GS2M_TSDF_EMPTY 0xffffffffffffffffull
#define GS2M_TSDF_KEY_BIAS (1 << 20)

struct TsdfVolume {  // device pointers + sizes, passed by value
    float* tsdf;                   // [max_blocks][4096]
    float* weight;                 // [max_blocks][4096]
    unsigned* rgb;                 // [max_blocks][3][4096]
    int* block_keys;               // [max_blocks][3]
    unsigned char* halo;           // [max_blocks]  1 = neighbour-only block (multi-GPU owner-side extraction)
    unsigned long long* hash_keys; // [hash_cap]
    int* hash_vals;                // [hash_cap]  slot or -1
    unsigned* stamp;               // [hash_cap]  last frame id that touched the entry
    unsigned* touched;             // [hash_cap]  hash indices touched this frame / batch
    unsigned long long* fmask;     // [hash_cap]  batch mode: bit f = frame f of the batch touched the block (0 between batches)
    unsigned* counters;            // [0] n_blocks  [1] touched_count  [2] overflow flags
    unsigned long long* totals;    // [0] block updates (sum over frames of touched blocks)
    unsigned hash_cap;             // power of two
    unsigned max_blocks;
    int has_color;
};

struct TsdfFrame {  // per-frame uniforms, passed by value in the launch packet
    double pose[12];  // camera->world, rows 0..2 of inverse(extrinsic)
    double fx, fy, cx, cy;
    double unit_length, sdf_trunc, depth_trunc;
    float E[12];  // (float)extrinsic rows 0..2
    float fx_f, fy_f, cx_f, cy_f, fx_inv_f, fy_inv_f;
    float voxel_length_f, half_voxel_length_f, sdf_trunc_f, sdf_trunc_inv_f;
    float Es02, Es12, Es22;  // E[:,2] * voxel_length (incremental z step)
    float safe_w, safe_h;
    float depth_scale_f, min_depth_f;
    float depth_trunc_up_f;  // smallest float >= depth_trunc:  (double)d >= depth_trunc  <=>  d >= depth_trunc_up_f
    int W, H, stride, nx, ny;
    unsigned frame_id;
    int use_mask, use_min;
};

// One frame of a batch (gs2m_tsdf_integrate_batch): the per-frame uniforms + its images.  An array of these lives in
// device memory; the frame index is wave-uniform, so the kernels read it with scalar loads.
struct TsdfBatchFrame {
    TsdfFrame f;
    const float* depth;
    const unsigned char* color;
    const unsigned char* mask;
};
#define GS2M_TSDF_MAX_BATCH 64   // frames per batch: one bit per frame in the per-block touch mask

// device-internal voxel index (see header); host code uses the same formula in gs2m_tsdf_download
#define GS2M_TSDF_VINDEX(x, y, z) (((((z) >> 2) * 16 + ((x) >> 2) * 4 + ((y) >> 2)) * 64) + ((z)&3) * 16 + ((x)&3) * 4 + ((y)&3))

GS2M_DEVICE unsigned long long tsdf_pack_key(int bx, int by, int bz) {
    return ((unsigned long long)(unsigned)(bx + GS2M_TSDF_KEY_BIAS) << 42) |
           ((unsigned long long)(unsigned)(by + GS2M_TSDF_KEY_BIAS) << 21) |
           (unsigned long long)(unsigned)(bz + GS2M_TSDF_KEY_BIAS);
}
GS2M_DEVICE bool tsdf_key_in_range(int bx, int by, int bz) {
    const int B = GS2M_TSDF_KEY_BIAS;
    return bx >= -B && bx < B && by >= -B && by < B && bz >= -B && bz < B;
}
GS2M_DEVICE unsigned tsdf_hash(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}
