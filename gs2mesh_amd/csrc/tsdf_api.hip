// tsdf_api.hip -- C ABI of the TSDF half (include/gs2mesh_amd.h).  Host code; kernels in
// tsdf_kernels.hip.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/gs2mesh_amd.h"
#include "tsdf_common.h"
#include "tsdf_internal.h"
#include "roctx_ranges.h"

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            gs2m_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

struct gs2m_tsdf {
    int device = 0;
    double voxel_length = 0, sdf_trunc = 0, unit_length = 0;
    int color_type = 1, resolution = 16, stride = 4;
    TsdfVolume V;
    unsigned frame_id = 0;  // monotonic, never reset (stamps of old frames can never alias)
    unsigned* h_counters = nullptr;         // pinned [4]
    unsigned long long* h_totals = nullptr; // pinned [1]
    int n_cu = 256;
    McDevTables* d_mc = nullptr;          // marching-cubes case table (generated at create)
    unsigned* d_blk_tris = nullptr;       // [max_blocks] per-block triangle counts / offsets
    unsigned long long* d_ntri = nullptr; // [1]
    // the last mesh gs2m_tsdf_extract_mesh produced, welded, on the device (read with gs2m_tsdf_mesh_copy)
    double* mesh_v = nullptr;      // [mesh_nv][3]
    double* mesh_c = nullptr;      // [mesh_nv][3]
    int* mesh_e = nullptr;         // [mesh_nv][4]
    int* mesh_tri = nullptr;       // [mesh_nt][3]
    int64_t mesh_nv = 0, mesh_nt = 0;
    size_t mesh_cap_v = 0, mesh_cap_t = 0;   // capacities (vertices / triangles) of the cached-mesh arrays: grow-only
    char* mesh_scratch = nullptr;            // grow-only arena of the extraction's temporaries (soup, weld table, scan)
    size_t mesh_scratch_cap = 0;
    int timing = 0;
    struct EvPair {
        int stage;
        hipEvent_t a, b;
        int frames;   // frames the timed launch covered (a batch counts as that many launches)
    };
    std::vector<EvPair> ev_live;
    std::vector<hipEvent_t> ev_free;
    TsdfBatchFrame* d_bframes = nullptr;   // [GS2M_TSDF_MAX_BATCH] frame descriptors of the batch in flight
    // pinned staging ring for the descriptors (a pageable source would make the "async" copy wait for the stream)
    static const int kRing = 8;
    TsdfBatchFrame* h_bframes = nullptr;   // [kRing][GS2M_TSDF_MAX_BATCH]
    hipEvent_t ring_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned ring_next = 0;
};

static hipEvent_t tsdf_ev_get(gs2m_tsdf* t) {
    if (!t->ev_free.empty()) {
        hipEvent_t e = t->ev_free.back();
        t->ev_free.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// General 4x4 inverse (cofactors), double.  The reference hands Open3D world->camera and
// Open3D inverts it again (PointCloudFactory.cpp: camera_pose = extrinsic.inverse()).
static bool invert4x4(const double* m, double* inv) {
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int i = 0; i < 16; i++) inv[i] = a[i] * det;
    return true;
}

// `whole_pool`: at creation (hipMalloc'ed memory is not zero); later resets only clear the slots that were handed out
static int zero_state(gs2m_tsdf* t, hipStream_t st, bool whole_pool, long long keep_first = -1) {
    TsdfVolume& V = t->V;
    if (whole_pool) {
        HIPCHK(hipMemsetAsync(V.tsdf, 0, sizeof(float) * (size_t)V.max_blocks * GS2M_TSDF_VOX, st));
        HIPCHK(hipMemsetAsync(V.weight, 0, sizeof(float) * (size_t)V.max_blocks * GS2M_TSDF_VOX, st));
        if (V.rgb) HIPCHK(hipMemsetAsync(V.rgb, 0, sizeof(unsigned) * 3 * (size_t)V.max_blocks * GS2M_TSDF_VOX, st));
        HIPCHK(hipMemsetAsync(V.halo, 0, (size_t)V.max_blocks, st));
    } else if (keep_first < 0) {
        gs2m_launch_tsdf_clear_used(st, V);   // reads counters[0] on the device: no host sync
    } else {
        gs2m_launch_tsdf_clear_from(st, V, (unsigned)keep_first);   // gs2m_tsdf_replace: slots [0, keep_first) are overwritten next
    }
    HIPCHK(hipMemsetAsync(V.hash_keys, 0xff, sizeof(unsigned long long) * (size_t)V.hash_cap, st));
    HIPCHK(hipMemsetAsync(V.hash_vals, 0xff, sizeof(int) * (size_t)V.hash_cap, st));
    HIPCHK(hipMemsetAsync(V.stamp, 0, sizeof(unsigned) * (size_t)V.hash_cap, st));
    HIPCHK(hipMemsetAsync(V.fmask, 0, sizeof(unsigned long long) * (size_t)V.hash_cap, st));
    HIPCHK(hipMemsetAsync(V.counters, 0, sizeof(unsigned) * 4, st));
    HIPCHK(hipMemsetAsync(V.totals, 0, sizeof(unsigned long long), st));
    return 0;
}

extern "C" int gs2m_tsdf_create(gs2m_tsdf** out, double voxel_length, double sdf_trunc, int color_type,
                                int volume_unit_resolution, int depth_sampling_stride, int64_t max_blocks,
                                int device) {
    if (!out) {
        gs2m_set_error("gs2m_tsdf_create: out is NULL");
        return 1;
    }
    if (volume_unit_resolution != GS2M_TSDF_RES) {
        gs2m_set_error("volume_unit_resolution %d not supported (kernels are specialised for 16, Open3D's default)",
                       volume_unit_resolution);
        return 1;
    }
    if (!(voxel_length > 0) || !(sdf_trunc > 0) || depth_sampling_stride < 1 || max_blocks < 1 ||
        max_blocks > (1ll << 30) || (color_type != GS2M_TSDF_COLOR_NONE && color_type != GS2M_TSDF_COLOR_RGB8)) {
        gs2m_set_error("gs2m_tsdf_create: bad argument");
        return 1;
    }
    HIPCHK(hipSetDevice(device));
    gs2m_tsdf* t = new gs2m_tsdf();
    t->device = device;
    t->voxel_length = voxel_length;
    t->sdf_trunc = sdf_trunc;
    t->unit_length = voxel_length * volume_unit_resolution;  // ScalableTSDFVolume ctor
    t->color_type = color_type;
    t->stride = depth_sampling_stride;
    memset(&t->V, 0, sizeof(t->V));
    TsdfVolume& V = t->V;
    V.max_blocks = (unsigned)max_blocks;
    unsigned cap = 1024;
    while ((uint64_t)cap < 2ull * (uint64_t)max_blocks) cap <<= 1;
    V.hash_cap = cap;
    V.has_color = color_type == GS2M_TSDF_COLOR_RGB8;
    const size_t nvox = (size_t)max_blocks * GS2M_TSDF_VOX;
    bool ok = hipMalloc((void**)&V.tsdf, sizeof(float) * nvox) == hipSuccess &&
              hipMalloc((void**)&V.weight, sizeof(float) * nvox) == hipSuccess &&
              (!V.has_color || hipMalloc((void**)&V.rgb, sizeof(unsigned) * 3 * nvox) == hipSuccess) &&
              hipMalloc((void**)&V.block_keys, sizeof(int) * 3 * (size_t)max_blocks) == hipSuccess &&
              hipMalloc((void**)&V.halo, (size_t)max_blocks) == hipSuccess &&
              hipMalloc((void**)&V.hash_keys, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
              hipMalloc((void**)&V.hash_vals, sizeof(int) * (size_t)cap) == hipSuccess &&
              hipMalloc((void**)&V.stamp, sizeof(unsigned) * (size_t)cap) == hipSuccess &&
              hipMalloc((void**)&V.touched, sizeof(unsigned) * (size_t)cap) == hipSuccess &&
              hipMalloc((void**)&V.fmask, sizeof(unsigned long long) * (size_t)cap) == hipSuccess &&
              hipMalloc((void**)&t->d_bframes, sizeof(TsdfBatchFrame) * GS2M_TSDF_MAX_BATCH) == hipSuccess &&
              hipHostMalloc((void**)&t->h_bframes, sizeof(TsdfBatchFrame) * GS2M_TSDF_MAX_BATCH * gs2m_tsdf::kRing) == hipSuccess &&
              hipMalloc((void**)&V.counters, sizeof(unsigned) * 4) == hipSuccess &&
              hipMalloc((void**)&V.totals, sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void**)&t->d_mc, gs2m_mc_tables_bytes()) == hipSuccess &&
              hipMalloc((void**)&t->d_blk_tris, sizeof(unsigned) * (size_t)max_blocks) == hipSuccess &&
              hipMalloc((void**)&t->d_ntri, sizeof(unsigned long long)) == hipSuccess &&
              hipHostMalloc((void**)&t->h_counters, sizeof(unsigned) * 4) == hipSuccess &&
              hipHostMalloc((void**)&t->h_totals, sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        gs2m_set_error("gs2m_tsdf_create: out of device memory for %lld blocks (%.1f MiB)", (long long)max_blocks,
                       (double)nvox * 20.0 / 1048576.0);
        gs2m_tsdf_destroy(t);
        return 1;
    }
    {
        std::vector<unsigned char> tab(gs2m_mc_tables_bytes());
        if (!gs2m_mc_tables_fill(tab.data()) ||
            hipMemcpy(t->d_mc, tab.data(), tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
            gs2m_set_error("gs2m_tsdf_create: marching-cubes table generation failed");
            gs2m_tsdf_destroy(t);
            return 1;
        }
    }
    if (zero_state(t, (hipStream_t)0, true)) {
        gs2m_tsdf_destroy(t);
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)0));
    *out = t;
    return 0;
}

extern "C" int gs2m_tsdf_destroy(gs2m_tsdf* t) {
    if (!t) return 0;
    TsdfVolume& V = t->V;
    (void)hipFree(V.tsdf);
    (void)hipFree(V.weight);
    (void)hipFree(V.rgb);
    (void)hipFree(V.block_keys);
    (void)hipFree(V.halo);
    (void)hipFree(V.hash_keys);
    (void)hipFree(V.hash_vals);
    (void)hipFree(V.stamp);
    (void)hipFree(V.touched);
    (void)hipFree(V.fmask);
    (void)hipFree(t->d_bframes);
    (void)hipHostFree(t->h_bframes);
    for (auto e : t->ring_done)
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(V.counters);
    (void)hipFree(V.totals);
    (void)hipFree(t->d_mc);
    (void)hipFree(t->d_blk_tris);
    (void)hipFree(t->d_ntri);
    (void)hipFree(t->mesh_scratch);
    (void)hipFree(t->mesh_v);
    (void)hipFree(t->mesh_c);
    (void)hipFree(t->mesh_e);
    (void)hipFree(t->mesh_tri);
    (void)hipHostFree(t->h_counters);
    (void)hipHostFree(t->h_totals);
    for (auto& p : t->ev_live) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : t->ev_free) (void)hipEventDestroy(e);
    delete t;
    return 0;
}

extern "C" int gs2m_tsdf_reset(gs2m_tsdf* t, gs2m_stream stream) {
    if (!t) {
        gs2m_set_error("null handle");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    return zero_state(t, (hipStream_t)stream, false);
}

// per-frame uniforms (intrinsics, pose, truncation constants) of one integrate call
static int fill_frame(gs2m_tsdf* t, TsdfFrame& f, int width, int height, double fx, double fy, double cx, double cy,
                      const double* extrinsic_w2c, double depth_scale, double depth_trunc, double min_depth, bool has_mask) {
    memset(&f, 0, sizeof(f));
    double pose[16];
    if (!invert4x4(extrinsic_w2c, pose)) {
        gs2m_set_error("gs2m_tsdf_integrate: singular extrinsic");
        return 1;
    }
    for (int i = 0; i < 12; ++i) {
        f.pose[i] = pose[i];
        f.E[i] = (float)extrinsic_w2c[i];
    }
    f.fx = fx;
    f.fy = fy;
    f.cx = cx;
    f.cy = cy;
    f.unit_length = t->unit_length;
    f.sdf_trunc = t->sdf_trunc;
    f.depth_trunc = depth_trunc;
    f.fx_f = (float)fx;
    f.fy_f = (float)fy;
    f.cx_f = (float)cx;
    f.cy_f = (float)cy;
    f.fx_inv_f = 1.0f / (float)fx;
    f.fy_inv_f = 1.0f / (float)fy;
    f.voxel_length_f = (float)t->voxel_length;
    f.half_voxel_length_f = f.voxel_length_f * 0.5f;
    f.sdf_trunc_f = (float)t->sdf_trunc;
    f.sdf_trunc_inv_f = 1.0f / f.sdf_trunc_f;
    f.Es02 = f.E[2] * f.voxel_length_f;
    f.Es12 = f.E[6] * f.voxel_length_f;
    f.Es22 = f.E[10] * f.voxel_length_f;
    f.safe_w = width - 0.0001f;
    f.safe_h = height - 0.0001f;
    f.depth_scale_f = (float)depth_scale;
    f.min_depth_f = (float)min_depth;
    // Image::ConvertDepthToFloatImage compares the float depth with the double threshold; the same decision in fp32:
    f.depth_trunc_up_f = (float)depth_trunc;
    if ((double)f.depth_trunc_up_f < depth_trunc) f.depth_trunc_up_f = nextafterf(f.depth_trunc_up_f, INFINITY);
    f.W = width;
    f.H = height;
    f.stride = t->stride;
    f.nx = (width + t->stride - 1) / t->stride;
    f.ny = (height + t->stride - 1) / t->stride;
    f.frame_id = ++t->frame_id;
    f.use_mask = has_mask;
    f.use_min = min_depth > 0;
    return 0;
}

extern "C" int gs2m_tsdf_integrate(gs2m_tsdf* t, const float* depth, const uint8_t* color, const uint8_t* mask,
                                   int width, int height, double fx, double fy, double cx, double cy,
                                   const double* extrinsic_w2c, double depth_scale, double depth_trunc,
                                   double min_depth, gs2m_stream stream) {
    if (!t || !depth || !extrinsic_w2c) {
        gs2m_set_error("gs2m_tsdf_integrate: NULL argument");
        return 1;
    }
    if (t->V.has_color && !color) {
        // Open3D: "[ScalableTSDFVolume::Integrate] Unsupported image format."
        gs2m_set_error("[ScalableTSDFVolume::Integrate] Unsupported image format.");
        return 1;
    }
    if (width <= 0 || height <= 0 || !(fx != 0) || !(fy != 0) || !(depth_scale != 0)) {
        gs2m_set_error("gs2m_tsdf_integrate: bad intrinsics / size");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    TsdfFrame f;
    if (fill_frame(t, f, width, height, fx, fy, cx, cy, extrinsic_w2c, depth_scale, depth_trunc, min_depth, mask != nullptr))
        return 1;
    HIPCHK(hipMemsetAsync(t->V.counters + 1, 0, sizeof(unsigned), st));  // touched_count = 0
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (t->timing) {
        e0 = tsdf_ev_get(t);
        e1 = tsdf_ev_get(t);
        e2 = tsdf_ev_get(t);
        e3 = tsdf_ev_get(t);
    }
    const bool tm = e0 && e1 && e2 && e3;
    if (tm) (void)hipEventRecord(e0, st);
    {
        Gs2mRange rg("gs2m:tsdf_touch");
        gs2m_launch_tsdf_touch(st, t->V, f, depth, mask);
    }
    if (tm) {
        (void)hipEventRecord(e1, st);
        (void)hipEventRecord(e2, st);
    }
    // persistent grid: enough workgroups to fill the chip; each loops over the touched list
    {
        Gs2mRange rg("gs2m:tsdf_integrate");
        gs2m_launch_tsdf_integrate(st, t->n_cu * 7, t->V, f, depth, color, mask);
    }
    if (tm) {
        (void)hipEventRecord(e3, st);
        t->ev_live.push_back({0, e0, e1, 1});
        t->ev_live.push_back({1, e2, e3, 1});
    }
    return 0;
}

extern "C" int gs2m_tsdf_integrate_batch(gs2m_tsdf* t, int n_frames, const float* const* depth, const uint8_t* const* color,
                                         const uint8_t* const* mask, int width, int height, double fx, double fy, double cx,
                                         double cy, const double* extrinsics_w2c, double depth_scale, double depth_trunc,
                                         double min_depth, gs2m_stream stream) {
    if (!t || n_frames < 0 || (n_frames > 0 && (!depth || !extrinsics_w2c))) {
        gs2m_set_error("gs2m_tsdf_integrate_batch: NULL argument");
        return 1;
    }
    if (t->V.has_color && n_frames > 0 && !color) {
        gs2m_set_error("[ScalableTSDFVolume::Integrate] Unsupported image format.");
        return 1;
    }
    if (n_frames > 0 && (width <= 0 || height <= 0 || width > 65535 || height > 65535 || !(fx != 0) || !(fy != 0) || !(depth_scale != 0))) {
        gs2m_set_error("gs2m_tsdf_integrate_batch: bad intrinsics / size (the sweep packs pixel coordinates in 16 bits: <= 65535)");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    for (int f0 = 0; f0 < n_frames; f0 += GS2M_TSDF_MAX_BATCH) {
        const int nf = n_frames - f0 < GS2M_TSDF_MAX_BATCH ? n_frames - f0 : GS2M_TSDF_MAX_BATCH;
        // next slot of the pinned staging ring; its previous copy (kRing batches ago) has long completed
        const unsigned slot = t->ring_next++ % gs2m_tsdf::kRing;
        if (t->ring_done[slot]) HIPCHK(hipEventSynchronize(t->ring_done[slot]));
        else HIPCHK(hipEventCreateWithFlags(&t->ring_done[slot], hipEventDisableTiming));
        TsdfBatchFrame* hb = t->h_bframes + (size_t)slot * GS2M_TSDF_MAX_BATCH;
        for (int k = 0; k < nf; ++k) {
            const int i = f0 + k;
            if (!depth[i] || (t->V.has_color && !color[i])) {
                gs2m_set_error("gs2m_tsdf_integrate_batch: frame %d has a NULL image", i);
                return 1;
            }
            const unsigned char* m = mask ? mask[i] : nullptr;
            if (fill_frame(t, hb[k].f, width, height, fx, fy, cx, cy, extrinsics_w2c + 16 * (size_t)i, depth_scale, depth_trunc,
                           min_depth, m != nullptr))
                return 1;
            hb[k].depth = depth[i];
            hb[k].color = color ? color[i] : nullptr;
            hb[k].mask = m;
        }
        HIPCHK(hipMemcpyAsync(t->d_bframes, hb, sizeof(TsdfBatchFrame) * nf, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(t->ring_done[slot], st));
        HIPCHK(hipMemsetAsync(t->V.counters + 1, 0, sizeof(unsigned), st));  // touched_count = 0
        HIPCHK(hipMemsetAsync(t->V.counters + 3, 0, sizeof(unsigned), st));  // work counter of the batch sweep
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
        if (t->timing) {
            e0 = tsdf_ev_get(t);
            e1 = tsdf_ev_get(t);
            e2 = tsdf_ev_get(t);
            e3 = tsdf_ev_get(t);
        }
        const bool tm = e0 && e1 && e2 && e3;
        if (tm) (void)hipEventRecord(e0, st);
        {
            Gs2mRange rg("gs2m:tsdf_touch_batch");
            gs2m_launch_tsdf_touch_batch(st, t->V, hb[0].f, nf, t->d_bframes);
        }
        if (tm) {
            (void)hipEventRecord(e1, st);
            (void)hipEventRecord(e2, st);
        }
        {
            Gs2mRange rg("gs2m:tsdf_integrate_batch");
            gs2m_launch_tsdf_integrate_batch(st, t->n_cu, t->V, t->d_bframes);
        }
        if (tm) {
            (void)hipEventRecord(e3, st);
            t->ev_live.push_back({0, e0, e1, nf});
            t->ev_live.push_back({1, e2, e3, nf});
        }
    }
    return 0;
}

extern "C" int gs2m_tsdf_set_stage_timing(gs2m_tsdf* t, int enable) {
    if (!t) {
        gs2m_set_error("null handle");
        return 1;
    }
    t->timing = enable != 0;
    return 0;
}

extern "C" int gs2m_tsdf_stage_times(gs2m_tsdf* t, gs2m_stream stream, double* total_ms, int64_t* launches) {
    if (!t || !total_ms || !launches) {
        gs2m_set_error("gs2m_tsdf_stage_times: NULL argument");
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (auto& p : t->ev_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess && (p.stage == 0 || p.stage == 1)) {
            total_ms[p.stage] += ms;
            launches[p.stage] += p.frames;   // a batch launch counts as its frames: averages stay per frame
        }
        t->ev_free.push_back(p.a);
        t->ev_free.push_back(p.b);
    }
    t->ev_live.clear();
    return 0;
}

extern "C" int gs2m_tsdf_status(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_blocks, int64_t* block_updates,
                                int* overflow) {
    if (!t) {
        gs2m_set_error("null handle");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(t->h_counters, t->V.counters, sizeof(unsigned) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(t->h_totals, t->V.totals, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    unsigned nb = t->h_counters[0];
    if (nb > t->V.max_blocks) nb = t->V.max_blocks;
    if (n_blocks) *n_blocks = nb;
    if (block_updates) *block_updates = (int64_t)t->h_totals[0];
    if (overflow) *overflow = (int)t->h_counters[2];
    return 0;
}

extern "C" int gs2m_tsdf_flags_device(gs2m_tsdf* t, uint32_t* flags_dev, gs2m_stream stream) {
    if (!t || !flags_dev) {
        gs2m_set_error("gs2m_tsdf_flags_device: NULL argument");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipMemcpyAsync(flags_dev, t->V.counters + 2, sizeof(unsigned), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int gs2m_tsdf_block_keys(gs2m_tsdf* t, int64_t n, int32_t* keys, gs2m_stream stream) {
    if (!t || (n > 0 && !keys) || n < 0 || n > (int64_t)t->V.max_blocks) {
        gs2m_set_error("gs2m_tsdf_block_keys: bad argument");
        return 1;
    }
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(t->device));
    gs2m_launch_tsdf_owned_keys((hipStream_t)stream, (unsigned)n, t->V, keys);   // halo copies -> sentinel key
    return 0;
}

// bytes of the exchange buffer of a window of dim[0] x dim[1] x dim[2] blocks and `world` ranks: the cells (one byte per block,
// rounded up to 16 so the header is aligned), then the header
static int64_t map_bytes(const int32_t* dim, int world, unsigned* n_cells_out) {
    if (!dim || dim[0] <= 0 || dim[1] <= 0 || dim[2] <= 0 || world <= 0) return -1;
    const int64_t cells = (int64_t)dim[0] * dim[1] * dim[2];
    if (cells > ((int64_t)1 << 30)) return -1;
    const unsigned nc = (unsigned)((cells + 15) / 16 * 16);
    if (n_cells_out) *n_cells_out = nc;
    return (int64_t)nc + GS2M_TSDF_MAP_HEADER_BYTES + 8 * (int64_t)world;
}

static unsigned window_hash(const int32_t* lo, const int32_t* dim) {
    unsigned h = 2166136261u;
    for (int k = 0; k < 3; ++k) {
        h = (h ^ (unsigned)lo[k]) * 16777619u;
        h = (h ^ (unsigned)dim[k]) * 16777619u;
    }
    return h;
}

extern "C" int64_t gs2m_tsdf_map_bytes(const int32_t* dim, int world) { return map_bytes(dim, world, nullptr); }

extern "C" int gs2m_tsdf_block_map(gs2m_tsdf* t, const int32_t* lo, const int32_t* dim, int rank, int world, int64_t frames_local,
                                   int64_t frames_base, int flags, uint8_t* cells, gs2m_stream stream) {
    unsigned nc = 0;
    const int64_t total = map_bytes(dim, world, &nc);
    if (!t || !lo || !cells || total < 0 || rank < 0 || rank >= world || frames_local < 0 || frames_base < 0 ||
        frames_local > 0x7fffffff || frames_base > 0x7fffffff) {
        gs2m_set_error("gs2m_tsdf_block_map: bad argument");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(cells, 0, (size_t)total, st));
    gs2m_launch_tsdf_block_map(st, t->V, lo, dim, cells, nc, (unsigned)flags, window_hash(lo, dim), rank, (unsigned)frames_local,
                               (unsigned)frames_base);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gs2m_tsdf_map_keys(gs2m_tsdf* t, const int32_t* lo, const int32_t* dim, int world, uint8_t* cells, int32_t* keys,
                                  int64_t max_keys, uint8_t* header_host, gs2m_stream stream) {
    unsigned nc = 0;
    const int64_t total = map_bytes(dim, world, &nc);
    if (!t || !lo || !cells || total < 0 || max_keys < 0 || (max_keys > 0 && !keys) || !header_host) {
        gs2m_set_error("gs2m_tsdf_map_keys: bad argument");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    gs2m_launch_tsdf_map_keys(st, lo, dim, cells, nc, keys, (unsigned)(max_keys > 0x7fffffff ? 0x7fffffff : max_keys));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(header_host, cells + nc, (size_t)(total - nc), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int gs2m_tsdf_download(gs2m_tsdf* t, gs2m_stream stream, int64_t n, int32_t* keys, float* tsdf,
                                  float* weight, uint32_t* rgb_sum) {
    if (!t || n < 0) {
        gs2m_set_error("gs2m_tsdf_download: bad argument");
        return 1;
    }
    int64_t nb = 0;
    if (gs2m_tsdf_status(t, stream, &nb, nullptr, nullptr)) return 1;
    if (n > nb) n = nb;
    if (n == 0) return 0;
    const size_t nv = (size_t)n * GS2M_TSDF_VOX;
    if (keys) HIPCHK(hipMemcpy(keys, t->V.block_keys, sizeof(int) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<float> tmp;
    // device layout (4x4x4 micro-blocks, tsdf_common.h)  ->  Open3D IndexOf x*256 + y*16 + z
    auto reorder_f = [&](const float* src, float* dst) {
        for (int64_t b = 0; b < n; ++b)
            for (int z = 0; z < 16; ++z)
                for (int x = 0; x < 16; ++x)
                    for (int y = 0; y < 16; ++y)
                        dst[(size_t)b * GS2M_TSDF_VOX + x * 256 + y * 16 + z] =
                            src[(size_t)b * GS2M_TSDF_VOX + GS2M_TSDF_VINDEX(x, y, z)];
    };
    if (tsdf) {
        tmp.resize(nv);
        HIPCHK(hipMemcpy(tmp.data(), t->V.tsdf, sizeof(float) * nv, hipMemcpyDeviceToHost));
        reorder_f(tmp.data(), tsdf);
    }
    if (weight) {
        tmp.resize(nv);
        HIPCHK(hipMemcpy(tmp.data(), t->V.weight, sizeof(float) * nv, hipMemcpyDeviceToHost));
        reorder_f(tmp.data(), weight);
    }
    if (rgb_sum) {
        if (!t->V.has_color) {
            memset(rgb_sum, 0, sizeof(uint32_t) * 3 * nv);
        } else {
            std::vector<unsigned> c(3 * nv);
            HIPCHK(hipMemcpy(c.data(), t->V.rgb, sizeof(unsigned) * 3 * nv, hipMemcpyDeviceToHost));
            for (int64_t b = 0; b < n; ++b)
                for (int ch = 0; ch < 3; ++ch)
                    for (int z = 0; z < 16; ++z)
                        for (int x = 0; x < 16; ++x)
                            for (int y = 0; y < 16; ++y)
                                rgb_sum[((size_t)b * GS2M_TSDF_VOX + x * 256 + y * 16 + z) * 3 + ch] =
                                    c[((size_t)b * 3 + ch) * GS2M_TSDF_VOX + GS2M_TSDF_VINDEX(x, y, z)];
        }
    }
    return 0;
}

extern "C" int gs2m_tsdf_pack(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, float* buf_f32, int64_t* buf_i64,
                              gs2m_stream stream) {
    if (!t || n < 0 || form < 0 || form > 2 || (n > 0 && (!keys || !buf_f32 || (form == 2 && !buf_i64)))) {
        gs2m_set_error("gs2m_tsdf_pack: bad argument");
        return 1;
    }
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(t->device));
    gs2m_launch_tsdf_pack((hipStream_t)stream, (unsigned)n, t->V, keys, form, buf_f32, (long long*)buf_i64);
    return 0;
}

extern "C" int gs2m_tsdf_unpack(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, const float* buf_f32,
                                const int64_t* buf_i64, int halo, gs2m_stream stream) {
    if (!t || n < 0 || form < 0 || form > 2 || (n > 0 && (!keys || !buf_f32 || (form == 2 && !buf_i64)))) {
        gs2m_set_error("gs2m_tsdf_unpack: bad argument");
        return 1;
    }
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(t->device));
    gs2m_launch_tsdf_unpack((hipStream_t)stream, (unsigned)n, t->V, keys, form, buf_f32, (const long long*)buf_i64, halo);
    return 0;
}

extern "C" int gs2m_tsdf_replace(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, const float* buf_f32, const int64_t* buf_i64,
                                 gs2m_stream stream) {
    if (!t || n < 0 || (n > 0 && (!keys || !buf_f32))) {
        gs2m_set_error("gs2m_tsdf_replace: bad argument");
        return 1;
    }
    if (n > (int64_t)t->V.max_blocks) {
        gs2m_set_error("gs2m_tsdf_replace: block pool exhausted (%lld blocks to place, max_blocks = %u)", (long long)n, t->V.max_blocks);
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    // a reset that leaves the voxel state of the first n slots alone (they are overwritten by the unpack below, which hands out
    // exactly the slots [0, n) to n distinct in-range keys), then the unpack
    if (zero_state(t, (hipStream_t)stream, false, n)) return 1;
    if (n == 0) return 0;
    if (gs2m_tsdf_unpack(t, keys, n, form, buf_f32, buf_i64, 0, stream)) return 1;
    // keys that repeat / lie outside the key range / do not fit the hash table leave slots [counters[0], n) un-handed-out AND
    // un-cleared: cleared on the device (no host read; a no-op for n distinct in-range keys)
    gs2m_launch_tsdf_clear_gap((hipStream_t)stream, t->V, (unsigned)n);
    return 0;
}

extern "C" int gs2m_tsdf_pack_sum(gs2m_tsdf* t, const int32_t* keys, int64_t n, float* buf, gs2m_stream stream) {
    return gs2m_tsdf_pack(t, keys, n, 0, buf, nullptr, stream);
}

extern "C" int gs2m_tsdf_unpack_sum(gs2m_tsdf* t, const int32_t* keys, int64_t n, const float* buf, int halo,
                                    gs2m_stream stream) {
    return gs2m_tsdf_unpack(t, keys, n, 0, buf, nullptr, halo, stream);
}

extern "C" int gs2m_tsdf_extract_count(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_triangles) {
    if (!t || !n_triangles) {
        gs2m_set_error("gs2m_tsdf_extract_count: NULL argument");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    int64_t nb = 0;
    if (gs2m_tsdf_status(t, stream, &nb, nullptr, nullptr)) return 1;
    *n_triangles = 0;
    if (nb == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    gs2m_launch_mc_count(st, t->V, t->d_mc, (unsigned)nb, t->d_blk_tris, t->d_ntri);
    unsigned long long n = 0;
    HIPCHK(hipMemcpyAsync(&n, t->d_ntri, sizeof(n), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n >= (1ull << 32)) {
        gs2m_set_error("mesh too large: %llu triangles (32-bit per-block offsets)", n);
        return 1;
    }
    *n_triangles = (int64_t)n;
    return 0;
}

extern "C" int gs2m_tsdf_extract(gs2m_tsdf* t, gs2m_stream stream, int64_t max_triangles, double* vertices,
                                 double* colors, int64_t* n_triangles) {
    return gs2m_tsdf_extract_indexed(t, stream, max_triangles, vertices, colors, nullptr, n_triangles);
}

extern "C" int gs2m_tsdf_extract_indexed(gs2m_tsdf* t, gs2m_stream stream, int64_t max_triangles, double* vertices,
                                         double* colors, int32_t* edge_index, int64_t* n_triangles) {
    if (!t || !vertices || max_triangles < 0) {
        gs2m_set_error("gs2m_tsdf_extract: bad argument");
        return 1;
    }
    int64_t n = 0;
    if (gs2m_tsdf_extract_count(t, stream, &n)) return 1;  // (re)computes the per-block offsets
    if (n_triangles) *n_triangles = n;
    if (n == 0 || max_triangles == 0) return 0;
    int64_t nb = 0;
    if (gs2m_tsdf_status(t, stream, &nb, nullptr, nullptr)) return 1;
    gs2m_launch_mc_emit((hipStream_t)stream, t->V, t->d_mc, (unsigned)nb, t->d_blk_tris, (unsigned long long)max_triangles,
                        t->voxel_length, t->unit_length, vertices, colors, edge_index);
    return 0;
}

// ---- device-side mesh: extraction + welding, connected components ----------------------------------------------------------------
namespace {
// Grow-only scratch arena of the mesh passes: sub-buffers are carved out of ONE allocation that survives the call (round 5
// hipMalloc'ed / hipFree'd ten temporaries per call: bench `cluster_ms` 69 for 0.9 ms of kernels).
struct Arena {
    char** p;
    size_t* cap;
    size_t used = 0;
    Arena(char** p_, size_t* cap_) : p(p_), cap(cap_) {}
    static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    bool reserve(size_t bytes) {          // total of pad(size) over the buffers about to be taken
        used = 0;
        if (bytes <= *cap && *p) return true;
        (void)hipFree(*p);                // synchronises: nothing of an earlier call is in flight
        *p = nullptr;
        *cap = 0;
        const size_t n = bytes + bytes / 8 + 256;
        if (hipMalloc((void**)p, n) != hipSuccess) return false;
        *cap = n;
        return true;
    }
    template <typename T> T* take(size_t count) {
        T* r = reinterpret_cast<T*>(*p + used);
        used += pad(sizeof(T) * count);
        return r;
    }
};
unsigned pow2_at_least(unsigned long long n) {
    unsigned c = 1024u;
    while ((unsigned long long)c < n && c < 0x80000000u) c <<= 1;
    return c;
}
// gs2m_mesh_cluster has no handle: one arena per device, used under a lock (the call synchronises before it returns)
struct DeviceScratch {
    char* p = nullptr;
    size_t cap = 0;
};
DeviceScratch g_cluster_scratch[64];
std::mutex g_cluster_lock;
}  // namespace

// forget the cached mesh (its arrays stay allocated: grow-only, freed with the handle)
static void drop_mesh(gs2m_tsdf* t) { t->mesh_nv = t->mesh_nt = 0; }
template <typename T>
static bool grow(T** p, size_t have, size_t need) {     // caller updates its capacity on success
    if (need <= have && *p) return true;
    (void)hipFree(*p);
    *p = nullptr;
    return hipMalloc((void**)p, sizeof(T) * (need + need / 8 + 16)) == hipSuccess;
}

extern "C" int gs2m_tsdf_extract_mesh(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_vertices, int64_t* n_triangles) {
    if (!t || !n_vertices || !n_triangles) {
        gs2m_set_error("gs2m_tsdf_extract_mesh: NULL argument");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    drop_mesh(t);
    *n_vertices = *n_triangles = 0;
    int64_t nt = 0;
    if (gs2m_tsdf_extract_count(t, stream, &nt)) return 1;
    if (nt == 0) return 0;
    if (nt > 0x2aaaaaaa) {
        gs2m_set_error("gs2m_tsdf_extract_mesh: %lld triangles exceed the 32-bit vertex numbering", (long long)nt);
        return 1;
    }
    const unsigned n = (unsigned)(3 * nt);
    const unsigned cap = pow2_at_least(2ull * n);
    const unsigned m = (n + 4095u) / 4096u;
    Arena A(&t->mesh_scratch, &t->mesh_scratch_cap);
    const size_t need = 2 * Arena::pad(sizeof(double) * 3 * (size_t)n) + Arena::pad(sizeof(int) * 4 * (size_t)n) +
                        Arena::pad(sizeof(unsigned long long) * (size_t)cap) + Arena::pad(sizeof(unsigned) * (size_t)cap) +
                        3 * Arena::pad(sizeof(unsigned) * (size_t)n) + Arena::pad(sizeof(unsigned) * ((size_t)m + 2)) + Arena::pad(sizeof(int) * 4);
    const bool tri_ok = grow(&t->mesh_tri, t->mesh_cap_t, (size_t)n);
    if (tri_ok && (size_t)n > t->mesh_cap_t) t->mesh_cap_t = (size_t)n + (size_t)n / 8 + 16;
    if (!tri_ok || !A.reserve(need)) {
        gs2m_set_error("gs2m_tsdf_extract_mesh: out of device memory for %lld triangles", (long long)nt);
        if (!tri_ok) t->mesh_cap_t = 0;
        drop_mesh(t);
        return 1;
    }
    double* soup_v = A.take<double>(3 * (size_t)n);
    double* soup_c = A.take<double>(3 * (size_t)n);
    int* soup_e = A.take<int>(4 * (size_t)n);
    unsigned long long* hkeys = A.take<unsigned long long>(cap);
    unsigned* hfirst = A.take<unsigned>(cap);
    unsigned* cell_of = A.take<unsigned>(n);
    unsigned* flag = A.take<unsigned>(n);
    unsigned* pos = A.take<unsigned>(n);
    unsigned* scratch = A.take<unsigned>((size_t)m + 2);
    int* small = A.take<int>(4);
    int64_t nb = 0;
    if (gs2m_tsdf_status(t, stream, &nb, nullptr, nullptr)) return 1;
    gs2m_launch_mc_emit(st, t->V, t->d_mc, (unsigned)nb, t->d_blk_tris, (unsigned long long)nt, t->voxel_length, t->unit_length,
                        soup_v, soup_c, soup_e);
    const int init[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0};      // mins[3], bad
    HIPCHK(hipMemcpyAsync(small, init, sizeof(init), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(hkeys, 0xff, sizeof(unsigned long long) * (size_t)cap, st));
    HIPCHK(hipMemsetAsync(hfirst, 0xff, sizeof(unsigned) * (size_t)cap, st));
    // the compact arrays cannot be sized before the scan: first the flags, the scan and its total, then the (grow-only) arrays and the emit
    {
        int* mins = small;
        unsigned* bad = reinterpret_cast<unsigned*>(mins + 3);
        gs2m_launch_mesh_weld_count(st, n, soup_e, mins, hkeys, hfirst, cap, cell_of, flag, pos, scratch, bad);
        unsigned total = 0;
        int hb[4];
        HIPCHK(hipMemcpyAsync(&total, scratch + m, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(hb, small, sizeof(hb), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hb[3]) {
            gs2m_set_error("gs2m_tsdf_extract_mesh: the surface spans more than 2^20 voxels along an axis (62-bit weld key)");
            drop_mesh(t);
            return 1;
        }
        const size_t nv3 = 3 * (size_t)total, nv4 = 4 * (size_t)total;
        if (nv4 > t->mesh_cap_v) {
            const bool ok = grow(&t->mesh_v, 0, nv3) && grow(&t->mesh_c, 0, nv3) && grow(&t->mesh_e, 0, nv4);
            t->mesh_cap_v = ok ? nv4 : 0;
            if (!ok) {
                gs2m_set_error("gs2m_tsdf_extract_mesh: out of device memory for %u vertices", total);
                drop_mesh(t);
                return 1;
            }
        }
        gs2m_launch_mesh_weld_emit(st, n, hfirst, cell_of, pos, soup_v, t->V.has_color ? soup_c : nullptr, soup_e, t->mesh_v, t->mesh_c,
                                   t->mesh_e, t->mesh_tri);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        t->mesh_nv = (int64_t)total;
        t->mesh_nt = nt;
    }
    *n_vertices = t->mesh_nv;
    *n_triangles = t->mesh_nt;
    return 0;
}

extern "C" int gs2m_tsdf_mesh_copy(gs2m_tsdf* t, gs2m_stream stream, double* vertices, double* colors, int32_t* edge_index, int32_t* triangles) {
    if (!t) {
        gs2m_set_error("null handle");
        return 1;
    }
    HIPCHK(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    if (t->mesh_nv > 0) {
        if (vertices) HIPCHK(hipMemcpyAsync(vertices, t->mesh_v, sizeof(double) * 3 * (size_t)t->mesh_nv, hipMemcpyDefault, st));
        if (colors) HIPCHK(hipMemcpyAsync(colors, t->mesh_c, sizeof(double) * 3 * (size_t)t->mesh_nv, hipMemcpyDefault, st));
        if (edge_index) HIPCHK(hipMemcpyAsync(edge_index, t->mesh_e, sizeof(int) * 4 * (size_t)t->mesh_nv, hipMemcpyDefault, st));
    }
    if (t->mesh_nt > 0 && triangles) HIPCHK(hipMemcpyAsync(triangles, t->mesh_tri, sizeof(int) * 3 * (size_t)t->mesh_nt, hipMemcpyDefault, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int gs2m_mesh_cluster(int device, gs2m_stream stream, int64_t n_triangles, const int32_t* triangles, int32_t* labels,
                                 int64_t* cluster_n_triangles, int64_t* n_clusters) {
    if (n_triangles < 0 || n_triangles > 0x7fffffff || !n_clusters || (n_triangles > 0 && (!triangles || !labels || !cluster_n_triangles))) {
        gs2m_set_error("gs2m_mesh_cluster: bad argument");
        return 1;
    }
    *n_clusters = 0;
    if (n_triangles == 0) return 0;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const unsigned nt = (unsigned)n_triangles;
    const unsigned cap = pow2_at_least(6ull * nt);
    const unsigned m = (nt + 4095u) / 4096u;
    if (device < 0 || device >= 64) {
        gs2m_set_error("gs2m_mesh_cluster: device %d not in 0..63", device);
        return 1;
    }
    std::lock_guard<std::mutex> hold(g_cluster_lock);
    Arena A(&g_cluster_scratch[device].p, &g_cluster_scratch[device].cap);
    const size_t need = Arena::pad(sizeof(unsigned long long) * (size_t)cap) + Arena::pad(sizeof(unsigned) * (size_t)cap) +
                        4 * Arena::pad(sizeof(unsigned) * (size_t)nt) + Arena::pad(sizeof(unsigned) * ((size_t)m + 2));
    if (!A.reserve(need)) {
        gs2m_set_error("gs2m_mesh_cluster: out of device memory for %u triangles", nt);
        return 1;
    }
    unsigned long long* hkeys = A.take<unsigned long long>(cap);
    unsigned* hval = A.take<unsigned>(cap);
    unsigned* parent = A.take<unsigned>(nt);
    unsigned* root = A.take<unsigned>(nt);
    unsigned* flag = A.take<unsigned>(nt);
    unsigned* pos = A.take<unsigned>(nt);
    unsigned* scratch = A.take<unsigned>((size_t)m + 2);
    HIPCHK(hipMemsetAsync(hkeys, 0xff, sizeof(unsigned long long) * (size_t)cap, st));
    HIPCHK(hipMemsetAsync(hval, 0xff, sizeof(unsigned) * (size_t)cap, st));
    HIPCHK(hipMemsetAsync(cluster_n_triangles, 0, sizeof(int64_t) * (size_t)nt, st));
    gs2m_launch_mesh_cluster(st, triangles, nt, hkeys, hval, cap, parent, root, flag, pos, scratch, labels,
                             reinterpret_cast<unsigned long long*>(cluster_n_triangles));
    unsigned total = 0;
    HIPCHK(hipMemcpyAsync(&total, scratch + m, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    *n_clusters = (int64_t)total;
    return 0;
}
