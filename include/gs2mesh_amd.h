/*
 * gs2mesh_amd.h -- C ABI of the MI355X-native render -> fuse hot path of GS2Mesh.
 *
 * This is the drop-in boundary: a plain C shared library (libgs2mesh_amd.so, HIP/gfx950)
 * with raw device pointers, sizes and a hipStream_t.  No torch types cross it.  The
 * Python host side (package gs2mesh_amd) binds it with ctypes; INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Citations are file:line in the reference tree (/root/reference), with
 *   DGR/ = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization/
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; gs2m_last_error()
 *     returns a thread-local message for the last failure on the calling thread;
 *   - all `const float*` / `float*` / `uint8_t*` data arguments are DEVICE pointers
 *     unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued on it, nothing synchronises the device except the functions that say
 *     so (gs2m_raster_status, gs2m_tsdf_status, the *_download helpers);
 *   - handles own persistent, grow-only scratch arenas (the reference re-allocates
 *     its three byte arenas every call: DGR/rasterize_points.cu:27-33,73-78).  A
 *     handle may only be used from one stream at a time.
 */
#ifndef GS2MESH_AMD_H
#define GS2MESH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS2M_VERSION 600 /* 0.6.0: round-6 ABI = the round-4 ABI (401) + the round-5 entry points that 401 never counted
                            (gs2m_tsdf_block_map / _map_keys / _map_bytes / _replace / _extract_mesh / _mesh_copy,
                            gs2m_mesh_cluster, gs2m_raster_blend_cycles, GS2M_OPT_BLEND_MODE / _PROFILE) + round 6
                            (gs2m_tsdf_flags_device, GS2M_OPT_BIN_LANE_TILES, GS2M_OPT_PROJECT_SHARED_READ, GS2M_OPT_EXACT_TILE_CULL level 2; GS2M_OPT_BLEND_MODE 1
                            removed).  The Python binding checks it at load time */

typedef void* gs2m_stream; /* hipStream_t */

/* ------------------------------------------------------------------------------------ */
/* library                                                                              */
/* ------------------------------------------------------------------------------------ */

int gs2m_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* gs2m_last_error(void);

/* ------------------------------------------------------------------------------------ */
/* rasteriser                                                                           */
/* ------------------------------------------------------------------------------------ */

typedef struct gs2m_raster gs2m_raster;

/* Flags for gs2m_raster_set_option */
enum {
    GS2M_OPT_EXACT_TILE_CULL = 1, /* 0 = reference AABB-of-3-sigma-circle instance list
                                     (DGR/cuda_rasterizer/auxiliary.h:46-56), 1 = also drop
                                     (Gaussian,tile) instances whose alpha < 1/255 on the
                                     whole tile (image unchanged, num_rendered smaller);
                                     2 (round 6) = the same, but rects of at most 4 binning tiles
                                     keep all their tiles (the per-tile test removes ~2 % of the
                                     instances of a small-splat scene and was 60 % of the vector
                                     instructions of the counting kernel)                 */
    GS2M_OPT_BLEND_VARIANT = 2,   /* compositing kernel: 4 (default) = one wave per 16x16 tile, 4 pixels per lane;
                                     0 = the reference's structure (16x16 tile per 256-thread workgroup,
                                     1 px/lane; GS2M_OPT_TILE_ROWS 1 only).  Same image (to rounding).
                                     (7, exponents on the matrix cores, was removed in round 3: slower than 4
                                     and wrong on strongly anisotropic splats -- DESIGN.md 3.) */
    GS2M_OPT_DEBUG_SYNC = 3,      /* 1 = synchronise + check after every launch (the
                                     reference's `debug`: auxiliary.h:166-173)            */
    GS2M_OPT_STAGE_TIMING = 4,    /* 1 = bracket every stage launch with hipEvents on the
                                     work stream (read with gs2m_raster_stage_times)      */
    GS2M_OPT_PAIR_BATCH = 8,      /* gs2m_render_views: up to `value` (2 .. 4; 0 / 1 = off, the default) stereo pairs share every launch
                                     of a pass, as far as the call has them (the projection / counting / scatter workgroups of a pair are
                                     1 / pairs as many, blockIdx.y picks the pair; scans, per-tile sort and compositing take all the views
                                     in one grid): the per-launch and per-workgroup fixed costs are paid once, and the compositing grid
                                     is long against its tail.  Same results. */
    /* Tuning options (results never change; defaults are the measured optima, profiles/r3_experiments.txt): */
    GS2M_OPT_BIN_WORKGROUPS = 9,  /* workgroups of the counting / scatter kernels per stereo pair (default 0 = 256, one per CU: the
                                     per-workgroup histogram rows / cursors scale with their number) */
    GS2M_OPT_BIN_WG_THREADS = 10, /* upper bound of their threads per workgroup, a multiple of 64 (default 0 = 1024) */
    GS2M_OPT_BLEND_MODE = 11,     /* compositing loop of variant 4 (three forms of the same arithmetic, bit-identical images):
                                     2 (default, round 5) = every staged instance evaluates all four 8x8 quadrants, candidate
                                     test = one v_cmp against the lane's own threshold, accumulate under the execution mask,
                                     the alpha-cap / power > 0 instances split out per run of the staged batch;
                                     3 (round 6) = 2 with min(0.99, alpha) applied to EVERY instance (the identity where the cap
                                     cannot bind) instead of splitting the runs at the instances whose opacity exceeds 0.98: for
                                     models with many saturated opacities (a trained splat: ~30 %; -10 % compositing time there,
                                     +3.6 % on a model with 2 %) -- rasterizer.auto_blend_mode picks by the model's share;
                                     0 = per-pixel decisions as lane masks in scalar registers + per-instance quadrant mask
                                     (the loop of rounds 1-4, kept as the cross-check).  (1, the execution-mask form of 0, was
                                     never the fastest anywhere and was removed in round 6.) */
    GS2M_OPT_BLEND_PROFILE = 12,  /* 1 = launch the s_memtime-instrumented build of the compositing kernel (mode 2 only) and sum
                                     its per-wave phase cycles in the handle; read with gs2m_raster_blend_cycles */
    GS2M_OPT_BIN_LANE_TILES = 13, /* tuning (results never change): a tile rect of at most `value` binning tiles (0 .. 16, default 4) is
                                     walked by its own lane in the counting / scatter kernels -- exact tile test in registers, the kept
                                     tiles as a small bit mask -- instead of through the wave-balanced staged walk; 0 = only the thin
                                     rects of round 4 (one tile wide or high, <= 4 tiles) */
    GS2M_OPT_PROJECT_SHARED_READ = 14, /* tuning (results never change): the stereo pairs of a launch (GS2M_OPT_PAIR_BATCH) share ONE read of
                                     the model in the projection kernel -- the thread that owns a Gaussian projects it for every pair of
                                     the launch -- instead of one grid row per pair.  0 (default) = with a spatially ordered packed model
                                     (gs2m_raster_pack_model) or from 1 000 000 Gaussians (a small unordered model re-reads from the
                                     last-level cache and prefers twice the waves), 1 = always, 2 = never */
    GS2M_OPT_TILE_ROWS = 5        /* binning tile = 16 x (16 * rows) pixels.  1 (default) = the reference's 16 x 16
                                     tiles: instance lists / num_rendered are the reference's.  2 = two reference
                                     tiles stacked: ~30 % fewer (Gaussian, tile) instances to count, scatter and
                                     sort; two waves share a list, each compositing its 16 x 16 half; same image -- the reference's
                                     16 x 16 tile rect still bounds every contribution.  The binning taps then
                                     describe the 16 x 32 tiles.                                            */
};

/* Stage order of gs2m_raster_stage_times */
enum {
    GS2M_STAGE_PROJECT = 0, /* k_project       */
    GS2M_STAGE_COLSCAN = 1, /* k_hist_colscan  */
    GS2M_STAGE_TILESCAN = 2,/* k_tile_scan     */
    GS2M_STAGE_SCATTER = 3, /* k_scatter       */
    GS2M_STAGE_SORT = 4,    /* k_sort_tiles    */
    GS2M_STAGE_BLEND = 5,   /* k_blend_*       */
    GS2M_STAGE_COUNT = 6,   /* k_count_tiles (runs between PROJECT and COLSCAN) */
    GS2M_N_STAGES = 7
};

int gs2m_raster_create(gs2m_raster** out, int device);
int gs2m_raster_destroy(gs2m_raster* r);
int gs2m_raster_set_option(gs2m_raster* r, int option, int value);

/* Pre-size the arenas (optional; every forward grows them on demand).
 * P Gaussians, n_views views of W x H rendered per call, `instances` (Gaussian,tile)
 * pairs per view (the reference's num_rendered). */
int gs2m_raster_reserve(gs2m_raster* r, int P, int n_views, int W, int H, int64_t instances);

/*
 * Operator-level entry point.  Replaces CudaRasterizer::Rasterizer::forward
 * (DGR/cuda_rasterizer/rasterizer.h:35-60, rasterizer_impl.cu:198-336) as bound by
 * RasterizeGaussiansCUDA (DGR/rasterize_points.cu:35-115): same argument meaning and order,
 * minus the three std::function arena callbacks (arenas live in the handle) plus the
 * handle and the stream.
 *
 *   P, D, M           #Gaussians, active SH degree (0..3), #SH coefficients per channel
 *   background[3]     device
 *   means3D[P,3]      device
 *   shs[P,M,3]        device or NULL (then colors_precomp must be given)
 *   colors_precomp[P,3] device or NULL
 *   opacities[P]      device (activated)
 *   scales[P,3], rotations[P,4] (wxyz)  device (activated) or NULL (then cov3D_precomp)
 *   cov3D_precomp[P,6] device or NULL
 *   viewmatrix[16], projmatrix[16], cam_pos[3]   device, the reference's transposed
 *                     row-major layout (GS/scene/cameras.py:54-57)
 *   out_color[3,H,W]  device, written completely
 *   radii[P]          device or NULL
 *   debug             as the reference: sync + check after every launch
 *
 * Asynchronous.  If the instance arena was too small the image is NOT valid; query
 * gs2m_raster_status() (which synchronises the stream), gs2m_raster_reserve() and
 * call again -- the Python binding does exactly that.
 */
int gs2m_rasterize_forward(gs2m_raster* r, int P, int D, int M, const float* background,
                           int width, int height, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations,
                           const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, float tan_fovx,
                           float tan_fovy, int prefiltered, float* out_color, int* radii,
                           int debug, gs2m_stream stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:26-31,
 * rasterizer_impl.cu:54-66,141-153).  present[P]: device, 1 byte per Gaussian. */
int gs2m_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, gs2m_stream stream);

/* Host-side camera record = the per-view fields of GaussianRasterizationSettings
 * (DGR/diff_gaussian_rasterization/__init__.py:157-169).  Matrices in the same
 * transposed row-major layout the reference passes. */
typedef struct gs2m_camera {
    int32_t width, height;
    float tanfovx, tanfovy;
    float viewmatrix[16];
    float projmatrix[16];
    float campos[3];
} gs2m_camera;

/* Gaussian store handed to the pipeline-level entry point.  `raw` != 0 means the arrays
 * hold the GaussianModel's pre-activation parameters (GS/scene/gaussian_model.py:95-115):
 * log-scales, unnormalised quaternions, logit opacities; exp / normalize / sigmoid are
 * fused into the projection kernel.  shs is [P,M,3] (get_features layout) or, when
 * shs_rest != NULL, shs = features_dc [P,1,3] and shs_rest = features_rest [P,M-1,3]
 * (saves the torch.cat of GS/scene/gaussian_model.py:108-111). */
typedef struct gs2m_gaussians {
    int32_t P, sh_degree, M, raw;
    const float* xyz;       /* [P,3] */
    const float* scales;    /* [P,3] */
    const float* rotations; /* [P,4] wxyz */
    const float* opacities; /* [P]   */
    const float* shs;       /* [P,M,3] or features_dc [P,1,3] */
    const float* shs_rest;  /* NULL or features_rest [P,M-1,3] */
} gs2m_gaussians;

/*
 * Pipeline-level entry point: renders n_views views (a stereo pair = 2) of the same
 * Gaussians in one fused pass.  Replaces the per-eye loop of Renderer.render_image_pair
 * (gs2mesh_utils/renderer_utils.py:378-389) -> render() (GS/gaussian_renderer/__init__.py:18-100)
 * -> rasterizer.  cams and bg are HOST pointers (copied at call time).
 *   out_color  [n_views,3,H,W] f32 device, or NULL
 *   out_rgb8   [n_views,H,W,3] u8 device, or NULL: saturate(rint(255*c)), the conversion
 *              cv2.imwrite applies to the float image (renderer_utils.py:389-390)
 *   out_radii  [n_views,P] i32 device, or NULL
 * All views must share width/height.  Asynchronous; same arena/overflow contract as
 * gs2m_rasterize_forward.
 *
 * STATED TOLERANCE against the reference rasteriser (DGR/cuda_rasterizer/forward.cu:261-374) on identical inputs; the same
 * statement holds for gs2m_rasterize_forward.  Projection, instance lists and sort order are exact (record, point_list,
 * ranges, radii, num_rendered bit-identical at the defaults).  The compositing stage evaluates alpha in the log2 domain
 * (v_exp_f32, 1 ulp) and accumulates rgb * (T - T'), so:
 *   - on every pixel where NO decision of renderCUDA (power > 0, alpha < 1/255, T' < 1e-4) is taken within a relative
 *     band of 1e-5 of its threshold, |out_color - reference| <= 2e-4 per value ("clean bar");
 *     measured <= 5.5e-7 on C2 .. C5 (profiles/r4_parity_*.json);
 *   - elsewhere a decision may flip; the difference is then bounded by 2e-4 + the contribution of the flipped
 *     instances (oracle/parity.py:flip_attribution computes that bound per pixel; the tests assert zero unexplained
 *     pixels).  Measured global maxima: 2.0e-3 on C2 (25 k of 3.84 M pixels carry a candidate flip), 3.8e-4 on C3,
 *     3.8e-3 on C4, 6.2e-4 on C5; the tests bound them by <= 2x measured (tests/test_fullsize_gpu.py:BENCH_PARITY_BOUNDS);
 *   - out_rgb8 differs from the quantised reference by at most 1 LSB, on <= 70 pixels of a C2 eye;
 *   - with g->raw != 0 (activations fused into the projection: exp / sigmoid / normalize in device arithmetic, 1 ulp from
 *     numpy's) about 1 radius in 3e5 Gaussians moves by one pixel and one (Gaussian, tile) instance with it; the
 *     bench line and the tests report the count (`radii_mismatches`), the image bound above already includes it.
 */
int gs2m_render_views(gs2m_raster* r, const gs2m_gaussians* g, const gs2m_camera* cams,
                      int n_views, const float* bg /* host[3] */, float scale_modifier,
                      float* out_color, uint8_t* out_rgb8, int* out_radii, gs2m_stream stream);

/*
 * Optional one-time preparation for gs2m_render_views (Renderer.prepare_renderer stage): keeps a
 * wave-transposed copy of the SH block ([P/64][12][64] float4, +192 B per Gaussian) inside the handle.
 * Later gs2m_render_views calls whose gs->shs / shs_rest / P match read that copy with fully
 * coalesced loads; the caller's arrays must not change in between (call again after an update;
 * a call with P == 0 or M != 16 just drops the copy).  Same results either way.
 */
int gs2m_raster_pack_sh(gs2m_raster* r, const gs2m_gaussians* g, gs2m_stream stream);

/*
 * Superset of gs2m_raster_pack_sh (same stage, same matching rule, extended to xyz / scales / rotations /
 * opacities): keeps a SPATIALLY ORDERED packed copy of the whole model inside the handle (+236 B per Gaussian).
 * order[P] (device, int32) is a permutation: position j of the copy holds Gaussian order[j] -- e.g. the Morton
 * order of xyz (gs2mesh_amd.rasterizer.morton_order).  A trained splat is stored in densification order
 * (GS/scene/gaussian_model.py:372-409 appends clones and splits), i.e. spatially random: every workgroup of the
 * counting sort then touches every tile.  In a spatially ordered copy consecutive Gaussians project to a few
 * neighbouring tiles: the keys a workgroup scatters form long runs and the records a tile gathers lie close together.
 * Results are those of the unordered model: the sort keys carry the Gaussian IDS (ties in depth resolve in id order,
 * as the reference's stable sort does), out_radii and the parity taps are indexed by id.  Synchronises `stream` once
 * (the permutation is verified); returns an error if `order` is not a permutation.
 */
int gs2m_raster_pack_model(gs2m_raster* r, const gs2m_gaussians* g, const int32_t* order /* device [P] */,
                           gs2m_stream stream);

/* The packed copies are matched to the caller's arrays by DEVICE POINTER (and P) only: after an in-place update of any
 * Gaussian parameter, or when tensors may have been freed and re-allocated at the same addresses, call this (or pack
 * again) -- later gs2m_render_views calls then read the caller's arrays until the next pack. */
int gs2m_raster_pack_invalidate(gs2m_raster* r);

/* Synchronises `stream` (pass the stream the handle is used on) and reports num_rendered[v] for
 * v < n_views of the LAST forward/render_views call on the handle (host array, may be NULL), and
 * whether the instance arena overflowed in ANY call since the previous status query
 * (*overflow = 1: the images of the overflowing calls are invalid; *required = the largest
 * per-view instance count any of those calls needed).  The overflow word is sticky on the device:
 * a later call that fits does not erase it; this query consumes it. */
int gs2m_raster_status(gs2m_raster* r, gs2m_stream stream, int n_views,
                       int64_t* num_rendered, int* overflow, int64_t* required);

/* With GS2M_OPT_STAGE_TIMING on: synchronises `stream`, then adds the hipEvent-measured
 * GPU time of every stage launch recorded since the last query to total_ms[GS2M_N_STAGES]
 * and the number of launches to launches[GS2M_N_STAGES] (host arrays, caller-zeroed). */
int gs2m_raster_stage_times(gs2m_raster* r, gs2m_stream stream, double* total_ms,
                            int64_t* launches);

/* Phase profile of the compositing kernel (GS2M_OPT_BLEND_PROFILE 1): synchronises `stream`, copies the counters summed
 * over every wave launched since the last query to counters[GS2M_BLEND_PROF_COUNTERS] (host) and clears them:
 *   [0] waves  [1] wave lifetime  [2] waiting for the record DMA  [3] staging a batch (transform, cull, compaction)
 *   [4] issuing the next batch's loads  [5] compositing loop  [6] epilogue (background, image stores)   -- shader cycles
 *   [7] staged batches  [8] staged instances (after the per-half / finished-quadrant cull)  [9] listed instances
 * The stamps themselves cost ~10 % of the wave's cycles: read the counters as shares. */
#define GS2M_BLEND_PROF_COUNTERS 10
int gs2m_raster_blend_cycles(gs2m_raster* r, gs2m_stream stream, uint64_t* counters);

/* Debug/parity taps: copy the projected per-Gaussian record of view `v` of the last call to
 * HOST buffers (any may be NULL): means2D[P,2], depths[P], conic_opacity[P,4], rgb[P,3],
 * rect[P,4] (u16 x0,y0,x1,y1), tiles_touched[P].  Synchronises. */
int gs2m_raster_download_geometry(gs2m_raster* r, gs2m_stream stream, int v, int P,
                                  float* means2D, float* depths, float* conic_opacity,
                                  float* rgb, uint16_t* rect, uint32_t* tiles_touched);
/* Copy view v's sorted instance list and tile ranges to HOST: point_list[n] (Gaussian ids,
 * n = num_rendered[v]), ranges[tiles,2].  Synchronises. */
int gs2m_raster_download_binning(gs2m_raster* r, gs2m_stream stream, int v, int64_t n,
                                 uint32_t* point_list, int32_t n_tiles, uint32_t* ranges);

/* ------------------------------------------------------------------------------------ */
/* TSDF fusion                                                                          */
/* ------------------------------------------------------------------------------------ */

typedef struct gs2m_tsdf gs2m_tsdf;

enum { GS2M_TSDF_COLOR_NONE = 0, GS2M_TSDF_COLOR_RGB8 = 1 };

/*
 * Replaces open3d.pipelines.integration.ScalableTSDFVolume(voxel_length, sdf_trunc,
 * color_type[, volume_unit_resolution=16, depth_sampling_stride=4]) as constructed at
 * gs2mesh_utils/tsdf_utils.py:53-56 (Open3D 0.17.0, requirements.txt:15; the C++ is not
 * in the reference tree -- algorithm restated in oracle/tsdf_oracle.cpp).
 * max_blocks = capacity of the 16^3-voxel block pool (20 B/voxel = 80 KiB per block).
 */
int gs2m_tsdf_create(gs2m_tsdf** out, double voxel_length, double sdf_trunc, int color_type,
                     int volume_unit_resolution, int depth_sampling_stride, int64_t max_blocks,
                     int device);
int gs2m_tsdf_destroy(gs2m_tsdf* t);
int gs2m_tsdf_reset(gs2m_tsdf* t, gs2m_stream stream);

/*
 * Replaces RGBDImage.create_from_color_and_depth(color, depth, depth_scale, depth_trunc,
 * convert_rgb_to_intensity=False) + PinholeCameraIntrinsic(w,h,fx,fy,cx,cy) +
 * volume.integrate(rgbd, intrinsic, extrinsic)  (tsdf_utils.py:88-93,106-107).
 *   depth   [H,W] f32 device (raw: the kernel applies d/depth_scale, d >= depth_trunc -> 0)
 *   color   [H,W,3] u8 device (may be NULL when color_type == NONE)
 *   mask    [H,W] u8 device or NULL: depth is multiplied by (mask != 0) first
 *           (tsdf_utils.py:68-81 object/occlusion masks)
 *   min_depth: depth < min_depth -> 0 before scaling (tsdf_utils.py:83); pass 0 to skip
 *   extrinsic_w2c: HOST, row-major 4x4 float64 world->camera (what the reference passes:
 *           np.linalg.inv(extrinsic_matrix))
 * Asynchronous.
 */
int gs2m_tsdf_integrate(gs2m_tsdf* t, const float* depth, const uint8_t* color,
                        const uint8_t* mask, int width, int height, double fx, double fy,
                        double cx, double cy, const double* extrinsic_w2c, double depth_scale,
                        double depth_trunc, double min_depth, gs2m_stream stream);

/*
 * The same for n_frames frames of one size / intrinsics in ONE sweep over the touched blocks (SURVEY.md 7 step 6 iii):
 * every block keeps its voxels in registers while the frames that touch it are applied IN FRAME ORDER, so the result is
 * bit-identical to n_frames gs2m_tsdf_integrate calls in that order, with one state read + write per batch instead of
 * per frame.  depth / color / mask: HOST arrays of n_frames DEVICE pointers (mask may be NULL, or hold NULLs);
 * extrinsics_w2c: HOST [n_frames][16].  Batches of up to 64 frames per sweep (longer lists are split).  Asynchronous; the
 * images must stay valid until the stream reaches the end of the call.
 */
int gs2m_tsdf_integrate_batch(gs2m_tsdf* t, int n_frames, const float* const* depth, const uint8_t* const* color,
                              const uint8_t* const* mask, int width, int height, double fx, double fy, double cx,
                              double cy, const double* extrinsics_w2c, double depth_scale, double depth_trunc,
                              double min_depth, gs2m_stream stream);

/* Stage timing as for the rasteriser: enable != 0 brackets k_tsdf_touch (index 0) and
 * k_tsdf_integrate (index 1) with hipEvents; gs2m_tsdf_stage_times synchronises and
 * accumulates into total_ms[2] / launches[2]. */
int gs2m_tsdf_set_stage_timing(gs2m_tsdf* t, int enable);   /* a batch counts as n_frames launches of each stage */
int gs2m_tsdf_stage_times(gs2m_tsdf* t, gs2m_stream stream, double* total_ms, int64_t* launches);

/* Synchronises; n_blocks = allocated blocks, block_updates = sum over frames of blocks
 * integrated (x 4096 = voxel-updates), overflow = flag bits (0 = fine): 1 block pool exhausted, 2 hash table full,
 * 4 block index outside the +-2^20 key range, 8 a voxel handed to gs2m_tsdf_pack(GS2M_XFORM_SUM_PACKED) did not fit the
 * packed fields (weight > 1023 or a colour sum >= 2^18: the summed exchange buffers are invalid, use SUM_F32). */
int gs2m_tsdf_status(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_blocks,
                     int64_t* block_updates, int* overflow);

/* The overflow flag word of gs2m_tsdf_status (bits 1 / 2 / 4 / 8) copied to a DEVICE word, asynchronously: lets a multi-GPU
 * caller all-reduce the verdict of a pack on the device and pay ONE host read for it (gs2mesh_amd.parallel.reduce_volume)
 * instead of a status synchronisation followed by the read of the reduced flag. */
int gs2m_tsdf_flags_device(gs2m_tsdf* t, uint32_t* flags_dev, gs2m_stream stream);

/* Copy allocated blocks to HOST (any pointer may be NULL): keys[n,3] (block index),
 * tsdf[n,4096], weight[n,4096] (voxel x*256+y*16+z, Open3D IndexOf), rgb_sum[n,4096,3] u32
 * (sum of u8 colours; mean colour = rgb_sum/weight).  n must be >= n_blocks. Synchronises. */
int gs2m_tsdf_download(gs2m_tsdf* t, gs2m_stream stream, int64_t n, int32_t* keys,
                       float* tsdf, float* weight, uint32_t* rgb_sum);

/* Block keys of the first n allocated blocks to a DEVICE buffer keys[n,3] (i32), in slot order.  Halo copies (blocks of
 * another rank brought in by gs2m_tsdf_unpack(..., halo = 1)) are reported as the out-of-range sentinel key
 * (2^20 - 1 in every component): they are not this volume's to exchange. */
int gs2m_tsdf_block_keys(gs2m_tsdf* t, int64_t n, int32_t* keys, gs2m_stream stream);

/*
 * Multi-GPU exchange (new; the reference is single-GPU).  pack_sum: for the n canonical block keys (device [n,3])
 * write this volume's accumulators in SUM form to ONE device buffer buf[n,5,4096] of fp32 planes
 * {wsum = tsdf*weight, weight, sum r, sum g, sum b} (zeros where the block is not allocated here).  Counts and colour
 * sums are integers < 2^24, exact in fp32: the host reduces the whole buffer with a single RCCL collective
 * (reduce-scatter or all-reduce), then unpack_sum replaces the state of the listed blocks (allocating as needed)
 * with tsdf = wsum/weight.  halo != 0 marks the blocks as neighbour-only: gs2m_tsdf_extract reads them for the +1
 * corners of its cubes but starts no cube in them (the rank that owns them does): owner-side mesh extraction.
 */
int gs2m_tsdf_pack_sum(gs2m_tsdf* t, const int32_t* keys, int64_t n, float* buf, gs2m_stream stream);
int gs2m_tsdf_unpack_sum(gs2m_tsdf* t, const int32_t* keys, int64_t n, const float* buf, int halo,
                         gs2m_stream stream);

/*
 * Block-map key exchange (multi-GPU): the union of the ranks' block sets as a dense map over a WINDOW of
 * block indices lo[k] <= b[k] < lo[k] + dim[k] (host int32[3] each; the same on every rank).  SURVEY asks for a bitmap under a
 * bitwise OR; RCCL has no bitwise reduction, so the map holds one BYTE per block and the caller's collective is MAX over uint8.
 *   gs2m_tsdf_map_bytes    bytes of the exchange buffer: the cells (dim0 * dim1 * dim2 rounded up to 16)
 *                          + GS2M_TSDF_MAP_HEADER_BYTES + 8 * world header bytes (-1: bad window).
 *   gs2m_tsdf_block_map    clears `cells` (device, 16-byte aligned) and marks this rank's blocks (halo copies are not its
 *                          blocks); the header behind the cells carries, in a form a bytewise MAX reduces: [0] a block outside the
 *                          window, [1..4] the volume's overflow flags 1 / 2 / 4 / 8, [5] / [6] bits 0 / 1 of `flags` (caller-defined),
 *                          [8..11] / [12..15] the bytes of max_blocks and of its complement (byte + complement byte == 255 after
 *                          the MAX <=> the ranks agree), [16..23] the same for a hash of the window, [32 + 8 r ..] frames_local and
 *                          frames_base of rank r as little-endian u32 (zeros in the other ranks' slots).  Asynchronous, no host
 *                          read.  The caller all-reduces the buffer with MAX.
 *   gs2m_tsdf_map_keys     on the reduced buffer: marked cells -> keys[n][3] (device) in cell order = ascending (x, y, z), the
 *                          canonical order of the exchange, identical on every rank; header bytes [24..27] = n (keys beyond
 *                          max_keys are counted, not written); copies the header to header_host[GS2M_TSDF_MAP_HEADER_BYTES + 8 * world]
 *                          and synchronises -- the one host read of the key exchange.
 * Replaces round 4's "gather every rank's key list, sort, unique" (gs2mesh_amd/parallel.py keeps it as the fallback for blocks
 * outside the window).
 */
#define GS2M_TSDF_MAP_HEADER_BYTES 32
int64_t gs2m_tsdf_map_bytes(const int32_t* dim, int world);
int gs2m_tsdf_block_map(gs2m_tsdf* t, const int32_t* lo, const int32_t* dim, int rank, int world, int64_t frames_local,
                        int64_t frames_base, int flags, uint8_t* cells, gs2m_stream stream);
int gs2m_tsdf_map_keys(gs2m_tsdf* t, const int32_t* lo, const int32_t* dim, int world, uint8_t* cells, int32_t* keys,
                       int64_t max_keys, uint8_t* header_host, gs2m_stream stream);

/* The same with a choice of exchange form (pack_sum / unpack_sum = form GS2M_XFORM_SUM_F32):
 *   GS2M_XFORM_SUM_F32     buf_f32[n,5,4096] in SUM form as above (one fp32 SUM collective), buf_i64 unused;
 *   GS2M_XFORM_RAW_F32     buf_f32[n,5,4096] with the planes verbatim {tsdf, weight, sum r, sum g, sum b}: for copies of blocks
 *                          that are already reduced (halo exchange) -- no tsdf * w / w round trip, seam voxels bit-identical;
 *   GS2M_XFORM_SUM_PACKED  buf_f32[n,4096] = wsum, buf_i64[n,4096] = weight | sum r << 10 | sum g << 28 | sum b << 46: an fp32 SUM
 *                          and an int64 SUM collective, 12 instead of 20 bytes per voxel.  Only valid while the volumes being
 *                          summed have integrated <= 1023 frames IN TOTAL (weight < 2^10, colour sums < 2^18: no carry between
 *                          the fields); the caller checks that bound (gs2mesh_amd.parallel does, and falls back to SUM_F32);
 *                          a LOCAL value that does not fit its field raises status bit 8 (gs2m_tsdf_status).
 * Halo copies held by the volume are packed as zeros in the SUM forms (they are another rank's blocks). */
enum { GS2M_XFORM_SUM_F32 = 0, GS2M_XFORM_RAW_F32 = 1, GS2M_XFORM_SUM_PACKED = 2 };
int gs2m_tsdf_pack(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, float* buf_f32, int64_t* buf_i64,
                   gs2m_stream stream);
int gs2m_tsdf_unpack(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, const float* buf_f32,
                     const int64_t* buf_i64, int halo, gs2m_stream stream);
/* gs2m_tsdf_reset + gs2m_tsdf_unpack(halo = 0) of n DISTINCT in-range keys in one call, without writing the reduced blocks' bytes
 * twice: the reset leaves the voxel state of the first n slots alone (the unpack overwrites exactly those), and clears the rest
 * of the slots that were in use.  The tail of the multi-GPU reduction (gs2mesh_amd.parallel.reduce_volume). */
int gs2m_tsdf_replace(gs2m_tsdf* t, const int32_t* keys, int64_t n, int form, const float* buf_f32, const int64_t* buf_i64,
                      gs2m_stream stream);

/*
 * Replaces volume.extract_triangle_mesh() (tsdf_utils.py:108; Open3D ScalableTSDFVolume::
 * ExtractTriangleMesh): marching cubes over the allocated blocks; cubes with a zero-weight corner are
 * skipped, inside = tsdf < 0, vertices interpolated on the cut edges, vertex colour = interpolated
 * mean colour / 255.  gs2m_tsdf_extract_count synchronises and returns the triangle count;
 * gs2m_tsdf_extract writes min(count, max_triangles) un-welded triangles to DEVICE buffers
 * vertices[n,3,3] and colors[n,3,3] (float64, colors may be NULL), asynchronously after its
 * own count pass.  Vertices shared by neighbouring triangles are bit-identical (weld on the host).
 */
int gs2m_tsdf_extract_count(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_triangles);
int gs2m_tsdf_extract(gs2m_tsdf* t, gs2m_stream stream, int64_t max_triangles, double* vertices,
                      double* colors, int64_t* n_triangles);
/* The same + edge_index[n,3,4] (int32, DEVICE, may be NULL): per emitted vertex Open3D's vertex key -- the global voxel
 * index of the lower corner of the cut edge and the edge's axis (ExtractTriangleMesh's `edge_index`).  Welding by this key
 * instead of by position reproduces Open3D's vertex set also where a tsdf value is exactly 0 (the vertices of up to
 * three cut edges then share one position and Open3D keeps them apart). */
int gs2m_tsdf_extract_indexed(gs2m_tsdf* t, gs2m_stream stream, int64_t max_triangles, double* vertices,
                              double* colors, int32_t* edge_index, int64_t* n_triangles);

/*
 * Device-side mesh post-processing.  volume.extract_triangle_mesh() (gs2mesh_utils/tsdf_utils.py:108) returns
 * an INDEXED mesh: Open3D welds the marching-cubes vertices through an edge -> vertex map while it extracts.
 *   gs2m_tsdf_extract_mesh   count + emit + weld on the device: the vertices are welded by Open3D's vertex identity (the cut edge:
 *                            global voxel index of its lower corner + axis), numbered by first appearance in the deterministic
 *                            emission order of gs2m_tsdf_extract_indexed (= what welding that soup on the host gives).  The mesh stays
 *                            in the handle; *n_vertices / *n_triangles size the caller's buffers.  Synchronises.
 *   gs2m_tsdf_mesh_copy      copies the cached mesh to the caller: vertices [nv][3] f64, colors [nv][3] f64 (zeros for a colourless
 *                            volume), edge_index [nv][4] i32, triangles [nt][3] i32.  Each destination may be NULL and may be a
 *                            HOST or a device pointer (the PLY writer wants it on the host; the clustering below on the device).
 *   gs2m_mesh_cluster        TriangleMesh::ClusterConnectedTriangles (tsdf_utils.py:133): triangles are connected when they share an
 *                            edge.  triangles [n][3] i32, labels [n] i32 and cluster_n_triangles [n] i64 (capacity n; the first
 *                            *n_clusters entries are written) are DEVICE pointers.  Clusters are numbered by their smallest triangle
 *                            index (the numbering of a sweep from triangle 0 upwards).  Lock-free union-find over an edge hash table.
 */
int gs2m_tsdf_extract_mesh(gs2m_tsdf* t, gs2m_stream stream, int64_t* n_vertices, int64_t* n_triangles);
int gs2m_tsdf_mesh_copy(gs2m_tsdf* t, gs2m_stream stream, double* vertices, double* colors, int32_t* edge_index, int32_t* triangles);
int gs2m_mesh_cluster(int device, gs2m_stream stream, int64_t n_triangles, const int32_t* triangles, int32_t* labels,
                      int64_t* cluster_n_triangles, int64_t* n_clusters);

/* ------------------------------------------------------------------------------------ */
/* stereo post-processing (between the stereo network and the TSDF)                     */
/* ------------------------------------------------------------------------------------ */

/*
 * Replaces Stereo.get_occlusion_mask(disparity_LR, disparity_RL, stereo_occlusion_threshold) and
 * depth = fx * baseline / disparity_LR (gs2mesh_utils/stereo_utils.py:132-133,149-179), fused.
 *   disp_lr, disp_rl  [H,W] f32 device (disp_rl may be NULL when mask_out is NULL)
 *   depth_out         [H,W] f32 device or NULL
 *   mask_out          [H,W] u8 device or NULL: 1 = visible (the reference returns ~occlusion_mask)
 * The outputs are the `depth` / `mask` inputs of gs2m_tsdf_integrate.  Asynchronous.
 */
int gs2m_stereo_depth_occlusion(const float* disp_lr, const float* disp_rl, int width, int height,
                                double fx_times_baseline, double occlusion_threshold, float* depth_out,
                                uint8_t* mask_out, gs2m_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* GS2MESH_AMD_H */
