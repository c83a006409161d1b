// raster_api.hip -- C ABI of the rasteriser (include/gs2mesh_amd.h): handles, grow-only arenas,
// stage sequencing.  Host code only; kernels live in raster_{project,bin,blend}.hip.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/gs2mesh_amd.h"
#include "raster_internal.h"
#include "roctx_ranges.h"

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local std::string g_err;
void gs2m_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
extern "C" const char* gs2m_last_error(void) { return g_err.c_str(); }
extern "C" int gs2m_version(void) { return GS2M_VERSION; }

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            gs2m_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

#define GS2M_SORT_CLASSES_API 3   // = GS2M_SORT_CLASSES (raster_sort.h)
#define GS2M_MAX_STATUS 64  // views per gs2m_render_views call whose status is kept

struct gs2m_raster {
    int device = 0;
    int opt_exact_cull = 0, opt_blend = 4, opt_debug = 0, opt_timing = 0, opt_tile_rows = 1, opt_pair_batch = 0;
    int opt_bin_workgroups = 256, opt_bin_wg_threads = 1024, opt_blend_mode = 2;   // tuning options (gs2m_raster_set_option)
    int opt_blend_profile = 0;
    int opt_project_shared = 0;     // GS2M_OPT_PROJECT_SHARED_READ: 0 auto (by model size), 1 always, 2 never
    int opt_bin_lane_tiles = 4;     // rects of at most this many binning tiles are walked by their own lane (round 6)
    unsigned long long* d_blend_prof = nullptr;   // [GS2M_BLEND_PROF_COUNTERS] phase-cycle sums of the profile build (GS2M_OPT_BLEND_PROFILE)
    struct EvPair {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<EvPair> ev_live;        // recorded, not yet read
    std::vector<hipEvent_t> ev_free;    // recycled events
    CamUniform* d_cams = nullptr;  // [GS2M_MAX_PASS_VIEWS]
    float4* d_recs = nullptr;   // [3 * recs_cap] float4: the 32-B parts of recs_cap records, then their 16-B parts (GeomRecs)
    size_t recs_cap = 0;  // float4s
    unsigned long long* d_tilemask = nullptr;
    float* d_shpack = nullptr;   // wave-transposed SH copy of the Gaussians last passed to gs2m_raster_pack_sh
    size_t shpack_cap = 0;
    const float* pack_src = nullptr;
    const float* pack_src_rest = nullptr;
    int pack_P = 0;
    // gs2m_raster_pack_model: spatially ordered packed copy of the per-Gaussian parameters (the SH copy above is then in
    // the same order), order = position -> id, rank = id -> position
    float* d_pk_xyz = nullptr;
    float* d_pk_scales = nullptr;
    float* d_pk_rots = nullptr;
    float* d_pk_opac = nullptr;
    int* d_order = nullptr;
    int* d_rank = nullptr;
    size_t pk_cap3 = 0, pk_cap3s = 0, pk_cap4 = 0, pk_cap1 = 0, order_cap = 0, rank_cap = 0;
    const float* model_src[4] = {nullptr, nullptr, nullptr, nullptr};  // xyz, scales, rotations, opacities packed from
    bool model_packed = false;
    bool hint_valid = false;         // h_status holds the class counts of an earlier pass of the same geometry
    bool last_packed = false;        // the last pass ran on the packed copy (parity taps map positions back to ids)
    const int* run_rank = nullptr;   // rank table of the pass being launched (null: keys carry record positions)
    size_t mask_cap = 0;
    unsigned* d_hist = nullptr;
    size_t hist_cap = 0;  // words
    unsigned* d_tile_count = nullptr;
    unsigned* d_tile_start = nullptr;
    size_t tile_cap = 0;  // words per array
    unsigned* d_sort_lists = nullptr;  // per view, per size class: count + tile ids (k_tile_scan -> k_sort_tiles_*)
    size_t sort_lists_cap = 0;
    unsigned long long* d_keys = nullptr;
    unsigned long long* d_tmp = nullptr;
    size_t keys_cap_total = 0;  // entries in each of d_keys / d_tmp
    unsigned inst_cap = 0;      // per-view capacity requested
    // [1 + GS2M_MAX_STATUS]: slot 0 is STICKY ({max instances any call needed, any call overflowed} since the last
    // gs2m_raster_status), slots 1.. are the views of the last call
    ViewStatus* d_status = nullptr;  // slot 0: the sticky word (device atomics)
    ViewStatus* h_status = nullptr;  // pinned, device-mapped: slots 1.. are WRITTEN BY k_tile_scan itself (round 4: no status copy
                                     // launch behind every pass); slot 0 mirrors the sticky word at gs2m_raster_status
    // last call
    int last_P = 0, last_nv = 0, last_tiles = 0, last_views_total = 0;
    unsigned last_cap = 0;
};

template <typename T>
static int ensure(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return 0;
    if (*p) {
        HIPCHK(hipFree(*p));  // synchronises: safe w.r.t. in-flight work
        *p = nullptr;
        *cap = 0;
    }
    size_t n = need + need / 8 + 64;
    HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
    *cap = n;
    return 0;
}

extern "C" int gs2m_raster_create(gs2m_raster** out, int device) {
    if (!out) {
        gs2m_set_error("gs2m_raster_create: out is NULL");
        return 1;
    }
    HIPCHK(hipSetDevice(device));
    gs2m_raster* r = new gs2m_raster();
    r->device = device;
    if (hipMalloc((void**)&r->d_cams, sizeof(CamUniform) * GS2M_MAX_PASS_VIEWS) != hipSuccess ||
        hipMalloc((void**)&r->d_status, sizeof(ViewStatus) * (GS2M_MAX_STATUS + 1)) != hipSuccess ||
        hipHostMalloc((void**)&r->h_status, sizeof(ViewStatus) * (GS2M_MAX_STATUS + 1)) != hipSuccess ||
        hipMemset(r->d_status, 0, sizeof(ViewStatus) * (GS2M_MAX_STATUS + 1)) != hipSuccess) {
        gs2m_set_error("gs2m_raster_create: allocation failed");
        delete r;
        return 1;
    }
    memset(r->h_status, 0, sizeof(ViewStatus) * (GS2M_MAX_STATUS + 1));
    *out = r;
    return 0;
}

extern "C" int gs2m_raster_destroy(gs2m_raster* r) {
    if (!r) return 0;
    (void)hipFree(r->d_cams);
    (void)hipFree(r->d_recs);
    (void)hipFree(r->d_tilemask);
    (void)hipFree(r->d_shpack);
    (void)hipFree(r->d_pk_xyz);
    (void)hipFree(r->d_pk_scales);
    (void)hipFree(r->d_pk_rots);
    (void)hipFree(r->d_pk_opac);
    (void)hipFree(r->d_order);
    (void)hipFree(r->d_rank);
    (void)hipFree(r->d_hist);
    (void)hipFree(r->d_tile_count);
    (void)hipFree(r->d_tile_start);
    (void)hipFree(r->d_sort_lists);
    (void)hipFree(r->d_keys);
    (void)hipFree(r->d_tmp);
    (void)hipFree(r->d_status);
    (void)hipFree(r->d_blend_prof);
    (void)hipHostFree(r->h_status);
    for (auto& p : r->ev_live) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : r->ev_free) (void)hipEventDestroy(e);
    delete r;
    return 0;
}

extern "C" int gs2m_raster_set_option(gs2m_raster* r, int option, int value) {
    if (!r) {
        gs2m_set_error("null handle");
        return 1;
    }
    switch (option) {
        case GS2M_OPT_EXACT_TILE_CULL:
            if (value < 0 || value > 2) {
                gs2m_set_error("GS2M_OPT_EXACT_TILE_CULL must be 0, 1 or 2");
                return 1;
            }
            r->opt_exact_cull = value;
            return 0;
        case GS2M_OPT_BLEND_VARIANT:
            if (value != 0 && value != 4) {
                gs2m_set_error("GS2M_OPT_BLEND_VARIANT must be 0 or 4");
                return 1;
            }
            r->opt_blend = value;
            return 0;
        case GS2M_OPT_TILE_ROWS:
            if (value != 1 && value != 2) {
                gs2m_set_error("GS2M_OPT_TILE_ROWS must be 1 or 2");
                return 1;
            }
            r->opt_tile_rows = value;
            return 0;
        case GS2M_OPT_PAIR_BATCH:
            if (value < 0 || value > GS2M_MAX_PAIRS) {
                gs2m_set_error("GS2M_OPT_PAIR_BATCH must be 0 .. %d (stereo pairs per launch)", GS2M_MAX_PAIRS);
                return 1;
            }
            r->opt_pair_batch = value;
            return 0;
        case GS2M_OPT_DEBUG_SYNC: r->opt_debug = value != 0; return 0;
        case GS2M_OPT_STAGE_TIMING: r->opt_timing = value != 0; return 0;
        case GS2M_OPT_BIN_WORKGROUPS:
            if (value < 0 || value > 4096) {
                gs2m_set_error("GS2M_OPT_BIN_WORKGROUPS must be 0 (default) .. 4096");
                return 1;
            }
            r->opt_bin_workgroups = value ? value : 256;
            return 0;
        case GS2M_OPT_BIN_WG_THREADS:
            if (value < 0 || value > 1024 || value % 64) {
                gs2m_set_error("GS2M_OPT_BIN_WG_THREADS must be 0 (default) or a multiple of 64 up to 1024");
                return 1;
            }
            r->opt_bin_wg_threads = value ? value : 1024;
            return 0;
        case GS2M_OPT_BLEND_MODE:
            if (value != 0 && value != 2 && value != 3) {
                gs2m_set_error("GS2M_OPT_BLEND_MODE must be 0, 2 or 3 (1 was removed in round 6)");
                return 1;
            }
            r->opt_blend_mode = value;
            return 0;
        case GS2M_OPT_BIN_LANE_TILES:
            if (value < 0 || value > 16) {
                gs2m_set_error("GS2M_OPT_BIN_LANE_TILES must be 0 .. 16");
                return 1;
            }
            r->opt_bin_lane_tiles = value;
            return 0;
        case GS2M_OPT_PROJECT_SHARED_READ:
            if (value < 0 || value > 2) {
                gs2m_set_error("GS2M_OPT_PROJECT_SHARED_READ must be 0 (by model size), 1 (always) or 2 (never)");
                return 1;
            }
            r->opt_project_shared = value;
            return 0;
        case GS2M_OPT_BLEND_PROFILE:
            r->opt_blend_profile = value != 0;
            if (value && !r->d_blend_prof) {
                HIPCHK(hipSetDevice(r->device));
                HIPCHK(hipMalloc((void**)&r->d_blend_prof, 64 * 16 * sizeof(unsigned long long)));   // 64 copies, one per 128 B
                HIPCHK(hipMemset(r->d_blend_prof, 0, 64 * 16 * sizeof(unsigned long long)));
            }
            return 0;
        default: gs2m_set_error("unknown option %d", option); return 1;
    }
}

// tiles of the binning grid: 16 x (16 * rows) pixels
static int binning_tiles(const gs2m_raster* r, int W, int H) {
    const int gx = (W + GS2M_TILE - 1) / GS2M_TILE, gy = (H + GS2M_TILE - 1) / GS2M_TILE;
    return gx * ((gy + r->opt_tile_rows - 1) / r->opt_tile_rows);
}

static void geometry(const gs2m_raster* r, int P, int* chunk, int* n_wg, int pairs = 1) {
    // ~256 workgroups (one per CU: the per-workgroup tile histogram rows / cursors scale with their number;
    // measured optimum on C2/C3; GS2M_OPT_BIN_WORKGROUPS), chunks a multiple of 256 Gaussians
    const int target = r->opt_bin_workgroups > 0 ? r->opt_bin_workgroups : 256;
    // `pairs` groups of views share a launch (blockIdx.y): the workgroups of a group are target / pairs, the launch still
    // fills the chip, and what a workgroup pays once (LDS clear, its histogram row, cursor set-up) is paid half as often
    const int tgt = target / pairs > 0 ? target / pairs : 1;
    int c = (P + tgt - 1) / tgt;
    c = (c + 255) / 256 * 256;
    if (c < 256) c = 256;
    if (c > 64000) c = 64000;  // k_count_tiles keeps 16-bit per-tile counters per workgroup (chunk + one sub-chunk < 65536)
    *chunk = c;
    *n_wg = (P + c - 1) / c;
    if (*n_wg < 1) *n_wg = 1;
}

extern "C" int gs2m_raster_reserve(gs2m_raster* r, int P, int n_views, int W, int H, int64_t instances) {
    if (!r) {
        gs2m_set_error("null handle");
        return 1;
    }
    HIPCHK(hipSetDevice(r->device));
    const int nv = n_views < GS2M_MAX_PASS_VIEWS ? (n_views < 1 ? 1 : n_views) : GS2M_MAX_PASS_VIEWS;
    const int tiles = ((W + GS2M_TILE - 1) / GS2M_TILE) * ((H + GS2M_TILE - 1) / GS2M_TILE);
    int chunk, n_wg;
    geometry(r, P, &chunk, &n_wg);
    if (ensure(&r->d_recs, &r->recs_cap, 3 * (size_t)nv * (size_t)(P > 0 ? P : 1))) return 1;
    if (ensure(&r->d_tilemask, &r->mask_cap, (size_t)nv * (size_t)(P > 0 ? P : 1))) return 1;
    if (ensure(&r->d_hist, &r->hist_cap, (size_t)nv * n_wg * tiles)) return 1;
    size_t tc = r->tile_cap;
    if (ensure(&r->d_tile_count, &tc, (size_t)nv * (tiles + 1))) return 1;
    if (ensure(&r->d_tile_start, &r->tile_cap, (size_t)nv * (tiles + 1))) return 1;
    if (ensure(&r->d_sort_lists, &r->sort_lists_cap, gs2m_sort_lists_words(nv, tiles))) return 1;
    if (instances > 0xfffffff0ll) {
        gs2m_set_error("instance count %lld exceeds the 32-bit offsets of the binning stage", (long long)instances);
        return 1;
    }
    if (instances > 0 && (unsigned)instances > r->inst_cap) r->inst_cap = (unsigned)instances;
    if (r->inst_cap < 1024) r->inst_cap = 1024;
    size_t kc = r->keys_cap_total;
    if (ensure(&r->d_keys, &kc, (size_t)nv * r->inst_cap)) return 1;
    if (ensure(&r->d_tmp, &r->keys_cap_total, (size_t)nv * r->inst_cap)) return 1;
    return 0;
}

static int dbg_check(gs2m_raster* r, hipStream_t st, const char* what) {
    if (!r->opt_debug) return 0;
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        gs2m_set_error("[debug] after %s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

static hipEvent_t ev_get(gs2m_raster* r) {
    if (!r->ev_free.empty()) {
        hipEvent_t e = r->ev_free.back();
        r->ev_free.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static const char* const kStageRange[GS2M_N_STAGES] = {"gs2m:project", "gs2m:hist_colscan", "gs2m:tile_scan", "gs2m:scatter",
                                                        "gs2m:sort_tiles", "gs2m:blend", "gs2m:count_tiles"};
struct StageTimer {  // RAII: a rocTX range (GS2M_ROCTX=1) and, when timing is on, an event pair around one stage launch
    gs2m_raster* r;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int stage;
    Gs2mRange range;
    StageTimer(gs2m_raster* r_, hipStream_t st_, int stage_) : r(r_), st(st_), stage(stage_), range(kStageRange[stage_]) {
        if (r->opt_timing) {
            a = ev_get(r);
            b = ev_get(r);
            if (a) (void)hipEventRecord(a, st);
        }
    }
    ~StageTimer() {
        if (a && b) {
            (void)hipEventRecord(b, st);
            r->ev_live.push_back({stage, a, b});
        }
    }
};

// One fused pass over `pairs` groups of nv (<= GS2M_MAX_VIEWS) views whose CamUniforms are already in r->d_cams (host_cams ==
// null) or travel with the projection launch (host_cams = the nv * pairs host-side uniforms).
// A group is what one projection / counting / scatter workgroup handles (a stereo pair: parameters and Sigma once for both
// eyes); with pairs > 1 (GS2M_OPT_PAIR_BATCH) that many groups share every launch (blockIdx.y), each with 1 / pairs of the workgroups.
static int run_views(gs2m_raster* r, const GaussIn& g, int nv, int pairs, int W, int H, float* out_color,
                     unsigned char* out_rgb8, int* out_radii, int status_slot, hipStream_t st, const CamUniform* host_cams) {
    const int nvt = nv * pairs;   // views of the pass: the scans, the per-tile sort and the compositing take them all (grid.y)
    const int gx = (W + GS2M_TILE - 1) / GS2M_TILE, gy = (H + GS2M_TILE - 1) / GS2M_TILE;
    const int gys = (gy + r->opt_tile_rows - 1) / r->opt_tile_rows;  // binning rows (tiles of 16 x 16*rows pixels)
    const int tiles = gx * gys;
    // Advisory hints from the PREVIOUS pass of the same geometry, snapshotted before this pass launches anything: k_tile_scan
    // writes {num_rendered, class counts} of a pass straight into the pinned, device-mapped mirror, so reading the mirror after
    // this pass's own launches would race with its writer.  Every value is legal -- the sort falls back exactly when
    // a class grid is too small, either dispatch order of the compositing grid is correct -- and a slot the previous pass has
    // not written yet reads as the pass before it (or zero: "no information").
    int hint[3] = {-1, -1, -1};
    int interleave = 1;
    if (r->hint_valid && r->last_tiles == tiles && r->last_nv == nvt) {
        unsigned nr_max = 0;
        for (int c = 0; c < 3; ++c) {
            unsigned m = 0;
            for (int v = 0; v < nvt; ++v) {
                const unsigned x = r->h_status[1 + status_slot + v].n_class[c];
                m = x > m ? x : m;
            }
            hint[c] = m > 0x3fffffffu ? -1 : (int)m;
        }
        for (int v = 0; v < nvt; ++v) {
            const unsigned x = r->h_status[1 + status_slot + v].num_rendered;
            nr_max = x > nr_max ? x : nr_max;
        }
        // Dispatch order of a multi-view compositing launch: chunk rank major / view minor ("interleaved": the heavy chunks of
        // every view start first) pays when the lists are long -- the launch then ends on the heavy chunks of its last view
        // (C3, ~1700 instances per list: 220 -> 205 us per pair) -- and costs ~2 % when they are short (C2, ~300 per list: four
        // views' records compete for each XCD's L2).  First pass / no information: interleaved.
        if (nr_max > 0u && nr_max <= 0x3fffffffu) interleave = (size_t)nr_max >= (size_t)800 * (size_t)tiles;
    }
    int chunk, n_wg;
    geometry(r, g.P, &chunk, &n_wg, pairs);
    // threads per counting / scatter workgroup: the chunk (<= 1024), halved until the wave staging fits next to the
    // tile cursors in the 160 KiB LDS (large images)
    int wg_threads = (gs2m_count_threads(chunk, r->opt_bin_wg_threads) + 63) / 64 * 64;
    while (wg_threads > 64 && (gs2m_scatter_lds_bytes(nv, tiles, wg_threads) > 160 * 1024 || gs2m_count_lds_bytes(nv, tiles, wg_threads) > 160 * 1024))
        wg_threads = (wg_threads / 2 + 63) / 64 * 64;  // stays a whole number of waves: the size checked is the size launched
    const size_t lds = gs2m_scatter_lds_bytes(nv, tiles, wg_threads);    // scatter: u32 cursors + wave staging
    const size_t lds_p = gs2m_count_lds_bytes(nv, tiles, wg_threads);    // count: u32 histogram + staging
    if (lds > 160 * 1024 || lds_p > 160 * 1024) {
        gs2m_set_error("image %dx%d: %d views x %d tiles do not fit the 160 KiB LDS tile cursors", W, H, nv, tiles);
        return 1;
    }
    if (r->inst_cap == 0) {
        // first call: 4 instances per Gaussian, at least 64k
        int64_t guess = (int64_t)g.P * 4;
        if (guess < 65536) guess = 65536;
        r->inst_cap = (unsigned)(guess > 0xfffffff0ll ? 0xfffffff0ll : guess);
    }
    if (gs2m_raster_reserve(r, g.P, nvt, W, H, 0)) return 1;
    const unsigned cap = r->inst_cap;
    // the records of the pass: [nvt * P] 32-B parts, then [nvt * P] 16-B parts
    const GeomRecs recs{r->d_recs, r->d_recs + 2 * (size_t)nvt * (size_t)g.P};
    const int cull_arg_p = r->opt_exact_cull, cull_arg_s = r->opt_exact_cull;  // same option for counting and scatter
    // (round 3: projection and counting fused into one kernel -- the counting workgroups projecting their own Gaussians and
    // going on from registers -- measured 51 vs 28 + 28 us on C2 and 212 vs 135 + 80 us on C3: the counting step is bound by
    // its own LDS atomics and tile tests, not by re-reading the records; 128 VGPRs for 1024-thread workgroups.  Not kept.)
    {
        StageTimer tm(r, st, GS2M_STAGE_PROJECT);
        gs2m_launch_project(nv, pairs, st, g, r->d_cams, recs, out_radii, cull_arg_p, host_cams, r->opt_project_shared);
    }
    if (dbg_check(r, st, "project")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_COUNT);
        if (gs2m_launch_count_tiles(nv, pairs, n_wg, wg_threads, lds_p, st, recs, g.P, r->d_cams, chunk, r->d_hist, r->d_tilemask,
                                    cull_arg_p, g.ids != nullptr, r->opt_bin_lane_tiles))
            return 1;
    }
    if (dbg_check(r, st, "count_tiles")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_COLSCAN);
        gs2m_launch_hist_colscan(st, nvt, r->d_hist, n_wg, tiles, r->d_tile_count);
    }
    if (dbg_check(r, st, "hist_colscan")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_TILESCAN);
        gs2m_launch_tile_scan(st, nvt, r->d_tile_count, r->d_tile_start, tiles, gx, r->h_status + 1 + status_slot, r->d_status, cap, r->d_sort_lists);
    }
    if (dbg_check(r, st, "tile_scan")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_SCATTER);
        if (gs2m_launch_scatter(nv, pairs, n_wg, wg_threads, lds, st, recs, g.P, r->d_cams, chunk, r->d_hist, r->d_tile_start,
                                r->d_tilemask, r->d_keys, cap, cull_arg_s, g.ids, g.ids != nullptr, r->opt_bin_lane_tiles))
            return 1;
    }
    if (dbg_check(r, st, "scatter")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_SORT);
        // class-grid hint: snapshotted at the top of the pass
        gs2m_launch_sort_tiles(st, nvt, r->d_keys, r->d_tmp, r->d_tile_start, tiles, cap, r->d_sort_lists, hint);
    }
    if (dbg_check(r, st, "sort_tiles")) return 1;
    {
        StageTimer tm(r, st, GS2M_STAGE_BLEND);
        if (gs2m_launch_blend(st, r->opt_blend, r->opt_tile_rows, nvt, gx, gy, r->d_keys, r->d_tile_start, recs, r->d_cams,
                              g.P, cap, out_color, out_rgb8, g.ids ? r->run_rank : nullptr,
                              r->d_sort_lists + (size_t)nvt * GS2M_SORT_CLASSES_API * (tiles + 1), r->opt_blend_mode,
                              r->opt_blend_profile ? r->d_blend_prof : nullptr, interleave))
            return 1;
    }
    if (dbg_check(r, st, "blend")) return 1;
    r->last_P = g.P;
    r->hint_valid = true;
    r->last_packed = g.ids != nullptr;
    r->last_nv = nvt;
    r->last_tiles = tiles;
    r->last_cap = cap;
    return 0;
}

extern "C" int gs2m_rasterize_forward(gs2m_raster* r, int P, int D, int M, const float* background, int width,
                                      int height, const float* means3D, const float* shs,
                                      const float* colors_precomp, const float* opacities, const float* scales,
                                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                      const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                      float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                                      int* radii, int debug, gs2m_stream stream) {
    (void)prefiltered;  // only changes the reference's in-kernel trap (auxiliary.h:156-160)
    if (!r) {
        gs2m_set_error("null handle");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(r->device));
    if (width <= 0 || height <= 0 || !out_color) {
        gs2m_set_error("gs2m_rasterize_forward: bad image arguments");
        return 1;
    }
    if (D < 0 || D > 3) {
        gs2m_set_error("SH degree %d not in 0..3", D);
        return 1;
    }
    r->last_views_total = 1;
    if (P == 0) {
        // rasterize_points.cu:68,81: the zero-filled image is returned untouched
        HIPCHK(hipMemsetAsync(out_color, 0, sizeof(float) * 3 * (size_t)width * height, st));
        HIPCHK(hipMemsetAsync(r->h_status + 1, 0, sizeof(ViewStatus), st));   // stream-ordered like a pass's own status write
        r->last_P = 0;
        r->last_nv = 1;
        return 0;
    }
    if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !cam_pos) {
        gs2m_set_error("gs2m_rasterize_forward: NULL required pointer");
        return 1;
    }
    if ((shs == nullptr) == (colors_precomp == nullptr)) {
        gs2m_set_error("Please provide excatly one of either SHs or precomputed colors!");
        return 1;
    }
    if (((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) ||
        ((scales != nullptr || rotations != nullptr) && cov3D_precomp != nullptr)) {
        gs2m_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return 1;
    }
    if (shs && (M < (D + 1) * (D + 1))) {
        gs2m_set_error("M = %d SH coefficients < (D+1)^2 = %d", M, (D + 1) * (D + 1));
        return 1;
    }
    GaussIn g;
    g.xyz = means3D;
    g.scales = scales;
    g.rots = rotations;
    g.opac = opacities;
    g.shs = shs;
    g.shs_rest = nullptr;
    g.shs_packed = nullptr;
    g.cov3D_precomp = cov3D_precomp;
    g.colors_precomp = colors_precomp;
    g.ids = nullptr;
    g.P = P;
    g.D = D;
    g.M = M;
    g.raw = 0;
    g.scale_modifier = scale_modifier;
    const int saved_debug = r->opt_debug;
    if (debug) r->opt_debug = 1;
    gs2m_launch_pack_camera(st, r->d_cams, 0, viewmatrix, projmatrix, cam_pos, background, tan_fovx, tan_fovy,
                            width, height, 16 * r->opt_tile_rows);
    int rc = run_views(r, g, 1, 1, width, height, out_color, nullptr, radii, 0, st, nullptr);
    r->opt_debug = saved_debug;
    if (rc) return rc;
    return 0;
}

extern "C" int gs2m_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                 uint8_t* present, gs2m_stream stream) {
    (void)projmatrix;
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) {
        gs2m_set_error("gs2m_mark_visible: NULL pointer");
        return 1;
    }
    gs2m_launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
    return 0;
}

extern "C" int gs2m_render_views(gs2m_raster* r, const gs2m_gaussians* gs, const gs2m_camera* cams, int n_views,
                                 const float* bg, float scale_modifier, float* out_color, uint8_t* out_rgb8,
                                 int* out_radii, gs2m_stream stream) {
    if (!r || !gs || !cams || !bg) {
        gs2m_set_error("gs2m_render_views: NULL argument");
        return 1;
    }
    if (n_views < 1 || n_views > GS2M_MAX_STATUS) {
        gs2m_set_error("n_views = %d not in 1..%d", n_views, GS2M_MAX_STATUS);
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(r->device));
    const int W = cams[0].width, H = cams[0].height;
    for (int v = 0; v < n_views; ++v)
        if (cams[v].width != W || cams[v].height != H || W <= 0 || H <= 0) {
            gs2m_set_error("all views of a batch must share one positive width/height");
            return 1;
        }
    if (gs->sh_degree < 0 || gs->sh_degree > 3 || gs->M < (gs->sh_degree + 1) * (gs->sh_degree + 1)) {
        gs2m_set_error("bad SH layout: degree %d, M %d", gs->sh_degree, gs->M);
        return 1;
    }
    r->last_views_total = n_views;
    const size_t img = (size_t)W * H;
    if (gs->P == 0) {
        if (out_color) HIPCHK(hipMemsetAsync(out_color, 0, sizeof(float) * 3 * img * n_views, st));
        if (out_rgb8) HIPCHK(hipMemsetAsync(out_rgb8, 0, 3 * img * n_views, st));
        HIPCHK(hipMemsetAsync(r->h_status + 1, 0, sizeof(ViewStatus) * n_views, st));
        return 0;
    }
    if (!gs->xyz || !gs->scales || !gs->rotations || !gs->opacities || !gs->shs) {
        gs2m_set_error("gs2m_render_views: NULL Gaussian array");
        return 1;
    }
    GaussIn g;
    g.xyz = gs->xyz;
    g.scales = gs->scales;
    g.rots = gs->rotations;
    g.opac = gs->opacities;
    g.shs = gs->shs;
    g.shs_rest = gs->shs_rest;
    g.shs_packed = (r->pack_src == gs->shs && r->pack_src_rest == gs->shs_rest && r->pack_P == gs->P && gs->M == 16)
                       ? r->d_shpack : nullptr;
    g.cov3D_precomp = nullptr;
    g.colors_precomp = nullptr;
    g.ids = nullptr;
    r->run_rank = nullptr;
    if (r->model_packed && g.shs_packed && r->model_src[0] == gs->xyz && r->model_src[1] == gs->scales &&
        r->model_src[2] == gs->rotations && r->model_src[3] == gs->opacities) {
        // the spatially ordered packed copy of THESE Gaussians (gs2m_raster_pack_model): every stage works on positions of
        // the copy; ids only enter the sort keys (tie order of the reference) and the radii output
        g.xyz = r->d_pk_xyz;
        g.scales = r->d_pk_scales;
        g.rots = r->d_pk_rots;
        g.opac = r->d_pk_opac;
        g.ids = r->d_order;
        r->run_rank = r->d_rank;
    }
    g.P = gs->P;
    g.D = gs->sh_degree;
    g.M = gs->M;
    g.raw = gs->raw;
    g.scale_modifier = scale_modifier;
    const int tiles = binning_tiles(r, W, H);
    // views fused per pass: as many (<= GS2M_MAX_VIEWS) as the LDS tile cursors of the scatter allow
    int per = GS2M_MAX_VIEWS;
    while (per > 1 && gs2m_scatter_lds_bytes(per, tiles, 64) > 160 * 1024) per--;
    for (int v0 = 0; v0 < n_views;) {
        const int nv = n_views - v0 < per ? n_views - v0 : per;
        // GS2M_OPT_PAIR_BATCH: up to that many full groups (stereo pairs) per pass, as far as the call has them
        int pairs = 1;
        if (r->opt_pair_batch > 1 && per == GS2M_MAX_VIEWS) {
            pairs = (n_views - v0) / per;
            if (pairs > r->opt_pair_batch) pairs = r->opt_pair_batch;
            if (pairs < 1) pairs = 1;
        }
        const int nvt = nv * pairs;
        CamUniform cu[GS2M_MAX_PASS_VIEWS];
        for (int k = 0; k < nvt; ++k) {
            const gs2m_camera& c = cams[v0 + k];
            CamUniform& u = cu[k];
            memcpy(u.view, c.viewmatrix, sizeof(u.view));
            memcpy(u.proj, c.projmatrix, sizeof(u.proj));
            memcpy(u.campos, c.campos, sizeof(u.campos));
            u.tanfovx = c.tanfovx;
            u.tanfovy = c.tanfovy;
            u.focal_y = H / (2.0f * c.tanfovy);  // rasterizer_impl.cu:222-223
            u.focal_x = W / (2.0f * c.tanfovx);
            u.W = W;
            u.H = H;
            u.gx = (W + GS2M_TILE - 1) / GS2M_TILE;
            u.gy = (H + GS2M_TILE - 1) / GS2M_TILE;
            u.bg[0] = bg[0];
            u.bg[1] = bg[1];
            u.bg[2] = bg[2];
            u.th = 16 * r->opt_tile_rows;
        }
        if (run_views(r, g, nv, pairs, W, H, out_color ? out_color + 3 * img * v0 : nullptr,
                      out_rgb8 ? out_rgb8 + 3 * img * v0 : nullptr,
                      out_radii ? out_radii + (size_t)gs->P * v0 : nullptr, v0, st, cu))   // the uniforms travel with k_project
            return 1;
        v0 += nvt;
    }
    return 0;
}

static int pack_common(gs2m_raster* r, const gs2m_gaussians* gs, const int32_t* order, gs2m_stream stream, const char* who) {
    if (!r || !gs) {
        gs2m_set_error("%s: NULL argument", who);
        return 1;
    }
    r->pack_src = nullptr;
    r->pack_src_rest = nullptr;
    r->pack_P = 0;
    r->model_packed = false;
    if (gs->P <= 0 || gs->M != 16 || !gs->shs) return 0;  // nothing to pack: the kernels read the caller's layout
    HIPCHK(hipSetDevice(r->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t P = (size_t)gs->P;
    const size_t groups = (P + 63) / 64;
    if (ensure(&r->d_shpack, &r->shpack_cap, groups * 12 * 64 * 4)) return 1;
    if (order) {
        if (!gs->xyz || !gs->scales || !gs->rotations || !gs->opacities) {
            gs2m_set_error("%s: NULL Gaussian array", who);
            return 1;
        }
        if (ensure(&r->d_pk_xyz, &r->pk_cap3, 3 * P) || ensure(&r->d_pk_scales, &r->pk_cap3s, 3 * P) ||
            ensure(&r->d_pk_rots, &r->pk_cap4, 4 * P) || ensure(&r->d_pk_opac, &r->pk_cap1, P) ||
            ensure(&r->d_order, &r->order_cap, P) || ensure(&r->d_rank, &r->rank_cap, P + 1))
            return 1;
        HIPCHK(hipMemcpyAsync(r->d_order, order, sizeof(int) * P, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemsetAsync(r->d_rank, 0xff, sizeof(int) * P, st));          // holes = -1
        HIPCHK(hipMemsetAsync(r->d_rank + P, 0, sizeof(int), st));             // word P: "not a permutation" flag
        gs2m_launch_pack_model(st, gs->P, r->d_order, gs->xyz, gs->scales, gs->rotations, gs->opacities, r->d_pk_xyz,
                               r->d_pk_scales, r->d_pk_rots, r->d_pk_opac, r->d_rank, reinterpret_cast<unsigned*>(r->d_rank + P));
        int bad = 0;
        HIPCHK(hipMemcpyAsync(&bad, r->d_rank + P, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));   // one-time preparation: a checked result is worth the sync
        if (bad) {
            gs2m_set_error("%s: order is not a permutation of 0..P-1", who);
            return 1;
        }
    }
    gs2m_launch_pack_sh(st, gs->P, gs->shs, gs->shs_rest, r->d_shpack, order ? r->d_order : nullptr);
    r->pack_src = gs->shs;
    r->pack_src_rest = gs->shs_rest;
    r->pack_P = gs->P;
    if (order) {
        r->model_src[0] = gs->xyz;
        r->model_src[1] = gs->scales;
        r->model_src[2] = gs->rotations;
        r->model_src[3] = gs->opacities;
        r->model_packed = true;
    }
    return 0;
}

extern "C" int gs2m_raster_pack_invalidate(gs2m_raster* r) {
    if (!r) {
        gs2m_set_error("null handle");
        return 1;
    }
    r->pack_src = nullptr;
    r->pack_src_rest = nullptr;
    r->pack_P = 0;
    r->model_packed = false;
    return 0;
}

extern "C" int gs2m_raster_pack_sh(gs2m_raster* r, const gs2m_gaussians* gs, gs2m_stream stream) {
    return pack_common(r, gs, nullptr, stream, "gs2m_raster_pack_sh");
}

extern "C" int gs2m_raster_pack_model(gs2m_raster* r, const gs2m_gaussians* gs, const int32_t* order, gs2m_stream stream) {
    if (!order) {
        gs2m_set_error("gs2m_raster_pack_model: order is NULL (gs2m_raster_pack_sh packs without reordering)");
        return 1;
    }
    return pack_common(r, gs, order, stream, "gs2m_raster_pack_model");
}

extern "C" int gs2m_raster_status(gs2m_raster* r, gs2m_stream stream, int n_views, int64_t* num_rendered,
                                  int* overflow, int64_t* required) {
    if (!r) {
        gs2m_set_error("null handle");
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(r->h_status, r->d_status, sizeof(ViewStatus), hipMemcpyDeviceToHost));   // the sticky word (the stream is idle)
    // slot 0 is sticky: an overflow in ANY call since the last query is reported (a later call on the same handle that
    // fits does not erase it), with the largest instance count any of those calls needed
    int ov = r->h_status[0].overflow != 0;
    int64_t req = r->h_status[0].num_rendered;
    const int n = n_views < r->last_views_total ? n_views : r->last_views_total;
    for (int v = 0; v < r->last_views_total && v < GS2M_MAX_STATUS; ++v) {
        ov |= r->h_status[1 + v].overflow != 0;
        if ((int64_t)r->h_status[1 + v].num_rendered > req) req = r->h_status[1 + v].num_rendered;
    }
    for (int v = 0; v < n && num_rendered; ++v) num_rendered[v] = r->h_status[1 + v].num_rendered;
    if (overflow) *overflow = ov;
    if (required) *required = req;
    // the query consumes the sticky word (the stream is idle here)
    HIPCHK(hipMemsetAsync(r->d_status, 0, sizeof(ViewStatus), (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    r->h_status[0].overflow = 0;
    r->h_status[0].num_rendered = 0;
    return 0;
}

extern "C" int gs2m_raster_stage_times(gs2m_raster* r, gs2m_stream stream, double* total_ms, int64_t* launches) {
    if (!r || !total_ms || !launches) {
        gs2m_set_error("gs2m_raster_stage_times: NULL argument");
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (auto& p : r->ev_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess && p.stage >= 0 && p.stage < GS2M_N_STAGES) {
            total_ms[p.stage] += ms;
            launches[p.stage] += 1;
        }
        r->ev_free.push_back(p.a);
        r->ev_free.push_back(p.b);
    }
    r->ev_live.clear();
    return 0;
}

extern "C" int gs2m_raster_blend_cycles(gs2m_raster* r, gs2m_stream stream, uint64_t* counters) {
    if (!r || !counters) {
        gs2m_set_error("gs2m_raster_blend_cycles: NULL argument");
        return 1;
    }
    for (int i = 0; i < GS2M_BLEND_PROF_COUNTERS; ++i) counters[i] = 0;
    if (!r->d_blend_prof) return 0;
    unsigned long long h[64 * 16];
    HIPCHK(hipMemcpyAsync(h, r->d_blend_prof, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipMemsetAsync(r->d_blend_prof, 0, sizeof(h), (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (int c = 0; c < 64; ++c)
        for (int i = 0; i < GS2M_BLEND_PROF_COUNTERS; ++i) counters[i] += h[c * 16 + i];
    return 0;
}

extern "C" int gs2m_raster_download_geometry(gs2m_raster* r, gs2m_stream stream, int v, int P, float* means2D,
                                             float* depths, float* conic_opacity, float* rgb, uint16_t* rect,
                                             uint32_t* tiles_touched) {
    if (!r || v < 0 || v >= r->last_nv || P != r->last_P) {
        gs2m_set_error("gs2m_raster_download_geometry: view %d / P %d do not match the last call", v, P);
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (P == 0) return 0;
    GeomRec* h = (GeomRec*)malloc(sizeof(GeomRec) * (size_t)P);
    float4* hab = (float4*)malloc(sizeof(float4) * 2 * (size_t)P);
    float4* hc = (float4*)malloc(sizeof(float4) * (size_t)P);
    if (!h || !hab || !hc) {
        free(h);
        free(hab);
        free(hc);
        gs2m_set_error("out of host memory");
        return 1;
    }
    // the two arrays of the last pass (GeomRecs: [last_nv * P] 32-B parts, then the 16-B parts) -> whole records
    hipError_t e = hipMemcpy(hab, r->d_recs + 2 * (size_t)v * P, sizeof(float4) * 2 * (size_t)P, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(hc, r->d_recs + 2 * (size_t)r->last_nv * P + (size_t)v * P, sizeof(float4) * (size_t)P, hipMemcpyDeviceToHost);
    for (int i = 0; e == hipSuccess && i < P; ++i) {
        memcpy(&h[i], &hab[2 * (size_t)i], 32);
        memcpy(reinterpret_cast<char*>(&h[i]) + 32, &hc[i], 16);
    }
    free(hab);
    free(hc);
    if (e != hipSuccess) {
        free(h);
        gs2m_set_error("hipMemcpy: %s", hipGetErrorString(e));
        return 1;
    }
    int* ord = nullptr;   // packed model: record position -> Gaussian id (the taps are indexed by id, like the reference's arrays)
    if (r->last_packed) {
        ord = (int*)malloc(sizeof(int) * (size_t)P);
        if (!ord || hipMemcpy(ord, r->d_order, sizeof(int) * (size_t)P, hipMemcpyDeviceToHost) != hipSuccess) {
            free(ord);
            free(h);
            gs2m_set_error("gs2m_raster_download_geometry: cannot read the packed order");
            return 1;
        }
    }
    for (int pos = 0; pos < P; ++pos) {
        const GeomRec& q = h[pos];
        const int i = ord ? ord[pos] : pos;
        const unsigned x0 = q.rect0 & 0xffffu, y0 = q.rect0 >> 16, x1 = q.rect1 & 0xffffu, y1 = q.rect1 >> 16;
        const bool vis = x1 > x0 && y1 > y0;
        if (means2D) {
            means2D[2 * i] = vis ? q.mx : 0.f;
            means2D[2 * i + 1] = vis ? q.my : 0.f;
        }
        if (depths) depths[i] = vis ? q.depth : 0.f;
        if (conic_opacity) {
            conic_opacity[4 * i] = vis ? q.ca : 0.f;
            conic_opacity[4 * i + 1] = vis ? q.cb : 0.f;
            conic_opacity[4 * i + 2] = vis ? q.cc : 0.f;
            conic_opacity[4 * i + 3] = vis ? q.op : 0.f;
        }
        if (rgb) {
            rgb[3 * i] = vis ? q.r : 0.f;
            rgb[3 * i + 1] = vis ? q.g : 0.f;
            rgb[3 * i + 2] = vis ? q.b : 0.f;
        }
        if (rect) {
            rect[4 * i] = (uint16_t)x0;
            rect[4 * i + 1] = (uint16_t)y0;
            rect[4 * i + 2] = (uint16_t)x1;
            rect[4 * i + 3] = (uint16_t)y1;
        }
        if (tiles_touched) tiles_touched[i] = vis ? (x1 - x0) * (y1 - y0) : 0u;
    }
    free(ord);
    free(h);
    return 0;
}

extern "C" int gs2m_raster_download_binning(gs2m_raster* r, gs2m_stream stream, int v, int64_t n,
                                            uint32_t* point_list, int32_t n_tiles, uint32_t* ranges) {
    if (!r || v < 0 || v >= r->last_nv || n_tiles != r->last_tiles) {
        gs2m_set_error("gs2m_raster_download_binning: arguments do not match the last call");
        return 1;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (ranges) {
        unsigned* ts = (unsigned*)malloc(sizeof(unsigned) * (size_t)(n_tiles + 1));
        HIPCHK(hipMemcpy(ts, r->d_tile_start + (size_t)v * (n_tiles + 1), sizeof(unsigned) * (size_t)(n_tiles + 1),
                         hipMemcpyDeviceToHost));
        for (int t = 0; t < n_tiles; ++t) {
            // the reference leaves {0,0} for empty tiles (cudaMemset, rasterizer_impl.cu:310)
            const bool empty = ts[t + 1] == ts[t];
            ranges[2 * t] = empty ? 0u : ts[t];
            ranges[2 * t + 1] = empty ? 0u : ts[t + 1];
        }
        free(ts);
    }
    if (point_list && n > 0) {
        if ((uint64_t)n > r->last_cap) n = r->last_cap;
        unsigned long long* k = (unsigned long long*)malloc(sizeof(unsigned long long) * (size_t)n);
        HIPCHK(hipMemcpy(k, r->d_keys + (size_t)v * r->last_cap, sizeof(unsigned long long) * (size_t)n,
                         hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) point_list[i] = (uint32_t)(k[i] & 0xffffffffull);
        free(k);
    }
    return 0;
}
