// raster_project.hip -- projection / counting / scatter stage (compiled with -ffp-contract=off).
#include "raster_project.h"
#include "raster_internal.h"

// k_count_tiles: threads per workgroup = the workgroup's chunk of Gaussians, at most 1024
int gs2m_count_threads(int chunk, int max_threads) {
    const int cap = max_threads >= 64 && max_threads <= 1024 ? max_threads : 1024;   // GS2M_OPT_BIN_WG_THREADS
    const int t = chunk < cap ? (chunk + 63) / 64 * 64 : cap;
    return t < 64 ? 64 : t;
}
size_t gs2m_count_lds_bytes(int nv, int tiles, int threads) {
    return (size_t)((nv * tiles + 3) & ~3) * sizeof(unsigned) + (size_t)(threads / 64) * GS2M_STAGE_BYTES_PER_WAVE;
}

size_t gs2m_scatter_lds_bytes(int nv, int tiles, int threads) {
    return (size_t)((nv * tiles + 3) & ~3) * sizeof(unsigned) + (size_t)(threads / 64) * GS2M_SCATTER_STAGE_BYTES_PER_WAVE;
}

void gs2m_launch_project(int nv, int pairs, hipStream_t st, const GaussIn& g, CamUniform* cams, GeomRecs recs, int* radii,
                         int exact_cull, const CamUniform* host_cams, int shared_read) {
    const unsigned nx = (unsigned)((g.P + 255) / 256);
    if (nx == 0) return;
    // SH rows through LDS (k_project<.., true>): with the spatially ordered packed model (large models, where (almost) every
    // Gaussian is visible: C3 148 -> 136 us); a model that only has the packed SH copy keeps the register path, which reads
    // no row of a culled Gaussian and runs 16 instead of 12 waves per CU (C2: 28 vs 30 us)
    const bool dma = g.shs_packed != nullptr && g.colors_precomp == nullptr && g.ids != nullptr;
    // the colour pass streams the SH row three float4 at a time where the row is 16-B aligned (packed copy, [P,16,3]) or not
    // needed (precomputed colours); dc + rest split layouts / M != 16 keep the 48-register path (k_project<.., .., false>)
    const bool stream = g.colors_precomp != nullptr || g.shs_packed != nullptr || (g.shs_rest == nullptr && g.M == 16);
    // the `pairs` groups of nv views of the launch (GS2M_OPT_PAIR_BATCH): walked inside the kernel by the thread that owns the
    // Gaussian (one model read per launch) for large models, one grid row per group otherwise (project_gaussian)
    // auto: with the DMA path (spatially ordered packed model: its LDS bounds the occupancy at 3 waves per SIMD anyway, so the loop's
    // registers cost nothing -- C2 ordered, 4 pairs per launch: 16.8 -> 15.5 us) or from GS2M_PROJECT_LOOP_MIN_P Gaussians
    const bool loop = pairs > 1 && (shared_read == 1 || (shared_read == 0 && (dma || g.P >= GS2M_PROJECT_LOOP_MIN_P))) && host_cams != nullptr &&
                      nv == 2 && (dma || stream);
    const dim3 n(nx, loop ? 1u : (unsigned)pairs);
    if (host_cams) {
        // pipeline-level API: the uniforms of the nv * pairs views travel in the launch packet (k_project_hc stores them to
        // `cams` for the later kernels of the pass)
        CamUniformArg a;
        const int n_views = nv * pairs;
        for (int k = 0; k < GS2M_MAX_PASS_VIEWS; ++k) a.c[k] = host_cams[k < n_views ? k : 0];
        if (nv == 2 && dma && loop) GS2M_LAUNCH((k_project_hc<2, true, true, true>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (nv == 2 && stream && loop) GS2M_LAUNCH((k_project_hc<2, false, true, true>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (nv == 2 && dma) GS2M_LAUNCH((k_project_hc<2, true>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (nv == 2 && stream) GS2M_LAUNCH((k_project_hc<2, false>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (nv == 2) GS2M_LAUNCH((k_project_hc<2, false, false>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (dma) GS2M_LAUNCH((k_project_hc<1, true>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else if (stream) GS2M_LAUNCH((k_project_hc<1, false>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        else GS2M_LAUNCH((k_project_hc<1, false, false>), n, dim3(256), 0, st, g, a, cams, recs, radii, exact_cull, pairs);
        return;
    }
    // operator-level API (one view, uniforms written to `cams` by k_pack_camera from the caller's device tensors)
    if (dma) GS2M_LAUNCH((k_project<1, true>), n, dim3(256), 0, st, g, cams, recs, radii, exact_cull, pairs);
    else if (stream) GS2M_LAUNCH((k_project<1, false>), n, dim3(256), 0, st, g, cams, recs, radii, exact_cull, pairs);
    else GS2M_LAUNCH((k_project<1, false, false>), n, dim3(256), 0, st, g, cams, recs, radii, exact_cull, pairs);
}

int gs2m_launch_count_tiles(int nv, int pairs, int n_wg, int threads, size_t lds_bytes, hipStream_t st, GeomRecs recs, int P,
                            const CamUniform* cams, int chunk, unsigned* hist, unsigned long long* tilemask,
                            int exact_cull, int interleave, int lane_tiles) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = nv == 2 ? hipFuncSetAttribute((const void*)k_count_tiles<2>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
                               : hipFuncSetAttribute((const void*)k_count_tiles<1>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            gs2m_set_error("hipFuncSetAttribute(k_count_tiles, %zu B LDS): %s", lds_bytes, hipGetErrorString(e));
            return 1;
        }
    }
    if (nv == 2)
        GS2M_LAUNCH((k_count_tiles<2>), dim3(n_wg, pairs), dim3(threads), lds_bytes, st, recs, P, cams, chunk, n_wg, hist, tilemask,
                    exact_cull, interleave, lane_tiles);
    else
        GS2M_LAUNCH((k_count_tiles<1>), dim3(n_wg, pairs), dim3(threads), lds_bytes, st, recs, P, cams, chunk, n_wg, hist, tilemask,
                    exact_cull, interleave, lane_tiles);
    return 0;
}

int gs2m_launch_scatter(int nv, int pairs, int n_wg, int threads, size_t lds_bytes, hipStream_t st, GeomRecs recs, int P,
                        const CamUniform* cams, int chunk, const unsigned* hist, const unsigned* tile_start,
                        const unsigned long long* tilemask, unsigned long long* keys, unsigned cap, int exact_cull,
                        const int* ids, int interleave, int lane_tiles) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = nv == 2 ? hipFuncSetAttribute((const void*)k_scatter<2>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
                               : hipFuncSetAttribute((const void*)k_scatter<1>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            gs2m_set_error("hipFuncSetAttribute(k_scatter, %zu B LDS): %s", lds_bytes, hipGetErrorString(e));
            return 1;
        }
    }
    if (nv == 2)
        GS2M_LAUNCH((k_scatter<2>), dim3(n_wg, pairs), dim3(threads), lds_bytes, st, recs, P, cams, chunk, n_wg, hist,
                    tile_start, tilemask, keys, cap, exact_cull, ids, interleave, lane_tiles);
    else
        GS2M_LAUNCH((k_scatter<1>), dim3(n_wg, pairs), dim3(threads), lds_bytes, st, recs, P, cams, chunk, n_wg, hist,
                    tile_start, tilemask, keys, cap, exact_cull, ids, interleave, lane_tiles);
    return 0;
}

void gs2m_launch_pack_sh(hipStream_t st, int P, const float* shs, const float* shs_rest, float* packed, const int* order) {
    const size_t n = (size_t)P * 48;
    GS2M_LAUNCH(k_pack_sh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, shs, shs_rest, packed, order);
}

void gs2m_launch_pack_model(hipStream_t st, int P, const int* order, const float* xyz, const float* scales, const float* rots,
                            const float* opac, float* p_xyz, float* p_scales, float* p_rots, float* p_opac, int* rank,
                            unsigned* bad) {
    const unsigned n = (unsigned)((P + 255) / 256);
    GS2M_LAUNCH(k_pack_model, dim3(n), dim3(256), 0, st, P, order, xyz, scales, rots, opac, p_xyz, p_scales, p_rots, p_opac, rank);
    GS2M_LAUNCH(k_check_rank, dim3(n), dim3(256), 0, st, P, rank, order, bad);
}

void gs2m_launch_mark_visible(hipStream_t st, int P, const float* xyz, const float* viewmatrix,
                              unsigned char* present) {
    GS2M_LAUNCH(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, st, P, xyz, viewmatrix, present);
}

void gs2m_launch_pack_camera(hipStream_t st, CamUniform* cams, int slot, const float* viewmatrix,
                             const float* projmatrix, const float* campos, const float* bg, float tanfovx,
                             float tanfovy, int W, int H, int th) {
    GS2M_LAUNCH(k_pack_camera, dim3(1), dim3(64), 0, st, cams, slot, viewmatrix, projmatrix, campos, bg, tanfovx,
                tanfovy, W, H, th);
}
