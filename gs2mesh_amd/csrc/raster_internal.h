// raster_internal.h -- launch wrappers shared between the translation units of the rasteriser.
// Each .hip file is compiled on its own so that per-stage floating-point contraction can
// differ (projection: -ffp-contract=off for a literal IEEE sequence; blend: default fast
// contraction, FMA).
#pragma once
#include "raster_common.h"

size_t gs2m_scatter_lds_bytes(int nv, int tiles, int threads);
// host_cams: null (the uniforms are already in `cams`, device memory) or the nv * pairs host-side uniforms of the pass (carried in
// the launch packet; the kernel also stores them to `cams`)
void gs2m_launch_project(int nv, int pairs, hipStream_t st, const GaussIn& g, CamUniform* cams, GeomRecs recs, int* radii,
                         int exact_cull, const CamUniform* host_cams, int shared_read);
int gs2m_launch_count_tiles(int nv, int pairs, int n_wg, int threads, size_t lds_bytes, hipStream_t st, GeomRecs recs, int P,
                            const CamUniform* cams, int chunk, unsigned* hist, unsigned long long* tilemask,
                            int exact_cull, int interleave, int lane_tiles);
int gs2m_count_threads(int chunk, int max_threads);
size_t gs2m_count_lds_bytes(int nv, int tiles, int threads);
int gs2m_launch_scatter(int nv, int pairs, int n_wg, int threads, size_t lds_bytes, hipStream_t st, GeomRecs recs, int P,
                        const CamUniform* cams, int chunk, const unsigned* hist, const unsigned* tile_start,
                        const unsigned long long* tilemask, unsigned long long* keys, unsigned cap, int exact_cull,
                        const int* ids, int interleave, int lane_tiles);
void gs2m_launch_pack_sh(hipStream_t st, int P, const float* shs, const float* shs_rest, float* packed, const int* order);
void gs2m_launch_pack_model(hipStream_t st, int P, const int* order, const float* xyz, const float* scales, const float* rots,
                            const float* opac, float* p_xyz, float* p_scales, float* p_rots, float* p_opac, int* rank,
                            unsigned* bad);
void gs2m_launch_mark_visible(hipStream_t st, int P, const float* xyz, const float* viewmatrix,
                              unsigned char* present);
void gs2m_launch_pack_camera(hipStream_t st, CamUniform* cams, int slot, const float* viewmatrix,
                             const float* projmatrix, const float* campos, const float* bg, float tanfovx,
                             float tanfovy, int W, int H, int th);

void gs2m_launch_hist_colscan(hipStream_t st, int nv, unsigned* hist, int n_wg, int tiles, unsigned* tile_count);
void gs2m_launch_tile_scan(hipStream_t st, int nv, const unsigned* tile_count, unsigned* tile_start, int tiles, int gx,
                           ViewStatus* status, ViewStatus* sticky, unsigned cap, unsigned* sort_lists);
size_t gs2m_sort_lists_words(int nv, int tiles);
void gs2m_launch_sort_tiles(hipStream_t st, int nv, unsigned long long* keys, unsigned long long* tmp,
                            const unsigned* tile_start, int tiles, unsigned cap, const unsigned* sort_lists, const int* class_hint);
int gs2m_launch_blend(hipStream_t st, int variant, int tile_rows, int nv, int gx, int gy, const unsigned long long* keys,
                      const unsigned* tile_start, GeomRecs recs, const CamUniform* cams, int P,
                      unsigned cap, float* out_color, unsigned char* out_rgb8, const int* rank, const unsigned* order,
                      int mode, unsigned long long* prof, int interleave_views);

// error plumbing (common_api.hip)
void gs2m_set_error(const char* fmt, ...);
