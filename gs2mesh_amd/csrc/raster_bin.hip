// raster_bin.hip -- tile offsets + per-tile depth sort (integer work).
#include "raster_sort.h"
#include "raster_internal.h"

void gs2m_launch_hist_colscan(hipStream_t st, int nv, unsigned* hist, int n_wg, int tiles, unsigned* tile_count) {
    GS2M_LAUNCH(k_hist_colscan, dim3((tiles + 63) / 64, nv), dim3(256), 0, st, hist, n_wg, tiles, tile_count);
}
void gs2m_launch_tile_scan(hipStream_t st, int nv, const unsigned* tile_count, unsigned* tile_start, int tiles,
                           ViewStatus* status, ViewStatus* sticky, unsigned cap) {
    GS2M_LAUNCH(k_tile_scan, dim3(nv), dim3(1024), 0, st, tile_count, tile_start, tiles, status, sticky, cap);
}
void gs2m_launch_sort_tiles(hipStream_t st, int nv, unsigned long long* keys, unsigned long long* tmp,
                            const unsigned* tile_start, int tiles, unsigned cap) {
    GS2M_LAUNCH(k_sort_tiles_small, dim3(tiles, nv), dim3(64), 0, st, keys, tile_start, tiles, cap);
    GS2M_LAUNCH(k_sort_tiles, dim3(tiles, nv), dim3(256), 0, st, keys, tmp, tile_start, tiles, cap);
}
