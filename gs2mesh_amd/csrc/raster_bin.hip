// raster_bin.hip -- tile offsets + per-tile depth sort (integer work).
#include "raster_sort.h"
#include "raster_internal.h"

void gs2m_launch_hist_colscan(hipStream_t st, int nv, unsigned* hist, int n_wg, int tiles, unsigned* tile_count) {
    GS2M_LAUNCH(k_hist_colscan, dim3((tiles + 63) / 64, nv), dim3(64 * GS2M_COLSCAN_SEGS), 0, st, hist, n_wg, tiles, tile_count);
}
void gs2m_launch_tile_scan(hipStream_t st, int nv, const unsigned* tile_count, unsigned* tile_start, int tiles, int gx,
                           ViewStatus* status, ViewStatus* sticky, unsigned cap, unsigned* sort_lists) {
    GS2M_LAUNCH(k_tile_scan, dim3(nv), dim3(1024), 0, st, tile_count, tile_start, tiles, gx, status, sticky, cap, sort_lists);
}
size_t gs2m_sort_lists_words(int nv, int tiles) { return (size_t)nv * (GS2M_SORT_CLASSES * (tiles + 1) + tiles); }  // class lists + schedule
void gs2m_launch_sort_tiles(hipStream_t st, int nv, unsigned long long* keys, unsigned long long* tmp,
                            const unsigned* tile_start, int tiles, unsigned cap, const unsigned* sort_lists, const int* class_hint) {
    // <= 512 instances: one wave per tile; larger lists are walked from the work lists k_tile_scan wrote, by grids
    // sized for residency (4 / 2 / 2 workgroups per CU by their LDS), not one mostly idle workgroup per tile.
    // class_hint (may be null) = lists per size class the PREVIOUS call on this handle found (read back with its status,
    // possibly stale): the class kernels walk their lists grid-stride, so any grid >= 1 is correct -- the hint only keeps a
    // class that is (almost) empty from launching hundreds of workgroups of 40-80 KiB LDS that wait for CU space just to
    // find nothing to do (C2 has no list above 512: 3 x ~35 us of stream latency per pair under the pipelined load).
    // every size class was empty last time: no class kernels; the first workgroups of the small-list kernel still sort whatever
    // larger lists they find, exactly and in bounded time even when the hint is stale (sort_class_lists_rank)
    const int fold = class_hint && class_hint[0] == 0 && class_hint[1] == 0 && class_hint[2] == 0;
    GS2M_LAUNCH(k_sort_tiles_small<4>, dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, tiles, cap, tmp, sort_lists, fold);
    if (fold) return;
    const int full[3] = {tiles < 1024 ? tiles : 1024, tiles < 512 ? tiles : 512, tiles < 512 ? tiles : 512};
    int g[3];
    for (int c = 0; c < 3; ++c) {
        g[c] = full[c];
        if (class_hint && class_hint[c] >= 0) {
            const int want = class_hint[c] + class_hint[c] / 2 + 16;  // 1.5 x the last count + a floor of 16 workgroups
            g[c] = want < full[c] ? want : full[c];
        }
        if (g[c] < 1) g[c] = 1;
    }
    GS2M_LAUNCH(k_sort_tiles_bucket_4096x8, dim3(g[0], nv), dim3(512), 0, st, keys, tile_start, tiles, cap, sort_lists);
    GS2M_LAUNCH(k_sort_tiles_bucket_8192x8, dim3(g[1], nv), dim3(1024), 0, st, keys, tile_start, tiles, cap, sort_lists);
    GS2M_LAUNCH(k_sort_tiles, dim3(g[2], nv), dim3(256), 0, st, keys, tmp, tile_start, tiles, cap, sort_lists);
}
