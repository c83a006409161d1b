// raster_blend.hip -- compositing stage (default FMA contraction; VALU-bound inner loop).
#include "raster_blend.h"
#include "raster_internal.h"

// gy = rows of the 16 x 16 reference grid; tile_rows = 16 x 16 tiles per instance list (GS2M_OPT_TILE_ROWS)
//   variant 4: wave-per-tile kernel (k_blend_wave4e),
//   0: the reference's structure (256-thread workgroup per tile, 1 pixel per lane; 16 x 16 lists only)
int gs2m_launch_blend(hipStream_t st, int variant, int tile_rows, int nv, int gx, int gy, const unsigned long long* keys,
                      const unsigned* tile_start, GeomRecs recs, const CamUniform* cams, int P, unsigned cap,
                      float* out_color, unsigned char* out_rgb8, const int* rank, const unsigned* order, int mode,
                      unsigned long long* prof, int interleave_views) {
    const dim3 block(256);
    if (variant == 4) {
        const int ltiles = gx * ((gy + tile_rows - 1) / tile_rows);
        // schedule: ceil(chunks / 8) chunks per XCD, GS2M_SCHED_CHUNK lists per chunk, tile_rows waves per list, 4 waves per workgroup
        const int nch = ((gx + GS2M_SCHED_CW - 1) / GS2M_SCHED_CW) * ((ltiles / gx + GS2M_SCHED_CH - 1) / GS2M_SCHED_CH);
        const dim3 g2(8u * ((unsigned)(((nch + 7) / 8) * GS2M_SCHED_CHUNK * tile_rows + 3) / 4u), nv);
        if (mode >= 2) {   // GS2M_OPT_BLEND_MODE 2 / 3 (round 5): all four quadrants per instance, flag-free runs; views interleaved along x
            // interleaved: one row of blocks, view = (blockIdx.x / 8) % nv; else the views along blockIdx.y
            const int nvx = interleave_views && nv > 1 ? nv : 0;
            const dim3 g1 = nvx ? dim3(g2.x * (unsigned)nv, 1) : g2;
            if (prof) {   // GS2M_OPT_BLEND_PROFILE: the same kernel with s_memtime phase stamps
                if (tile_rows == 2) GS2M_LAUNCH((k_blend_wave4e<4, 2, 7, 2, 1>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, prof, nvx);
                else GS2M_LAUNCH((k_blend_wave4e<4, 1, 7, 2, 1>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, prof, nvx);
                return 0;
            }
            if (mode == 3) {   // mode 2 with the alpha cap for every instance (no run split at capped instances)
                if (tile_rows == 2) GS2M_LAUNCH((k_blend_wave4e<4, 2, 7, 3>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr, nvx);
                else GS2M_LAUNCH((k_blend_wave4e<4, 1, 7, 3>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr, nvx);
                return 0;
            }
            if (tile_rows == 2) GS2M_LAUNCH((k_blend_wave4e<4, 2, 7, 2>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr, nvx);
            else GS2M_LAUNCH((k_blend_wave4e<4, 1, 7, 2>), g1, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr, nvx);
            return 0;
        }
        // GS2M_OPT_BLEND_MODE 0: the lane-mask loop of rounds 1-4 (cross-check of mode 2)
        if (tile_rows == 2) GS2M_LAUNCH((k_blend_wave4e<4, 2, 7>), g2, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr);
        else GS2M_LAUNCH((k_blend_wave4e<4, 1, 7>), g2, block, 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank, order, nullptr);
        return 0;
    }
    if (variant == 0 && tile_rows == 1) {
        GS2M_LAUNCH(k_blend_tile256, dim3(gx, gy, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8, rank);
        return 0;
    }
    gs2m_set_error("blend variant %d is not available with GS2M_OPT_TILE_ROWS %d (variants: 0 [rows 1 only], 4)", variant, tile_rows);
    return 1;
}
