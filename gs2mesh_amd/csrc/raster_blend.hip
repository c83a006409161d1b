// raster_blend.hip -- compositing stage (default FMA contraction; VALU-bound inner loop).
#include <cstdlib>
#include "raster_blend.h"
#include "raster_internal.h"

// gy = rows of the 16 x 16 reference grid; variants 5 / 6 read instance lists of 16 x 32 tiles (GS2M_OPT_TILE_ROWS 2)
void gs2m_launch_blend(hipStream_t st, int variant, int nv, int gx, int gy, const unsigned long long* keys,
                       const unsigned* tile_start, const GeomRec* recs, const CamUniform* cams, int P,
                       unsigned cap, float* out_color, unsigned char* out_rgb8) {
    if (variant == 6) {  // 16 x 32 lists, one wave per 16 x 16 half
        const int tiles = gx * gy;
        GS2M_LAUNCH((k_blend_wave4e<4, 1, 2>), dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P,
                    cap, out_color, out_rgb8);
        return;
    }
    if (variant == 5) {  // 16 x 32 lists, one wave per 16 x 32 tile (8 pixels per lane)
        const int tiles = gx * ((gy + 1) / 2);
        GS2M_LAUNCH((k_blend_wave4e<4, 2, 2>), dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P,
                    cap, out_color, out_rgb8);
        return;
    }
    if (variant == 4) {
        const int tiles = gx * gy;
        GS2M_LAUNCH((k_blend_wave4e<4, 1, 1>), dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
                    out_color, out_rgb8);
        return;
    }
    if (variant == 3) {
        const int tiles = gx * gy;
        GS2M_LAUNCH(k_blend_wave4q, dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
                    out_color, out_rgb8);
        return;
    }
    if (variant == 2) {
        const int tiles = gx * gy;
        GS2M_LAUNCH(k_blend_wave4p, dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
                    out_color, out_rgb8);
        return;
    }
    if (variant == 1) {
        const int tiles = gx * gy;
        GS2M_LAUNCH(k_blend_wave4, dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
                    out_color, out_rgb8);
        return;
    }
    GS2M_LAUNCH(k_blend_tile256, dim3(gx, gy, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
                out_color, out_rgb8);
}
