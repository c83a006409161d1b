// raster_blend.hip -- compositing stage (default FMA contraction; VALU-bound inner loop).
#include <cstdlib>
#include "raster_blend.h"
#include "raster_internal.h"

void gs2m_launch_blend(hipStream_t st, int variant, int nv, int gx, int gy, const unsigned long long* keys,
                       const unsigned* tile_start, const GeomRec* recs, const CamUniform* cams, int P,
                       unsigned cap, float* out_color, unsigned char* out_rgb8) {
    if (variant == 4) {
        const int tiles = gx * gy;
        GS2M_LAUNCH((k_blend_wave4e<4>), dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
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
