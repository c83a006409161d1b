// raster_blend.hip -- compositing stage (default FMA contraction; VALU-bound inner loop).
#include <cstdlib>
#include "raster_blend.h"
#include "raster_internal.h"

void gs2m_launch_blend(hipStream_t st, int variant, int nv, int gx, int gy, const unsigned long long* keys,
                       const unsigned* tile_start, const GeomRec* recs, const CamUniform* cams, int P,
                       unsigned cap, float* out_color, unsigned char* out_rgb8) {
#ifdef GS2M_DEV_ABLATE
    if (variant >= 40) {
        const int tiles = gx * gy;
#define ABL_LAUNCH(A, W) GS2M_LAUNCH((k_blend_wave4e<A, W>), dim3((tiles + W - 1) / W, nv), dim3(64 * W), 0, st, keys, tile_start, recs, cams, P, cap, out_color, out_rgb8)
        if (variant == 41) ABL_LAUNCH(1, 4);
        if (variant == 42) ABL_LAUNCH(2, 4);
        if (variant == 43) ABL_LAUNCH(3, 4);
        if (variant == 44) ABL_LAUNCH(0, 1);
        if (variant == 45) ABL_LAUNCH(0, 2);
        if (variant == 46) ABL_LAUNCH(0, 8);
        return;
    }
#endif
    if (variant == 4) {
        const int tiles = gx * gy;
        GS2M_LAUNCH((k_blend_wave4e<0, 4>), dim3((tiles + 3) / 4, nv), dim3(256), 0, st, keys, tile_start, recs, cams, P, cap,
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
