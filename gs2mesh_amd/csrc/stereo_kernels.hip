// stereo_kernels.hip -- disparity -> depth + left-right consistency mask, the small
// data-parallel step between the stereo network and the TSDF (gs2mesh_utils/stereo_utils.py:132-133,
// 149-179).  Fused into one pass; outputs stay on the device and feed gs2m_tsdf_integrate directly
// (depth f32, mask u8) instead of going through depth.npy / occlusion_mask.npy.
//
// Per pixel (x, y), following Stereo.get_occlusion_mask's numpy arithmetic (int64 grid - float32
// disparity promotes to float64; astype(int32) truncates toward zero):
//   xp   = (int32)((double)x - L[y][x])                      x projected into the right image
//   xc   = clip(xp, 0, W-1)
//   xr   = clip((double)xc + R[y][xc], 0, W-1)                re-projected into the left image
//   occluded = |x - xr| > threshold  or  xp < 0  or  xp >= W
//   mask = !occluded                                           (1 = visible)
//   depth = (float)(fx * baseline) / L[y][x]                   (float32 division, stereo_utils.py:133)
#include <math.h>

#include "../../include/gs2mesh_amd.h"
#include "platform.h"

void gs2m_set_error(const char* fmt, ...);

GS2M_KERNEL void __launch_bounds__(256)
k_stereo_depth_occlusion(const float* __restrict__ disp_lr, const float* __restrict__ disp_rl, int W, int H,
                         float fb, double threshold, float* __restrict__ depth, unsigned char* __restrict__ mask) {
    const int x = (int)(blockIdx.x * 256u + threadIdx.x);
    const int y = (int)blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const float l = disp_lr[p];
    if (depth) depth[p] = fb / l;
    if (mask) {
        const int xp = (int)((double)x - (double)l);
        const int xc = xp < 0 ? 0 : (xp > W - 1 ? W - 1 : xp);
        double xr = (double)xc + (double)disp_rl[(size_t)y * W + xc];
        xr = xr < 0.0 ? 0.0 : (xr > (double)(W - 1) ? (double)(W - 1) : xr);
        const bool occluded = fabs((double)x - xr) > threshold || xp < 0 || xp >= W;
        mask[p] = occluded ? 0 : 1;
    }
}

extern "C" int gs2m_stereo_depth_occlusion(const float* disp_lr, const float* disp_rl, int width, int height,
                                           double fx_times_baseline, double occlusion_threshold, float* depth_out,
                                           uint8_t* mask_out, gs2m_stream stream) {
    if (!disp_lr || (mask_out && !disp_rl) || width <= 0 || height <= 0) {
        gs2m_set_error("gs2m_stereo_depth_occlusion: bad argument");
        return 1;
    }
    GS2M_LAUNCH(k_stereo_depth_occlusion, dim3((width + 255) / 256, height), dim3(256), 0, stream, disp_lr, disp_rl,
                width, height, (float)fx_times_baseline, occlusion_threshold, depth_out, mask_out);
    return 0;
}
