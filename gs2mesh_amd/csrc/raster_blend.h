// raster_blend.h -- per-tile front-to-back alpha compositing (renderCUDA, forward.cu:261-374).
//
// Per pixel, for each instance of the tile in sorted order:
//   d = mean2D - pixel;  power = -0.5*(a dx^2 + c dy^2) - b dx dy;   power > 0      -> skip
//   alpha = min(0.99, opacity*exp(power));                           alpha < 1/255  -> skip
//   T' = T*(1-alpha);                                                T' < 1e-4      -> pixel done
//   C += rgb*alpha*T;  T = T'
// out = C + T*bg.  final_T / n_contrib (backward-only in the reference) are not written.
//
// Variant 0 (this file, k_blend_tile256): one 256-thread workgroup (4 waves) per 16x16 tile, one
// pixel per lane; instances are staged 256 at a time into LDS as full 36-B records (mean, conic,
// opacity AND rgb -- the reference gathers rgb from global memory per contributing pixel,
// forward.cu:355), the workgroup leaves as soon as all its pixels are saturated.
#pragma once
#include "raster_common.h"

GS2M_DEVICE unsigned char quantize_u8(float c) {
    // cv2.imwrite on a float image: saturate_cast<uchar>(v) = clamp(round-half-even(v), 0, 255)
    // (renderer_utils.py:389-390 multiplies by 255 first)
    float v = rintf(c * 255.0f);
    v = fminf(255.0f, fmaxf(0.0f, v));
    return (unsigned char)(int)v;
}

GS2M_KERNEL void __launch_bounds__(256)
k_blend_tile256(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
                const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
                float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8) {
    __shared__ float4 s_a[256];  // mx, my, ca, cb
    __shared__ float4 s_b[256];  // cc, op, r, g
    __shared__ float s_c[256];   // b
    const int v = (int)blockIdx.z;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * cam.gy;
    const int tile = (int)blockIdx.y * gx + (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int pxi = (int)blockIdx.x * GS2M_TILE + lx, pyi = (int)blockIdx.y * GS2M_TILE + ly;
    const bool inside = pxi < W && pyi < H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    unsigned r0 = tile_start[(size_t)v * (tiles + 1) + tile];
    unsigned r1 = tile_start[(size_t)v * (tiles + 1) + tile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    bool done = !inside;
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    int todo = (int)(r1 - r0);
    for (unsigned base = r0; base < r1; base += 256, todo -= 256) {
        const int num_done = gs2m_syncthreads_count(done ? 1 : 0);
        if (num_done == 256) break;
        if (base + (unsigned)tid < r1) {
            const unsigned gid = (unsigned)(kv[base + tid] & 0xffffffffull);
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            s_a[tid] = r4[0];
            s_b[tid] = r4[1];
            s_c[tid] = r4[2].x;
        }
        __syncthreads();
        const int nb = todo < 256 ? todo : 256;
        for (int j = 0; !done && j < nb; ++j) {
            const float4 A = s_a[j];
            const float4 B = s_b[j];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, B.y * gs2m_fast_exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            C0 += B.z * alpha * T;  // (rgb*alpha)*T, the reference's association (forward.cu:355)
            C1 += B.w * alpha * T;
            C2 += s_c[j] * alpha * T;
            T = test_T;
        }
    }
    if (inside) {
        const float o0 = C0 + T * cam.bg[0], o1 = C1 + T * cam.bg[1], o2 = C2 + T * cam.bg[2];
        const size_t pix = (size_t)pyi * W + pxi;
        if (out_color) {
            float* oc = out_color + (size_t)v * 3 * H * W;
            oc[pix] = o0;
            oc[(size_t)H * W + pix] = o1;
            oc[2 * (size_t)H * W + pix] = o2;
        }
        if (out_rgb8) {
            unsigned char* o8 = out_rgb8 + ((size_t)v * H * W + pix) * 3;
            o8[0] = quantize_u8(o0);
            o8[1] = quantize_u8(o1);
            o8[2] = quantize_u8(o2);
        }
    }
}
