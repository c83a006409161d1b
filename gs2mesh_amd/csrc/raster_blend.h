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
#include <type_traits>
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

// ---------------------------------------------------------------------------------------------
// Variant 1 (k_blend_wave4): ONE WAVE PER 16x16 TILE, 4 pixels per lane (column lane&15, rows
// (lane>>4) + 4k).  A 256-thread workgroup is four independent waves = four consecutive tiles:
// no __syncthreads at all, a wave leaves as soon as ITS 256 pixels are saturated.
//   * per instance the wave pays ONE pair of LDS broadcast reads (2 x ds_read_b128) for 256 pixel
//     evaluations instead of four (variant 0: one pair per 64 pixels) -> VALU-bound, not LDS-bound;
//   * dx, a*dx*dx and b*dx are shared by the lane's 4 pixels;
//   * the accumulate path is entered only if some lane of the wave has a contributing pixel;
//   * the next 64 records are gathered into registers while the current 64 are composited;
//   * workgroup -> tile-group mapping is XCD-aware (block b runs on XCD b%8: give each XCD a
//     contiguous run of tiles so neighbouring tiles, which share Gaussians, hit the same L2).
// ---------------------------------------------------------------------------------------------
GS2M_KERNEL void __launch_bounds__(256)
k_blend_wave4(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
              const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
              float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8) {
    __shared__ float4 s_a[4][64];
    __shared__ float4 s_b[4][64];
    __shared__ float s_c[4][64];
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int v = (int)blockIdx.y;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * cam.gy;
    // bijective XCD swizzle (guide T1)
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, idx = bid / 8u;
    const unsigned grp = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
    const int tile = (int)(grp * 4u) + wave;
    if (tile >= tiles) return;  // whole wave; no workgroup barriers in this kernel
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * GS2M_TILE + (lane & 15);
    const int py0 = ty * GS2M_TILE + (lane >> 4);
    const float pxf = (float)pxi;
    float pyf[4];
    bool done[4];
    float T[4], C0[4], C1[4], C2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pyf[k] = (float)(py0 + 4 * k);
        done[k] = !(pxi < W && py0 + 4 * k < H);
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
    }
    unsigned r0 = tile_start[(size_t)v * (tiles + 1) + tile];
    unsigned r1 = tile_start[(size_t)v * (tiles + 1) + tile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    float4 ra, rb;
    float rc = 0.0f;
    ra.x = ra.y = ra.z = ra.w = 0.0f;
    rb = ra;
    unsigned base = r0;
    if (base + (unsigned)lane < r1) {
        const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2].x;
    }
    while (base < r1) {
        const bool all_done = done[0] && done[1] && done[2] && done[3];
        if (gs2m_ballot(all_done ? 0 : 1) == 0ull) break;
        gs2m_wave_sync();  // the previous batch has been read by every lane
        s_a[wave][lane] = ra;
        s_b[wave][lane] = rb;
        s_c[wave][lane] = rc;
        gs2m_wave_sync();
        const int nb = (int)(r1 - base) < 64 ? (int)(r1 - base) : 64;
        base += 64u;
        if (base + (unsigned)lane < r1) {  // gather the next batch while this one is composited
            const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            ra = r4[0];
            rb = r4[1];
            rc = r4[2].x;
        }
        for (int j = 0; j < nb; ++j) {
            const float4 A = s_a[wave][j];
            const float4 B = s_b[wave][j];
            const float dx = A.x - pxf;
            const float adx2 = A.z * dx * dx;
            const float bdx = A.w * dx;
            float alpha[4];
            bool hit[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = A.y - pyf[k];
                const float power = -0.5f * (adx2 + B.x * dy * dy) - bdx * dy;
                alpha[k] = fminf(0.99f, B.y * gs2m_fast_exp(power));
                hit[k] = !done[k] && !(power > 0.0f) && !(alpha[k] < 1.0f / 255.0f);
                any = any || hit[k];
            }
            if (gs2m_ballot(any ? 1 : 0) != 0ull) {
                const float cb = s_c[wave][j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (hit[k]) {
                        const float test_T = T[k] * (1.0f - alpha[k]);
                        if (test_T < 0.0001f) {
                            done[k] = true;
                        } else {
                            C0[k] += B.z * alpha[k] * T[k];
                            C1[k] += B.w * alpha[k] * T[k];
                            C2[k] += cb * alpha[k] * T[k];
                            T[k] = test_T;
                        }
                    }
                }
            }
        }
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pyi = py0 + 4 * k;
        if (pxi < W && pyi < H) {
            const float o0 = C0[k] + T[k] * cam.bg[0], o1 = C1[k] + T[k] * cam.bg[1], o2 = C2[k] + T[k] * cam.bg[2];
            const size_t pix = (size_t)pyi * W + pxi;
            if (out_color) {
                float* oc = out_color + (size_t)v * 3 * plane;
                oc[pix] = o0;
                oc[plane + pix] = o1;
                oc[2 * plane + pix] = o2;
            }
            if (out_rgb8) {
                unsigned char* o8 = out_rgb8 + ((size_t)v * plane + pix) * 3;
                o8[0] = quantize_u8(o0);
                o8[1] = quantize_u8(o1);
                o8[2] = quantize_u8(o2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 2 (k_blend_wave4p): variant 1's layout with a cheaper inner loop.  The kernel is
// VALU-bound (measured: ~300 issue cycles per instance x 256 pixels in variant 1), so the loop is
// rebuilt around instruction count:
//   * alpha >= 1/255  <=>  power >= -ln(255*opacity).  A per-instance limit lim = -ln(255 o) - 1e-4
//     (computed once per instance while staging, kept in LDS) gives a conservative one-compare
//     pre-filter per pixel; exp() and the exact alpha test of the reference run only inside the
//     accumulate path, and only for the 16x4 strips (the lane's pixel index k) in which some lane
//     passed the pre-filter;
//   * a finished pixel is parked at y = 1e18 (its power becomes hugely negative), which removes
//     every per-pixel "done" flag from the hot path;
//   * the accumulate path is straight-line predicated arithmetic (weights selected to 0), no
//     per-pixel exec-mask regions.
// Decisions are the reference's (same alpha, same thresholds, same order): the pre-filter only
// skips pixels whose exact test would fail.
// ---------------------------------------------------------------------------------------------
#define GS2M_PARKED 1.0e18f

GS2M_KERNEL void __launch_bounds__(256)
k_blend_wave4p(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
               const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
               float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8) {
    __shared__ float4 s_a[4][64];  // mx, my, ca, cb
    __shared__ float4 s_b[4][64];  // cc, op, r, g
    __shared__ float2 s_c[4][64];  // b, lim
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int v = (int)blockIdx.y;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * cam.gy;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, idx = bid / 8u;
    const unsigned grp = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
    const int tile = (int)(grp * 4u) + wave;
    if (tile >= tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * GS2M_TILE + (lane & 15);
    const int py0 = ty * GS2M_TILE + (lane >> 4);
    const float pxf = (float)pxi;
    float pyf[4], T[4], C0[4], C1[4], C2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pyf[k] = (pxi < W && py0 + 4 * k < H) ? (float)(py0 + 4 * k) : GS2M_PARKED;
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
    }
    unsigned r0 = tile_start[(size_t)v * (tiles + 1) + tile];
    unsigned r1 = tile_start[(size_t)v * (tiles + 1) + tile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    float4 ra, rb;
    float rc = 0.0f;
    ra.x = ra.y = ra.z = ra.w = 0.0f;
    rb = ra;
    unsigned base = r0;
    if (base + (unsigned)lane < r1) {
        const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2].x;
    }
    while (base < r1) {
        const bool live = pyf[0] < 1.0e17f || pyf[1] < 1.0e17f || pyf[2] < 1.0e17f || pyf[3] < 1.0e17f;
        if (gs2m_ballot(live ? 1 : 0) == 0ull) break;
        gs2m_wave_sync();
        s_a[wave][lane] = ra;
        s_b[wave][lane] = rb;
        float2 cl;
        cl.x = rc;
        // alpha >= 1/255 <=> power >= -ln(255*o); opacity*255 < 1 never contributes (lim > 0 >= power)
        cl.y = -gs2m_fast_log(rb.y * 255.0f) - 1.0e-4f;
        s_c[wave][lane] = cl;
        gs2m_wave_sync();
        const int nb = (int)(r1 - base) < 64 ? (int)(r1 - base) : 64;
        base += 64u;
        if (base + (unsigned)lane < r1) {
            const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            ra = r4[0];
            rb = r4[1];
            rc = r4[2].x;
        }
        for (int j = 0; j < nb; ++j) {
            const float4 A = s_a[wave][j];
            const float4 B = s_b[wave][j];
            const float2 CL = s_c[wave][j];
            const float dx = A.x - pxf;
            const float adx2 = A.z * dx * dx;
            const float bdx = A.w * dx;
            float power[4];
            bool cand[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = A.y - pyf[k];
                power[k] = -0.5f * (adx2 + B.x * dy * dy) - bdx * dy;
                cand[k] = power[k] >= CL.y && !(power[k] > 0.0f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (gs2m_ballot(cand[k] ? 1 : 0) != 0ull) {  // wave-uniform: strip k has a candidate
                    const float alpha = fminf(0.99f, B.y * gs2m_fast_exp(power[k]));
                    const bool hit = cand[k] && !(alpha < 1.0f / 255.0f);
                    const float test_T = T[k] * (1.0f - alpha);
                    const bool sat = hit && test_T < 0.0001f;
                    const bool acc = hit && !sat;
                    const float w = acc ? alpha : 0.0f;
                    C0[k] += B.z * w * T[k];
                    C1[k] += B.w * w * T[k];
                    C2[k] += CL.x * w * T[k];
                    T[k] = acc ? test_T : T[k];
                    pyf[k] = sat ? GS2M_PARKED : pyf[k];
                }
            }
        }
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pyi = py0 + 4 * k;
        if (pxi < W && pyi < H) {
            const float o0 = C0[k] + T[k] * cam.bg[0], o1 = C1[k] + T[k] * cam.bg[1], o2 = C2[k] + T[k] * cam.bg[2];
            const size_t pix = (size_t)pyi * W + pxi;
            if (out_color) {
                float* oc = out_color + (size_t)v * 3 * plane;
                oc[pix] = o0;
                oc[plane + pix] = o1;
                oc[2 * plane + pix] = o2;
            }
            if (out_rgb8) {
                unsigned char* o8 = out_rgb8 + ((size_t)v * plane + pix) * 3;
                o8[0] = quantize_u8(o0);
                o8[1] = quantize_u8(o1);
                o8[2] = quantize_u8(o2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 3 (k_blend_wave4q): variant 2 with the lane's four pixels laid out as the four 8x8
// QUADRANTS of the tile (lane l: x = l&7, y = l>>3, plus (0|8, 0|8)), and a per-instance 4-bit
// quadrant mask computed while staging from the bounding box of the alpha >= 1/255 ellipse
// (|dx| <= sqrt(2 t cov_xx), |dy| <= sqrt(2 t cov_yy), t = ln(255 o); cov = conic^-1).  A quadrant
// whose 8x8 pixels lie outside that box is skipped with a scalar branch: for a splat of radius r the
// evaluated pixel groups drop from 4 per touched tile to ~(1 + 2r/8)^2 / (1 + 2r/16)^2 (44 % at
// r = 15 px, 40 % at r = 5 px).  Same decisions as the reference: the mask, like the pre-filter, only
// removes pixels whose exact alpha test would fail.
// ---------------------------------------------------------------------------------------------
GS2M_KERNEL void __launch_bounds__(256)
k_blend_wave4q(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
               const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
               float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8) {
    __shared__ float4 s_a[4][64];  // mx, my, ca, cb
    __shared__ float4 s_b[4][64];  // cc, op, r, g
    __shared__ float4 s_c[4][64];  // b, lim, quadrant mask (bits), -
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int v = (int)blockIdx.y;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * cam.gy;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, idx = bid / 8u;
    const unsigned grp = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
    const int tile = (int)(grp * 4u) + wave;
    if (tile >= tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int px0 = tx * GS2M_TILE + (lane & 7), py0 = ty * GS2M_TILE + (lane >> 3);
    // pixel k: x = px0 + 8*(k&1), y = py0 + 8*(k>>1)
    const float pxf0 = (float)px0, pxf1 = (float)(px0 + 8);
    float pyf[4], T[4], C0[4], C1[4], C2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
        pyf[k] = (x < W && y < H) ? (float)y : GS2M_PARKED;
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
    }
    // quadrant pixel ranges (tile-uniform)
    const float qx0 = (float)(tx * GS2M_TILE), qy0 = (float)(ty * GS2M_TILE);
    unsigned r0 = tile_start[(size_t)v * (tiles + 1) + tile];
    unsigned r1 = tile_start[(size_t)v * (tiles + 1) + tile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    float4 ra, rb;
    float rc = 0.0f;
    ra.x = ra.y = ra.z = ra.w = 0.0f;
    rb = ra;
    unsigned base = r0;
    if (base + (unsigned)lane < r1) {
        const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2].x;
    }
    while (base < r1) {
        const bool live = pyf[0] < 1.0e17f || pyf[1] < 1.0e17f || pyf[2] < 1.0e17f || pyf[3] < 1.0e17f;
        if (gs2m_ballot(live ? 1 : 0) == 0ull) break;
        gs2m_wave_sync();
        {
            // per-instance constants, computed once by the staging lane
            const float lim = -gs2m_fast_log(rb.y * 255.0f) - 1.0e-4f;   // alpha >= 1/255 <=> power >= lim
            const float t2 = fmaxf(-2.0f * lim, 0.0f);                    // 2 ln(255 o) (+ margin)
            const float det = ra.z * rb.x - ra.w * ra.w;                  // conic determinant (> 0)
            const float inv = 1.0f / det;
            const float hx = sqrtf(t2 * rb.x * inv) * 1.001f + 0.01f;     // cov_xx = cc/det
            const float hy = sqrtf(t2 * ra.z * inv) * 1.001f + 0.01f;     // cov_yy = ca/det
            const bool xl = ra.x - hx <= qx0 + 7.0f, xr = ra.x + hx >= qx0 + 8.0f;
            const bool yt = ra.y - hy <= qy0 + 7.0f, yb = ra.y + hy >= qy0 + 8.0f;
            unsigned m = 0u;
            if (!(lim > 0.0f) && det > 0.0f) {
                if (xl && yt) m |= 1u;
                if (xr && yt) m |= 2u;
                if (xl && yb) m |= 4u;
                if (xr && yb) m |= 8u;
            } else if (!(det > 0.0f)) {
                m = 15u;  // degenerate conic: no box, test every pixel
            }
            float4 cl;
            cl.x = rc;
            cl.y = lim;
            cl.z = __uint_as_float(m);
            cl.w = 0.0f;
            s_c[wave][lane] = cl;
            // conic pre-scaled by -0.5 (exact: power of two), so that
            // power = -0.5*(a dx^2 + c dy^2) - b dx dy = (a' dx^2 + c' dy^2) - b dx dy bit for bit
            float4 sa = ra, sb = rb;
            sa.z = -0.5f * ra.z;
            sb.x = -0.5f * rb.x;
            s_a[wave][lane] = sa;
            s_b[wave][lane] = sb;
        }
        gs2m_wave_sync();
        const int nb = (int)(r1 - base) < 64 ? (int)(r1 - base) : 64;
        base += 64u;
        if (base + (unsigned)lane < r1) {
            const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            ra = r4[0];
            rb = r4[1];
            rc = r4[2].x;
        }
        for (int j = 0; j < nb; ++j) {
            const float4 CL = s_c[wave][j];
            const int qm = gs2m_uniform((int)__float_as_uint(CL.z));
            if (qm == 0) continue;
            const float4 A = s_a[wave][j];
            const float4 B = s_b[wave][j];
            const float dx0 = A.x - pxf0, dx1 = A.x - pxf1;
            const float adx2[2] = {A.z * dx0 * dx0, A.z * dx1 * dx1};
            const float bdx[2] = {A.w * dx0, A.w * dx1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (qm & (1 << k)) {  // scalar branch: quadrant k intersects the splat's box
                    const float dy = A.y - pyf[k];
                    const float power = (adx2[k & 1] + B.x * dy * dy) - bdx[k & 1] * dy;
                    const bool cand = power >= CL.y && !(power > 0.0f);
                    if (gs2m_ballot(cand ? 1 : 0) != 0ull) {
                        const float alpha = fminf(0.99f, B.y * gs2m_fast_exp(power));
                        const bool hit = cand && !(alpha < 1.0f / 255.0f);
                        const float test_T = T[k] * (1.0f - alpha);
                        const bool sat = hit && test_T < 0.0001f;
                        const bool acc = hit && !sat;
                        const float w = acc ? alpha : 0.0f;
                        C0[k] += B.z * w * T[k];
                        C1[k] += B.w * w * T[k];
                        C2[k] += CL.x * w * T[k];
                        T[k] = acc ? test_T : T[k];
                        pyf[k] = sat ? GS2M_PARKED : pyf[k];
                    }
                }
            }
        }
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pxi = px0 + 8 * (k & 1), pyi = py0 + 8 * (k >> 1);
        if (pxi < W && pyi < H) {
            const float o0 = C0[k] + T[k] * cam.bg[0], o1 = C1[k] + T[k] * cam.bg[1], o2 = C2[k] + T[k] * cam.bg[2];
            const size_t pix = (size_t)pyi * W + pxi;
            if (out_color) {
                float* oc = out_color + (size_t)v * 3 * plane;
                oc[pix] = o0;
                oc[plane + pix] = o1;
                oc[2 * plane + pix] = o2;
            }
            if (out_rgb8) {
                unsigned char* o8 = out_rgb8 + ((size_t)v * plane + pix) * 3;
                o8[0] = quantize_u8(o0);
                o8[1] = quantize_u8(o1);
                o8[2] = quantize_u8(o2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 4 (k_blend_wave4e): variant 3 with the exponent kept in the log2 domain.  The staging lane
// folds log2(e) into the conic and log2(opacity) into the constant term,
//   q = (a' dx^2 + log2 o) + c' dy^2 - b' dx dy,   alpha = min(0.99, exp2(q))        (1 v_exp_f32),
// so the per-pixel path has no multiply by log2(e) and none by the opacity; the alpha >= 1/255
// pre-filter becomes q >= -log2(255) (constant), power > 0 becomes q > log2 o.  The accumulate path is
// written for the fewest VALU ops: test_T = fma(-T, alpha, T), one w*T product shared by the three
// colour FMAs.  Batch bounds are wave-uniform scalars (scalar loop control).  Differences from
// variant 3 are roundings of ~1 ulp in q (|q| <= 8) -> relative 1e-6 in alpha; same tolerance.
// ---------------------------------------------------------------------------------------------
struct alignas(16) BlendInst {
    float4 a, b;
    float2 c, pad;
};

template <int WPB, int ROWS, int LROWS>
GS2M_KERNEL void __launch_bounds__(64 * WPB)
k_blend_wave4e(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
               const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
               float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8) {
    // 48-B staged instance: a = {mx, my, a' = -0.5 log2e ca, b' = log2e cb}, b = {c' = -0.5 log2e cc, log2 o, r, g},
    // c = {b, quadrant mask (bits)}; one LDS address + immediate offsets per instance.  2 pad slots: the
    // software-pipelined reads run up to 2 instances ahead.
    // ROWS = reference tiles (16 x 16) composited by one wave, stacked vertically: 1 = 4 pixels per lane, 2 = 8.
    // LROWS = reference tiles per instance list (GS2M_OPT_TILE_ROWS): with 2 the binning stages handle ~30 % fewer
    // (Gaussian, tile) instances.  <ROWS 1, LROWS 2>: two waves walk the same 16 x 32 list, each compositing its own
    // 16 x 16 half (instances that miss the half cost a skipped iteration); <2, 2>: one wave, 8 pixels per lane
    // (fewer instructions, but 103 VGPRs -> 4 waves/SIMD: measured slower).  A quadrant outside the instance's
    // 16 x 16 tile rect is masked, so the reference's rect still bounds every contribution.
    constexpr int NQ = 4 * ROWS, TH = GS2M_TILE * ROWS;
    __shared__ BlendInst s_i[WPB][64 + 2];
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int v = (int)blockIdx.y;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * ((cam.gy + ROWS - 1) / ROWS);        // tiles composited by waves
    const int ltiles = gx * ((cam.gy + LROWS - 1) / LROWS);    // instance lists
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned qq = nwg / 8u, rr = nwg % 8u, xcd = bid % 8u, idx = bid / 8u;
    const unsigned grp = (xcd < rr ? xcd * (qq + 1u) : rr * (qq + 1u) + (xcd - rr) * qq) + idx;
    const int tile = gs2m_uniform((int)(grp * (unsigned)WPB) + wave);
    if (tile >= tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int px0 = tx * GS2M_TILE + (lane & 7), py0 = ty * TH + (lane >> 3);
    float pxf0 = (float)px0, pxf1 = (float)(px0 + 8);
    GS2M_KEEP_F32(pxf0);
    GS2M_KEEP_F32(pxf1);
    float pyf[NQ], T[NQ], C0[NQ], C1[NQ], C2[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
        pyf[k] = (x < W && y < H) ? (float)y : GS2M_PARKED;
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
    }
    const float qx0 = (float)(tx * GS2M_TILE), qy0 = (float)(ty * TH);
    const int ltile = (ty * ROWS / LROWS) * gx + tx;
    unsigned r0 = tile_start[(size_t)v * (ltiles + 1) + ltile];
    unsigned r1 = tile_start[(size_t)v * (ltiles + 1) + ltile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    r0 = (unsigned)gs2m_uniform((int)r0);
    r1 = (unsigned)gs2m_uniform((int)r1);
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    const float LOG2E = 1.44269504088896340736f;
    const float QMIN = -7.99435343685885793770f;  // -log2(255): alpha >= 1/255 <=> q >= QMIN (decided in the log2 domain)
    float4 ra, rb, rc;  // the three 16-B vectors of a GeomRec
    ra.x = ra.y = ra.z = ra.w = 0.0f;
    rb = ra;
    rc = ra;
    rb.y = 1.0f;
    unsigned base = r0;
    if (base + (unsigned)lane < r1) {
        const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2];
    }
    while (base < r1) {
        bool live = false;
#pragma unroll
        for (int k = 0; k < NQ; ++k) live = live || pyf[k] < 1.0e17f;
        if (gs2m_ballot(live ? 1 : 0) == 0ull) break;
        gs2m_wave_sync();
        int nb_staged = 0;
        {
            const float lo = gs2m_fast_log2(rb.y);                         // log2 o
            const float t2 = fmaxf(2.0f * (gs2m_fast_log(rb.y * 255.0f) + 1.0e-4f), 0.0f);  // 2 ln(255 o) (+ margin)
            const float det = ra.z * rb.x - ra.w * ra.w;                   // conic determinant (> 0)
            const float inv = 1.0f / det;
            const float hx = sqrtf(t2 * rb.x * inv) * 1.001f + 0.01f;      // cov_xx = cc/det
            const float hy = sqrtf(t2 * ra.z * inv) * 1.001f + 0.01f;      // cov_yy = ca/det
            const bool xl = ra.x - hx <= qx0 + 7.0f, xr = ra.x + hx >= qx0 + 8.0f;
            const bool box = rb.y * 255.0f >= 0.9999f && det > 0.0f;
            const bool degenerate = !(det > 0.0f);  // no box: test every pixel of the tile rect
            const int ry0 = (int)(__float_as_uint(rc.z) >> 16), ry1 = (int)(__float_as_uint(rc.w) >> 16);  // rect rows, 16-px units
            unsigned m = 0u;
#pragma unroll
            for (int qr = 0; qr < 2 * ROWS; ++qr) {  // 8-pixel quadrant rows of the tile
                const float top = qy0 + 8.0f * (float)qr;
                bool row = degenerate || (box && ra.y - hy <= top + 7.0f && ra.y + hy >= top);
                if (LROWS > 1) {  // the quadrant must lie in a 16 x 16 tile of the instance's rect
                    const int y16 = ty * ROWS + (qr >> 1);
                    row = row && y16 >= ry0 && y16 < ry1;
                }
                if (row && (degenerate || xl)) m |= 1u << (2 * qr);
                if (row && (degenerate || xr)) m |= 2u << (2 * qr);
            }
            BlendInst bi;
            bi.a = ra;
            bi.b = rb;
            bi.a.z = (-0.5f * LOG2E) * ra.z;
            bi.a.w = LOG2E * ra.w;
            bi.b.x = (-0.5f * LOG2E) * rb.x;
            bi.b.y = lo;
            bi.c.x = rc.x;
            bi.c.y = __uint_as_float(m);
            bi.pad = bi.c;
            if (LROWS > ROWS) {
                // the list also serves the other half of the 16 x 32 tile: stage only the instances that reach this
                // half (ballot compaction), so the compositing loop never iterates over the others
                const bool mine = m != 0u && base + (unsigned)lane < r1;
                const unsigned long long keep = gs2m_ballot(mine ? 1 : 0);
                if (mine) s_i[wave][gs2m_popc64(keep & ((1ull << lane) - 1ull))] = bi;
                nb_staged = gs2m_popc64(keep);
            } else {
                s_i[wave][lane] = bi;
            }
        }
        gs2m_wave_sync();
        const int nb = LROWS > ROWS ? nb_staged : ((int)(r1 - base) < 64 ? (int)(r1 - base) : 64);
        base += 64u;
        if (base + (unsigned)lane < r1) {
            const unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            ra = r4[0];
            rb = r4[1];
            rc = r4[2];
        }
        // software-pipelined broadcast reads, unrolled by two with ping-pong registers: instance j+1's
        // constants are in flight while instance j is composited (no LDS wait on the critical path).
        auto step = [&](const int qm, const float2 CL, const float4 A, const float4 B) __attribute__((always_inline)) {
            if (qm != 0) {
                const float dx0 = A.x - pxf0, dx1 = A.x - pxf1;
                const float e[2] = {fmaf(A.z * dx0, dx0, B.y), fmaf(A.z * dx1, dx1, B.y)};
                const float nbdx[2] = {-(A.w * dx0), -(A.w * dx1)};
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    if (qm & (1 << k)) {  // scalar branch: quadrant k intersects the splat's box
                        const float dy = A.y - pyf[k];
                        // q = e + dy (c' dy - b' dx): two FMAs
                        const float qv = fmaf(fmaf(B.x, dy, nbdx[k & 1]), dy, e[k & 1]);
                        const bool pre = qv >= QMIN;
                        if (gs2m_ballot(pre ? 1 : 0) != 0ull) {
                            const bool cand = pre && !(qv > B.y);  // power > 0 (numerically non-PSD conic): skipped
                            const float alpha = fminf(0.99f, gs2m_fast_exp2(qv));
                            const float test_T = fmaf(-T[k], alpha, T[k]);
                            const bool sat = cand && test_T < 0.0001f;
                            const float Tn = (cand && !sat) ? test_T : T[k];
                            const float wT = T[k] - Tn;  // = alpha * T for an accepted contribution, else 0
                            C0[k] = fmaf(B.z, wT, C0[k]);
                            C1[k] = fmaf(B.w, wT, C1[k]);
                            C2[k] = fmaf(CL.x, wT, C2[k]);
                            T[k] = Tn;
                            pyf[k] = sat ? GS2M_PARKED : pyf[k];
                        }
                    }
                }
            }
        };
        // The quadrant mask is read (= the LDS wait) BEFORE the next instance's reads are issued, so the
        // wait never covers a read that was just issued (the compiler's waitcnt is lgkmcnt(0) in this loop).
        const BlendInst* sp = &s_i[wave][0];
        float4 A0 = sp[0].a, B0 = sp[0].b, A1, B1;
        float2 K0 = sp[0].c, K1;
        int j = 0;
        for (; j + 1 < nb; j += 2) {
            const int qm0 = gs2m_uniform((int)__float_as_uint(K0.y));
            GS2M_SCHED_BARRIER();
            A1 = sp[j + 1].a;
            B1 = sp[j + 1].b;
            K1 = sp[j + 1].c;
            GS2M_SCHED_BARRIER();
            step(qm0, K0, A0, B0);
            const int qm1 = gs2m_uniform((int)__float_as_uint(K1.y));
            GS2M_SCHED_BARRIER();
            A0 = sp[j + 2].a;
            B0 = sp[j + 2].b;
            K0 = sp[j + 2].c;
            GS2M_SCHED_BARRIER();
            step(qm1, K1, A1, B1);
        }
        if (j < nb) step(gs2m_uniform((int)__float_as_uint(K0.y)), K0, A0, B0);
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int pxi = px0 + 8 * (k & 1), pyi = py0 + 8 * (k >> 1);
        if (pxi < W && pyi < H) {
            const float o0 = C0[k] + T[k] * cam.bg[0], o1 = C1[k] + T[k] * cam.bg[1], o2 = C2[k] + T[k] * cam.bg[2];
            const size_t pix = (size_t)pyi * W + pxi;
            if (out_color) {
                float* oc = out_color + (size_t)v * 3 * plane;
                oc[pix] = o0;
                oc[plane + pix] = o1;
                oc[2 * plane + pix] = o2;
            }
            if (out_rgb8) {
                unsigned char* o8 = out_rgb8 + ((size_t)v * plane + pix) * 3;
                o8[0] = quantize_u8(o0);
                o8[1] = quantize_u8(o1);
                o8[2] = quantize_u8(o2);
            }
        }
    }
}
