// raster_math.h -- per-Gaussian projection maths of the hot path (device functions).
//
// Written from the algorithm in SURVEY.md Appendix A; every function cites the reference
// lines whose RESULT it must reproduce (DGR/ = third_party/gaussian-splatting/submodules/
// diff-gaussian-rasterization/).  Operand order follows the reference so that, compiled with
// -ffp-contract=off (this translation unit is), the fp32 sequence is the literal IEEE one and
// matches oracle/raster_oracle.c bit for bit (HIP's fp32 sqrt / divide are correctly rounded
// by default).  The stage is HBM-bound, so giving up FMA contraction here costs nothing.
#pragma once
#include "raster_common.h"

GS2M_DEVICE int gs2m_imin(int a, int b) { return a < b ? a : b; }
GS2M_DEVICE int gs2m_imax(int a, int b) { return a > b ? a : b; }

// DGR/cuda_rasterizer/auxiliary.h:58-66 transformPoint4x3
GS2M_DEVICE void xform4x3(const float* m, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = m[0] * x + m[4] * y + m[8] * z + m[12];
    oy = m[1] * x + m[5] * y + m[9] * z + m[13];
    oz = m[2] * x + m[6] * y + m[10] * z + m[14];
}
// auxiliary.h:68-77 transformPoint4x4
GS2M_DEVICE void xform4x4(const float* m, float x, float y, float z, float& ox, float& oy, float& oz, float& ow) {
    ox = m[0] * x + m[4] * y + m[8] * z + m[12];
    oy = m[1] * x + m[5] * y + m[9] * z + m[13];
    oz = m[2] * x + m[6] * y + m[10] * z + m[14];
    ow = m[3] * x + m[7] * y + m[11] * z + m[15];
}

// auxiliary.h:41-44: evaluated in double (the literals are doubles)
GS2M_DEVICE float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// forward.cu:118-152 computeCov3D: Sigma = Rq diag(mod*s)^2 Rq^T, q = (r,x,y,z) used as given.
// cov[6] = [S00,S01,S02,S11,S12,S22]; each entry a k = 0,1,2 left-to-right sum.
GS2M_DEVICE void cov3d_from_scale_rot(float sx, float sy, float sz, float mod, float qr, float qx, float qy,
                                      float qz, float* cov) {
    const float s0 = mod * sx, s1 = mod * sy, s2 = mod * sz;
    const float r = qr, x = qx, y = qy, z = qz;
    const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    const float M00 = s0 * R00, M01 = s1 * R01, M02 = s2 * R02;
    const float M10 = s0 * R10, M11 = s1 * R11, M12 = s2 * R12;
    const float M20 = s0 * R20, M21 = s1 * R21, M22 = s2 * R22;
    cov[0] = M00 * M00 + M01 * M01 + M02 * M02;
    cov[1] = M00 * M10 + M01 * M11 + M02 * M12;
    cov[2] = M00 * M20 + M01 * M21 + M02 * M22;
    cov[3] = M10 * M10 + M11 * M11 + M12 * M12;
    cov[4] = M10 * M20 + M11 * M21 + M12 * M22;
    cov[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

// forward.cu:74-113 computeCov2D.  The reference builds GLM (column-major) matrices
// J, W, T = W*J, cov = T^T * Vrk^T * T; unfolded here entry by entry in GLM's evaluation
// order (3-term sums, left to right; products with J's structural zeros are kept because
// 0*x + y is not always bit-identical to y when x is inf/nan -- they are not, for finite input,
// but keeping them costs nothing and keeps the sequence literal).
GS2M_DEVICE void cov2d_ewa(const float* vm, float tx_in, float ty_in, float tz, float focal_x, float focal_y,
                           float tan_fovx, float tan_fovy, const float* c3, float& oa, float& ob, float& oc) {
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = tx_in / tz;
    const float tytz = ty_in / tz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    // J_glm columns
    const float J00 = focal_x / tz, J01 = 0.0f, J02 = -(focal_x * tx) / (tz * tz);
    const float J10 = 0.0f, J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
    // W_glm[col][row]: col0 = (v0,v4,v8), col1 = (v1,v5,v9), col2 = (v2,v6,v10)
    const float W00 = vm[0], W01 = vm[4], W02 = vm[8];
    const float W10 = vm[1], W11 = vm[5], W12 = vm[9];
    const float W20 = vm[2], W21 = vm[6], W22 = vm[10];
    // T = W*J: T[c][r] = W[0][r]*J[c][0] + W[1][r]*J[c][1] + W[2][r]*J[c][2]
    const float T00 = W00 * J00 + W10 * J01 + W20 * J02;
    const float T01 = W01 * J00 + W11 * J01 + W21 * J02;
    const float T02 = W02 * J00 + W12 * J01 + W22 * J02;
    const float T10 = W00 * J10 + W10 * J11 + W20 * J12;
    const float T11 = W01 * J10 + W11 * J11 + W21 * J12;
    const float T12 = W02 * J10 + W12 * J11 + W22 * J12;
    // third column of T is W*0 = 0 (J's third column is zero): 0*a + 0*b + 0*c = 0
    // A = transpose(T): A[c][r] = T[r][c];  B = transpose(Vrk): Vrk symmetric so B = Vrk:
    // V[c][r] with V[0] = (c0,c1,c2), V[1] = (c1,c3,c4), V[2] = (c2,c4,c5)
    const float V00 = c3[0], V01 = c3[1], V02 = c3[2];
    const float V10 = c3[1], V11 = c3[3], V12 = c3[4];
    const float V20 = c3[2], V21 = c3[4], V22 = c3[5];
    // Bt[c][r] = V[r][c]
    // tmp = A * Bt: tmp[c][r] = A[0][r]*Bt[c][0] + A[1][r]*Bt[c][1] + A[2][r]*Bt[c][2]
    //             = T[r][0]*V[0][c] + T[r][1]*V[1][c] + T[r][2]*V[2][c]
    // we need tmp rows r = 0,1 only (rows of cov used: cov[0][0], cov[0][1], cov[1][1])
    const float P00 = T00 * V00 + T01 * V10 + T02 * V20;  // tmp[c=0][r=0]
    const float P10 = T00 * V01 + T01 * V11 + T02 * V21;  // tmp[c=1][r=0]
    const float P20 = T00 * V02 + T01 * V12 + T02 * V22;  // tmp[c=2][r=0]
    const float P01 = T10 * V00 + T11 * V10 + T12 * V20;  // tmp[c=0][r=1]
    const float P11 = T10 * V01 + T11 * V11 + T12 * V21;  // tmp[c=1][r=1]
    const float P21 = T10 * V02 + T11 * V12 + T12 * V22;  // tmp[c=2][r=1]
    // cov = tmp * T: cov[c][r] = tmp[0][r]*T[c][0] + tmp[1][r]*T[c][1] + tmp[2][r]*T[c][2]
    const float cov00 = P00 * T00 + P10 * T01 + P20 * T02;  // cov[0][0]
    const float cov01 = P01 * T00 + P11 * T01 + P21 * T02;  // cov[0][1]
    const float cov11 = P01 * T10 + P11 * T11 + P21 * T12;  // cov[1][1]
    oa = cov00 + 0.3f;
    ob = cov01;
    oc = cov11 + 0.3f;
}

// forward.cu:20-71 computeColorFromSH for one channel-interleaved coefficient block
// sh[k*3 + c]; returns max(0, SH + 0.5).  dir must already be normalised.
#define GS2M_SH_C0 0.28209479177387814f
#define GS2M_SH_C1 0.4886025119029199f
// STRIDE = distance between the coefficients of one channel: 3 for a [16][3] row (sh[3 k + c]), 1 for the 16
// coefficients of channel c gathered into an array (c = 0).  Same expression, same order, either way.
template <int STRIDE = 3>
GS2M_DEVICE float sh_channel(int deg, const float* sh, int c, float x, float y, float z) {
#define SHK(k) sh[STRIDE * (k) + c]
    float result = GS2M_SH_C0 * SHK(0);
    if (deg > 0) {
        result = result - GS2M_SH_C1 * y * SHK(1) + GS2M_SH_C1 * z * SHK(2) - GS2M_SH_C1 * x * SHK(3);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            result = result + 1.0925484305920792f * xy * SHK(4) + -1.0925484305920792f * yz * SHK(5) +
                     0.31539156525252005f * (2.0f * zz - xx - yy) * SHK(6) + -1.0925484305920792f * xz * SHK(7) +
                     0.5462742152960396f * (xx - yy) * SHK(8);
            if (deg > 2) {
                result = result + -0.5900435899266435f * y * (3.0f * xx - yy) * SHK(9) +
                         2.890611442640554f * xy * z * SHK(10) +
                         -0.4570457994644658f * y * (4.0f * zz - xx - yy) * SHK(11) +
                         0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHK(12) +
                         -0.4570457994644658f * x * (4.0f * zz - xx - yy) * SHK(13) +
                         1.445305721320277f * z * (xx - yy) * SHK(14) +
                         -0.5900435899266435f * x * (xx - 3.0f * yy) * SHK(15);
            }
        }
    }
#undef SHK
    result += 0.5f;
    return result < 0.0f ? 0.0f : result;
}

// The same three channels for NV view directions at once, STREAMING the 192-B row: float4 j of the row comes from
// `row(j)` (LDS: the wave-transposed landing zone of k_project<.., true>, one ds_read_b128 per float4 -- a 16-B lane stride is
// conflict-free for b128 reads, the per-channel b32 read-back of round 2 hit every bank four times; HBM: the packed copy or the
// [P,16,3] layout, one dwordx4 load per float4).  Element e = 3 k + c of the row = component e % 4 of float4 e / 4; three
// float4 = four coefficients x three channels are in flight at a time, so no 48 registers hold the row (k_project<2, false>:
// 108 -> see DESIGN.md).  The basis factors are formed once per view for the three channels; every product and every addition
// is the one sh_channel performs, in its order (a * b * c = (a * b) * c, sums left to right): bit-identical.
template <int NV, typename Row>
GS2M_DEVICE void sh_rgb_stream(int deg, Row row, const float (*dir)[3], float (*out)[3]) {
    float acc[NV][3];
    const float4 F0 = row(0);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        acc[v][0] = GS2M_SH_C0 * F0.x;
        acc[v][1] = GS2M_SH_C0 * F0.y;
        acc[v][2] = GS2M_SH_C0 * F0.z;
    }
#define GS2M_SH_ADD(v, bk, s0, s1, s2)    \
    do {                                  \
        acc[v][0] = acc[v][0] + (bk) * (s0); \
        acc[v][1] = acc[v][1] + (bk) * (s1); \
        acc[v][2] = acc[v][2] + (bk) * (s2); \
    } while (0)
#define GS2M_SH_SUB(v, bk, s0, s1, s2)    \
    do {                                  \
        acc[v][0] = acc[v][0] - (bk) * (s0); \
        acc[v][1] = acc[v][1] - (bk) * (s1); \
        acc[v][2] = acc[v][2] - (bk) * (s2); \
    } while (0)
    if (deg > 0) {
        const float4 F1 = row(1), F2 = row(2);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float x = dir[v][0], y = dir[v][1], z = dir[v][2];
            const float b1 = GS2M_SH_C1 * y, b2 = GS2M_SH_C1 * z, b3 = GS2M_SH_C1 * x;
            GS2M_SH_SUB(v, b1, F0.w, F1.x, F1.y);   // k = 1: elements 3, 4, 5
            GS2M_SH_ADD(v, b2, F1.z, F1.w, F2.x);   // k = 2: 6, 7, 8
            GS2M_SH_SUB(v, b3, F2.y, F2.z, F2.w);   // k = 3: 9, 10, 11
        }
        if (deg > 1) {
            {
                const float4 F3 = row(3), F4 = row(4), F5 = row(5);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float x = dir[v][0], y = dir[v][1], z = dir[v][2];
                    const float xx = x * x, yy = y * y, zz = z * z;
                    const float xy = x * y, yz = y * z, xz = x * z;
                    const float b4 = 1.0925484305920792f * xy, b5 = -1.0925484305920792f * yz;
                    const float b6 = 0.31539156525252005f * (2.0f * zz - xx - yy), b7 = -1.0925484305920792f * xz;
                    GS2M_SH_ADD(v, b4, F3.x, F3.y, F3.z);   // k = 4: 12, 13, 14
                    GS2M_SH_ADD(v, b5, F3.w, F4.x, F4.y);   // k = 5: 15, 16, 17
                    GS2M_SH_ADD(v, b6, F4.z, F4.w, F5.x);   // k = 6: 18, 19, 20
                    GS2M_SH_ADD(v, b7, F5.y, F5.z, F5.w);   // k = 7: 21, 22, 23
                }
            }
            const float4 F6 = row(6);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float x = dir[v][0], y = dir[v][1];
                const float xx = x * x, yy = y * y;
                const float b8 = 0.5462742152960396f * (xx - yy);
                GS2M_SH_ADD(v, b8, F6.x, F6.y, F6.z);       // k = 8: 24, 25, 26
            }
            if (deg > 2) {
                {
                    const float4 F7 = row(7), F8 = row(8);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float x = dir[v][0], y = dir[v][1], z = dir[v][2];
                        const float xx = x * x, yy = y * y, zz = z * z;
                        const float xy = x * y;
                        const float b9 = -0.5900435899266435f * y * (3.0f * xx - yy), b10 = 2.890611442640554f * xy * z;
                        const float b11 = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
                        GS2M_SH_ADD(v, b9, F6.w, F7.x, F7.y);    // k = 9: 27, 28, 29
                        GS2M_SH_ADD(v, b10, F7.z, F7.w, F8.x);   // k = 10: 30, 31, 32
                        GS2M_SH_ADD(v, b11, F8.y, F8.z, F8.w);   // k = 11: 33, 34, 35
                    }
                }
                const float4 F9 = row(9), F10 = row(10), F11 = row(11);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float x = dir[v][0], y = dir[v][1], z = dir[v][2];
                    const float xx = x * x, yy = y * y, zz = z * z;
                    const float b12 = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                    const float b13 = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
                    const float b14 = 1.445305721320277f * z * (xx - yy), b15 = -0.5900435899266435f * x * (xx - 3.0f * yy);
                    GS2M_SH_ADD(v, b12, F9.x, F9.y, F9.z);      // k = 12: 36, 37, 38
                    GS2M_SH_ADD(v, b13, F9.w, F10.x, F10.y);    // k = 13: 39, 40, 41
                    GS2M_SH_ADD(v, b14, F10.z, F10.w, F11.x);   // k = 14: 42, 43, 44
                    GS2M_SH_ADD(v, b15, F11.y, F11.z, F11.w);   // k = 15: 45, 46, 47
                }
            }
        }
    }
#undef GS2M_SH_ADD
#undef GS2M_SH_SUB
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float r = acc[v][c] + 0.5f;
            out[v][c] = r < 0.0f ? 0.0f : r;
        }
    }
}

// ---- exact tile test (extension, GS2M_OPT_EXACT_TILE_CULL; image-preserving) --------------
// Same arithmetic as oracle_tile_may_contribute (oracle/raster_oracle.c).
GS2M_DEVICE float q_form(float a, float b, float c, float dx, float dy) {
    return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
}
// rx = -b / a, ry = -b / c: slopes of the minimiser lines, formed once per Gaussian (cull_slopes)
GS2M_DEVICE float edge_min_x(float a, float b, float c, float rx, float dy, float x0, float x1) {
    float dxs = a > 0.0f ? rx * dy : x0;
    dxs = fminf(x1, fmaxf(x0, dxs));
    return q_form(a, b, c, dxs, dy);
}
GS2M_DEVICE float edge_min_y(float a, float b, float c, float ry, float dx, float y0, float y1) {
    float dys = c > 0.0f ? ry * dx : y0;
    dys = fminf(y1, fmaxf(y0, dys));
    return q_form(a, b, c, dx, dys);
}
GS2M_DEVICE void cull_slopes(float ca, float cb, float cc, float& rx, float& ry) {
    rx = -cb / ca;
    ry = -cb / cc;
}
// thresh = ln(255*o)*1.0001 + 0.001 (precomputed per Gaussian), or < 0 if o*255 < 1.
// th = tile height in pixels (16 = the reference tile; 32 = two stacked, GS2M_OPT_TILE_ROWS 2), ty in units of th
GS2M_DEVICE bool tile_may_contribute(float mx, float my, float ca, float cb, float cc, float rx, float ry,
                                     float thresh, int tx, int ty, int th = GS2M_TILE) {
    const float dx0 = mx - (float)(tx * GS2M_TILE + GS2M_TILE - 1);
    const float dx1 = mx - (float)(tx * GS2M_TILE);
    const float dy0 = my - (float)(ty * th + th - 1);
    const float dy1 = my - (float)(ty * th);
    const bool in_x = dx0 <= 0.0f && dx1 >= 0.0f, in_y = dy0 <= 0.0f && dy1 >= 0.0f;   // the centre's column / row of tiles
    if (in_x && in_y) return true;
    // Round 6: only the edges that FACE the centre are evaluated (two of the four, one when the centre lies in the tile's column
    // or row).  q is convex with its minimum at the centre, outside the tile: the smallest level ellipse that touches the tile
    // touches it in a point visible from the centre, i.e. on the vertical edge nearest in x (if the centre is outside the
    // tile's column) or on the horizontal edge nearest in y (if outside its row); the far edges can only tie.  Half the
    // arithmetic of the four-edge form (the counting kernel is vector-issue-bound); same decisions (oracle: same form).
    const float ex = dx0 > 0.0f ? dx0 : dx1;      // d.x on the vertical edge facing the centre
    const float ey = dy0 > 0.0f ? dy0 : dy1;      // d.y on the horizontal edge facing the centre
    const float qv = edge_min_y(ca, cb, cc, ry, ex, dy0, dy1);
    const float qh = edge_min_x(ca, cb, cc, rx, ey, dx0, dx1);
    const float qmin = in_x ? qh : (in_y ? qv : fminf(qv, qh));
    return qmin <= thresh;
}
GS2M_DEVICE float cull_threshold(float opacity) {
    if (!(opacity * 255.0f >= 1.0f)) return -1.0f;  // alpha < 1/255 everywhere: never contributes
    return logf(opacity * 255.0f) * 1.0001f + 0.001f;
}
