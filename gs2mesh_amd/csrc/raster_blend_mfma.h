// raster_blend_mfma.h -- compositing with the per-pixel exponents computed on the MATRIX CORES (blend variant 7).
//
// Same per-pixel decisions, order and arithmetic as k_blend_wave4e (raster_blend.h, renderCUDA forward.cu:261-374);
// what changes is where q = log2(alpha) of a (pixel, instance) pair comes from.  In tile-centred pixel coordinates
// (u, v) in {-7.5 .. 7.5}^2 the exponent is a quadratic form with 6 per-instance coefficients,
//   q(u,v) = k_c + k_u u + k_v v + k_uu u^2 + k_vv v^2 + k_uv u v,
// i.e. a [pixels x 6] * [6 x instances] product.  The pixel features are small half-integers and their products
// (<= 56.25, two fractional bits) are EXACT in bf16; each fp32 coefficient is split into three bf16 pieces
// (hi + mid + lo = 24 bits, truncation), so 18 exact bf16 products accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (K = 32: two K-steps) reproduce the fp32 polynomial to ~1e-7 of its largest term.
// One wave per 16x16 tile; per 8x8 quadrant (64 pixels = one per lane) and per 32 staged instances: 4 MFMAs give a
// 64 x 32 block of exponents, 16 v_permlane32_swap put the 32 exponents of a lane's own pixel into its registers,
// and the compositing walks them in order.  The VALU then only pays for the alpha >= 1/255 compare per
// (instance, quadrant) and for the accumulate path of contributing ones (exp2, T update, 3 colour FMAs); the
// 5 VALU ops per evaluated quadrant and the 8 per instance of the all-VALU kernel are gone, and finished pixels
// are tracked in lane masks (scalar registers), so a saturated quadrant costs nothing.
// Measured motivation (tools, C2): the all-VALU kernel issues 8 I + 5 Qe + 12 Qa wave instructions
// (I = 1.5 M instance-waves, Qe = 4.6 M evaluated, Qa = 3.55 M contributing quadrants per eye).
#pragma once
#include "raster_blend.h"

struct alignas(16) MfmaInst {
    float4 geo;   // mx, my, a' = -0.5 log2e ca, b' = log2e cb
    float4 aux;   // c' = -0.5 log2e cc, log2 o, quadrant mask (bits), flags (bit 0: general path)
    float4 col;   // r, g, b, log2 o
};

// three truncated bf16 pieces of an fp32 value, returned as fp32 bit patterns whose high halves are the pieces
GS2M_DEVICE void split3_bf16(float k, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(k) & 0xffff0000u;
    const float r1 = k - __uint_as_float(h);            // exact
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);           // exact
    l = __float_as_uint(r2);
}
// dword of two bf16: low half = high 16 bits of x, high half = high 16 bits of y
GS2M_DEVICE unsigned pack_hi16(unsigned x, unsigned y) { return (x >> 16) | (y & 0xffff0000u); }
GS2M_DEVICE unsigned bf16_of_exact(float f) { return __float_as_uint(f) >> 16; }  // f representable in bf16

// Sequential compositing of one quadrant's 64 pixels (one per lane) over the 32 instances whose exponents sit in D1 / D2
// (instance i: register (i&3) + 4 (i>>3) of D1 if bit 2 of i is clear, else of D2).
//   MODE 0: scalar branch on the instance's quadrant-mask bit, then a branch on "any lane passes alpha >= 1/255";
//   MODE 1: only the mask-bit branch (the body runs predicated);
//   MODE 2: straight-line: no branch per instance, the mask bit is folded into the lane predicate.
// GEN: the unit may contain general-path instances (power > 0 test + alpha cap), selected per instance by bit i of gh.
template <int MODE, bool GEN>
GS2M_DEVICE void composite_unit(const gs2m_f32x16& D1, const gs2m_f32x16& D2, const MfmaInst* sp, unsigned mh, unsigned gh,
                                float QMIN, float& T, float& C0, float& C1, float& C2, unsigned long long& dn) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const bool bit = (mh >> i) & 1u;                                  // scalar: the instance reaches this quadrant
        if (MODE != 2 && !bit) continue;
        const int r = (i & 3) + 4 * (i >> 3);
        const float qv = (i & 4) ? D2[r] : D1[r];
        unsigned long long prem = gs2m_ballot_b(qv >= QMIN) & ~dn;       // v_cmp + s_andn2
        if (MODE == 2) prem &= bit ? ~0ull : 0ull;
        if (MODE == 0 && prem == 0ull) continue;
        const float4 CL = sp[i].col;                                      // r, g, b, log2 o (LDS broadcast)
        float alpha = gs2m_fast_exp2(qv);
        unsigned long long candm = prem;
        if (GEN) {
            if (MODE == 2) {  // predicated
                const unsigned long long gm = ((gh >> i) & 1u) ? ~0ull : 0ull;
                candm = prem & ~(gs2m_ballot_b(qv > CL.w) & gm);
                alpha = gm ? fminf(0.99f, alpha) : alpha;
            } else if ((gh >> i) & 1u) {                                  // scalar branch: general path (rare)
                GS2M_NO_IF_CONVERT();
                candm = prem & ~gs2m_ballot_b(qv > CL.w);                 // power > 0: skipped (forward.cu:336-337)
                alpha = fminf(0.99f, alpha);
            }
        }
        const float test_T = fmaf(-T, alpha, T);
        const unsigned long long satm = gs2m_ballot_b(test_T < 0.0001f) & candm;
        const float Tn = gs2m_lanes(candm & ~satm) ? test_T : T;
        const float wT = T - Tn;  // = alpha * T for an accepted contribution, else 0
        C0 = fmaf(CL.x, wT, C0);
        C1 = fmaf(CL.y, wT, C1);
        C2 = fmaf(CL.z, wT, C2);
        T = Tn;
        dn |= satm;
    }
}

template <int WPB, int LROWS, int MODE>
GS2M_KERNEL void __launch_bounds__(64 * WPB)
k_blend_mfma(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
             const GeomRec* __restrict__ recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
             float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8, const int* __restrict__ rank) {
    __shared__ MfmaInst s_i[WPB][64];
    __shared__ uint4 s_B[4][2][64];  // pixel-feature operand of quadrant k, pixel half h (the same for every tile)
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // ---- pixel features, once per workgroup -------------------------------------------------------------------
    for (int idx = tid; idx < 4 * 2 * 64; idx += 64 * WPB) {
        const int k = idx >> 7, h = (idx >> 6) & 1, l = idx & 63;
        const int n = (l & 31) + 32 * h;                  // pixel of the quadrant (= the lane that owns it afterwards)
        const float u = (float)(8 * (k & 1) + (n & 7)) - 7.5f, v = (float)(8 * (k >> 1) + (n >> 3)) - 7.5f;
        const unsigned bu = bf16_of_exact(u), bv = bf16_of_exact(v), buu = bf16_of_exact(u * u), bvv = bf16_of_exact(v * v),
                       buv = bf16_of_exact(u * v);
        uint4 b;
        if ((l >> 5) == 0) {  // slots: u u u v v v uu uu
            b.x = bu | (bu << 16);
            b.y = bu | (bv << 16);
            b.z = bv | (bv << 16);
            b.w = buu | (buu << 16);
        } else {              // slots: uu vv vv vv uv uv uv 0
            b.x = buu | (bvv << 16);
            b.y = bvv | (bvv << 16);
            b.z = buv | (buv << 16);
            b.w = buv;
        }
        s_B[k][h][l] = b;
    }
    __syncthreads();  // the only workgroup barrier; waves are independent from here on
    const int v = (int)blockIdx.y;
    const CamUniform& cam = cams[v];
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int tiles = gx * cam.gy;
    const int ltiles = gx * ((cam.gy + LROWS - 1) / LROWS);
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned qq = nwg / 8u, rr = nwg % 8u, xcd = bid % 8u, idx8 = bid / 8u;
    const unsigned grp = (xcd < rr ? xcd * (qq + 1u) : rr * (qq + 1u) + (xcd - rr) * qq) + idx8;
    const int tile = gs2m_uniform((int)(grp * (unsigned)WPB) + wave);
    if (tile >= tiles) return;
    const int tx = tile % gx, ty = tile / gx;
    const int px0 = tx * GS2M_TILE + (lane & 7), py0 = ty * GS2M_TILE + (lane >> 3);
    float T[4], C0[4], C1[4], C2[4];
    unsigned long long dn[4];  // finished pixels of each quadrant: explicit lane masks in scalar registers
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
        dn[k] = gs2m_ballot_b(!(x < W && y < H));
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
    }
    const float qx0 = (float)(tx * GS2M_TILE), qy0 = (float)(ty * GS2M_TILE);
    const float tcx = qx0 + 7.5f, tcy = qy0 + 7.5f;
    const int ltile = (ty / LROWS) * gx + tx;
    unsigned r0 = tile_start[(size_t)v * (ltiles + 1) + ltile];
    unsigned r1 = tile_start[(size_t)v * (ltiles + 1) + ltile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    r0 = (unsigned)gs2m_uniform((int)r0);
    r1 = (unsigned)gs2m_uniform((int)r1);
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRec* rv = recs + (size_t)v * P;
    const float LOG2E = 1.44269504088896340736f;
    const float QMIN = -7.99435343685885793770f;  // -log2(255)
    // constant-term operand of the second K-step: slots 1 1 1 0 0 0 0 0 in slot group 0, nothing in group 1
    uint4 B1;
    B1.x = lane < 32 ? 0x3f803f80u : 0u;
    B1.y = lane < 32 ? 0x00003f80u : 0u;
    B1.z = 0u;
    B1.w = 0u;
    float4 ra, rb, rc;
    ra.x = ra.y = ra.z = ra.w = 0.0f;
    rb = ra;
    rc = ra;
    rb.y = 1.0f;
    unsigned base = r0;
    if (base + (unsigned)lane < r1) {
        unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        if (rank) gid = (unsigned)rank[gid];
        const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
        ra = r4[0];
        rb = r4[1];
        rc = r4[2];
    }
    while (base < r1) {
        if ((dn[0] & dn[1] & dn[2] & dn[3]) == ~0ull) break;
        gs2m_wave_sync();
        int nb = 0;
        {
            // ---- staging: per-instance constants by the lane that gathered the record; compaction to this wave's tile
            const float lo = gs2m_fast_log2(rb.y);
            const float t2 = fmaxf(2.0f * (gs2m_fast_log(rb.y * 255.0f) + 1.0e-4f), 0.0f);
            const float det = ra.z * rb.x - ra.w * ra.w;
            const float inv = 1.0f / det;
            const float hx = sqrtf(t2 * rb.x * inv) * 1.001f + 0.01f;
            const float hy = sqrtf(t2 * ra.z * inv) * 1.001f + 0.01f;
            const bool xl = ra.x - hx <= qx0 + 7.0f, xr = ra.x + hx >= qx0 + 8.0f;
            const bool box = rb.y * 255.0f >= 0.9999f && det > 0.0f;
            const bool degenerate = !(det > 0.0f);
            const int ry0 = (int)(__float_as_uint(rc.z) >> 16), ry1 = (int)(__float_as_uint(rc.w) >> 16);
            unsigned m = 0u;
#pragma unroll
            for (int qr = 0; qr < 2; ++qr) {
                const float top = qy0 + 8.0f * (float)qr;
                bool row = degenerate || (box && ra.y - hy <= top + 7.0f && ra.y + hy >= top);
                if (LROWS > 1) row = row && ty >= ry0 && ty < ry1;  // the 16 x 16 tile must be in the instance's rect
                if (row && (degenerate || xl)) m |= 1u << (2 * qr);
                if (row && (degenerate || xr)) m |= 2u << (2 * qr);
            }
            // general path (power > 0 test, alpha cap) only where it can matter: opacity near the 0.99 cap, or a conic
            // so close to singular that rounding could make the quadratic form negative (SURVEY.md App. A thresholds)
            const bool general = !(rb.y <= 0.98f) || !(ra.z > 0.0f && rb.x > 0.0f && det >= 1.0e-3f * ra.z * rb.x);
            MfmaInst mi;
            mi.geo.x = ra.x;
            mi.geo.y = ra.y;
            mi.geo.z = (-0.5f * LOG2E) * ra.z;
            mi.geo.w = LOG2E * ra.w;
            mi.aux.x = (-0.5f * LOG2E) * rb.x;
            mi.aux.y = lo;
            mi.aux.z = __uint_as_float(m);
            mi.aux.w = __uint_as_float(general ? 1u : 0u);
            mi.col.x = rb.z;
            mi.col.y = rb.w;
            mi.col.z = rc.x;
            mi.col.w = lo;
            const bool mine = m != 0u && base + (unsigned)lane < r1;
            const unsigned long long keep = gs2m_ballot(mine ? 1 : 0);
            if (mine) s_i[wave][gs2m_popc64(keep & ((1ull << lane) - 1ull))] = mi;
            nb = gs2m_popc64(keep);
        }
        gs2m_wave_sync();
        base += 64u;
        if (base + (unsigned)lane < r1) {  // gather the next batch while this one is composited
            unsigned gid = (unsigned)(kv[base + lane] & 0xffffffffull);
            if (rank) gid = (unsigned)rank[gid];
            const float4* r4 = reinterpret_cast<const float4*>(rv + gid);
            ra = r4[0];
            rb = r4[1];
            rc = r4[2];
        }
        if (nb == 0) continue;
        // ---- matrix operand of the instances: lane i < nb owns compacted instance i ---------------------------------
        unsigned long long mk[4], gen;
        uint4 A0[2], A1[2];
        {
            const bool valid = lane < nb;
            float4 geo, aux;
            geo.x = geo.y = geo.z = geo.w = 0.0f;
            aux = geo;
            if (valid) {
                geo = s_i[wave][lane].geo;
                aux = s_i[wave][lane].aux;
            }
            const unsigned m = __float_as_uint(aux.z);
#pragma unroll
            for (int k = 0; k < 4; ++k) mk[k] = gs2m_ballot(((m >> k) & 1u) ? 1 : 0);
            gen = gs2m_ballot((__float_as_uint(aux.w) & 1u) ? 1 : 0);
            const float Dx = geo.x - tcx, Dy = geo.y - tcy;
            const float ap = geo.z, bp = geo.w, cp = aux.x;
            // q = lo + a'(Dx-u)^2 + c'(Dy-v)^2 - b'(Dx-u)(Dy-v)
            const float k_u = fmaf(-2.0f * ap, Dx, bp * Dy);
            const float k_v = fmaf(-2.0f * cp, Dy, bp * Dx);
            const float k_c = fmaf(ap * Dx, Dx, aux.y) + Dy * fmaf(cp, Dy, -(bp * Dx));
            unsigned uh, um, ul, vh, vm, vl, ah, am, al, ch, cm, cl, bh, bm, bl, kh, km, kl;
            split3_bf16(k_u, uh, um, ul);
            split3_bf16(k_v, vh, vm, vl);
            split3_bf16(ap, ah, am, al);     // u^2
            split3_bf16(cp, ch, cm, cl);     // v^2
            split3_bf16(-bp, bh, bm, bl);    // u v
            split3_bf16(k_c, kh, km, kl);
            uint4 R0, R1;
            R0.x = pack_hi16(uh, um);
            R0.y = pack_hi16(ul, vh);
            R0.z = pack_hi16(vm, vl);
            R0.w = pack_hi16(ah, am);
            R1.x = pack_hi16(al, ch);
            R1.y = pack_hi16(cm, cl);
            R1.z = pack_hi16(bh, bm);
            R1.w = bl >> 16;
            unsigned S0 = pack_hi16(kh, km), S1 = kl >> 16, Z0 = 0u, Z1 = 0u;
            // instances 0-31 live in lanes 0-31, 32-63 in lanes 32-63; an operand wants slot group 0 of 32 instances in its
            // lower half and slot group 1 of the SAME instances in its upper half: one half-wave swap per register
            gs2m_permlane32_swap(R0.x, R1.x);
            gs2m_permlane32_swap(R0.y, R1.y);
            gs2m_permlane32_swap(R0.z, R1.z);
            gs2m_permlane32_swap(R0.w, R1.w);
            gs2m_permlane32_swap(S0, Z0);
            gs2m_permlane32_swap(S1, Z1);
            A0[0] = R0;   // instances 0-31
            A0[1] = R1;   // instances 32-63
            A1[0].x = S0;
            A1[0].y = S1;
            A1[0].z = 0u;
            A1[0].w = 0u;
            A1[1].x = Z0;
            A1[1].y = Z1;
            A1[1].z = 0u;
            A1[1].w = 0u;
        }
        // ---- per quadrant: exponents on the matrix cores, then the sequential compositing of its 64 pixels -------------
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (mk[k] == 0ull) continue;                                     // no staged instance reaches this quadrant
            if (dn[k] == ~0ull) continue;                                    // every pixel of the quadrant is finished
            const uint4 Bq0 = s_B[k][0][lane], Bq1 = s_B[k][1][lane];
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
                const unsigned mh = (unsigned)(mk[k] >> (32 * hh));
                if (mh == 0u) continue;
                const unsigned gh = (unsigned)(gen >> (32 * hh));
                const uint4 a0 = hh ? A0[1] : A0[0], a1 = hh ? A1[1] : A1[0];
                gs2m_f32x16 D1, D2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    D1[r] = 0.0f;
                    D2[r] = 0.0f;
                }
                D1 = gs2m_mfma_32x32x16_bf16(a0, Bq0, D1);
                D2 = gs2m_mfma_32x32x16_bf16(a0, Bq1, D2);
                D1 = gs2m_mfma_32x32x16_bf16(a1, B1, D1);
                D2 = gs2m_mfma_32x32x16_bf16(a1, B1, D2);
                // lane l < 32 holds pixel l (D1) and pixel 32 + l (D2) for instance rows {0-3, 8-11, ..}, lane 32 + l the
                // same pixels for rows {4-7, 12-15, ..}: swap halves so that every lane holds all 32 rows of ITS pixel
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = D1[r], y = D2[r];
                    gs2m_permlane32_swap(x, y);
                    D1[r] = x;
                    D2[r] = y;
                }
                const MfmaInst* sp = &s_i[wave][32 * hh];
                if (MODE == 2 && gh != 0u) composite_unit<2, true>(D1, D2, sp, mh, gh, QMIN, T[k], C0[k], C1[k], C2[k], dn[k]);
                else composite_unit<MODE, (MODE != 2)>(D1, D2, sp, mh, gh, QMIN, T[k], C0[k], C1[k], C2[k], dn[k]);
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
