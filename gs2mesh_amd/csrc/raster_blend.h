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
                const GeomRecs recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
                float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8, const int* __restrict__ rank) {
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
    const GeomRecs rv = gs2m_recs_at(recs, (size_t)v * P);
    bool done = !inside;
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    int todo = (int)(r1 - r0);
    for (unsigned base = r0; base < r1; base += 256, todo -= 256) {
        const int num_done = gs2m_syncthreads_count(done ? 1 : 0);
        if (num_done == 256) break;
        if (base + (unsigned)tid < r1) {
            unsigned gid = (unsigned)(kv[base + tid] & 0xffffffffull);
            if (rank) gid = (unsigned)rank[gid];   // packed model: id in the key -> position of the record
            s_a[tid] = rv.ab[2 * (size_t)gid];
            s_b[tid] = rv.ab[2 * (size_t)gid + 1];
            s_c[tid] = rv.c[gid].x;
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
// Variant 4 (k_blend_wave4e): ONE WAVE PER 16x16 TILE, 4 pixels per lane laid out as the tile's four 8x8 QUADRANTS
// (lane l: x = l&7, y = l>>3, plus (0|8, 0|8)).  A 256-thread workgroup is four independent waves = four consecutive
// tiles: no __syncthreads at all, a wave leaves as soon as ITS 256 pixels are saturated.
//   * per instance the wave pays one set of LDS broadcast reads for 256 pixel evaluations; dx terms are shared by the
//     lane's pixels;
//   * the 48-B records of the NEXT 64 instances are gathered from HBM straight into LDS by the DMA path
//     (global_load_lds_dwordx4: no VGPR holds them while the current 64 are composited), their ids one batch earlier;
//   * a per-instance 4-bit quadrant mask, computed while staging from the bounding box of the alpha >= 1/255 ellipse
//     (|dx| <= sqrt(2 t cov_xx), |dy| <= sqrt(2 t cov_yy), t = ln(255 o); cov = conic^-1), skips quadrants with a scalar
//     branch;
//   * the exponent is kept in the log2 domain: the staging lane folds log2(e) into the conic and log2(opacity) into the
//     constant term,  q = (a' dx^2 + log2 o) + c' dy^2 - b' dx dy,  alpha = exp2(q)  (1 v_exp_f32), so the
//     alpha >= 1/255 pre-filter is q >= -log2(255);
//   * every per-pixel decision is a LANE MASK in scalar registers (v_cmp + s_and*): the finished pixels of each quadrant
//     (parking a pixel costs no vector op; a finished quadrant is removed from every instance's quadrant mask), the
//     candidates, the saturating ones; v_cndmask takes the mask directly (gs2m_lanes);
//   * the reference's power > 0 test and alpha cap (forward.cu:336-343) only run for the instances where they can matter
//     (opacity > 0.98, or a conic within 1e-3 of singular), flagged while staging: 9 vector ops per contributing
//     (instance, quadrant) -- exp, T' = fma(-T, alpha, T), compare, select, w = T - T', 3 colour FMAs, move;
//   * wave -> tile mapping = k_tile_scan's schedule (`order`): chunks of 4 x 2 neighbouring lists (they share Gaussians:
//     one L2) ranked by descending weight and dealt to the XCDs round-robin (block b runs on XCD b % 8, blocks are
//     dispatched in order) -- a wave lives ~100 us and a launch has ~2 generations of them, so the kernel used to end on
//     whatever long lists came last and on the XCD that owned the densest part of the image (C2 238 -> 218 us, C3 320 ->
//     277 us).
// Decisions are the reference's (same thresholds, same order); roundings of ~1 ulp in q (|q| <= 8) -> relative 1e-6 in alpha.
// LROWS = reference tiles per instance list (GS2M_OPT_TILE_ROWS): with 2 the binning stages handle ~30 % fewer
// (Gaussian, tile) instances; two waves walk the same 16 x 32 list, each compositing its own 16 x 16 half (instances that
// miss the half are dropped while staging, by ballot compaction).  A quadrant outside the instance's 16 x 16 tile rect is
// masked, so the reference's rect still bounds every contribution: the image is the 16 x 16 image bit for bit.
// Measured ladder (C2, us per pair): r1 all-VALU kernel 250 -> lane masks + flagged general path 245 -> 7 waves/SIMD 235
// -> DMA prefetch (this file).  PMC (profiles/r2b_pmc.json): 138 M VALU + 85 M SALU + 36 M branch instructions per
// launch, 41 % of the wave cycles issuing / 27 % issue stalls / 32 % waits.  tools/ubench/valu_rates.hip prices the VALU
// mix at 7 waves per SIMD: plain fp32 op 2.4 cycles per wave64 instruction per SIMD, v_exp_f32 8.1, v_cmp + v_cndmask
// through a lane mask ~4 each, v_pk_fma_f32 7 (no gain over two v_fma) -> the mix of this loop costs >= 166 us at
// 2.4 GHz; the kernel runs at ~0.76 of that bound, the rest is the scalar / branch traffic of the skips.
// Rejected after A/B on the GPU (same image, slower): exponents + compares of all four quadrants hoisted in front of
// the branches (+17 %: 2.8 more VALU ops per instance), accepted lanes by execution mask instead of v_cndmask (+3 %:
// one VALU op less, two scalar ops and a branch more), column terms only for the columns an instance reaches (+6 %),
// 1 / 2 / 8 waves per workgroup (+-1 %), packed-f32 SLP (+18 %).
// ---------------------------------------------------------------------------------------------
struct BlendInst {
    float4 a, b;
    float2 c;
};

// PROF = 1 (GS2M_OPT_BLEND_PROFILE): the same kernel with s_memtime stamps at the phase boundaries of every wave, summed
// into prof[GS2M_BLEND_PROF_N] with one atomic per counter per wave (the stamps cost ~10 % themselves: shares, not times).
#define GS2M_BLEND_PROF_N 10

// One wave composites one 16 x 16 tile (tx, ty) of view v.  MODE 0 = per-pixel decisions as lane masks in scalar registers +
// per-instance quadrant mask ("reference-structure" loop of rounds 1-4, kept as the cross-check of MODE 2); MODE 2 (round 5,
// the default) = all four quadrants per staged instance, flag-free runs; MODE 3 (round 6) = MODE 2 with the alpha cap applied to
// every instance instead of splitting the runs at the capped ones (models with many saturated opacities).  (MODE 1, the execution-mask form of MODE 0, was never
// the fastest anywhere and was removed in round 6.)
template <int WPB, int LROWS, int MODE, int PROF>
GS2M_DEVICE void blend_tile(const int v, const int tx, const int ty, const CamUniform& cam, const unsigned long long* __restrict__ keys,
                            const unsigned* __restrict__ tile_start, const GeomRecs recs, const int P, const unsigned cap,
                            float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8, const int* __restrict__ rank,
                            unsigned long long* __restrict__ prof, float4* __restrict__ s_a, float4* __restrict__ s_b,
                            float2* __restrict__ s_c, float4* __restrict__ s_raw, const int lane) {
    static_assert(MODE == 0 || MODE == 2 || MODE == 3, "loop forms: 0 (lane masks in scalar registers), 2 (all quadrants, flag-free runs), 3 (2 + alpha cap for every instance)");
    constexpr bool CAP_ALL = MODE == 3;   // see the staging step
    unsigned long long pt_wait = 0, pt_stage = 0, pt_issue = 0, pt_loop = 0, pn_batches = 0, pn_staged = 0;
    // Phase stamps of the profile build: the interval since the previous stamp is added to the phase that ENDS here.
    unsigned long long pt_pro = 0, pt_epi = 0;
    const unsigned long long pt_begin = PROF ? gs2m_clock() : 0ull;
    unsigned long long pt_mark = pt_begin;
    auto stamp = [&](unsigned long long& acc) __attribute__((always_inline)) {
        if (PROF) {
            const unsigned long long now = gs2m_clock();
            acc += now - pt_mark;
            pt_mark = now;
        }
    };
    const int W = cam.W, H = cam.H, gx = cam.gx;
    const int ltiles = gx * ((cam.gy + LROWS - 1) / LROWS);    // instance lists
    const int px0 = tx * GS2M_TILE + (lane & 7), py0 = ty * GS2M_TILE + (lane >> 3);
    float pxf0 = (float)px0, pxf1 = (float)(px0 + 8), pyf0 = (float)py0, pyf1 = (float)(py0 + 8);
    GS2M_KEEP_F32(pxf0);
    GS2M_KEEP_F32(pxf1);
    GS2M_KEEP_F32(pyf0);
    GS2M_KEEP_F32(pyf1);
    float T[4], C0[4], C1[4], C2[4];
    unsigned long long dn[4];  // MODE 0: finished (or outside-the-image) pixels of quadrant k: lane mask in scalar registers
    // MODE 2: no lane masks in scalar registers -- the scalar unit issues ONE instruction per cycle for the four SIMDs of a
    // CU (tools/ubench/salu_rates.hip) and the mask bookkeeping of MODE 0 (7 scalar instructions + 3 branches per
    // contributing quadrant) made the kernel scalar-issue-bound.  A finished pixel raises its own threshold, the candidate
    // test is one v_cmp against that threshold and the accumulate path runs under the execution mask.
    float thr[4];
    const float QMIN_ = -7.99435343685885793770f;
    // threshold of a finished pixel: q <= log2(opacity) <= 0 for every accepted contribution (an inline constant, no register)
    const float THR_DONE = 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
        T[k] = 1.0f;
        C0[k] = C1[k] = C2[k] = 0.0f;
        dn[k] = gs2m_ballot_b(!(x < W && y < H));
        thr[k] = (x < W && y < H) ? QMIN_ : THR_DONE;
    }
    const float qx0 = (float)(tx * GS2M_TILE), qy0 = (float)(ty * GS2M_TILE);
    const int ltile = (ty / LROWS) * gx + tx;
    unsigned r0 = tile_start[(size_t)v * (ltiles + 1) + ltile];
    unsigned r1 = tile_start[(size_t)v * (ltiles + 1) + ltile + 1];
    if (r0 > cap) r0 = cap;
    if (r1 > cap) r1 = cap;
    r0 = (unsigned)gs2m_uniform((int)r0);
    r1 = (unsigned)gs2m_uniform((int)r1);
    const unsigned long long* kv = keys + (size_t)v * cap;
    const GeomRecs rv = gs2m_recs_at(recs, (size_t)v * P);
    const float LOG2E = 1.44269504088896340736f;
    const float QMIN = -7.99435343685885793770f;  // -log2(255): alpha >= 1/255 <=> q >= QMIN (decided in the log2 domain)
    // ---- prefetch pipeline: ids two batches ahead (one VGPR), records one batch ahead (DMA into LDS).  With a packed
    // (spatially ordered) model the low word of a key is the Gaussian's id and `rank` maps it to the position of its
    // record: one more stage, keys three batches ahead (`kid_next2`, ids) and their ranks two batches ahead.
    unsigned base = r0;
    unsigned gid_next = 0u;   // record index of instance base + 64 + lane
    unsigned kid_next2 = 0u;  // (rank != null) id of instance base + 128 + lane
    {
        unsigned gid = 0u;
        if (base + (unsigned)lane < r1) gid = (unsigned)(kv[base + lane] & 0xffffffffull);
        if (base + 64u + (unsigned)lane < r1) gid_next = (unsigned)(kv[base + 64u + lane] & 0xffffffffull);
        if (rank) {
            if (base + 128u + (unsigned)lane < r1) kid_next2 = (unsigned)(kv[base + 128u + lane] & 0xffffffffull);
            if (base + (unsigned)lane < r1) gid = (unsigned)rank[gid];
            if (base + 64u + (unsigned)lane < r1) gid_next = (unsigned)rank[gid_next];
        }
        if (base + (unsigned)lane < r1) {
            const float4* r4 = rv.ab + 2 * (size_t)gid;
            gs2m_global_load_lds16(r4, &s_raw[0]);
            gs2m_global_load_lds16(r4 + 1, &s_raw[64]);
            gs2m_global_load_lds16(rv.c + gid, &s_raw[128]);
        }
    }
    stamp(pt_pro);   // prologue: first ids / ranks / record DMA
    while (base < r1) {
        unsigned lq = 0u;  // quadrants that still have unfinished pixels
#pragma unroll
        for (int k = 0; k < 4; ++k)
            lq |= ((MODE != 0 ? gs2m_ballot_b(thr[k] < 0.0f) != 0ull : dn[k] != ~0ull) ? 1u : 0u) << k;
        if (lq == 0u) break;
        stamp(pt_loop);        // (tail of the previous batch's loop + the all-finished test)
        gs2m_wait_dma();       // this batch's records have landed in s_raw
        stamp(pt_wait);
        gs2m_wave_sync();      // ... and the previous batch's staged instances have been read by every lane
        int nb_staged = 0;
        unsigned long long gslots = 0ull;   // MODE 2: staged slots that take the general path
        {
            const float4 ra = s_raw[lane], rb = s_raw[64 + lane], rc = s_raw[128 + lane];
            const bool have = base + (unsigned)lane < r1;
            const float op = have ? rb.y : 1.0f;
            const float lo = gs2m_fast_log2(op);                           // log2 o
            // The bounding box is a conservative pre-filter (margins: 1e-4 on the threshold, 0.1 % + 0.01 px on the half
            // extents): 1-ulp hardware rcp / sqrt and ln(255 o) = (log2 o + log2 255) ln 2 instead of a second logarithm, a
            // correctly rounded division and two correctly rounded square roots (-30 % of the staging step's vector ops).
            const float t2 = fmaxf(2.0f * ((lo + 7.99435343685885793770f) * 0.69314718055994530942f + 1.0e-4f), 0.0f);  // 2 ln(255 o) (+ margin)
            const float det = ra.z * rb.x - ra.w * ra.w;                   // conic determinant (> 0)
            const float inv = gs2m_fast_rcp(det);
            const float hx = gs2m_fast_sqrt(t2 * rb.x * inv) * 1.001f + 0.01f;      // cov_xx = cc/det
            const float hy = gs2m_fast_sqrt(t2 * ra.z * inv) * 1.001f + 0.01f;      // cov_yy = ca/det
            const bool xl = ra.x - hx <= qx0 + 7.0f, xr = ra.x + hx >= qx0 + 8.0f;
            const bool box = op * 255.0f >= 0.9999f && det > 0.0f;
            const bool degenerate = !(det > 0.0f);  // no box: test every pixel of the tile rect
            const int ry0 = (int)(__float_as_uint(rc.z) >> 16), ry1 = (int)(__float_as_uint(rc.w) >> 16);  // rect rows, 16-px units
            unsigned m = 0u;
#pragma unroll
            for (int qr = 0; qr < 2; ++qr) {  // 8-pixel quadrant rows of the tile
                const float top = qy0 + 8.0f * (float)qr;
                bool row = degenerate || (box && ra.y - hy <= top + 7.0f && ra.y + hy >= top);
                if (LROWS > 1) row = row && ty >= ry0 && ty < ry1;  // the 16 x 16 tile must lie in the instance's rect
                if (row && (degenerate || xl)) m |= 1u << (2 * qr);
                if (row && (degenerate || xr)) m |= 2u << (2 * qr);
            }
            // general path (power > 0 test + alpha cap) only where it can matter: opacity near the 0.99 cap, or a conic so
            // close to singular that rounding could make the quadratic form negative
            const bool singular = !(ra.z > 0.0f && rb.x > 0.0f && det >= 1.0e-3f * ra.z * rb.x);
            const bool capped = !(op <= 0.98f);                // the alpha cap can bind (forward.cu:343)
            const bool general = capped || singular;           // MODE 2: resolved per RUN of the batch
            BlendInst bi;
            bi.a = ra;
            bi.b = rb;
            bi.a.z = (-0.5f * LOG2E) * ra.z;
            bi.a.w = LOG2E * ra.w;
            bi.b.x = (-0.5f * LOG2E) * rb.x;
            bi.b.y = lo;
            bi.c.x = rc.x;
            bi.c.y = __uint_as_float(m | (general ? 0x100u : 0u));
            if (LROWS > 1 || MODE >= 2) {
                // the list also serves the other half of the 16 x 32 tile: stage only the instances that reach this
                // half (ballot compaction), so the compositing loop never iterates over the others
                // MODE 2 has no per-instance mask test in the loop: instances that only reach finished quadrants go here too
                const bool mine = (MODE >= 2 ? (m & lq) != 0u : m != 0u) && have;
                const unsigned long long keep = gs2m_ballot_b(mine);
                const int slot = gs2m_popc64(keep & ((1ull << lane) - 1ull));
                if (mine) {
                    s_a[slot] = bi.a;
                    s_b[slot] = bi.b;
                    s_c[slot] = bi.c;
                }
                nb_staged = gs2m_popc64(keep);
                if (MODE >= 2) {
                    // slots (positions after compaction) of the instances that need the general path: scalar bit loop
                    // over the (few) flagged lanes -- slot = number of kept lanes below
                    // MODE 3 (round 6): only the near-singular instances (rare) split the batch into runs; the alpha cap is applied
                    // to EVERY instance in the pipelined runs (min(0.99, alpha): +1 vector instruction per contributing quadrant,
                    // the identity where the cap cannot bind -- same image).  A trained splat has ~30 % of its opacities at the
                    // cap: as flagged instances they cut the batch every ~2.4 staged instances, i.e. no pipelined run at all
                    // (C2-sized trained-like scene: 190 -> 172 us per pair; synth_v1, 2 % capped: 170 -> 176, so mode 2 stays the
                    // default and the pipeline level picks by the model's share of capped opacities, rasterizer.auto_blend_mode).
                    unsigned long long gl = gs2m_ballot_b((CAP_ALL ? singular : general) && mine);
                    gslots = 0ull;
                    while (gl != 0ull) {
                        const int g = __ffsll((long long)gl) - 1;
                        gl &= gl - 1ull;
                        gslots |= 1ull << gs2m_popc64(keep & ((1ull << g) - 1ull));
                    }
                }
            } else {
                s_a[lane] = bi.a;
                s_b[lane] = bi.b;
                s_c[lane] = bi.c;
            }
        }
        gs2m_wave_sync();   // s_raw has been consumed, the staged batch is complete
        stamp(pt_stage);
        const int nb = (LROWS > 1 || MODE >= 2) ? nb_staged : ((int)(r1 - base) < 64 ? (int)(r1 - base) : 64);
        base += 64u;
        if (base + (unsigned)lane < r1) {  // records of the next batch -> LDS while this one is composited
            const float4* r4 = rv.ab + 2 * (size_t)gid_next;
            gs2m_global_load_lds16(r4, &s_raw[0]);
            gs2m_global_load_lds16(r4 + 1, &s_raw[64]);
            gs2m_global_load_lds16(rv.c + gid_next, &s_raw[128]);
        }
        if (rank) {
            // both loads are consumed after the next gs2m_wait_dma: the rank of the ids fetched a batch ago, and new ids
            if (base + 64u + (unsigned)lane < r1) gid_next = (unsigned)rank[kid_next2];
            if (base + 128u + (unsigned)lane < r1) kid_next2 = (unsigned)(kv[base + 128u + lane] & 0xffffffffull);
        } else if (base + 64u + (unsigned)lane < r1) {
            gid_next = (unsigned)(kv[base + 64u + lane] & 0xffffffffull);
        }
        stamp(pt_issue);
        if (PROF) {
            pn_batches += 1;
            pn_staged += (unsigned long long)nb;
        }
        if (MODE >= 2) {
            // ---- MODE 2 (round 5): "evaluate all four quadrants".  The loop of MODE 0 spends more scalar than vector issue
            // (per instance-wave on C2: 134 vector cycles per SIMD, but 28 scalar-ALU instructions at ~4.2 SIMD-cycles each
            // when the CU's one scalar unit is the bottleneck + 12 branches; tools/ubench/salu_rates.hip, r5_rates.hip):
            // quadrant-mask bit tests, lane-mask bookkeeping, the per-quadrant flag test.  Here an instance that reaches
            // the loop gets all four quadrants evaluated (2 FMAs + 1 compare each; the 8-px column / row terms are shared),
            // the candidate test is ONE v_cmp against the lane's own threshold (a finished pixel carries THR_DONE) followed by
            // a branch on VCC, the accumulate path runs under the execution mask, and the instances that need the
            // reference's power > 0 test / alpha cap (forward.cu:336-343) are split out per RUN: the staged batch is a
            // sequence of flag-free runs (software-pipelined fast copy, no cap: op <= 0.98) separated by single flagged
            // instances (general copy), found with scalar bit scans of `gslots` -- no flag test per instance or quadrant.
            struct Pre2 {
                float e0, e1, n0, n1, dy0, dy1;
            };
            // column / row terms of one instance (the first use of its broadcast registers: the LDS wait sits here, BEFORE
            // the next instance's reads are issued, so it never covers a read that was just issued)
            auto pre2 = [&](const float4 A, const float4 B) __attribute__((always_inline)) {
                const float dx0 = A.x - pxf0, dx1 = A.x - pxf1;
                Pre2 p;
                p.dy0 = A.y - pyf0;
                p.dy1 = A.y - pyf1;
                p.e0 = fmaf(A.z * dx0, dx0, B.y);
                p.e1 = fmaf(A.z * dx1, dx1, B.y);
                p.n0 = -(A.w * dx0);
                p.n1 = -(A.w * dx1);
                return p;
            };
            auto quads2 = [&](auto gen_tag, const Pre2 p, const float4 B, const float CLx) __attribute__((always_inline)) {
                constexpr int GTAG = (int)decltype(gen_tag)::value;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dy = (k >> 1) ? p.dy1 : p.dy0;
                    const float qv = fmaf(fmaf(B.x, dy, (k & 1) ? p.n1 : p.n0), dy, (k & 1) ? p.e1 : p.e0);   // same products and sums as MODE 0
                    const bool cand = qv >= thr[k];
                    if (gs2m_any_active_lane(cand)) {     // v_cmp + s_cbranch_vccz
                        GS2M_NO_IF_CONVERT();
                        if (cand) {
                            float alpha = gs2m_fast_exp2(qv);
                            if (GTAG != 0) alpha = fminf(0.99f, alpha);                // alpha cap (forward.cu:343)
                            if (GTAG == 1) alpha = qv > B.y ? 0.0f : alpha;            // power > 0: skipped (forward.cu:336-337)
                            // Round 6: w = alpha T (the reference's own weight, forward.cu:355) and T -= alpha T IN PLACE -- the round-5
                            // form (T' = fma(-T, alpha, T), w = T - T') kept T and T' alive together and ended every contributing
                            // quadrant on a register copy (1 of its 8 vector instructions).  A pixel that saturates (T - alpha T <
                            // 1e-4: its contribution is dropped, it is finished and keeps T) takes the subtraction back in the rare
                            // path: fl(fl(T - w) + w) may differ from T in the last bit, on a transmittance < ~1e-2 that only
                            // enters out = C + T bg; MODE 0 does the same, lane for lane.
                            float wT = T[k] * alpha;
                            T[k] = fmaf(-T[k], alpha, T[k]);     // T - alpha T with one rounding
                            if (gs2m_any_active_lane(T[k] < 0.0001f)) {   // a pixel saturates once: behind a wave-uniform branch
                                GS2M_NO_IF_CONVERT();
                                const bool sat = T[k] < 0.0001f;
                                thr[k] = sat ? THR_DONE : thr[k];
                                T[k] = sat ? gs2m_add_rn(T[k], wT) : T[k];
                                wT = sat ? 0.0f : wT;
                            }
                            C0[k] = fmaf(B.z, wT, C0[k]);
                            C1[k] = fmaf(B.w, wT, C1[k]);
                            C2[k] = fmaf(CLx, wT, C2[k]);
                        }
                    }
                }
            };
            const float4 *spa = s_a, *spb = s_b;
            const float2* spc = s_c;
            int j0 = 0;
            while (j0 < nb) {
                // next flagged slot at or after j0 (nb if none)
                const unsigned long long gq = gslots >> j0;
                const int j1 = gq != 0ull ? j0 + (__ffsll((long long)gq) - 1) : nb;
                if (j0 < j1) {
                    float4 A0 = spa[j0], B0 = spb[j0], A1, B1;
                    float K0 = spc[j0].x, K1;
                    int j = j0;
                    for (; j + 1 < j1; j += 2) {
                        const Pre2 p0 = pre2(A0, B0);
                        GS2M_SCHED_BARRIER();
                        A1 = spa[j + 1];
                        B1 = spb[j + 1];
                        K1 = spc[j + 1].x;
                        GS2M_SCHED_BARRIER();
                        quads2(std::integral_constant<int, CAP_ALL ? 2 : 0>{}, p0, B0, K0);
                        const Pre2 p1 = pre2(A1, B1);
                        GS2M_SCHED_BARRIER();
                        A0 = spa[j + 2];
                        B0 = spb[j + 2];
                        K0 = spc[j + 2].x;
                        GS2M_SCHED_BARRIER();
                        quads2(std::integral_constant<int, CAP_ALL ? 2 : 0>{}, p1, B1, K1);
                    }
                    if (j < j1) quads2(std::integral_constant<int, CAP_ALL ? 2 : 0>{}, pre2(A0, B0), B0, K0);
                }
                if (j1 < nb) {
                    const float4 Ag = spa[j1], Bg = spb[j1];
                    quads2(std::integral_constant<int, 1>{}, pre2(Ag, Bg), Bg, spc[j1].x);
                }
                j0 = j1 + 1;
            }
            continue;
        }
        // ---- MODE 0: per-pixel decisions as lane masks in scalar registers, per-instance quadrant mask, the instance's flag
        // (bit 8 of the mask word) selects the general path inside the body.  Software-pipelined broadcast reads, unrolled by
        // two with ping-pong registers; the quadrant mask is read (= the LDS wait) BEFORE the next instance's reads are issued.
        auto step_body = [&](const int qmf, const float2 CL, const float4 A, const float4 B) __attribute__((always_inline)) {
            int qm = qmf & (int)lq;
            GS2M_OPAQUE_SGPR(qm);   // one s_and per instance, then s_bitcmp per quadrant (not an s_and + s_cmp per quadrant)
            const float dx0 = A.x - pxf0, dx1 = A.x - pxf1;
            const float e[2] = {fmaf(A.z * dx0, dx0, B.y), fmaf(A.z * dx1, dx1, B.y)};
            const float nbdx[2] = {-(A.w * dx0), -(A.w * dx1)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (qm & (1 << k)) {  // scalar branch: quadrant k intersects the splat's box and has unfinished pixels
                    const float dy = A.y - ((k >> 1) ? pyf1 : pyf0);
                    // q = e + dy (c' dy - b' dx): two FMAs
                    const float qv = fmaf(fmaf(B.x, dy, nbdx[k & 1]), dy, e[k & 1]);
                    const unsigned long long prem = gs2m_ballot_b(qv >= QMIN) & ~dn[k];   // v_cmp + s_andn2
                    if (prem != 0ull) {
                        float alpha = gs2m_fast_exp2(qv);
                        unsigned long long candm = prem;
                        if (qmf & 0x100) {  // scalar branch: general path (rare)
                            GS2M_NO_IF_CONVERT();
                            candm = prem & ~gs2m_ballot_b(qv > B.y);       // power > 0: skipped (forward.cu:336-337)
                            alpha = fminf(0.99f, alpha);
                        }
                        // the same arithmetic as MODE 2, lane for lane: w = alpha T, T - w with one rounding, and a saturating lane
                        // takes the subtraction back (fl(fl(T - w) + w)); lanes that are no candidates keep T untouched
                        const float w0 = T[k] * alpha;
                        const float Ts = fmaf(-T[k], alpha, T[k]);
                        const unsigned long long satm = gs2m_ballot_b(Ts < 0.0001f) & candm;
                        const float Tn = gs2m_lanes(satm) ? gs2m_add_rn(Ts, w0) : Ts;
                        const float wT = gs2m_lanes(candm & ~satm) ? w0 : 0.0f;
                        C0[k] = fmaf(B.z, wT, C0[k]);
                        C1[k] = fmaf(B.w, wT, C1[k]);
                        C2[k] = fmaf(CL.x, wT, C2[k]);
                        T[k] = gs2m_lanes(candm) ? Tn : T[k];
                        dn[k] |= satm;
                    }
                }
            }
        };
        {
            const float4 *spa = s_a, *spb = s_b;
            const float2* spc = s_c;
            float4 A0 = spa[0], B0 = spb[0], A1, B1;
            float2 K0 = spc[0], K1;
            int j = 0;
            for (; j + 1 < nb; j += 2) {
                const int qm0 = gs2m_uniform((int)__float_as_uint(K0.y));
                GS2M_SCHED_BARRIER();
                A1 = spa[j + 1];
                B1 = spb[j + 1];
                K1 = spc[j + 1];
                GS2M_SCHED_BARRIER();
                if (qm0 & (int)lq) step_body(qm0, K0, A0, B0);
                const int qm1 = gs2m_uniform((int)__float_as_uint(K1.y));
                GS2M_SCHED_BARRIER();
                A0 = spa[j + 2];
                B0 = spb[j + 2];
                K0 = spc[j + 2];
                GS2M_SCHED_BARRIER();
                if (qm1 & (int)lq) step_body(qm1, K1, A1, B1);
            }
            if (j < nb) {
                const int qm0 = gs2m_uniform((int)__float_as_uint(K0.y));
                if (qm0 & (int)lq) step_body(qm0, K0, A0, B0);
            }
        }
    }
    stamp(pt_loop);
    gs2m_wait_dma();   // never leave (or re-use the landing zone) with a DMA write to this wave's LDS in flight
    stamp(pt_wait);
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
    if (PROF && prof) {
        stamp(pt_epi);
        if (lane == 0) {
            // 64 copies of the counter block, one per 128 B: a launch ends ~30 k waves, all on the same 10 words otherwise
            unsigned long long* pc = prof + (size_t)(blockIdx.x & 63u) * 16;
            atomicAdd(&pc[0], 1ull);
            atomicAdd(&pc[1], pt_mark - pt_begin);
            atomicAdd(&pc[2], pt_wait);
            atomicAdd(&pc[3], pt_stage);
            atomicAdd(&pc[4], pt_issue + pt_pro);
            atomicAdd(&pc[5], pt_loop);
            atomicAdd(&pc[6], pt_epi);      // background blend + image stores
            atomicAdd(&pc[7], pn_batches);
            atomicAdd(&pc[8], pn_staged);
            atomicAdd(&pc[9], (unsigned long long)(r1 - r0));
        }
    }
}

// Wave slot j of XCD x of view v -> tile, by k_tile_scan's schedule (`order`): chunks of GS2M_SCHED_CW x GS2M_SCHED_CH neighbouring
// lists ranked by descending weight; rank p is the (p / 8)-th chunk of XCD p % 8; slot j takes half (j % LROWS) of list
// (j / LROWS) % CHUNK of that XCD's (j / (CHUNK * LROWS))-th chunk.  false: no such tile.
template <int LROWS>
GS2M_DEVICE bool blend_slot_tile(const CamUniform& cam, const unsigned* __restrict__ order, const int v, const int xcd, const int j,
                                 int& tx, int& ty) {
    const int gx = cam.gx;
    const int ltiles = gx * ((cam.gy + LROWS - 1) / LROWS);
    const int lrows = ltiles / gx, cpr = (gx + GS2M_SCHED_CW - 1) / GS2M_SCHED_CW;
    const int nch = cpr * ((lrows + GS2M_SCHED_CH - 1) / GS2M_SCHED_CH);
    const int rank_c = (j / (GS2M_SCHED_CHUNK * LROWS)) * 8 + xcd;
    if (rank_c >= nch) return false;
    const int c = (int)order[(size_t)v * ltiles + rank_c];
    const int k = (j / LROWS) % GS2M_SCHED_CHUNK;                 // list of the chunk
    const int crow = c / cpr, lx = (c - crow * cpr) * GS2M_SCHED_CW + k % GS2M_SCHED_CW;
    const int ly = crow * GS2M_SCHED_CH + k / GS2M_SCHED_CW;
    if (lx >= gx || ly >= lrows) return false;
    tx = lx;
    ty = ly * LROWS + j % LROWS;
    return ty < cam.gy;
}

// One wave per tile; the grid covers the schedule (block b runs on XCD b % 8, blocks are dispatched in order).  Round 6 measured a
// RESIDENT grid whose waves pull the slots of their XCD's schedule from a ticket counter (dynamic hand-out): C2
// 174 -> 207 us per pair, C3 205 -> 253, trained-like 191 -> 231, pipelined step 0.272 -> 0.300 ms, images bit-identical
// (profiles/r6_traces/ab_blend_persistent.jsonl) -- as in round 2 (+30 %), the hardware's dispatch order IS the better dealer;
// removed again.
template <int WPB, int LROWS, int OCC, int MODE = 0, int PROF = 0>
GS2M_KERNEL void __launch_bounds__(64 * WPB, OCC)
k_blend_wave4e(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ tile_start,
               const GeomRecs recs, const CamUniform* __restrict__ cams, int P, unsigned cap,
               float* __restrict__ out_color, unsigned char* __restrict__ out_rgb8, const int* __restrict__ rank,
               const unsigned* __restrict__ order, unsigned long long* __restrict__ prof = nullptr, int nv_x = 0) {
    // staged instance (40 B in three arrays): a = {mx, my, a' = -0.5 log2e ca, b' = log2e cb}, b = {c' = -0.5 log2e cc,
    // log2 o, r, g}, c = {b, quadrant mask | general << 8 (bits)}; 2 pad slots: the software-pipelined reads run 2
    // instances ahead.  5.6 KiB of LDS per wave with the DMA landing zone: 7 waves per SIMD fit the 160 KiB.
    __shared__ float4 s_a[WPB][64 + 2], s_b[WPB][64 + 2];
    __shared__ float2 s_c[WPB][64 + 2];
    __shared__ float4 s_raw[WPB][3 * 64];   // DMA landing zone: the three 16-B vectors of the next batch's GeomRecs
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // Views interleaved along the schedule (nv_x > 1, MODE 2): per XCD the order is chunk rank major, view minor, so the heavy
    // chunks of ALL views of the launch start first.  With the views one after the other the heavy chunks of the last view
    // were dispatched in the last quarter of a 4-view launch and the kernel ended on them.
    const int nvx = gs2m_uniform(nv_x);   // > 1: that many views interleaved
    const int v = nvx > 1 ? (int)((blockIdx.x / 8u) % (unsigned)nvx) : (int)blockIdx.y;
    const unsigned bid = nvx > 1 ? (blockIdx.x / 8u / (unsigned)nvx) * 8u + blockIdx.x % 8u : blockIdx.x;
    int tx, ty;
    if (!blend_slot_tile<LROWS>(cams[v], order, v, (int)(bid % 8u), gs2m_uniform((int)(bid / 8u) * WPB + wave), tx, ty)) return;
    blend_tile<WPB, LROWS, MODE, PROF>(v, tx, ty, cams[v], keys, tile_start, recs, P, cap, out_color, out_rgb8, rank, prof, &s_a[wave][0],
                                       &s_b[wave][0], &s_c[wave][0], &s_raw[wave][0], lane);
}
