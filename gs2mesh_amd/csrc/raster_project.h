// raster_project.h -- kernels of the projection / tile-counting / instance-scatter stages.
//
// Stage map (reference -> here):
//   preprocessCUDA            forward.cu:155-256          -> k_project (fused activations, Sigma once per
//                                                            Gaussian for all views of the batch)
//   cub InclusiveSum + D2H    rasterizer_impl.cu:277-281  -> k_count_tiles (per-workgroup LDS tile histogram
//   duplicateWithKeys         rasterizer_impl.cu:70-111      instead of a per-Gaussian scan)
//                                                         -> k_scatter (LDS cursors, 8-B keys)
// Instead of emitting (tile<<32|depth, id) pairs in Gaussian order and radix-sorting 64-bit keys
// through HBM 5-6 times, instances are counting-sorted by tile with workgroup-private LDS
// histograms (160 KiB LDS holds a 40k-tile cursor array), then depth-sorted per tile in
// registers (raster_sort.h).  The final order is identical to the reference's: ascending depth bits, ties
// by ascending Gaussian id (= what the stable radix sort over emission order produces).
#pragma once
#include "raster_math.h"

// Workgroup-private tile histogram of k_count_tiles: one u32 counter per (view, tile) in LDS.  (Rounds 2-5 packed two 16-bit
// counters per word to halve the LDS; the kernel runs one 1024-thread workgroup per CU either way -- the scatter kernel's u32
// cursors set the LDS budget of the pair -- and the hi / lo select cost 4 of the 6 vector instructions of a bump in a kernel that
// is vector-issue-bound.  Round 6 also measured, and dropped, wave-aggregated bumps (one atomic per run of neighbouring lanes
// with the same tile) and 2 / 4 lane-interleaved copies of the histogram against the same-address serialisation of a spatially
// ordered model's atomics: C3 count 29.7 -> 36.8 us (aggregated), 30.4 -> 29.6 (copies) -- profiles/r6_experiments.txt.)
GS2M_DEVICE void hist_bump(unsigned* hh, int t) { atomicAdd(&hh[t], 1u); }

// ---- which rects of the binning grid get what (k_count_tiles and k_scatter must agree on `tested`: it decides whether a
// tile mask exists) ---------------------------------------------------------------------------------------------------------
// cull = GS2M_OPT_EXACT_TILE_CULL: 0 = every tile of the (reference) rect is an instance; 1 = the rect is the bounding box of the
// alpha >= 1/255 ellipse and every tile of a rect with corners (>= 2 x 2 tiles) is tested against the ellipse; 2 (round 6) = the
// same, but rects of at most 4 tiles keep all their tiles -- the test only removes ~2 % of the instances of a small-splat
// scene (C3: 3.85 M -> 3.77 M per eye) and was 60 % of the vector instructions of the counting kernel, which is VALU-issue-bound
// (PMC round 5: 12 k vector instructions per wave, 0.8 of the SIMD's issue slots).  Rects of more than 64 tiles are walked by the
// whole wave and test every tile.  A thin rect (one tile wide or high) is its own bounding box: never tested.
GS2M_DEVICE bool gs2m_rect_tested(unsigned w, unsigned h, unsigned area, int cull) {
    return cull != 0 && (area > 64u || (w >= 2u && h >= 2u && (cull == 1 || area > 4u)));
}
// A rect of the binning grid that is one tile wide or one tile high and has at most GS2M_THIN_MAX tiles (round 4).
#define GS2M_THIN_MAX 4u
GS2M_DEVICE bool gs2m_thin_rect(unsigned w, unsigned h, unsigned area) {
    return area != 0u && area <= GS2M_THIN_MAX && (w == 1u || h == 1u);
}
// Rects walked by THEIR OWN LANE (no staging, no item space): the thin ones and, round 6, every rect of at most `lane_max` tiles
// (GS2M_OPT_BIN_LANE_TILES) -- in the counting kernel only when it is not tested (the test costs ~35 vector instructions per
// tile: a lane running it 4 times in sequence lost to the staged walk, which spreads the tiles over the lanes; measured), in
// the scatter kernel always (it replays the mask the counting pass wrote: a bit test per tile).
#define GS2M_LANE_TILES_MAX 16
GS2M_DEVICE bool gs2m_lane_rect(unsigned w, unsigned h, unsigned area, unsigned lane_max) {
    return area != 0u && (area <= lane_max || gs2m_thin_rect(w, h, area));
}

struct ProjView {
    float mx, my, ca, cb, cc, depth, cova, covc;
    int x0, y0, x1, y1, radius;
    bool ok;
};

// Geometry of one Gaussian in one view: forward.cu:186-237 up to the zero-area test.
GS2M_DEVICE void project_view(const CamUniform& cam, float px, float py, float pz, const float* cov3, ProjView& o) {
    o.ok = false;
    o.radius = 0;
    o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.mx = o.my = o.ca = o.cb = o.cc = o.depth = o.cova = o.covc = 0.0f;
    float tvx, tvy, tvz;
    xform4x3(cam.view, px, py, pz, tvx, tvy, tvz);
    if (tvz <= 0.2f) return;  // auxiliary.h:154 near cull
    float hx, hy, hz, hw;
    xform4x4(cam.proj, px, py, pz, hx, hy, hz, hw);
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ppx = hx * p_w, ppy = hy * p_w;
    float a, b, c;
    cov2d_ewa(cam.view, tvx, tvy, tvz, cam.focal_x, cam.focal_y, cam.tanfovx, cam.tanfovy, cov3, a, b, c);
    const float det = (a * c - b * b);
    if (det == 0.0f) return;
    const float det_inv = 1.f / det;
    o.cova = a;
    o.covc = c;
    o.ca = c * det_inv;
    o.cb = -b * det_inv;
    o.cc = a * det_inv;
    const float mid = 0.5f * (a + c);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    o.mx = ndc2pix(ppx, cam.W);
    o.my = ndc2pix(ppy, cam.H);
    const int r = (int)my_radius;
    // auxiliary.h:46-56 getRect
    o.x0 = gs2m_imin(cam.gx, gs2m_imax(0, (int)((o.mx - r) / GS2M_TILE)));
    o.y0 = gs2m_imin(cam.gy, gs2m_imax(0, (int)((o.my - r) / GS2M_TILE)));
    o.x1 = gs2m_imin(cam.gx, gs2m_imax(0, (int)((o.mx + r + GS2M_TILE - 1) / GS2M_TILE)));
    o.y1 = gs2m_imin(cam.gy, gs2m_imax(0, (int)((o.my + r + GS2M_TILE - 1) / GS2M_TILE)));
    if ((o.x1 - o.x0) * (o.y1 - o.y0) == 0) {
        o.x0 = o.y0 = o.x1 = o.y1 = 0;
        return;
    }
    o.depth = tvz;
    o.radius = r;
    o.ok = true;
}

// ---- wave-balanced tile expansion ------------------------------------------------------------
// Every Gaussian touches a different number of tiles (1 ... hundreds).  Walking the rect per lane
// (as duplicateWithKeys does, rasterizer_impl.cu:98-108) leaves most lanes idle while the wave waits for
// its largest rect (C2: mean 8.6 tiles, mean of the per-wave maximum 38).  Instead:
//   * the owners of small rects (<= 64 tiles) are compacted (rank k by ballot) and an exclusive wave scan
//     of their areas flattens all their (Gaussian, tile) pairs into one item space of <= 4096 items;
//   * a bit array marks the first item of every owner ("heads"); lane l of batch b0 takes item b0 + l and
//     finds its owner as  k = (#heads before the batch) + popcount(heads_word & lanes <= l) - 1
//     -- one broadcast LDS read and a popcount instead of a binary search;
//   * rects of more than 64 tiles (rare) are walked by the whole wave, one owner at a time.
// Each kept pair bumps the workgroup's LDS tile histogram; the kept tiles of a small rect are recorded as a
// bit mask (assembled from ballots by the first lane of each owner's run) that the scatter kernel replays.
struct WaveStage {
    float mx[64], my[64], ca[64], cb[64], cc[64], thr[64], rx[64], ry[64];  // indexed by owner rank k
    unsigned swh[64];   // first item (16 bits) | w << 16 | h << 24
    unsigned xy0[64];   // x0 | y0 << 16
    unsigned mlo[64], mhi[64];  // kept-tile mask of the owner's rect
    unsigned heads[130];        // bit i: item i is the first item of an owner (+ slack for the 64-bit window)
    unsigned pad[2];
};
#define GS2M_STAGE_BYTES_PER_WAVE ((int)sizeof(WaveStage))

GS2M_DEVICE unsigned long long lanes_le(int lane) { return (2ull << lane) - 1ull; }
GS2M_DEVICE unsigned long long lanes_lt(int lane) { return (1ull << lane) - 1ull; }

// li -> (rx, ry) of a rect of width ow without an integer division (li < 2^16, exact after one fix-up)
GS2M_DEVICE void rect_coords(unsigned li, unsigned ow, float inv_w, unsigned& rx, unsigned& ry) {
    ry = (unsigned)((float)li * inv_w);
    int r = (int)li - (int)(ry * ow);
    if (r < 0) {
        ry -= 1u;
        r += (int)ow;
    } else if (r >= (int)ow) {
        ry += 1u;
        r -= (int)ow;
    }
    rx = (unsigned)r;
}

GS2M_DEVICE unsigned wave_inclusive_scan(unsigned x) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned y = gs2m_shfl_up(x, d);
        if (gs2m_lane() >= d) x += y;
    }
    return x;
}

// Projection + colour, one thread per Gaussian -- a streaming kernel (no LDS, no loop): activations, Sigma once
// for all views of the batch, EWA projection, SH colour, the (cull-tightened) tile rect -> GeomRec.
#define GS2M_PROJECT_THREADS 256
// One Gaussian (array position gi) for the `groups` x NV views of the launch: parameter loads (+ fused activations) and Sigma
// ONCE, then per group of NV views (a stereo pair) the EWA projection, SH colour, the (cull-tightened) tile rect and the GeomRec
// stores.  Round 6: the groups of a launch (GS2M_OPT_PAIR_BATCH) used to be blockIdx.y -- every group re-read the 44 B of
// parameters and the 192-B SH row of every Gaussian, 71 % of the kernel's bytes; now the thread that owns the Gaussian walks the
// groups (LOOP), the row stays in LDS (DMA_SH) or is re-read through the caches (streamed path): C3 project 113.5 -> 86.5 us per
// pair.  The loop costs registers (84 -> 128, 106 -> 152: 4 -> 3 waves per SIMD on the streamed path) and halves the waves of the
// launch, and a model that fits the 256 MB last-level cache re-reads cheaply anyway (C2: 18.4 -> 19.5 us with the loop): the
// launcher keeps the groups on blockIdx.y (LOOP = false, `groups` = 1) for models below GS2M_PROJECT_LOOP_MIN_P Gaussians.
// s_sh = this wave's DMA landing zone, [12][64] float4 (DMA_SH only).
#define GS2M_PROJECT_LOOP_MIN_P 1000000
template <int NV, bool DMA_SH, bool STREAM = true, bool LOOP = false>
GS2M_DEVICE void project_gaussian(const GaussIn& g, const CamUniform* __restrict__ cams, GeomRecs recs,
                                  int* __restrict__ radii, int exact_cull, int gi, bool valid, float4* s_sh, int lane_id,
                                  int groups) {
    const int ncoef = (g.D + 1) * (g.D + 1);
    constexpr bool dma_sh = DMA_SH;
    {
        if (dma_sh && valid) {
            const float4* s4 = reinterpret_cast<const float4*>(g.shs_packed) + (size_t)(gi >> 6) * (12 * 64) + (gi & 63);
#pragma unroll
            for (int k = 0; k < 12; ++k)
                if (k * 4 < ncoef * 3) gs2m_global_load_lds16(s4 + k * 64, &s_sh[k * 64]);
        }
        if (valid) {
            const float px = g.xyz[3 * (size_t)gi], py = g.xyz[3 * (size_t)gi + 1], pz = g.xyz[3 * (size_t)gi + 2];
            float cov3[6];
            if (g.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) cov3[k] = g.cov3D_precomp[6 * (size_t)gi + k];
            } else {
                float sx = g.scales[3 * (size_t)gi], sy = g.scales[3 * (size_t)gi + 1], sz = g.scales[3 * (size_t)gi + 2];
                const float4 q4 = *reinterpret_cast<const float4*>(g.rots + 4 * (size_t)gi);
                float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
                if (g.raw) {
                    // GaussianModel getters (GS/scene/gaussian_model.py:95-101): exp, F.normalize
                    sx = expf(sx);
                    sy = expf(sy);
                    sz = expf(sz);
                    const float n = fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
                    qr = qr / n;
                    qx = qx / n;
                    qy = qy / n;
                    qz = qz / n;
                }
                cov3d_from_scale_rot(sx, sy, sz, g.scale_modifier, qr, qx, qy, qz, cov3);
            }
            float op = g.opac[gi];
            if (g.raw) op = 1.0f / (1.0f + expf(-op));  // gaussian_model.py:113-115 sigmoid
            const float thr = exact_cull ? cull_threshold(op) : 0.0f;
            const CamUniform* __restrict__ cams0 = cams;
            const GeomRecs recs0 = recs;
            int* __restrict__ const radii0 = radii;
#pragma unroll 1
            for (int grp = 0; grp < (LOOP ? groups : 1); ++grp) {
            cams = cams0 + NV * grp;
            recs = gs2m_recs_at(recs0, (size_t)NV * grp * g.P);
            radii = radii0 ? radii0 + (size_t)NV * grp * g.P : nullptr;
            // the SH row is loop-invariant: re-read it per group (LDS / caches) instead of letting the compiler hoist 48 registers
            // of it over the loop (4 -> 3 waves per SIMD)
            int gi_sh = gi, lane_sh = lane_id;
            if (LOOP) {
                GS2M_OPAQUE_VGPR(gi_sh);
                GS2M_OPAQUE_VGPR(lane_sh);
            }
            ProjView pv[NV];
            bool any = false;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                project_view(cams[v], px, py, pz, cov3, pv[v]);
                any = any || pv[v].ok;
            }
            // colour: the 192-B SH row is STREAMED for all views at once where it is 16-B aligned (packed copy, [P,16,3] or
            // the DMA landing zone): three float4 in flight instead of 48 registers for the row.  dc + rest split layouts
            // and M != 16 keep the register path below.
            // STREAM (chosen by the launcher from the layout) compiles the 48-register path out.
            const bool stream_sh = STREAM && (g.colors_precomp == nullptr) && (dma_sh || g.shs_packed != nullptr || (g.shs_rest == nullptr && g.M == 16));
            float rgb_v[NV][3];
#pragma unroll
            for (int v = 0; v < NV; ++v) rgb_v[v][0] = rgb_v[v][1] = rgb_v[v][2] = 0.0f;
            if (dma_sh && grp == 0) gs2m_wait_dma();   // this lane's row has landed (a lane only reads its own column: no barrier)
            if (stream_sh && any && !dma_sh) {
                float dirs[NV][3];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    // forward.cu:25-27: dir = (pos - campos) / length
                    float dx = px - cams[v].campos[0], dy = py - cams[v].campos[1], dz = pz - cams[v].campos[2];
                    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dirs[v][0] = dx / len;
                    dirs[v][1] = dy / len;
                    dirs[v][2] = dz / len;
                }
                if (g.shs_packed) {
                    // wave-transposed copy (k_pack_sh): float4 j of 64 consecutive Gaussians is 1 KiB contiguous
                    const float4* s4 = reinterpret_cast<const float4*>(g.shs_packed) + (size_t)(gi_sh >> 6) * (12 * 64) + (gi_sh & 63);
                    sh_rgb_stream<NV>(g.D, [&](int j) { return s4[j * 64]; }, dirs, rgb_v);
                } else {
                    const float4* s4 = reinterpret_cast<const float4*>(g.shs + 48 * (size_t)gi_sh);   // 192-B row, 16-B aligned
                    sh_rgb_stream<NV>(g.D, [&](int j) { return s4[j]; }, dirs, rgb_v);
                }
            }
            float sh[STREAM ? 1 : 48];
            const bool need_sh = !STREAM && any && (g.colors_precomp == nullptr) && !stream_sh;
            if constexpr (!STREAM) if (need_sh) {
                if (g.shs_rest == nullptr) {
                    const float* s = g.shs + (size_t)gi_sh * g.M * 3;
#pragma unroll
                    for (int k = 0; k < 48; ++k)
                        if (k < ncoef * 3) sh[k] = s[k];
                } else {
                    const float* s0 = g.shs + (size_t)gi_sh * 3;
                    const float* s1 = g.shs_rest + (size_t)gi_sh * (g.M - 1) * 3;
                    sh[0] = s0[0];
                    sh[1] = s0[1];
                    sh[2] = s0[2];
#pragma unroll
                    for (int k = 3; k < 48; ++k)
                        if (k < ncoef * 3) sh[k] = s1[k - 3];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const size_t ri = (size_t)v * g.P + gi;
                if (radii) radii[(size_t)v * g.P + (g.ids ? g.ids[gi] : gi)] = pv[v].radius;
                if (!pv[v].ok) {
                    // invisible: only the vector holding the (empty) rect is written
                    float4 w2;
                    w2.x = 0.0f;
                    w2.y = 0.0f;
                    w2.z = 0.0f;
                    w2.w = 0.0f;
                    recs.c[ri] = w2;
                    continue;
                }
                float cr, cg, cb;
                if (g.colors_precomp) {
                    cr = g.colors_precomp[3 * (size_t)gi];
                    cg = g.colors_precomp[3 * (size_t)gi + 1];
                    cb = g.colors_precomp[3 * (size_t)gi + 2];
                } else if (dma_sh) {
                    // forward.cu:25-27: dir = (pos - campos) / length; the row is read back from LDS view by view
                    float dirs[1][3], col[1][3];
                    const float dx = px - cams[v].campos[0], dy = py - cams[v].campos[1], dz = pz - cams[v].campos[2];
                    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dirs[0][0] = dx / len;
                    dirs[0][1] = dy / len;
                    dirs[0][2] = dz / len;
                    sh_rgb_stream<1>(g.D, [&](int j) { return s_sh[j * 64 + lane_sh]; }, dirs, col);
                    cr = col[0][0];
                    cg = col[0][1];
                    cb = col[0][2];
                } else if (stream_sh) {
                    cr = rgb_v[v][0];
                    cg = rgb_v[v][1];
                    cb = rgb_v[v][2];
                } else {
                    cr = cg = cb = 0.0f;
                    if constexpr (!STREAM) {
                        // forward.cu:25-27: dir = (pos - campos) / length
                        float dx = px - cams[v].campos[0], dy = py - cams[v].campos[1], dz = pz - cams[v].campos[2];
                        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                        dx = dx / len;
                        dy = dy / len;
                        dz = dz / len;
                        cr = sh_channel(g.D, sh, 0, dx, dy, dz);
                        cg = sh_channel(g.D, sh, 1, dx, dy, dz);
                        cb = sh_channel(g.D, sh, 2, dx, dy, dz);
                    }
                }
                // GS2M_OPT_EXACT_TILE_CULL (image-preserving extension): shrink the reference's AABB-of-
                // the-3-sigma-circle rect to the bounding box of the alpha >= 1/255 ellipse,
                // |dx| <= sqrt(2 t cov_xx), |dy| <= sqrt(2 t cov_yy), t = ln(255 o) (+ margin).  The
                // stored rect is what the counting sort and the scatter walk.
                if (exact_cull) {
                    if (thr < 0.0f) {
                        pv[v].x1 = pv[v].x0 = pv[v].y0 = pv[v].y1 = 0;  // never reaches 1/255 anywhere
                    } else {
                        const float hx = sqrtf(2.0f * thr * pv[v].cova) + 0.01f;
                        const float hy = sqrtf(2.0f * thr * pv[v].covc) + 0.01f;
                        const int bx0 = (int)ceilf((pv[v].mx - hx - (float)(GS2M_TILE - 1)) / GS2M_TILE);
                        const int bx1 = (int)floorf((pv[v].mx + hx) / GS2M_TILE) + 1;
                        const int by0 = (int)ceilf((pv[v].my - hy - (float)(GS2M_TILE - 1)) / GS2M_TILE);
                        const int by1 = (int)floorf((pv[v].my + hy) / GS2M_TILE) + 1;
                        pv[v].x0 = gs2m_imax(pv[v].x0, bx0);
                        pv[v].y0 = gs2m_imax(pv[v].y0, by0);
                        pv[v].x1 = gs2m_imin(pv[v].x1, bx1);
                        pv[v].y1 = gs2m_imin(pv[v].y1, by1);
                        if (pv[v].x1 <= pv[v].x0 || pv[v].y1 <= pv[v].y0) pv[v].x1 = pv[v].x0 = pv[v].y0 = pv[v].y1 = 0;
                    }
                }
                float4 w0, w1, w2;
                w0.x = pv[v].mx;
                w0.y = pv[v].my;
                w0.z = pv[v].ca;
                w0.w = pv[v].cb;
                w1.x = pv[v].cc;
                w1.y = op;
                w1.z = cr;
                w1.w = cg;
                w2.x = cb;
                w2.y = pv[v].depth;
                w2.z = __uint_as_float((unsigned)pv[v].x0 | ((unsigned)pv[v].y0 << 16));
                w2.w = __uint_as_float((unsigned)pv[v].x1 | ((unsigned)pv[v].y1 << 16));
                recs.ab[2 * ri] = w0;
                recs.ab[2 * ri + 1] = w1;
                recs.c[ri] = w2;
            }
            }   // groups
        }
    }
}

// Per-view uniforms from HOST values carried in the kernel arguments (pipeline-level API): the launch packet is the
// transport, so no pinned staging buffer / lifetime hazard.  Round 4: the packet is k_project's own (its workgroups read the
// uniforms of their views straight from the kernel-argument segment -- scalar loads -- and workgroup 0 of every group of views
// stores them to `cams_out` for the later kernels of the pass): one launch less on the critical chain of every pass (round 3:
// k_set_cameras, 4.5 us + a launch gap).
struct CamUniformArg {
    CamUniform c[GS2M_MAX_PASS_VIEWS];
};

// Projection + colour, one thread per Gaussian -- a streaming kernel (no loop): see project_gaussian.
// Packed SH copy (DMA_SH): the 192-B row of every Gaussian of the wave goes straight from HBM into LDS (global_load_lds, 12 x
// 1 KiB per wave), issued BEFORE the parameter loads and the projection: one memory round trip per thread instead of two in
// sequence (the wave spent 69 % of its life waiting, PMC) and no 48 registers holding the row while it is in flight.  The
// colour pass then reads the 16 coefficients of one channel at a time back from LDS.
template <int NV, bool DMA_SH, bool STREAM, bool HOSTCAMS, bool LOOP>
GS2M_DEVICE void project_kernel_body(const GaussIn& g, const CamUniform* __restrict__ cams, GeomRecs recs,
                                     int* __restrict__ radii, int exact_cull, CamUniform* __restrict__ cams_out, int groups) {
    __shared__ float4 s_sh[DMA_SH ? GS2M_PROJECT_THREADS / 64 : 1][DMA_SH ? 12 : 1][DMA_SH ? 64 : 1];
    // the groups of NV views of the launch (GS2M_OPT_PAIR_BATCH: two stereo pairs per launch): LOOP -- all `groups` of them are
    // walked by the thread that owns the Gaussian (project_gaussian); else blockIdx.y = the group of this workgroup
    const int g0 = LOOP ? 0 : (int)blockIdx.y, ng = LOOP ? groups : 1;
    cams += NV * g0;
    if (HOSTCAMS && blockIdx.x == 0) {
        // the uniforms of this workgroup's views -> device memory, for the later kernels of the pass (dword copy)
        const unsigned* src = reinterpret_cast<const unsigned*>(cams);
        unsigned* dst = reinterpret_cast<unsigned*>(cams_out + NV * g0);
        for (unsigned i = threadIdx.x; i < (unsigned)(ng * NV) * (unsigned)(sizeof(CamUniform) / 4u); i += GS2M_PROJECT_THREADS) dst[i] = src[i];
    }
    recs = gs2m_recs_at(recs, (size_t)NV * g0 * g.P);
    if (radii) radii += (size_t)NV * g0 * g.P;
    const int gi = (int)(blockIdx.x * (unsigned)GS2M_PROJECT_THREADS + threadIdx.x);
    const int wave_id = (int)(threadIdx.x >> 6), lane_id = (int)(threadIdx.x & 63u);
    project_gaussian<NV, DMA_SH, STREAM, LOOP>(g, cams, recs, radii, exact_cull, gi, gi < g.P, &s_sh[DMA_SH ? wave_id : 0][0][0], lane_id, ng);
}
// uniforms in device memory (operator-level API: the caller's matrices are device tensors, k_pack_camera)
template <int NV, bool DMA_SH, bool STREAM = true, bool LOOP = false>
GS2M_KERNEL void __launch_bounds__(GS2M_PROJECT_THREADS)   // forcing 5 waves per SIMD on the round-2 kernel (96 VGPRs, spills): C2 30 -> 36 us, C3 149 -> 204
k_project(GaussIn g, const CamUniform* __restrict__ cams, GeomRecs recs, int* __restrict__ radii,
          int exact_cull, int groups) {
    project_kernel_body<NV, DMA_SH, STREAM, false, LOOP>(g, cams, recs, radii, exact_cull, nullptr, groups);
}
// uniforms in the kernel arguments (pipeline-level API, host-side cameras)
template <int NV, bool DMA_SH, bool STREAM = true, bool LOOP = false>
GS2M_KERNEL void __launch_bounds__(GS2M_PROJECT_THREADS)
k_project_hc(GaussIn g, CamUniformArg hc, CamUniform* __restrict__ cams_out, GeomRecs recs, int* __restrict__ radii,
             int exact_cull, int groups) {
    project_kernel_body<NV, DMA_SH, STREAM, true, LOOP>(g, &hc.c[0], recs, radii, exact_cull, cams_out, groups);
}

// Gaussian -> workgroup assignment of the counting sort (k_count_tiles and k_scatter must agree: the histogram row of a
// workgroup describes exactly the Gaussians it scatters).  Step `it` of wave `wave` of the workgroup with histogram row
// `row` starts at the returned index (64 consecutive Gaussians, one per lane); *end = one past the last index it may take;
// returns -1 when the wave is done.
//   contiguous (models in their stored order): the workgroup owns [row * chunk, (row + 1) * chunk) and walks it
//     blockDim.x Gaussians at a time;
//   interleaved (spatially ordered packed model, gs2m_raster_pack_model): blocks of 64 consecutive Gaussians are dealt
//     round-robin to the rows.  A block is a compact screen region (its instances fall into a few tiles: the keys of a
//     store instruction form runs), while every workgroup samples the whole model, so the work stays balanced (contiguous
//     chunks of a Morton-ordered model differ by 2.4x in instances on C2).
GS2M_DEVICE int bin_step_begin(int it, int wave, int nwaves, int row, int n_wg, int chunk, int P, int interleave, int* end) {
    if (interleave) {
        const long long first = (((long long)it * nwaves + wave) * n_wg + row) * 64ll;
        *end = P;
        return first < (long long)P ? (int)first : -1;
    }
    const int stop = gs2m_imin(P, (row + 1) * chunk);   // chunk <= 64000, rows < 2^15: no overflow
    const long long first = (long long)row * chunk + ((long long)it * nwaves + wave) * 64ll;
    *end = stop;
    return first < (long long)stop ? (int)first : -1;
}

// Per-(view, Gaussian) input of the counting step: the geometry half of the GeomRec + the rect in BINNING rows.
struct CountIn {
    float mx, my, ca, cb, cc;
    int x0, y0, x1, y1;
    bool ok;
};
// The tile rect of a record's binning part (16 x 16 tile units) in the binning grid (tiles of 16 x 16 * 2^rs pixels): rs = 0 / 1
// (a shift: the signed divisions by `rows` this replaces were ~100 of the ~800 vector instructions of a C3 counting step).
struct BinRect {
    int x0, y0;
    unsigned w, h, area;
};
GS2M_DEVICE BinRect bin_rect_of(const float4& c, int rs) {
    const unsigned rect0 = __float_as_uint(c.z), rect1 = __float_as_uint(c.w);
    const int x0 = (int)(rect0 & 0xffffu), x1 = (int)(rect1 & 0xffffu);
    const int y0 = (int)(rect0 >> 16), y1 = (int)(rect1 >> 16);
    const bool ok = x1 > x0 && y1 > y0;
    BinRect r;
    r.x0 = x0;
    r.y0 = y0 >> rs;
    r.w = ok ? (unsigned)(x1 - x0) : 0u;
    r.h = ok ? (unsigned)(((y1 + (1 << rs) - 1) >> rs) - r.y0) : 0u;
    r.area = r.w * r.h;
    return r;
}

// The counting step for the 64 Gaussians of one wave step (wave collectives: EVERY lane calls it): balanced
// (Gaussian, tile) expansion, the workgroup's LDS tile histogram, the kept-tile masks of the tested rects.
template <int NV>
GS2M_DEVICE void count_expand(const CountIn* pv, float thr, bool valid, int gi, int P, unsigned* lhist, int tiles, WaveStage* stage,
                              unsigned long long* __restrict__ tilemask, int gx, int th, int cull, int lane, unsigned lane_max) {
    // ---- balanced (Gaussian, tile) expansion, one view at a time (wave collectives: every lane) ----
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const unsigned w = (unsigned)(pv[v].x1 - pv[v].x0), h = (unsigned)(pv[v].y1 - pv[v].y0);
        const unsigned area = (valid && pv[v].ok) ? w * h : 0u;
        if (gs2m_ballot(area != 0u ? 1 : 0) == 0ull) continue;  // wave-uniform
        unsigned* hh = lhist + v * tiles;
        const unsigned xy0 = (unsigned)pv[v].x0 | ((unsigned)pv[v].y0 << 16);
        const bool tested = gs2m_rect_tested(w, h, area, cull);
        // ---- rects walked by their own lane: thin ones (round 4) and untested ones of <= lane_max tiles (round 6).  Every tile
        // is kept: no staging, no item space, no tile mask.  A 2 M-Gaussian scene of small splats is 85 % thin + 14 % 2 x 2
        // rects (C3, 16 x 32 binning tiles).
        const bool lane_rect = !tested && gs2m_lane_rect(w, h, area, lane_max);
        if (gs2m_ballot(lane_rect ? 1 : 0) != 0ull) {
            // row-major walk with a running tile index (no multiplication per tile)
            const unsigned area_l = lane_rect ? area : 0u;
            int tile = pv[v].y0 * gx + pv[v].x0;
            unsigned rx = 0u;
            for (unsigned t = 0; t < area_l; ++t) {
                hist_bump(hh, tile);
                ++tile;
                if (++rx == w) {
                    rx = 0u;
                    tile += gx - (int)w;
                }
            }
        }
        // ---- the other rects of <= 64 tiles: flattened item space ----
        const bool small = area != 0u && area <= 64u && !lane_rect;
        const unsigned long long smalls = gs2m_ballot(small ? 1 : 0);
        if (smalls != 0ull) {
            const int k = gs2m_popc64(smalls & lanes_lt(lane));
            const unsigned incl = wave_inclusive_scan(small ? area : 0u);
            const unsigned total = gs2m_shfl(incl, 63);
            const unsigned start = incl - (small ? area : 0u);
            gs2m_wave_sync();
            stage->heads[lane] = 0u;
            stage->heads[lane + 64] = 0u;
            if (lane < 2) stage->heads[128 + lane] = 0u;
            gs2m_wave_sync();
            if (small) {
                if (tested) {
                    stage->mx[k] = pv[v].mx;
                    stage->my[k] = pv[v].my;
                    stage->ca[k] = pv[v].ca;
                    stage->cb[k] = pv[v].cb;
                    stage->cc[k] = pv[v].cc;
                    stage->thr[k] = thr;
                    cull_slopes(pv[v].ca, pv[v].cb, pv[v].cc, stage->rx[k], stage->ry[k]);
                }
                stage->swh[k] = start | (w << 16) | (h << 23) | (tested ? 0x80000000u : 0u);   // w, h <= 64: 7 bits each
                stage->xy0[k] = xy0;
                stage->mlo[k] = 0u;
                stage->mhi[k] = 0u;
                atomicOr(&stage->heads[start >> 5], 1u << (start & 31u));
            }
            gs2m_wave_sync();
            int kbase = 0;
            for (unsigned b0 = 0; b0 < total; b0 += 64u) {
                const unsigned long long H =
                    (unsigned long long)stage->heads[b0 >> 5] | ((unsigned long long)stage->heads[(b0 >> 5) + 1] << 32);
                const int kk = kbase + gs2m_popc64(H & lanes_le(lane)) - 1;
                kbase += gs2m_popc64(H);
                const unsigned item = b0 + (unsigned)lane;
                const bool act = item < total;
                bool keep = false;
                unsigned li = 0u;
                if (act) {
                    const unsigned swh = stage->swh[kk];
                    const unsigned ow = (swh >> 16) & 0x7fu;
                    li = item - (swh & 0xffffu);
                    unsigned rx, ry;
                    rect_coords(li, ow, gs2m_fast_rcp((float)ow), rx, ry);
                    const unsigned oxy = stage->xy0[kk];
                    const int tx = (int)(oxy & 0xffffu) + (int)rx, ty = (int)(oxy >> 16) + (int)ry;
                    keep = true;
                    if (swh & 0x80000000u)  // only tested rects (corners to cut)
                        keep = tile_may_contribute(stage->mx[kk], stage->my[kk], stage->ca[kk], stage->cb[kk],
                                                   stage->cc[kk], stage->rx[kk], stage->ry[kk], stage->thr[kk], tx, ty, th);
                    if (keep) hist_bump(hh, ty * gx + tx);
                }
                // the first lane of each owner's run in this batch folds the run's keep bits into the owner's mask
                const unsigned long long kept = gs2m_ballot(keep ? 1 : 0);
                if (act && (lane == 0 || ((H >> lane) & 1ull))) {
                    const unsigned long long rest = lane == 63 ? 0ull : (H >> (lane + 1));
                    const int len = rest ? __ffsll(rest) : 64 - lane;
                    const unsigned long long run = (kept >> lane) & (len >= 64 ? ~0ull : ((1ull << len) - 1ull));
                    const unsigned long long bits = run << li;
                    stage->mlo[kk] |= (unsigned)bits;
                    stage->mhi[kk] |= (unsigned)(bits >> 32);
                }
            }
            gs2m_wave_sync();
            if (small && tested)   // an untested rect keeps every tile: the scatter does not read a mask for it
                tilemask[(size_t)v * P + gi] = (unsigned long long)stage->mlo[k] | ((unsigned long long)stage->mhi[k] << 32);
        }
        // ---- rects of more than 64 tiles: the whole wave walks one owner at a time ----
        unsigned long long bigs = gs2m_ballot(area > 64u ? 1 : 0);
        while (bigs != 0ull) {
            const int o = __ffsll(bigs) - 1;
            bigs &= bigs - 1ull;
            const unsigned ow = gs2m_shfl(w, o), oa = gs2m_shfl(area, o), oxy = gs2m_shfl(xy0, o);
            const float omx = gs2m_shfl(pv[v].mx, o), omy = gs2m_shfl(pv[v].my, o), oca = gs2m_shfl(pv[v].ca, o),
                        ocb = gs2m_shfl(pv[v].cb, o), occ = gs2m_shfl(pv[v].cc, o), othr = gs2m_shfl(thr, o);
            const float oinv = gs2m_fast_rcp((float)ow);
            float orx, ory;
            cull_slopes(oca, ocb, occ, orx, ory);
            for (unsigned li = (unsigned)lane; li < oa; li += 64u) {
                unsigned rx, ry;
                rect_coords(li, ow, oinv, rx, ry);
                const int tx = (int)(oxy & 0xffffu) + (int)rx, ty = (int)(oxy >> 16) + (int)ry;
                if (!cull || tile_may_contribute(omx, omy, oca, ocb, occ, orx, ory, othr, tx, ty, th)) hist_bump(hh, ty * gx + tx);
            }
        }
    }
}

// Tile counting: same Gaussian -> workgroup assignment as k_scatter.  Re-reads the binning part of the GeomRecs written by
// k_project (and the geometry part of the rects it tests), expands every rect into (Gaussian, tile) pairs (count_expand),
// bumps the workgroup-private LDS tile histogram, records the kept tiles of tested rects as bit masks and writes the
// workgroup's histogram row.
template <int NV>
GS2M_KERNEL void __launch_bounds__(1024)
k_count_tiles(GeomRecs recs, int P, const CamUniform* __restrict__ cams, int chunk, int n_wg,
              unsigned* __restrict__ hist, unsigned long long* __restrict__ tilemask, int exact_cull, int interleave, int lane_tiles) {
    GS2M_DYN_LDS(unsigned, lds);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = gs2m_uniform(tid >> 6);
    const int gx = cams[0].gx, th = cams[0].th, rs = GS2M_CAM_ROWS(cams[0]) >> 1;   // rows = 1 / 2 -> shift 0 / 1
    const int tiles = gx * GS2M_CAM_GYS(cams[0]);
    const unsigned lane_max = (unsigned)gs2m_uniform(lane_tiles);
    const int cull = gs2m_uniform(exact_cull);
    // blockIdx.y = group of NV views of the launch (GS2M_OPT_PAIR_BATCH): its own records, masks and histogram rows
    cams += NV * blockIdx.y;
    recs = gs2m_recs_at(recs, (size_t)NV * blockIdx.y * P);
    tilemask += (size_t)NV * blockIdx.y * P;
    hist += (size_t)NV * blockIdx.y * n_wg * tiles;
    // workgroup-private tile histogram: one u32 counter per (view, tile)
    const int nthreads = (int)blockDim.x;
    unsigned* lhist = lds;
    WaveStage* stage = reinterpret_cast<WaveStage*>(lds + ((NV * tiles + 3) & ~3)) + wave;
    for (int i = tid; i < NV * tiles; i += nthreads) lhist[i] = 0u;
    __syncthreads();
    // histogram row of this workgroup: the workgroups of an XCD own consecutive rows (see k_scatter)
    const int row = (int)gs2m_xcd_contiguous(blockIdx.x, (unsigned)n_wg);
    // Two-stage prefetch (round 6).  The 16-B binning part of the records (depth + rect) is loaded TWO wave steps ahead; the 32-B
    // geometry part one step ahead and ONLY for the lanes whose rect is tested -- nothing else of the counting step reads it
    // (round 5 streamed 48 B per (view, Gaussian) through this kernel, 32 of them unused for the thin rects that make up
    // 85 % of a small-splat scene).
    const int nw = nthreads >> 6;
    auto fetch_c = [&](int f0, int e0, float4* c) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            c[v] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (f0 >= 0 && f0 + lane < e0) c[v] = recs.c[(size_t)v * P + f0 + lane];
        }
    };
    auto fetch_ab = [&](int f0, int e0, const float4* c, float4* a0, float4* a1) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            a0[v] = a1[v] = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (cull && f0 >= 0 && f0 + lane < e0) {
                const BinRect q = bin_rect_of(c[v], rs);
                if (gs2m_rect_tested(q.w, q.h, q.area, cull)) {
                    const size_t ri = (size_t)v * P + f0 + lane;
                    a0[v] = recs.ab[2 * ri];
                    a1[v] = recs.ab[2 * ri + 1];
                }
            }
        }
    };
    float4 c_cur[NV], a0_cur[NV], a1_cur[NV], c_nxt[NV], a0_nxt[NV], a1_nxt[NV], c_nn[NV];
    int end = 0, end_n = 0, end_nn = 0;
    int first = bin_step_begin(0, wave, nw, row, n_wg, chunk, P, interleave, &end);
    int first_n = bin_step_begin(1, wave, nw, row, n_wg, chunk, P, interleave, &end_n);
    fetch_c(first, end, c_cur);
    fetch_c(first_n, end_n, c_nxt);
    fetch_ab(first, end, c_cur, a0_cur, a1_cur);
    for (int it = 0;; ++it) {
        if (first < 0) break;   // wave-uniform; the loop body only uses wave collectives
        const int gi = first + lane;
        const bool valid = gi < end;
        const int first_nn = bin_step_begin(it + 2, wave, nw, row, n_wg, chunk, P, interleave, &end_nn);
        fetch_c(first_nn, end_nn, c_nn);                        // step it + 2: rects
        fetch_ab(first_n, end_n, c_nxt, a0_nxt, a1_nxt);        // step it + 1: geometry where its rect (loaded a step ago) asks for it
        CountIn pv[NV];
        float op = 0.0f;
        bool any_tested = false;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const BinRect q = bin_rect_of(c_cur[v], rs);
            pv[v].ok = valid && q.area != 0u;
            pv[v].x0 = pv[v].ok ? q.x0 : 0;
            pv[v].y0 = pv[v].ok ? q.y0 : 0;
            pv[v].x1 = pv[v].ok ? q.x0 + (int)q.w : 0;
            pv[v].y1 = pv[v].ok ? q.y0 + (int)q.h : 0;
            pv[v].mx = pv[v].my = pv[v].ca = pv[v].cb = pv[v].cc = 0.0f;
            if (pv[v].ok && gs2m_rect_tested(q.w, q.h, q.area, cull)) {
                pv[v].mx = a0_cur[v].x;
                pv[v].my = a0_cur[v].y;
                pv[v].ca = a0_cur[v].z;
                pv[v].cb = a0_cur[v].w;
                pv[v].cc = a1_cur[v].x;
                op = a1_cur[v].y;          // one opacity per Gaussian, whichever view delivered it
                any_tested = true;
            }
        }
        const float thr = any_tested ? cull_threshold(op) : 0.0f;   // one logarithm per tested Gaussian (it was one per view and record)
        count_expand<NV>(pv, thr, valid, gi, P, lhist, tiles, stage, tilemask, gx, th, cull, lane, lane_max);
        first = first_n;
        end = end_n;
        first_n = first_nn;
        end_n = end_nn;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            c_cur[v] = c_nxt[v];
            a0_cur[v] = a0_nxt[v];
            a1_cur[v] = a1_nxt[v];
            c_nxt[v] = c_nn[v];
        }
    }
    __syncthreads();
    for (int v = 0; v < NV; ++v)
        for (int t = tid; t < tiles; t += nthreads) hist[((size_t)v * n_wg + row) * tiles + t] = lhist[v * tiles + t];
}

struct ScatterStage {
    unsigned swh[64], xy0[64], dbits[64], gid[64], mlo[64], mhi[64];  // indexed by owner rank k
    unsigned heads[130];
    unsigned pad[2];
};
#define GS2M_SCATTER_STAGE_BYTES_PER_WAVE ((int)sizeof(ScatterStage))

// Instance scatter: same Gaussian -> workgroup assignment as k_count_tiles; cursors start at
// tile_start[v][t] + (exclusive prefix over workgroups, left in `hist` by k_hist_colscan).
// Key = depth_bits << 32 | gaussian_id (unique => order after the per-tile sort is deterministic
// although LDS-atomic arrival order is not).  Rects of <= lane_max tiles are emitted by their own lane, the other rects of
// <= 64 tiles through the balanced walk of k_count_tiles; tested rects replay the tile mask written there, rects of more than
// 64 tiles repeat the same per-tile test.
template <int NV>
GS2M_KERNEL void __launch_bounds__(1024)
k_scatter(GeomRecs recs, int P, const CamUniform* __restrict__ cams, int chunk, int n_wg,
          const unsigned* __restrict__ hist, const unsigned* __restrict__ tile_start,
          const unsigned long long* __restrict__ tilemask, unsigned long long* __restrict__ keys, unsigned cap,
          int exact_cull, const int* __restrict__ ids, int interleave, int lane_tiles) {
    GS2M_DYN_LDS(unsigned, cursor);
    const int tid = (int)threadIdx.x;
    const int nthreads = (int)blockDim.x;
    const int gx = cams[0].gx, th = cams[0].th, rs = GS2M_CAM_ROWS(cams[0]) >> 1;
    const int tiles = gx * GS2M_CAM_GYS(cams[0]);
    const unsigned lane_max = (unsigned)gs2m_uniform(lane_tiles);
    const int cull = gs2m_uniform(exact_cull);
    // blockIdx.y = group of NV views of the launch (GS2M_OPT_PAIR_BATCH)
    cams += NV * blockIdx.y;
    recs = gs2m_recs_at(recs, (size_t)NV * blockIdx.y * P);
    tilemask += (size_t)NV * blockIdx.y * P;
    hist += (size_t)NV * blockIdx.y * n_wg * tiles;
    tile_start += (size_t)NV * blockIdx.y * (tiles + 1);
    keys += (size_t)NV * blockIdx.y * cap;
    // Chunk (= histogram row = position of this workgroup's segment inside every tile's key range).  The workgroups of
    // one XCD own consecutive rows, so the segments an XCD writes into a tile are adjacent: its 8-B key stores fill
    // whole lines in ITS L2 instead of leaving 1/8-written lines in eight L2s (4x write amplification, PMC WRITE_SIZE).
    const int row = (int)gs2m_xcd_contiguous(blockIdx.x, (unsigned)n_wg);
    for (int v = 0; v < NV; ++v)
        for (int t = tid; t < tiles; t += nthreads)
            cursor[v * tiles + t] = tile_start[(size_t)v * (tiles + 1) + t] + hist[((size_t)v * n_wg + row) * tiles + t];
    __syncthreads();
    const int lane = tid & 63, wave = gs2m_uniform(tid >> 6);
    const int nw = nthreads >> 6;
    ScatterStage* stage = reinterpret_cast<ScatterStage*>(cursor + ((NV * tiles + 3) & ~3)) + wave;
    // one view after the other: the key lines this XCD is filling at any time belong to ONE view's array (half the L2
    // working set of a walk that alternates between the views)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        // next-step prefetch (rect + depth vector, id, tile mask) as in k_count_tiles.  (Round 6 measured the counting kernel's
        // two-stage form here -- the mask one step ahead and only for the lanes that replay one: C3 39.8 -> 48 us per pair; this
        // kernel is issue-bound and the second rect decode + the longer rotation cost more than the 8 of 28 bytes saved.)
        float4 n_w2;
        unsigned n_kid;
        unsigned long long n_msk;
        int end = 0;
        int first = bin_step_begin(0, wave, nw, row, n_wg, chunk, P, interleave, &end);
        auto fetch = [&](int f0, int e0) __attribute__((always_inline)) {
            n_w2 = float4{0.0f, 0.0f, 0.0f, 0.0f};
            n_kid = (unsigned)(f0 + lane);
            n_msk = ~0ull;                                       // an untested rect keeps every tile
            if (f0 >= 0 && f0 + lane < e0) {
                if (ids) n_kid = (unsigned)ids[f0 + lane];
                n_w2 = recs.c[(size_t)v * P + f0 + lane];   // the 16-B binning part only: contiguous, fully used lines
                if (cull) n_msk = tilemask[(size_t)v * P + f0 + lane];   // only meaningful (and only used) for tested rects of <= 64 tiles
            }
        };
        fetch(first, end);
        for (int it = 0;; ++it) {
            if (first < 0) break;   // wave-uniform
            const int gi = first + lane;
            const bool valid = gi < end;
            const float4 w2 = n_w2;
            const unsigned kid = n_kid;   // low word of the sort key: the Gaussian's id (ties in depth resolve as in the reference)
            unsigned long long msk_pre = n_msk;
            {
                int end_next = 0;
                const int first_next = bin_step_begin(it + 1, wave, nw, row, n_wg, chunk, P, interleave, &end_next);
                fetch(first_next, end_next);
                first = first_next;      // rotated here: the wave-uniform `continue` below skips nothing of the pipeline
                end = end_next;
            }
            const BinRect rc = bin_rect_of(w2, rs);
            const unsigned dbits = __float_as_uint(w2.y);
            const int x0 = rc.x0, y0 = rc.y0;
            const unsigned rect0 = (unsigned)x0 | ((unsigned)y0 << 16);
            const unsigned w = rc.w, h = rc.h, area = valid ? rc.area : 0u;
            if (gs2m_ballot(area != 0u ? 1 : 0) == 0ull) continue;  // wave-uniform
            if (!gs2m_rect_tested(w, h, area, cull)) msk_pre = ~0ull;   // the word loaded for an untested rect is not a mask
            unsigned* cur = cursor + v * tiles;
            unsigned long long* kv = keys + (size_t)v * cap;
            // rects of <= lane_max tiles (and the thin ones): the lane emits its own keys, replaying its mask (all ones when the
            // rect was not tested), row-major bits
            const bool lane_rect = gs2m_lane_rect(w, h, area, lane_max);
            if (gs2m_ballot(lane_rect ? 1 : 0) != 0ull) {
                const unsigned area_l = lane_rect ? area : 0u;
                const unsigned long long key = ((unsigned long long)dbits << 32) | kid;
                int tile = y0 * gx + x0;
                unsigned rx = 0u;
                // four tiles per round: the (returning) LDS atomics of a round are issued back to back and waited for ONCE, then
                // the key stores (rounds 4-5 ran one atomic -> wait -> store chain per tile: four LDS round trips in sequence
                // for a 2 x 2 rect)
                for (unsigned t0 = 0; t0 < area_l; t0 += 4u) {
                    unsigned pos[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned t = t0 + (unsigned)u;
                        pos[u] = cap;
                        if (t < area_l && ((msk_pre >> t) & 1ull) != 0ull) pos[u] = atomicAdd(&cur[tile], 1u);
                        ++tile;
                        if (++rx == w) {
                            rx = 0u;
                            tile += gx - (int)w;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (pos[u] < cap) kv[pos[u]] = key;
                }
            }
            const bool small = area != 0u && area <= 64u && !lane_rect;
            const unsigned long long smalls = gs2m_ballot(small ? 1 : 0);
            if (smalls != 0ull) {
                const unsigned long long msk = small ? msk_pre : 0ull;
                const int k = gs2m_popc64(smalls & lanes_lt(lane));
                const unsigned incl = wave_inclusive_scan(small ? area : 0u);
                const unsigned total = gs2m_shfl(incl, 63);
                const unsigned start = incl - (small ? area : 0u);
                gs2m_wave_sync();
                stage->heads[lane] = 0u;
                stage->heads[lane + 64] = 0u;
                if (lane < 2) stage->heads[128 + lane] = 0u;
                gs2m_wave_sync();
                if (small) {
                    stage->swh[k] = start | (w << 16) | (h << 24);
                    stage->xy0[k] = rect0;
                    stage->dbits[k] = dbits;
                    stage->gid[k] = kid;
                    stage->mlo[k] = (unsigned)msk;
                    stage->mhi[k] = (unsigned)(msk >> 32);
                    atomicOr(&stage->heads[start >> 5], 1u << (start & 31u));
                }
                gs2m_wave_sync();
                int kbase = 0;
                // one 64-item batch: owner lookup, mask bit, cursor bump -> (position, key) of this lane's item (position `cap`: none)
                auto emit = [&](unsigned b0, unsigned& pos, unsigned long long& key) __attribute__((always_inline)) {
                    const unsigned long long H =
                        (unsigned long long)stage->heads[b0 >> 5] | ((unsigned long long)stage->heads[(b0 >> 5) + 1] << 32);
                    const int kk = kbase + gs2m_popc64(H & lanes_le(lane)) - 1;
                    kbase += gs2m_popc64(H);
                    const unsigned item = b0 + (unsigned)lane;
                    pos = cap;
                    key = 0ull;
                    if (item < total) {
                        const unsigned swh = stage->swh[kk];
                        const unsigned li = item - (swh & 0xffffu);
                        const unsigned long long om = (unsigned long long)stage->mlo[kk] | ((unsigned long long)stage->mhi[kk] << 32);
                        if ((om >> li) & 1ull) {
                            const unsigned ow = (swh >> 16) & 0xffu;
                            unsigned rx, ry;
                            rect_coords(li, ow, gs2m_fast_rcp((float)ow), rx, ry);
                            const unsigned oxy = stage->xy0[kk];
                            const int tx = (int)(oxy & 0xffffu) + (int)rx, ty = (int)(oxy >> 16) + (int)ry;
                            key = ((unsigned long long)stage->dbits[kk] << 32) | stage->gid[kk];
                            pos = atomicAdd(&cur[ty * gx + tx], 1u);
                        }
                    }
                };
                // two batches per round, the key store of a batch BEHIND the next batch's cursor bump: the returning LDS atomic of
                // batch A is waited for together with the stage reads of batch B (LDS operations return in order), not on its own
                // (rounds 1-5: reads -> wait -> atomic -> wait -> store, per batch)
                unsigned pos_a, pos_b = cap;
                unsigned long long key_a, key_b = 0ull;
                for (unsigned b0 = 0; b0 < total; b0 += 128u) {
                    emit(b0, pos_a, key_a);
                    if (pos_b < cap) kv[pos_b] = key_b;
                    pos_b = cap;
                    if (b0 + 64u < total) emit(b0 + 64u, pos_b, key_b);
                    if (pos_a < cap) kv[pos_a] = key_a;
                }
                if (pos_b < cap) kv[pos_b] = key_b;
            }
            // rects of more than 64 tiles (rare): the whole wave walks one owner at a time, repeating the test
            unsigned long long bigs = gs2m_ballot(area > 64u ? 1 : 0);
            while (bigs != 0ull) {
                const int o = __ffsll(bigs) - 1;
                bigs &= bigs - 1ull;
                const unsigned ow = gs2m_shfl(w, o), oa = gs2m_shfl(area, o), oxy = gs2m_shfl(rect0, o);
                const unsigned long long key = ((unsigned long long)gs2m_shfl(dbits, o) << 32) | gs2m_shfl(kid, o);
                float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, thr = 0.f;
                if (cull) {
                    const float4* r4 = recs.ab + 2 * ((size_t)v * P + gs2m_shfl(gi, o));
                    const float4 w0 = r4[0];
                    const float4 w1 = r4[1];
                    mx = w0.x;
                    my = w0.y;
                    ca = w0.z;
                    cb = w0.w;
                    cc = w1.x;
                    thr = cull_threshold(w1.y);
                }
                float srx, sry;
                cull_slopes(ca, cb, cc, srx, sry);
                const float oinv = gs2m_fast_rcp((float)ow);
                for (unsigned li = (unsigned)lane; li < oa; li += 64u) {
                    unsigned rx, ry;
                    rect_coords(li, ow, oinv, rx, ry);
                    const int tx = (int)(oxy & 0xffffu) + (int)rx, ty = (int)(oxy >> 16) + (int)ry;
                    if (!cull || tile_may_contribute(mx, my, ca, cb, cc, srx, sry, thr, tx, ty, th)) {
                        const unsigned pos = atomicAdd(&cur[ty * gx + tx], 1u);
                        if (pos < cap) kv[pos] = key;
                    }
                }
            }
        }
    }
}

// One-time re-layout of the SH block for the pipeline-level path (Renderer.prepare_renderer): the
// reference layout is one 192-B row per Gaussian ([P,16,3], or dc [P,1,3] + rest [P,15,3]), which a
// thread-per-Gaussian kernel reads as 12 float4 loads strided by 192 B (64 cache lines per wave
// instruction, L1 thrash).  Packed: [P/64][12][64] float4 -- every load instruction of a wave is one
// contiguous KiB.
// `order` (null = identity): position gi of the packed copy holds Gaussian order[gi] (gs2m_raster_pack_model).
GS2M_KERNEL void __launch_bounds__(256)
k_pack_sh(int P, const float* __restrict__ shs, const float* __restrict__ shs_rest, float* __restrict__ packed,
          const int* __restrict__ order) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // one float of the [P,48] matrix
    if (i >= (size_t)P * 48) return;
    const size_t gi = i / 48;
    const int c = (int)(i - gi * 48);
    const size_t src = order ? (size_t)order[gi] : gi;
    float v;
    if (shs_rest == nullptr) v = shs[src * 48 + c];
    else v = c < 3 ? shs[src * 3 + c] : shs_rest[src * 45 + (c - 3)];
    const int k = c >> 2, e = c & 3;
    packed[(((gi >> 6) * 12 + k) * 64 + (gi & 63)) * 4 + e] = v;
}

// Spatially ordered packed copy of the per-Gaussian parameters (gs2m_raster_pack_model): position j <- Gaussian
// order[j]; rank[order[j]] = j is the inverse the compositing stage uses to find a record from the id in a sort key.
GS2M_KERNEL void __launch_bounds__(256)
k_pack_model(int P, const int* __restrict__ order, const float* __restrict__ xyz, const float* __restrict__ scales,
             const float* __restrict__ rots, const float* __restrict__ opac, float* __restrict__ p_xyz,
             float* __restrict__ p_scales, float* __restrict__ p_rots, float* __restrict__ p_opac, int* __restrict__ rank) {
    const int j = (int)(blockIdx.x * 256u + threadIdx.x);
    if (j >= P) return;
    const int id = order[j];
    if (id < 0 || id >= P) return;   // not a permutation: rank keeps a hole, k_check_rank reports it
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p_xyz[3 * (size_t)j + c] = xyz[3 * (size_t)id + c];
        p_scales[3 * (size_t)j + c] = scales[3 * (size_t)id + c];
    }
    *reinterpret_cast<float4*>(p_rots + 4 * (size_t)j) = *reinterpret_cast<const float4*>(rots + 4 * (size_t)id);
    p_opac[j] = opac[id];
    rank[id] = j;
}

// order is a permutation of 0..P-1 <=> every rank entry was written exactly once (holes stay -1)
GS2M_KERNEL void __launch_bounds__(256)
k_check_rank(int P, const int* __restrict__ rank, const int* __restrict__ order, unsigned* __restrict__ bad) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= P) return;
    const int j = rank[i];
    if (j < 0 || j >= P || order[j] != i) atomicOr(bad, 1u);
}

// checkFrustum (rasterizer_impl.cu:54-66): present = z_view > 0.2
GS2M_KERNEL void __launch_bounds__(256)
k_mark_visible(int P, const float* __restrict__ xyz, const float* __restrict__ viewmatrix,
               unsigned char* __restrict__ present) {
    const int gi = (int)(blockIdx.x * 256u + threadIdx.x);
    if (gi < P) {
        float tx, ty, tz;
        xform4x3(viewmatrix, xyz[3 * (size_t)gi], xyz[3 * (size_t)gi + 1], xyz[3 * (size_t)gi + 2], tx, ty, tz);
        present[gi] = tz <= 0.2f ? 0 : 1;
    }
}

// Fill the device CamUniform of view `slot` from DEVICE pointers (operator-level API, where
// viewmatrix / projmatrix / campos / bg are device tensors as in the reference).
GS2M_KERNEL void k_pack_camera(CamUniform* cams, int slot, const float* viewmatrix, const float* projmatrix,
                               const float* campos, const float* bg, float tanfovx, float tanfovy, int W, int H,
                               int th) {
    const int t = (int)threadIdx.x;
    CamUniform* c = cams + slot;
    if (t < 16) {
        c->view[t] = viewmatrix[t];
        c->proj[t] = projmatrix[t];
    }
    if (t < 3) {
        c->campos[t] = campos[t];
        c->bg[t] = bg[t];
    }
    if (t == 0) {
        c->tanfovx = tanfovx;
        c->tanfovy = tanfovy;
        c->focal_y = H / (2.0f * tanfovy);  // rasterizer_impl.cu:222-223
        c->focal_x = W / (2.0f * tanfovx);
        c->W = W;
        c->H = H;
        c->gx = (W + GS2M_TILE - 1) / GS2M_TILE;
        c->gy = (H + GS2M_TILE - 1) / GS2M_TILE;
        c->th = th;
    }
}
