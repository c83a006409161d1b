// raster_common.h -- data layout of the rasteriser in HBM (see DESIGN.md "Data layout").
#pragma once
#include "platform.h"

#define GS2M_TILE 16        // DGR/cuda_rasterizer/config.h:16-17 (BLOCK_X = BLOCK_Y = 16)
#define GS2M_MAX_VIEWS 2    // views fused per workgroup (a stereo pair: the Gaussian's parameters and Sigma are read once for both)
#define GS2M_MAX_PAIRS 4    // such groups per launch (GS2M_OPT_PAIR_BATCH): blockIdx.y of the projection / counting / scatter kernels
#define GS2M_MAX_PASS_VIEWS (GS2M_MAX_VIEWS * GS2M_MAX_PAIRS)
#define GS2M_SORT_LDS 4096  // keys sorted per workgroup in LDS (32 KiB)
// chunk of the compositing schedule (k_tile_scan -> blend): GS2M_SCHED_CW x GS2M_SCHED_CH neighbouring lists
#ifndef GS2M_SCHED_CW
#define GS2M_SCHED_CW 4
#endif
#ifndef GS2M_SCHED_CH
#define GS2M_SCHED_CH 2
#endif
#define GS2M_SCHED_CHUNK (GS2M_SCHED_CW * GS2M_SCHED_CH)

// Per-view uniforms (the per-view fields of GaussianRasterizationSettings,
// DGR/diff_gaussian_rasterization/__init__.py:157-169, + derived focal / tile grid,
// rasterizer_impl.cu:222-223,234).  Lives in device memory, one per view of the batch.
struct CamUniform {
    float view[16];  // "viewmatrix": transposed row-major, m[4c+r] = M[r][c]
    float proj[16];  // "projmatrix"
    float campos[3];
    float tanfovx, tanfovy, focal_x, focal_y;
    int W, H, gx, gy;  // gx x gy = the reference's 16 x 16 tile grid (rect units of the GeomRec)
    float bg[3];
    int th;            // binning tile height in pixels: 16 (reference tiles) or 32 (two reference tiles stacked)
};
// binning grid rows for a camera: tiles of 16 x th pixels
#define GS2M_CAM_ROWS(cam) ((cam).th >> 4)
#define GS2M_CAM_GYS(cam) (((cam).gy + ((cam).th >> 4) - 1) / ((cam).th >> 4))

// Projected per-(view, Gaussian) record: 48 B = three 16-B vectors
//   {mx, my, ca, cb} {cc, op, r, g} | {b, depth, rect0, rect1}
// (reference: means2D 8 B + conic_opacity 16 B + rgb 12 B + depth 4 B in four arrays, rasterizer_impl.h:30-45; rgb gathered
// from global per contributing pixel, forward.cu:355).  Round 4: stored as TWO arrays -- the 32-B geometry / colour part `ab`
// and the 16-B binning part `c` (depth + tile rect, + the blue channel to fill the vector) -- because the scatter kernel needs
// the binning part only: out of the 48-B records its 16-B loads still pulled every line (C3: 245 MB fetched for 64 MB used,
// PMC round 3).  The compositing stage gathers both parts of an instance (three 16-B DMA loads, as before).
struct GeomRec {             // host-side / emulator view of one whole record (parity taps)
    float mx, my, ca, cb;    // mean2D, conic.x, conic.y
    float cc, op, r, g;      // conic.z, opacity, rgb.r, rgb.g
    float b, depth;          // rgb.b, view-space z
    unsigned rect0, rect1;   // tile rect: rect0 = x0 | y0<<16, rect1 = x1 | y1<<16 (x1==x0 => invisible)
};
struct GeomRecs {            // device view: both arrays hold [views * P] records, passed by value
    float4* ab;              // [i][2]
    float4* c;               // [i]
};
GS2M_DEVICE GeomRecs gs2m_recs_at(const GeomRecs& r, size_t first) { return GeomRecs{r.ab + 2 * first, r.c + first}; }

// Inputs of the projection stage (device pointers).
struct GaussIn {
    const float* xyz;             // [P,3]
    const float* scales;          // [P,3] (log-scale if raw)
    const float* rots;            // [P,4] wxyz (unnormalised if raw)
    const float* opac;            // [P]   (logit if raw)
    const float* shs;             // [P,M,3], or features_dc [P,1,3] when shs_rest != null
    const float* shs_rest;        // null or features_rest [P,M-1,3]
    const float* shs_packed;      // null or wave-transposed SH [ceil(P/64)][12][64] float4 (M = 16 only)
    const float* cov3D_precomp;   // null or [P,6]
    const float* colors_precomp;  // null or [P,3]
    const int* ids;               // null, or [P]: Gaussian id of array position j (a spatially ordered packed copy of the
                                  // model, gs2m_raster_pack_model): ids go into the sort keys (the reference's order among
                                  // equal depths) and index the radii output; everything else is indexed by position
    int P, D, M, raw;
    float scale_modifier;
};

// Workgroup -> work-item index such that the workgroups of one XCD (block b runs on XCD b % 8) own a CONTIGUOUS run of
// indices: a bijection of [0, nwg).  Used for the tile groups of the compositing kernel (neighbouring tiles share
// Gaussians -> same L2) and for the Gaussian chunks of the counting sort (the segments one XCD writes into a tile's key
// range are then adjacent, so the partial lines of its 8-B key stores merge in ITS L2 before they are written back).
GS2M_DEVICE unsigned gs2m_xcd_contiguous(unsigned bid, unsigned nwg) {
    const unsigned qq = nwg / 8u, rr = nwg % 8u, xcd = bid % 8u, idx = bid / 8u;
    return (xcd < rr ? xcd * (qq + 1u) : rr * (qq + 1u) + (xcd - rr) * qq) + idx;
}

// Per-view status words written by the tile scan.
struct ViewStatus {
    unsigned num_rendered;  // instances this view produced
    unsigned overflow;      // 1 if num_rendered > arena capacity (results invalid)
    unsigned n_class[3];    // lists of the view in the three size classes of the per-tile sort (> 512, > 4096, > 8192 instances)
    unsigned pad;
};
