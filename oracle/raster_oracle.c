/*
 * raster_oracle.c -- CPU restatement of the reference rasteriser FORWARD pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (package gs2mesh_amd, the
 * C-ABI library) may import, link or execute this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * What it restates (file:line in /root/reference, DGR/ =
 * third_party/gaussian-splatting/submodules/diff-gaussian-rasterization/):
 *   DGR/cuda_rasterizer/auxiliary.h:22-56    SH constants, ndc2Pix (double!), getRect
 *   DGR/cuda_rasterizer/auxiliary.h:58-77    transformPoint4x3 / 4x4 (flat m[4c+r])
 *   DGR/cuda_rasterizer/auxiliary.h:139-164  in_frustum (near cull z <= 0.2)
 *   DGR/cuda_rasterizer/forward.cu:20-71     computeColorFromSH
 *   DGR/cuda_rasterizer/forward.cu:74-113    computeCov2D (GLM column-major unfolded)
 *   DGR/cuda_rasterizer/forward.cu:118-152   computeCov3D
 *   DGR/cuda_rasterizer/forward.cu:155-256   preprocessCUDA control flow / early outs
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:70-138   duplicateWithKeys, identifyTileRanges
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:300-308  stable radix sort on (tile|depth) keys
 *   DGR/cuda_rasterizer/forward.cu:261-374   renderCUDA per-pixel compositing
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:198-336  Rasterizer::forward sequencing
 *
 * Pinning status: the reference ships no golden vectors and its CUDA kernels cannot be
 * built here (no nvcc).  Sub-steps with an independent in-tree Python implementation are
 * pinned by tests/golden fixtures generated from the reference (SH -> RGB via
 * GS/utils/sh_utils.py:eval_sh, Sigma via build_covariance_from_scaling_rotation,
 * camera matrices via GS/utils/graphics_utils.py).  The EWA projection, binning and
 * compositing stages have no second implementation in the reference:
 * PARITY UNPINNED for those (analytic known-answer tests only).
 *
 * Arithmetic: every expression is written in the reference's operand order, fp32 unless
 * the reference promotes to double (ndc2Pix).  Build with -ffp-contract=off so that no
 * FMA contraction happens (nvcc would contract some; that freedom is why RGB parity is a
 * tolerance, not bitwise).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:16-17 */
#define BLOCK_Y 16

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* auxiliary.h:41-44 -- literals 1.0 / 0.5 are doubles, so the whole expression is double */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:46-56 -- (int) casts truncate toward zero; max_radius is int */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin,
                    uint32_t* rmax) {
    rmin[0] = (uint32_t)imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:58-66 */
static void transformPoint4x3(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
/* auxiliary.h:68-77 */
static void transformPoint4x4(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* forward.cu:20-71.  glm::vec3 arithmetic is component-wise, left to right. */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means,
                               const float* campos, const float* shs, float* out_rgb) {
    float pos[3] = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    /* glm::length = sqrt(dot(v,v)), dot = x*x + y*y + z*z (left to right) */
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len;
    dir[1] = dir[1] / len;
    dir[2] = dir[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    for (int c = 0; c < 3; ++c) {
#define SH(k) sh[3 * (k) + c]
        float result = SH_C0 * SH(0);
        if (deg > 0) {
            float x = dir[0], y = dir[1], z = dir[2];
            result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                         SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                             SH_C3[1] * xy * z * SH(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                             SH_C3[5] * z * (xx - yy) * SH(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        result += 0.5f;
        out_rgb[c] = result < 0.0f ? 0.0f : result; /* glm::max(result, 0.0f) */
    }
}

/* forward.cu:118-152.  GLM is column-major: glm::mat3(a,b,c, d,e,f, g,h,i) has COLUMNS
 * (a,b,c),(d,e,f),(g,h,i).  So the `R` written there is the transpose of the standard
 * rotation matrix Rq; M = S*R (GLM) = (Rq*S)^T in conventional terms; Sigma = M^T*M (GLM)
 * = Rq S S Rq^T.  Element-wise: Sigma[i][j] = sum_k (Rq[i][k] s_k)(Rq[j][k] s_k), summed
 * k = 0,1,2 left to right (glm mat3*mat3 evaluates a0*b0 + a1*b1 + a2*b2). */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    /* Rq[i][k], standard wxyz rotation matrix (rows i) */
    float Rq[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    /* GLM: M = S * R with R_glm[col][row]; M_glm[col c][row r] = S[r][r] * R_glm[c][r]
     *      = s_r * Rq[c][r]  (S diagonal: the other products are exact zeros added).
     * Sigma_glm = transpose(M) * M; Sigma_glm[col j][row i] = sum_k Mt[k][i]*M[j][k]
     *      where Mt_glm[col k][row i] = M_glm[col i][row k] = s_k*Rq[i][k],
     *      M_glm[col j][row k] = s_k*Rq[j][k].
     * glm's mat3*mat3 computes each entry as A[0][i]*B[j][0] + A[1][i]*B[j][1] + A[2][i]*B[j][2]. */
    float Mm[3][3]; /* Mm[i][k] = s_k * Rq[i][k] */
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) Mm[i][k] = s[k] * Rq[i][k];
#define SIG(i, j) (Mm[i][0] * Mm[j][0] + Mm[i][1] * Mm[j][1] + Mm[i][2] * Mm[j][2])
    cov3D[0] = SIG(0, 0);
    cov3D[1] = SIG(0, 1);
    cov3D[2] = SIG(0, 2);
    cov3D[3] = SIG(1, 1);
    cov3D[4] = SIG(1, 2);
    cov3D[5] = SIG(2, 2);
#undef SIG
}

/* forward.cu:74-113, GLM unfolded.  With conventional (row,col) matrices:
 *   J_glm columns = (fx/tz,0,-fx*tx/tz^2), (0,fy/tz,-fy*ty/tz^2), (0,0,0)
 *   W_glm columns = (v0,v4,v8), (v1,v5,v9), (v2,v6,v10)
 *   T = W*J (GLM): T_glm[col c][row r] = sum_k W_glm[k][r] * J_glm[c][k]
 *   cov = transpose(T) * transpose(Vrk) * T
 * We evaluate exactly that with glm's left-to-right 3-term sums. */
static void mat3_mul(const float A[3][3], const float B[3][3], float C[3][3]) {
    /* GLM storage: X[col][row].  (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) C[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2];
}
static void mat3_T(const float A[3][3], float B[3][3]) {
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) B[c][r] = A[r][c];
}
static void computeCov2D(const float* mean, float focal_x, float focal_y, float tan_fovx,
                         float tan_fovy, const float* cov3D, const float* vm, float* cov) {
    float t[3];
    transformPoint4x3(mean, vm, t);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    float J[3][3] = {{focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2])},
                     {0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])},
                     {0.0f, 0.0f, 0.0f}};
    float W[3][3] = {{vm[0], vm[4], vm[8]}, {vm[1], vm[5], vm[9]}, {vm[2], vm[6], vm[10]}};
    float T[3][3], Tt[3][3], Vrk[3][3], Vt[3][3], tmp[3][3], c2[3][3];
    mat3_mul(W, J, T);
    float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]},
                     {cov3D[1], cov3D[3], cov3D[4]},
                     {cov3D[2], cov3D[4], cov3D[5]}};
    memcpy(Vrk, V, sizeof(V));
    mat3_T(T, Tt);
    mat3_T(Vrk, Vt);
    mat3_mul(Tt, Vt, tmp);
    mat3_mul(tmp, T, c2);
    c2[0][0] += 0.3f;
    c2[1][1] += 0.3f;
    cov[0] = c2[0][0];
    cov[1] = c2[0][1];
    cov[2] = c2[1][1];
}

/*
 * preprocessCUDA (forward.cu:155-256) for all P Gaussians of one view.
 * Outputs (all caller-allocated, size P unless noted; zero-initialised here where the
 * reference initialises): radii[P], means2D[P,2], depths[P], cov3D[P,6], rgb[P,3],
 * conic_opacity[P,4], tiles_touched[P], rect[P,4] (x0,y0,x1,y1; extra, for tests).
 * Entries of skipped Gaussians keep whatever the caller put there except radii and
 * tiles_touched (set to 0), as in the reference.
 */
void oracle_preprocess(int P, int D, int M, const float* orig_points, const float* scales,
                       float scale_modifier, const float* rotations, const float* opacities,
                       const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       int W, int H, float tan_fovx, float tan_fovy, int* radii, float* means2D,
                       float* depths, float* cov3Ds, float* rgb, float* conic_opacity,
                       uint32_t* tiles_touched, uint32_t* rect) {
    /* rasterizer_impl.cu:222-223 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    for (int idx = 0; idx < P; ++idx) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        if (rect) rect[4 * idx] = rect[4 * idx + 1] = rect[4 * idx + 2] = rect[4 * idx + 3] = 0;
        const float* p_orig = orig_points + 3 * idx;
        /* in_frustum, auxiliary.h:139-164 */
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue;
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float* cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + 6 * (size_t)idx;
        } else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
                         cov3Ds + 6 * (size_t)idx);
            cov3D = cov3Ds + 6 * (size_t)idx;
        }
        float cov[3];
        computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
        float det = (cov[0] * cov[2] - cov[1] * cov[1]);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        getRect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            computeColorFromSH(idx, D, M, orig_points, cam_pos, shs, rgb + 3 * (size_t)idx);
        }
        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        means2D[2 * idx] = px;
        means2D[2 * idx + 1] = py;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
        if (rect) {
            rect[4 * idx + 0] = rmin[0];
            rect[4 * idx + 1] = rmin[1];
            rect[4 * idx + 2] = rmax[0];
            rect[4 * idx + 3] = rmax[1];
        }
    }
}

/* checkFrustum (rasterizer_impl.cu:54-66) */
void oracle_mark_visible(int P, const float* orig_points, const float* viewmatrix,
                         const float* projmatrix, uint8_t* present) {
    (void)projmatrix;
    for (int idx = 0; idx < P; ++idx) {
        float p_view[3];
        transformPoint4x3(orig_points + 3 * idx, viewmatrix, p_view);
        present[idx] = p_view[2] <= 0.2f ? 0 : 1;
    }
}

/* ---- binning: duplicateWithKeys + stable sort + identifyTileRanges ---------------- */
typedef struct {
    uint64_t key;
    uint32_t val;
    uint32_t seq; /* emission order: makes qsort stable like cub::DeviceRadixSort */
} kv_t;
static int kv_cmp(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a;
    const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
    return 0;
}

/*
 * Exact (conservative-free) tile test used by the optional GS2M_OPT_EXACT_TILE_CULL mode of
 * the HIP path -- NOT part of the reference.  Kept here only so tests can predict the culled
 * instance list; returns 1 if the reference instance (idx,tile) is kept.  Same arithmetic
 * as the kernel (gs2mesh_amd/csrc/raster_math.h: tile_may_contribute).
 */
int oracle_tile_may_contribute(float mx, float my, float ca, float cb, float cc, float opacity,
                               int tx, int ty);

/*
 * Returns num_rendered; fills point_list[num_rendered] (caller sized via tiles_touched sum)
 * and ranges[2*tiles] (zeroed first, rasterizer_impl.cu:310).
 * exact_cull != 0 applies oracle_tile_may_contribute (extension, see above): 1 = to every rect with corners (>= 2 x 2 tiles),
 * 2 = only to those of more than 4 tiles.
 */
int64_t oracle_bin(int P, int W, int H, const int* radii, const float* means2D,
                   const float* depths, const float* conic_opacity, int exact_cull,
                   uint32_t* point_list, uint32_t* ranges, int64_t capacity) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    int64_t n = 0;
    for (int idx = 0; idx < P; ++idx) {
        if (radii[idx] > 0) {
            uint32_t rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            n += (int64_t)(rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
        }
    }
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    int64_t off = 0;
    for (int idx = 0; idx < P; ++idx) {
        if (radii[idx] > 0) {
            uint32_t rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            int per_tile = 0;
            if (exact_cull) {
                /* the HIP path's rule (raster_project.h): shrink the rect to the bbox of the
                 * alpha >= 1/255 ellipse, then test tiles individually only for >= 2x2 rects */
                const float ca = conic_opacity[4 * idx], cb = conic_opacity[4 * idx + 1];
                const float cc = conic_opacity[4 * idx + 2], op = conic_opacity[4 * idx + 3];
                if (!(op * 255.0f >= 1.0f)) continue;
                const float thr = logf(op * 255.0f) * 1.0001f + 0.001f;
                const float detc = ca * cc - cb * cb;
                const float cova = cc / detc, covc = ca / detc; /* cov = conic^-1 */
                const float hx = sqrtf(2.0f * thr * cova) + 0.01f, hy = sqrtf(2.0f * thr * covc) + 0.01f;
                const float mx = means2D[2 * idx], my = means2D[2 * idx + 1];
                const int bx0 = (int)ceilf((mx - hx - (float)(BLOCK_X - 1)) / BLOCK_X);
                const int bx1 = (int)floorf((mx + hx) / BLOCK_X) + 1;
                const int by0 = (int)ceilf((my - hy - (float)(BLOCK_Y - 1)) / BLOCK_Y);
                const int by1 = (int)floorf((my + hy) / BLOCK_Y) + 1;
                if ((int)rmin[0] < bx0) rmin[0] = (uint32_t)bx0;
                if ((int)rmin[1] < by0) rmin[1] = (uint32_t)by0;
                if ((int)rmax[0] > bx1) rmax[0] = (uint32_t)(bx1 < 0 ? 0 : bx1);
                if ((int)rmax[1] > by1) rmax[1] = (uint32_t)(by1 < 0 ? 0 : by1);
                if (rmax[0] <= rmin[0] || rmax[1] <= rmin[1]) continue;
                per_tile = (rmax[0] - rmin[0]) >= 2 && (rmax[1] - rmin[1]) >= 2;
                /* level 2 (round 6): rects of at most 4 tiles keep all their tiles (gs2m_rect_tested, raster_project.h) */
                if (exact_cull == 2 && (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) <= 4) per_tile = 0;
            }
            for (uint32_t y = rmin[1]; y < rmax[1]; y++) {
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    if (per_tile &&
                        !oracle_tile_may_contribute(means2D[2 * idx], means2D[2 * idx + 1],
                                                    conic_opacity[4 * idx], conic_opacity[4 * idx + 1],
                                                    conic_opacity[4 * idx + 2],
                                                    conic_opacity[4 * idx + 3], (int)x, (int)y))
                        continue;
                    uint64_t key = (uint64_t)y * gx + x;
                    key <<= 32;
                    uint32_t dbits;
                    memcpy(&dbits, &depths[idx], 4);
                    key |= dbits;
                    kv[off].key = key;
                    kv[off].val = (uint32_t)idx;
                    kv[off].seq = (uint32_t)off;
                    off++;
                }
            }
        }
    }
    n = off;
    qsort(kv, (size_t)n, sizeof(kv_t), kv_cmp);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (int64_t i = 0; i < n; ++i) {
        if (i < capacity) point_list[i] = kv[i].val;
        uint32_t currtile = (uint32_t)(kv[i].key >> 32);
        if (i == 0)
            ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(kv[i - 1].key >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)i;
                ranges[2 * currtile] = (uint32_t)i;
            }
        }
        if (i == n - 1) ranges[2 * currtile + 1] = (uint32_t)n;
    }
    free(kv);
    return n;
}

/* ---- renderCUDA (forward.cu:261-374), one pixel at a time -------------------------- */
void oracle_render(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                   const float* means2D, const float* features, const float* conic_opacity,
                   const float* bg_color, float* out_color, float* final_T,
                   uint32_t* n_contrib) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
    for (int py = 0; py < H; ++py) {
        for (int px = 0; px < W; ++px) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixf[2] = {(float)px, (float)py};
            float T = 1.0f;
            float C[3] = {0, 0, 0};
            uint32_t contributor = 0, last_contributor = 0;
            for (uint32_t k = r0; k < r1; ++k) {
                contributor++;
                const uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixf[0];
                float dy = means2D[2 * g + 1] - pixf[1];
                const float* co = conic_opacity + 4 * (size_t)g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break; /* done = true */
                for (int ch = 0; ch < 3; ++ch) C[ch] += features[3 * (size_t)g + ch] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            const size_t pix_id = (size_t)W * py + px;
            if (final_T) final_T[pix_id] = T;
            if (n_contrib) n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ++ch)
                out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
        }
    }
}

/*
 * Rasterizer::forward (rasterizer_impl.cu:198-336).  All pointers are host memory.
 * Optional taps (may be NULL): radii[P], and the geometry buffers.  Returns num_rendered.
 * P == 0 writes nothing (the caller zero-fills out_color: rasterize_points.cu:68,81).
 */
/* ---------------------------------------------------------------------------------------------
 * Flip attribution (test infrastructure of the parity tests; no counterpart in the reference).
 * renderCUDA (forward.cu:330-362) takes three per-(pixel, instance) DECISIONS on fp32 values:
 *     power > 0 -> skip,    alpha < 1/255 -> skip,    T(1 - alpha) < 1e-4 -> stop.
 * An implementation whose exp / products differ from libm's in the last ulps takes the other side of a
 * decision when the value sits within its rounding error of the threshold; the image then changes by the
 * size of the dropped / added contribution, not by a rounding error.  This pass replays the reference
 * loop and, per pixel, counts the decisions taken within `rel_eps` of their threshold and bounds the
 * change a flip of each can cause:
 *     alpha / power candidate:  2 * alpha * T * cmax   (its own contribution + the (1 - alpha) factor on
 *                                                        everything behind it, which sums to <= T * cmax)
 *     T candidate:              T * cmax               (everything behind the stop, background included)
 * cmax = largest colour value that can multiply a weight (features and background).  A parity test can
 * then assert the rounding-level bar on the pixels with bound == 0 and |delta| <= bound elsewhere.
 * Conditioning: `power` is a sum of three products that cancel along the ridge of a long thin splat (terms
 * ~1e5 for a result ~1 at 300 px from the centre of a background splat): ANY fp32 evaluation order, the
 * reference's included, is then only good to ~4 ulps of the largest term (pscale), i.e. alpha to a relative
 * kappa * pscale with kappa = 2^-22.  cond[pixel] sums 2 * alpha * T * cmax * kappa * pscale over the
 * evaluations where that exceeds rel_eps: the part of |delta| that is the reference's own rounding noise.
 * ------------------------------------------------------------------------------------------- */
void oracle_render_flip_bounds(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                               const float* means2D, const float* conic_opacity, float rel_eps, float cmax,
                               uint16_t* n_alpha, uint16_t* n_T, uint16_t* n_power, float* bound, float* cond) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; ++py) {
        for (int px = 0; px < W; ++px) {
            const int tile = (py / BLOCK_Y) * gx + (px / BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixf[2] = {(float)px, (float)py};
            float T = 1.0f;
            unsigned ca = 0, cT = 0, cp = 0;
            double b = 0.0, cnd = 0.0;
            for (uint32_t k = r0; k < r1; ++k) {
                const uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixf[0];
                float dy = means2D[2 * g + 1] - pixf[1];
                const float* co = conic_opacity + 4 * (size_t)g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                /* |power| within the rounding of its three products of the sign change */
                const float pscale = 0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy);
                /* relative band of alpha in THIS evaluation: rel_eps, or the reference's own rounding noise on an ill-conditioned
                 * sum (an absolute error kappa * pscale of power = that relative error of alpha) */
                const float rel_c = pscale * (1.0f / 4194304.0f);   /* kappa = 2^-22 */
                const float eps_i = rel_c > rel_eps ? (rel_c < 0.5f ? rel_c : 0.5f) : rel_eps;
                const int near_p = pscale > 0.0f && fabsf(power) <= rel_eps * pscale;   /* dx = dy = 0: exactly 0 for everyone */
                if (near_p) {
                    cp++;
                    b += 2.0 * (double)fminf(0.99f, co[3]) * T * cmax;
                }
                if (power > 0.0f) continue;
                float alpha = fminf(0.99f, co[3] * expf(power));
                if (fabsf(alpha - 1.0f / 255.0f) <= eps_i * (1.0f / 255.0f)) {
                    ca++;
                    b += 2.0 * (double)alpha * T * cmax;
                }
                if (alpha < 1.0f / 255.0f) continue;
                if (rel_c > rel_eps) cnd += 2.0 * (double)alpha * T * cmax * eps_i;
                float test_T = T * (1 - alpha);
                if (fabsf(test_T - 0.0001f) <= rel_eps * 0.0001f * 100.0f) { /* T carries the rounding of every (1 - alpha) before it */
                    cT++;
                    b += (double)T * cmax;
                }
                if (test_T < 0.0001f) break;
                T = test_T;
            }
            const size_t pix_id = (size_t)W * py + px;
            n_alpha[pix_id] = (uint16_t)(ca > 65535u ? 65535u : ca);
            n_T[pix_id] = (uint16_t)(cT > 65535u ? 65535u : cT);
            n_power[pix_id] = (uint16_t)(cp > 65535u ? 65535u : cp);
            bound[pix_id] = (float)b;
            if (cond) cond[pix_id] = (float)cnd;
        }
    }
}

int64_t oracle_rasterize_forward(int P, int D, int M, const float* background, int W, int H,
                                 const float* means3D, const float* shs,
                                 const float* colors_precomp, const float* opacities,
                                 const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp,
                                 const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, float tan_fovx, float tan_fovy,
                                 int exact_cull, float* out_color, int* radii_out) {
    if (P == 0) return 0;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    int* radii = (int*)calloc((size_t)P, sizeof(int));
    float* means2D = (float*)calloc((size_t)P * 2, sizeof(float));
    float* depths = (float*)calloc((size_t)P, sizeof(float));
    float* cov3Ds = (float*)calloc((size_t)P * 6, sizeof(float));
    float* rgb = (float*)calloc((size_t)P * 3, sizeof(float));
    float* conic_opacity = (float*)calloc((size_t)P * 4, sizeof(float));
    uint32_t* tiles_touched = (uint32_t*)calloc((size_t)P, sizeof(uint32_t));
    oracle_preprocess(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                      cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos, W, H,
                      tan_fovx, tan_fovy, radii, means2D, depths, cov3Ds, rgb, conic_opacity,
                      tiles_touched, NULL);
    int64_t n_ref = 0;
    for (int i = 0; i < P; ++i) n_ref += tiles_touched[i];
    uint32_t* point_list = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_ref > 0 ? n_ref : 1));
    uint32_t* ranges = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)gx * gy);
    int64_t n = oracle_bin(P, W, H, radii, means2D, depths, conic_opacity, exact_cull,
                           point_list, ranges, n_ref);
    const float* feat = colors_precomp ? colors_precomp : rgb;
    oracle_render(W, H, ranges, point_list, means2D, feat, conic_opacity, background, out_color,
                  NULL, NULL);
    if (radii_out) memcpy(radii_out, radii, sizeof(int) * (size_t)P);
    free(radii);
    free(means2D);
    free(depths);
    free(cov3Ds);
    free(rgb);
    free(conic_opacity);
    free(tiles_touched);
    free(point_list);
    free(ranges);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* Extension shared with the HIP path (not in the reference): exact tile test.          */
/* A (Gaussian, tile) instance can only contribute if somewhere on the tile's pixel     */
/* lattice alpha = min(0.99, o*exp(power)) >= 1/255, i.e. power >= ln(1/(255 o)).       */
/* We lower-bound q = -power over the CONTINUOUS tile rectangle [x0,x0+15]x[y0,y0+15]   */
/* (pixel centres are integers, forward.cu:282) and drop the instance only when         */
/* q_min > ln(255 o) + margin.  Conservative: a kept instance may still contribute 0.   */
/* ------------------------------------------------------------------------------------ */
static float q_form(float a, float b, float c, float dx, float dy) {
    return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
}
/* min over t in [t0,t1] of q along an axis-aligned segment */
static float edge_min_x(float a, float b, float c, float rx, float dy, float x0, float x1) {
    /* q(dx) = 0.5 a dx^2 + b dy dx + 0.5 c dy^2, minimiser dx* = (-b / a) dy; rx = -b / a is formed once per
       Gaussian (any point of the edge bounds the minimum from above, so its rounding only costs margin) */
    float dxs = a > 0.0f ? rx * dy : x0;
    dxs = fminf(x1, fmaxf(x0, dxs));
    return q_form(a, b, c, dxs, dy);
}
static float edge_min_y(float a, float b, float c, float ry, float dx, float y0, float y1) {
    float dys = c > 0.0f ? ry * dx : y0;
    dys = fminf(y1, fmaxf(y0, dys));
    return q_form(a, b, c, dx, dys);
}
int oracle_tile_may_contribute(float mx, float my, float ca, float cb, float cc, float opacity,
                               int tx, int ty) {
    /* d = mean - pixel; pixel in [tx*16, tx*16+15] */
    const float dx0 = mx - (float)(tx * BLOCK_X + BLOCK_X - 1); /* smallest d.x */
    const float dx1 = mx - (float)(tx * BLOCK_X);               /* largest d.x  */
    const float dy0 = my - (float)(ty * BLOCK_Y + BLOCK_Y - 1);
    const float dy1 = my - (float)(ty * BLOCK_Y);
    if (!(opacity * 255.0f >= 1.0f)) return 0; /* alpha <= opacity < 1/255 everywhere */
    const float thresh = logf(opacity * 255.0f) * 1.0001f + 0.001f;
    if (dx0 <= 0.0f && dx1 >= 0.0f && dy0 <= 0.0f && dy1 >= 0.0f) return 1; /* centre inside */
    const float rx = -cb / ca, ry = -cb / cc;
    /* only the edges that FACE the centre (q is convex with its minimum at the centre, outside the tile: the constrained
       minimum lies on the part of the boundary visible from the centre); same form as the kernel (raster_math.h, round 6) */
    const int in_x = dx0 <= 0.0f && dx1 >= 0.0f, in_y = dy0 <= 0.0f && dy1 >= 0.0f;
    const float ex = dx0 > 0.0f ? dx0 : dx1, ey = dy0 > 0.0f ? dy0 : dy1;
    const float qv = edge_min_y(ca, cb, cc, ry, ex, dy0, dy1);
    const float qh = edge_min_x(ca, cb, cc, rx, ey, dx0, dx1);
    const float qmin = in_x ? qh : (in_y ? qv : fminf(qv, qh));
    return qmin <= thresh ? 1 : 0;
}
