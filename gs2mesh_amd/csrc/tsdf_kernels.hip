// tsdf_kernels.hip -- TSDF integration stage (compiled with -ffp-contract=off).
#include "tsdf_kernels.h"
#include "tsdf_extract.h"
#include "mesh_kernels.h"
#include "tsdf_internal.h"
#include <math.h>

void gs2m_launch_tsdf_touch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, const float* depth,
                            const unsigned char* mask) {
    const int n = f.nx * f.ny;
    GS2M_LAUNCH(k_tsdf_touch, dim3((n + 255) / 256), dim3(256), 0, st, V, f, depth, mask);
    GS2M_LAUNCH(k_tsdf_compact, dim3((V.hash_cap + 1023u) / 1024u), dim3(1024), 0, st, V, f.frame_id);
}
void gs2m_launch_tsdf_integrate(hipStream_t st, int n_wg, const TsdfVolume& V, const TsdfFrame& f,
                                const float* depth, const unsigned char* color, const unsigned char* mask) {
    GS2M_LAUNCH(k_tsdf_integrate, dim3(n_wg), dim3(256), 0, st, V, f, depth, color, mask);
}
void gs2m_launch_tsdf_touch_batch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, int n_frames,
                                  const TsdfBatchFrame* frames) {
    const int n = f.nx * f.ny;
    GS2M_LAUNCH(k_tsdf_touch_batch, dim3((n + 255) / 256, n_frames), dim3(256), 0, st, V, frames);
    GS2M_LAUNCH(k_tsdf_compact, dim3((V.hash_cap + 1023u) / 1024u), dim3(1024), 0, st, V, 0u);
}
void gs2m_launch_tsdf_integrate_batch(hipStream_t st, int n_cu, const TsdfVolume& V, const TsdfBatchFrame* frames) {
    GS2M_LAUNCH(k_tsdf_integrate_batch, dim3(n_cu * 6), dim3(256), 0, st, V, frames);
    GS2M_LAUNCH(k_tsdf_clear_fmask, dim3(16), dim3(256), 0, st, V);
}
void gs2m_launch_tsdf_clear_used(hipStream_t st, const TsdfVolume& V) {
    GS2M_LAUNCH(k_tsdf_clear_used, dim3(2048), dim3(256), 0, st, V);
}
void gs2m_launch_tsdf_pack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, int form, float* buf, long long* ibuf) {
    if (form == GS2M_XF_SUM_PACKED) GS2M_LAUNCH(k_tsdf_pack<GS2M_XF_SUM_PACKED>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf);
    else if (form == GS2M_XF_RAW_F32) GS2M_LAUNCH(k_tsdf_pack<GS2M_XF_RAW_F32>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf);
    else GS2M_LAUNCH(k_tsdf_pack<GS2M_XF_SUM_F32>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf);
}
void gs2m_launch_tsdf_unpack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, int form, const float* buf,
                             const long long* ibuf, int halo) {
    if (form == GS2M_XF_SUM_PACKED) GS2M_LAUNCH(k_tsdf_unpack<GS2M_XF_SUM_PACKED>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf, halo);
    else if (form == GS2M_XF_RAW_F32) GS2M_LAUNCH(k_tsdf_unpack<GS2M_XF_RAW_F32>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf, halo);
    else GS2M_LAUNCH(k_tsdf_unpack<GS2M_XF_SUM_F32>, dim3(n), dim3(256), 0, st, V, keys, buf, ibuf, halo);
}
void gs2m_launch_tsdf_clear_from(hipStream_t st, const TsdfVolume& V, unsigned first) {
    GS2M_LAUNCH(k_tsdf_clear_from, dim3(1024), dim3(256), 0, st, V, first);
}
void gs2m_launch_tsdf_clear_gap(hipStream_t st, const TsdfVolume& V, unsigned upto) {
    GS2M_LAUNCH(k_tsdf_clear_gap, dim3(256), dim3(256), 0, st, V, upto);
}
void gs2m_launch_tsdf_owned_keys(hipStream_t st, unsigned n, const TsdfVolume& V, int* keys) {
    GS2M_LAUNCH(k_tsdf_owned_keys, dim3((n + 255u) / 256u), dim3(256), 0, st, V, n, keys);
}

void gs2m_launch_tsdf_block_map(hipStream_t st, const TsdfVolume& V, const int* lo, const int* dim, unsigned char* cells, unsigned n_cells,
                                unsigned flags, unsigned win_hash, int rank, unsigned frames_local, unsigned frames_base) {
    const unsigned wg = (V.max_blocks + 255u) / 256u;
    GS2M_LAUNCH(k_tsdf_block_map, dim3(wg < 256u ? (wg ? wg : 1u) : 256u), dim3(256), 0, st, V, lo[0], lo[1], lo[2], dim[0], dim[1], dim[2], cells,
                n_cells, flags, win_hash, rank, frames_local, frames_base);
}
void gs2m_launch_tsdf_map_keys(hipStream_t st, const int* lo, const int* dim, unsigned char* cells, unsigned n_cells, int* keys, unsigned max_keys) {
    GS2M_LAUNCH(k_tsdf_map_keys, dim3(1), dim3(1024), 0, st, lo[0], lo[1], lo[2], dim[1], dim[2], cells, n_cells, keys, max_keys);
}

size_t gs2m_mc_tables_bytes() { return sizeof(McDevTables); }
bool gs2m_mc_tables_fill(void* host_buf) {
    McTables T;
    if (!mc_generate(&T)) return false;
    McDevTables* D = (McDevTables*)host_buf;
    memcpy(D->tri, T.tri, sizeof(T.tri));
    memcpy(D->ntri, T.ntri, sizeof(T.ntri));
    return true;
}
void gs2m_launch_mc_count(hipStream_t st, const TsdfVolume& V, const McDevTables* T, unsigned n_blocks,
                          unsigned* blk_tris, unsigned long long* n_total) {
    GS2M_LAUNCH(k_mc_count, dim3(n_blocks), dim3(256), 0, st, V, T, n_blocks, blk_tris);
    GS2M_LAUNCH(k_mc_scan, dim3(1), dim3(1024), 0, st, blk_tris, n_blocks, n_total);
}
void gs2m_launch_mc_emit(hipStream_t st, const TsdfVolume& V, const McDevTables* T, unsigned n_blocks,
                         const unsigned* blk_off, unsigned long long max_tris, double voxel_length, double unit_length,
                         double* vertices, double* colors, int* edge_index) {
    McGeom G;
    for (int i = 0; i < 8; ++i)
        for (int a = 0; a < 3; ++a) G.corner[i][a] = mc_corner[i][a];
    for (int e = 0; e < 12; ++e)
        for (int a = 0; a < 3; ++a) G.edge[e][a] = mc_edge[e][a];
    GS2M_LAUNCH(k_mc_emit, dim3(n_blocks), dim3(256), 0, st, V, T, G, n_blocks, blk_off, max_tris, voxel_length,
                unit_length, vertices, colors, edge_index);
}

// ---- mesh post-processing (mesh_kernels.h) ----------------------------------------------------------------------------------
static unsigned grid_for(unsigned n) {
    const unsigned g = (n + 255u) / 256u;
    return g < 1u ? 1u : (g > 4096u ? 4096u : g);
}
// exclusive scan of in[n] -> out[n]; scratch >= ceil(n / 4096) + 1 words; the total -> scratch[ceil(n / 4096)]
void gs2m_launch_scan_u32(hipStream_t st, const unsigned* in, unsigned n, unsigned* out, unsigned* scratch) {
    const unsigned m = (n + GS2M_SCAN_TILE - 1u) / GS2M_SCAN_TILE;
    GS2M_LAUNCH(k_scan_tile_sums, dim3(m), dim3(1024), 0, st, in, n, scratch);
    GS2M_LAUNCH(k_scan_sums, dim3(1), dim3(1024), 0, st, scratch, m, scratch + m);
    GS2M_LAUNCH(k_scan_apply, dim3(m), dim3(1024), 0, st, in, n, scratch, out);
}
// weld, first half: key minima, table, "first carrier" flags and their exclusive scan (total -> scratch[ceil(n / 4096)])
void gs2m_launch_mesh_weld_count(hipStream_t st, unsigned n, const int* edge_index, int* mins, unsigned long long* hkeys, unsigned* hfirst,
                                 unsigned cap, unsigned* cell_of, unsigned* flag, unsigned* pos, unsigned* scratch, unsigned* bad) {
    GS2M_LAUNCH(k_mesh_key_min, dim3(grid_for(n)), dim3(256), 0, st, edge_index, n, mins);
    GS2M_LAUNCH(k_mesh_weld_insert, dim3(grid_for(n)), dim3(256), 0, st, edge_index, n, mins, hkeys, hfirst, cap, cell_of, bad);
    GS2M_LAUNCH(k_mesh_weld_flag, dim3(grid_for(n)), dim3(256), 0, st, n, hfirst, cell_of, flag);
    gs2m_launch_scan_u32(st, flag, n, pos, scratch);
}
// second half, once the caller has sized the compact arrays
void gs2m_launch_mesh_weld_emit(hipStream_t st, unsigned n, const unsigned* hfirst, const unsigned* cell_of, const unsigned* pos, const double* verts,
                                const double* cols, const int* edge_index, double* out_v, double* out_c, int* out_e, int* out_tri) {
    GS2M_LAUNCH(k_mesh_weld_emit, dim3(grid_for(n)), dim3(256), 0, st, n, hfirst, cell_of, pos, verts, cols, edge_index, out_v, out_c, out_e,
                out_tri);
}
void gs2m_launch_mesh_cluster(hipStream_t st, const int* tri, unsigned n_tri, unsigned long long* hkeys, unsigned* hval, unsigned cap,
                              unsigned* parent, unsigned* root, unsigned* flag, unsigned* pos, unsigned* scratch, int* labels,
                              unsigned long long* cluster_n) {
    GS2M_LAUNCH(k_mesh_uf_init, dim3(grid_for(n_tri)), dim3(256), 0, st, n_tri, parent);
    GS2M_LAUNCH(k_mesh_uf_edges, dim3(grid_for(n_tri)), dim3(256), 0, st, tri, n_tri, hkeys, hval, cap, parent);
    GS2M_LAUNCH(k_mesh_uf_roots, dim3(grid_for(n_tri)), dim3(256), 0, st, n_tri, parent, root, flag);
    gs2m_launch_scan_u32(st, flag, n_tri, pos, scratch);
    GS2M_LAUNCH(k_mesh_uf_labels, dim3(grid_for(n_tri)), dim3(256), 0, st, n_tri, root, pos, labels, cluster_n);
}
