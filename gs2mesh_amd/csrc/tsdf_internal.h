// tsdf_internal.h -- launch wrappers of the TSDF kernels.
#pragma once
#include "tsdf_common.h"

void gs2m_launch_tsdf_touch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, const float* depth,
                            const unsigned char* mask);
void gs2m_launch_tsdf_integrate(hipStream_t st, int n_wg, const TsdfVolume& V, const TsdfFrame& f,
                                const float* depth, const unsigned char* color, const unsigned char* mask);
void gs2m_launch_tsdf_touch_batch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, int n_frames,
                                  const TsdfBatchFrame* frames);
void gs2m_launch_tsdf_integrate_batch(hipStream_t st, int n_wg, const TsdfVolume& V, const TsdfBatchFrame* frames);
void gs2m_launch_tsdf_clear_used(hipStream_t st, const TsdfVolume& V);
void gs2m_launch_tsdf_pack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, int form, float* buf, long long* ibuf);
void gs2m_launch_tsdf_unpack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, int form, const float* buf,
                             const long long* ibuf, int halo);
void gs2m_launch_tsdf_clear_from(hipStream_t st, const TsdfVolume& V, unsigned first);
void gs2m_launch_tsdf_clear_gap(hipStream_t st, const TsdfVolume& V, unsigned upto);
void gs2m_launch_tsdf_owned_keys(hipStream_t st, unsigned n, const TsdfVolume& V, int* keys);
void gs2m_launch_tsdf_block_map(hipStream_t st, const TsdfVolume& V, const int* lo, const int* dim, unsigned char* cells, unsigned n_cells,
                                unsigned flags, unsigned win_hash, int rank, unsigned frames_local, unsigned frames_base);
void gs2m_launch_tsdf_map_keys(hipStream_t st, const int* lo, const int* dim, unsigned char* cells, unsigned n_cells, int* keys, unsigned max_keys);
// marching cubes (tsdf_extract.h)
struct McDevTables;
size_t gs2m_mc_tables_bytes();
bool gs2m_mc_tables_fill(void* host_buf);  // generates the case table (host); false on internal error
void gs2m_launch_mc_count(hipStream_t st, const TsdfVolume& V, const McDevTables* T, unsigned n_blocks,
                          unsigned* blk_tris, unsigned long long* n_total);
void gs2m_launch_mc_emit(hipStream_t st, const TsdfVolume& V, const McDevTables* T, unsigned n_blocks,
                         const unsigned* blk_off, unsigned long long max_tris, double voxel_length, double unit_length,
                         double* vertices, double* colors, int* edge_index);
// mesh post-processing (mesh_kernels.h)
void gs2m_launch_scan_u32(hipStream_t st, const unsigned* in, unsigned n, unsigned* out, unsigned* scratch);
void gs2m_launch_mesh_weld_count(hipStream_t st, unsigned n, const int* edge_index, int* mins, unsigned long long* hkeys, unsigned* hfirst,
                                 unsigned cap, unsigned* cell_of, unsigned* flag, unsigned* pos, unsigned* scratch, unsigned* bad);
void gs2m_launch_mesh_weld_emit(hipStream_t st, unsigned n, const unsigned* hfirst, const unsigned* cell_of, const unsigned* pos, const double* verts,
                                const double* cols, const int* edge_index, double* out_v, double* out_c, int* out_e, int* out_tri);
void gs2m_launch_mesh_cluster(hipStream_t st, const int* tri, unsigned n_tri, unsigned long long* hkeys, unsigned* hval, unsigned cap,
                              unsigned* parent, unsigned* root, unsigned* flag, unsigned* pos, unsigned* scratch, int* labels,
                              unsigned long long* cluster_n);
void gs2m_set_error(const char* fmt, ...);
