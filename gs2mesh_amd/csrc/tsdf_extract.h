// tsdf_extract.h -- triangle-mesh extraction from the block-sparse TSDF (marching cubes).
//
// Replaces volume.extract_triangle_mesh() (gs2mesh_utils/tsdf_utils.py:108) = Open3D 0.17
// ScalableTSDFVolume::ExtractTriangleMesh: for every allocated block and every voxel (x,y,z) the cube
// spanned by the voxel and its +1 neighbours (reaching into neighbouring blocks) is skipped if any of
// its 8 corners has weight 0; corner i is inside when tsdf < 0; a vertex on a cut edge lies at
//   lower_corner_centre + |f_lo| / (|f_lo| + |f_hi|) * voxel_length  along the edge axis,
// its colour is the same interpolation of the corner colours (mean colour = rgb_sum / weight, / 255).
// Two passes over the blocks (count, then emit after an exclusive scan of the per-block triangle
// counts); triangles are emitted un-welded (3 vertices each) in a deterministic order -- the host
// mirror welds identical vertices (shared edges produce bit-identical positions because every edge is
// always evaluated from its lower to its upper corner).
#pragma once
#include "mc_tables.h"
#include "tsdf_kernels.h"

struct McDevTables {
    signed char tri[256][GS2M_MC_MAX_TRIS * 3];
    unsigned char ntri[256];
};

// cube corner offsets and edge endpoints (device copies of mc_tables.h)
struct McGeom {
    int corner[8][3];
    int edge[12][3];  // lower corner, upper corner, axis
};

GS2M_DEVICE int mc_cube_case(const TsdfVolume& V, const int* nb_slot, int x, int y, int z, float* f, float* w) {
    int ci = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cx = x + ((i == 1 || i == 2 || i == 5 || i == 6) ? 1 : 0);
        const int cy = y + ((i == 2 || i == 3 || i == 6 || i == 7) ? 1 : 0);
        const int cz = z + (i >= 4 ? 1 : 0);
        const int slot = nb_slot[(cx >> 4) | ((cy >> 4) << 1) | ((cz >> 4) << 2)];
        if (slot < 0) return 0;
        const size_t vi = (size_t)slot * GS2M_TSDF_VOX + GS2M_TSDF_VINDEX(cx & 15, cy & 15, cz & 15);
        w[i] = V.weight[vi];
        if (w[i] == 0.0f) return 0;
        f[i] = V.tsdf[vi];
        if (f[i] < 0.0f) ci |= 1 << i;
    }
    return ci == 255 ? 0 : ci;
}

GS2M_DEVICE void mc_neighbour_slots(const TsdfVolume& V, int slot, int* nb_slot /* LDS [8] */, int tid) {
    if (tid < 8) {
        const int bx = V.block_keys[3 * (size_t)slot] + (tid & 1);
        const int by = V.block_keys[3 * (size_t)slot + 1] + ((tid >> 1) & 1);
        const int bz = V.block_keys[3 * (size_t)slot + 2] + ((tid >> 2) & 1);
        nb_slot[tid] = tid == 0 ? slot : (tsdf_key_in_range(bx, by, bz) ? tsdf_lookup(V, tsdf_pack_key(bx, by, bz)) : -1);
    }
}

// pass 1: triangles per block
GS2M_KERNEL void __launch_bounds__(256)
k_mc_count(TsdfVolume V, const McDevTables* __restrict__ T, unsigned n_blocks, unsigned* __restrict__ blk_tris) {
    __shared__ int nb_slot[8];
    __shared__ unsigned total;
    const int tid = (int)threadIdx.x;
    const int slot = (int)blockIdx.x;
    if ((unsigned)slot >= n_blocks) return;
    if (V.halo[slot]) {   // neighbour-only block: another rank starts the cubes of this block
        if (tid == 0) blk_tris[slot] = 0u;
        return;
    }
    if (tid == 0) total = 0u;
    mc_neighbour_slots(V, slot, nb_slot, tid);
    __syncthreads();
    unsigned cnt = 0;
    for (int v = tid; v < GS2M_TSDF_VOX; v += 256) {
        const int x = v >> 8, y = (v >> 4) & 15, z = v & 15;
        float f[8], w[8];
        const int ci = mc_cube_case(V, nb_slot, x, y, z, f, w);
        cnt += T->ntri[ci];
    }
    if (cnt) atomicAdd(&total, cnt);
    __syncthreads();
    if (tid == 0) blk_tris[slot] = total;
}

// exclusive scan of blk_tris (single 1024-thread workgroup); total -> *n_total
GS2M_KERNEL void __launch_bounds__(1024)
k_mc_scan(unsigned* __restrict__ blk_tris, unsigned n_blocks, unsigned long long* __restrict__ n_total) {
    __shared__ unsigned long long part[1024];
    const int tid = (int)threadIdx.x;
    const unsigned per = (n_blocks + 1023u) / 1024u;
    const unsigned lo = (unsigned)tid * per;
    const unsigned hi = lo + per < n_blocks ? lo + per : n_blocks;
    unsigned long long s = 0;
    for (unsigned i = lo; i < hi; ++i) s += blk_tris[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long add = tid >= off ? part[tid - off] : 0ull;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (unsigned i = lo; i < hi; ++i) {
        const unsigned c = blk_tris[i];
        blk_tris[i] = (unsigned)run;  // offsets fit 32 bits: checked by the host (total < 2^32)
        run += c;
    }
    if (tid == 1023) *n_total = part[1023];
}

// pass 2: emit.  vertices / colours: [n_tri][3][3] float64 (Open3D meshes are double); edge_index (optional): [n_tri][3][4] int32.
GS2M_KERNEL void __launch_bounds__(256)
k_mc_emit(TsdfVolume V, const McDevTables* __restrict__ T, McGeom G, unsigned n_blocks,
          const unsigned* __restrict__ blk_off, unsigned long long max_tris, double voxel_length, double unit_length,
          double* __restrict__ vertices, double* __restrict__ colors, int* __restrict__ edge_index) {
    __shared__ int nb_slot[8];
    __shared__ unsigned scan[256];
    const int tid = (int)threadIdx.x;
    const int slot = (int)blockIdx.x;
    if ((unsigned)slot >= n_blocks || V.halo[slot]) return;
    mc_neighbour_slots(V, slot, nb_slot, tid);
    __syncthreads();
    const int bx = V.block_keys[3 * (size_t)slot], by = V.block_keys[3 * (size_t)slot + 1],
              bz = V.block_keys[3 * (size_t)slot + 2];
    // deterministic order: thread t owns voxels t*16 .. t*16+15 (x*256 + y*16 + z order); positions by a
    // workgroup scan of the per-thread triangle counts
    unsigned cnt = 0;
    for (int k = 0; k < 16; ++k) {
        const int v = tid * 16 + k;
        float f[8], w[8];
        cnt += T->ntri[mc_cube_case(V, nb_slot, v >> 8, (v >> 4) & 15, v & 15, f, w)];
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned add = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    unsigned long long out = (unsigned long long)blk_off[slot] + (scan[tid] - cnt);
    const double half = voxel_length * 0.5;
    for (int k = 0; k < 16; ++k) {
        const int v = tid * 16 + k;
        const int x = v >> 8, y = (v >> 4) & 15, z = v & 15;
        float f[8], w[8];
        const int ci = mc_cube_case(V, nb_slot, x, y, z, f, w);
        const int nt = T->ntri[ci];
        if (nt == 0) continue;
        // corner colours (mean = sum / weight), needed only here
        for (int t = 0; t < nt; ++t, ++out) {
            if (out >= max_tris) continue;
            for (int j = 0; j < 3; ++j) {
                const int e = T->tri[ci][3 * t + j];
                const int c_lo = G.edge[e][0], c_hi = G.edge[e][1], axis = G.edge[e][2];
                const double f0 = fabs((double)f[c_lo]), f1 = fabs((double)f[c_hi]);   // Open3D: |double(f)|, sums in double
                // global voxel index of the lower corner
                const int gx = bx * GS2M_TSDF_RES + x + G.corner[c_lo][0];
                const int gy = by * GS2M_TSDF_RES + y + G.corner[c_lo][1];
                const int gz = bz * GS2M_TSDF_RES + z + G.corner[c_lo][2];
                double p[3] = {half + voxel_length * gx, half + voxel_length * gy, half + voxel_length * gz};
                p[axis] += f0 * voxel_length / (f0 + f1);
                double* vo = vertices + (out * 3 + j) * 3;
                vo[0] = p[0];
                vo[1] = p[1];
                vo[2] = p[2];
                if (edge_index) {   // Open3D's vertex key: global voxel index of the edge's lower corner + axis
                    int* eo = edge_index + (out * 3 + j) * 4;
                    eo[0] = gx;
                    eo[1] = gy;
                    eo[2] = gz;
                    eo[3] = axis;
                }
                if (colors) {
                    double col[3] = {0, 0, 0};
                    if (V.has_color) {
                        double cend[2][3];   // corner colours: mean colour (sum / weight) / 255
                        for (int s = 0; s < 2; ++s) {
                            const int cc = s ? c_hi : c_lo;
                            const int cx = x + G.corner[cc][0], cy = y + G.corner[cc][1], cz = z + G.corner[cc][2];
                            const int sl = nb_slot[(cx >> 4) | ((cy >> 4) << 1) | ((cz >> 4) << 2)];
                            const size_t vi = GS2M_TSDF_VINDEX(cx & 15, cy & 15, cz & 15);
                            const double wv = (double)V.weight[(size_t)sl * GS2M_TSDF_VOX + vi];
                            for (int ch = 0; ch < 3; ++ch)
                                cend[s][ch] = ((double)V.rgb[((size_t)sl * 3 + ch) * GS2M_TSDF_VOX + vi] / wv) / 255.0;
                        }
                        for (int ch = 0; ch < 3; ++ch) col[ch] = (f1 * cend[0][ch] + f0 * cend[1][ch]) / (f0 + f1);   // Open3D's form
                    }
                    double* co = colors + (out * 3 + j) * 3;
                    co[0] = col[0];
                    co[1] = col[1];
                    co[2] = col[2];
                }
            }
        }
    }
    (void)unit_length;
}
