// mesh_kernels.h -- device-side mesh post-processing of the fuse half:
//   * welding the marching-cubes triangle soup by Open3D's vertex identity (the cut edge: global voxel index of its lower corner
//     + axis; ScalableTSDFVolume::ExtractTriangleMesh keeps an edge -> vertex map) -- gs2mesh_utils/tsdf_utils.py:108;
//   * connected components over shared edges (TriangleMesh::ClusterConnectedTriangles) -- tsdf_utils.py:133.
// Both used to run on the host (numpy unique / scipy connected_components) on a 144 B-per-triangle soup copied over PCIe.
//
// Weld: an open-addressing hash table keyed by the 62-bit edge key keeps, per key, the SMALLEST soup index that carries it
// (atomicMin): the vertex order of the result is first appearance in the soup, which is what the host path produced and what makes
// the output independent of the arrival order of the threads.  An exclusive scan of the "I am the first" flags numbers the
// vertices; a last pass writes the compact arrays and the triangle indices.
// Cluster: lock-free union-find (hook the larger root under the smaller with atomicCAS, path halving).  A second hash table maps
// an undirected edge (vmin, vmax) to the first triangle that registered it; every other triangle with that edge is united with it.
// The root of a component is its smallest triangle index, so numbering the roots in index order reproduces the labels of a
// breadth-first sweep from triangle 0 upwards (scipy / Open3D number clusters by their first triangle).
#pragma once
#include "platform.h"

#define GS2M_MESH_EMPTY 0xffffffffffffffffull

GS2M_DEVICE unsigned mesh_hash64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}

// find-or-insert `key` (never GS2M_MESH_EMPTY) -> cell index; cap is a power of two > the number of distinct keys
GS2M_DEVICE unsigned mesh_table_cell(unsigned long long* __restrict__ hkeys, unsigned cap, unsigned long long key) {
    const unsigned mask = cap - 1u;
    unsigned h = mesh_hash64(key) & mask;
    for (;;) {
        const unsigned long long k = hkeys[h];
        if (k == key) return h;
        if (k == GS2M_MESH_EMPTY) {
            const unsigned long long prev = atomicCAS(&hkeys[h], GS2M_MESH_EMPTY, key);
            if (prev == GS2M_MESH_EMPTY || prev == key) return h;
        }
        h = (h + 1u) & mask;
    }
}

// ---- weld ---------------------------------------------------------------------------------------------------------------
// per-axis minimum of the soup's voxel indices (the keys are made relative to it): mins[3] start at INT_MAX
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_key_min(const int* __restrict__ edge_index, unsigned n, int* __restrict__ mins) {
    __shared__ int s_min[3];
    if (threadIdx.x < 3) s_min[threadIdx.x] = 0x7fffffff;
    __syncthreads();
    int m0 = 0x7fffffff, m1 = 0x7fffffff, m2 = 0x7fffffff;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const int x = edge_index[4 * (size_t)i], y = edge_index[4 * (size_t)i + 1], z = edge_index[4 * (size_t)i + 2];
        m0 = x < m0 ? x : m0;
        m1 = y < m1 ? y : m1;
        m2 = z < m2 ? z : m2;
    }
    // one LDS atomic per thread, one global atomic per workgroup (every thread on the three global words: 0.56 ms for 2.6 M keys)
    atomicMin(&s_min[0], m0);
    atomicMin(&s_min[1], m1);
    atomicMin(&s_min[2], m2);
    __syncthreads();
    if (threadIdx.x < 3 && s_min[threadIdx.x] != 0x7fffffff) atomicMin(&mins[threadIdx.x], s_min[threadIdx.x]);
}

// 62-bit key of soup vertex i: 20 bits per axis relative to mins, 2 bits axis; bad[0] |= 1 if the soup spans more than 2^20 voxels
GS2M_DEVICE unsigned long long mesh_edge_key(const int* __restrict__ edge_index, size_t i, const int* __restrict__ mins, unsigned* bad) {
    const unsigned x = (unsigned)(edge_index[4 * i] - mins[0]), y = (unsigned)(edge_index[4 * i + 1] - mins[1]);
    const unsigned z = (unsigned)(edge_index[4 * i + 2] - mins[2]), a = (unsigned)edge_index[4 * i + 3];
    if ((x | y | z) >> 20 || a > 2u) atomicOr(bad, 1u);
    return (unsigned long long)(x & 0xfffffu) | ((unsigned long long)(y & 0xfffffu) << 20) | ((unsigned long long)(z & 0xfffffu) << 40) |
           ((unsigned long long)(a & 3u) << 60);
}

// pass 1: cell of every soup vertex; per cell the smallest soup index (hfirst starts at 0xffffffff)
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_weld_insert(const int* __restrict__ edge_index, unsigned n, const int* __restrict__ mins, unsigned long long* __restrict__ hkeys,
                   unsigned* __restrict__ hfirst, unsigned cap, unsigned* __restrict__ cell_of, unsigned* __restrict__ bad) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned c = mesh_table_cell(hkeys, cap, mesh_edge_key(edge_index, i, mins, bad));
        cell_of[i] = c;
        atomicMin(&hfirst[c], i);
    }
}
// pass 2: flag[i] = 1 where soup vertex i is the first carrier of its key
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_weld_flag(unsigned n, const unsigned* __restrict__ hfirst, const unsigned* __restrict__ cell_of, unsigned* __restrict__ flag) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) flag[i] = hfirst[cell_of[i]] == i ? 1u : 0u;
}
// pass 3 (after the exclusive scan of flag -> pos): triangle indices; the first carriers write the compact vertex arrays
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_weld_emit(unsigned n, const unsigned* __restrict__ hfirst, const unsigned* __restrict__ cell_of, const unsigned* __restrict__ pos,
                 const double* __restrict__ verts, const double* __restrict__ cols, const int* __restrict__ edge_index,
                 double* __restrict__ out_v, double* __restrict__ out_c, int* __restrict__ out_e, int* __restrict__ out_tri) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned f = hfirst[cell_of[i]];
        const unsigned p = pos[f];
        out_tri[i] = (int)p;
        if (f == i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                out_v[3 * (size_t)p + k] = verts[3 * (size_t)i + k];
                if (out_c) out_c[3 * (size_t)p + k] = cols ? cols[3 * (size_t)i + k] : 0.0;
            }
            if (out_e) {
#pragma unroll
                for (int k = 0; k < 4; ++k) out_e[4 * (size_t)p + k] = edge_index[4 * (size_t)i + k];
            }
        }
    }
}

// ---- exclusive scan of u32 (three launches: tile sums, scan of the sums by one workgroup, apply) --------------------------
#define GS2M_SCAN_TILE 4096u   // items per workgroup of 1024 threads (4 each)
GS2M_DEVICE unsigned mesh_wg_exclusive_scan_1024(unsigned v, unsigned* wave_sum /* LDS [16] */, unsigned* total) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = gs2m_shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();   // wave_sum may still be read by the previous use
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    unsigned base = 0, all = 0;
    for (int k = 0; k < 16; ++k) {
        if (k < wave) base += wave_sum[k];
        all += wave_sum[k];
    }
    if (total) *total = all;
    return base + inc - v;
}
GS2M_KERNEL void __launch_bounds__(1024)
k_scan_tile_sums(const unsigned* __restrict__ in, unsigned n, unsigned* __restrict__ sums) {
    __shared__ unsigned wave_sum[16];
    const unsigned base = blockIdx.x * GS2M_SCAN_TILE + threadIdx.x * 4u;
    unsigned s = 0;
    for (unsigned k = 0; k < 4u; ++k) s += base + k < n ? in[base + k] : 0u;
    unsigned total = 0;
    (void)mesh_wg_exclusive_scan_1024(s, wave_sum, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: exclusive scan of sums[m] in place; the grand total -> total_out[0]
GS2M_KERNEL void __launch_bounds__(1024)
k_scan_sums(unsigned* __restrict__ sums, unsigned m, unsigned* __restrict__ total_out) {
    __shared__ unsigned wave_sum[16];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (unsigned b0 = 0; b0 < m; b0 += 1024u) {
        const unsigned i = b0 + threadIdx.x;
        const unsigned v = i < m ? sums[i] : 0u;
        unsigned total = 0;
        const unsigned ex = mesh_wg_exclusive_scan_1024(v, wave_sum, &total);
        const unsigned c = carry;
        if (i < m) sums[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = carry;
}
GS2M_KERNEL void __launch_bounds__(1024)
k_scan_apply(const unsigned* __restrict__ in, unsigned n, const unsigned* __restrict__ sums, unsigned* __restrict__ out) {
    __shared__ unsigned wave_sum[16];
    const unsigned base = blockIdx.x * GS2M_SCAN_TILE + threadIdx.x * 4u;
    unsigned v[4], s = 0;
    for (unsigned k = 0; k < 4u; ++k) {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    unsigned run = sums[blockIdx.x] + mesh_wg_exclusive_scan_1024(s, wave_sum, nullptr);
    for (unsigned k = 0; k < 4u; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---- connected components over shared edges ------------------------------------------------------------------------------
GS2M_DEVICE unsigned mesh_uf_find(unsigned* __restrict__ parent, unsigned i) {
    unsigned p = gs2m_load_agent(&parent[i]);
    while (p != i) {
        const unsigned g = gs2m_load_agent(&parent[p]);
        if (g != p) parent[i] = g;     // path halving (a benign race: any ancestor is a valid parent)
        i = p;
        p = g;
    }
    return i;
}
GS2M_DEVICE void mesh_uf_union(unsigned* __restrict__ parent, unsigned a, unsigned b) {
    for (;;) {
        a = mesh_uf_find(parent, a);
        b = mesh_uf_find(parent, b);
        if (a == b) return;
        if (a > b) {
            const unsigned t = a;
            a = b;
            b = t;
        }
        // hook the larger root under the smaller: the root of a component ends up its smallest triangle
        if (atomicCAS(&parent[b], b, a) == b) return;
    }
}
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_uf_init(unsigned n, unsigned* __restrict__ parent) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) parent[i] = i;
}
// every triangle registers its three undirected edges; hval[cell] = a triangle that carries the edge (the first to arrive),
// everybody else is united with it
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_uf_edges(const int* __restrict__ tri, unsigned n_tri, unsigned long long* __restrict__ hkeys, unsigned* __restrict__ hval,
                unsigned cap, unsigned* __restrict__ parent) {
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n_tri; t += gridDim.x * 256u) {
        const unsigned v[3] = {(unsigned)tri[3 * (size_t)t], (unsigned)tri[3 * (size_t)t + 1], (unsigned)tri[3 * (size_t)t + 2]};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const unsigned a = v[e], b = v[(e + 1) % 3];
            const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
            const unsigned c = mesh_table_cell(hkeys, cap, ((unsigned long long)lo << 32) | hi);
            const unsigned prev = atomicCAS(&hval[c], 0xffffffffu, t);
            if (prev != 0xffffffffu && prev != t) mesh_uf_union(parent, prev, t);
        }
    }
}
// root of every triangle (the component's smallest triangle) + flag[t] = 1 where t is a root
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_uf_roots(unsigned n_tri, unsigned* __restrict__ parent, unsigned* __restrict__ root, unsigned* __restrict__ flag) {
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n_tri; t += gridDim.x * 256u) {
        const unsigned r = mesh_uf_find(parent, t);
        root[t] = r;
        flag[t] = r == t ? 1u : 0u;
    }
}
// labels = rank of the root among the roots (pos = exclusive scan of flag); cluster sizes by atomics, aggregated per wave: the
// triangles of a wave mostly belong to ONE component (a scene is one big surface + debris), and 0.87 M single atomics on the
// big component's counter took 10 ms
GS2M_KERNEL void __launch_bounds__(256)
k_mesh_uf_labels(unsigned n_tri, const unsigned* __restrict__ root, const unsigned* __restrict__ pos, int* __restrict__ labels,
                 unsigned long long* __restrict__ cluster_n) {
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned stride = gridDim.x * 256u;
    const unsigned rounds = (n_tri + stride - 1u) / stride;      // the same for every lane: the ballots below are collectives
    for (unsigned r = 0; r < rounds; ++r) {
        const unsigned t = r * stride + blockIdx.x * 256u + threadIdx.x;
        const bool have = t < n_tri;
        const unsigned l = have ? pos[root[t]] : 0xffffffffu;
        if (have) labels[t] = (int)l;
        unsigned long long todo = gs2m_ballot(have ? 1 : 0);
        while (todo != 0ull) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned ll = gs2m_shfl(l, leader);
            const unsigned long long same = gs2m_ballot(have && l == ll ? 1 : 0);
            if (lane == leader) atomicAdd(&cluster_n[ll], (unsigned long long)gs2m_popc64(same));
            todo &= ~same;
        }
    }
}
