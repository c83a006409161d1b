/*
 * tsdf_oracle.cpp -- CPU restatement of Open3D 0.17.0 ScalableTSDFVolume::Integrate and ::ExtractTriangleMesh.
 *
 * TEST INFRASTRUCTURE ONLY (and bench.py's cpu_baseline leg).  Nothing in the product
 * path may import, link or execute this file.
 *
 * The reference calls Open3D through Python (gs2mesh_utils/tsdf_utils.py:53-56 ctor,
 * :88-93 RGBDImage.create_from_color_and_depth, :106-107 PinholeCameraIntrinsic +
 * volume.integrate).  Open3D is a pip dependency pinned `open3d==0.17.0`
 * (requirements.txt:15); its C++ is NOT in /root/reference and the module is not
 * installed here, so this file restates the published v0.17.0 algorithm
 *   cpp/open3d/pipelines/integration/ScalableTSDFVolume.cpp   (Integrate, OpenVolumeUnit,
 *                                                              LocateVolumeUnit)
 *   cpp/open3d/pipelines/integration/UniformTSDFVolume.cpp    (IntegrateWithDepthToCamera
 *                                                              DistanceMultiplier)
 *   cpp/open3d/geometry/RGBDImageFactory.cpp, ImageFactory.cpp (CreateFromColorAndDepth,
 *       ConvertDepthToFloatImage, CreateDepthToCameraDistanceMultiplierFloatImage)
 *   cpp/open3d/geometry/PointCloudFactory.cpp                 (CreatePointCloudFromFloatDepthImage)
 * from knowledge of upstream.  PARITY UNPINNED: the reference holds no golden vectors,
 * tests or fixtures for this boundary; the oracle is pinned only by analytic cases
 * (tests/test_oracle_tsdf.py) until a real open3d==0.17.0 wheel can be A/B'd.
 *
 * Structure mirrors upstream on purpose (hash map of 16^3 blocks, blocks integrated
 * serially at first touch within a frame, OpenMP only over the x index of one block) so
 * that, timed on host cores, it is a fair stand-in for the reference's CPU TSDF path
 * ("restated Open3D 0.17 (CPU)" in BASELINE.md section 3).
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Vec3i {
    int x, y, z;
    bool operator==(const Vec3i& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct Vec3iHash { /* utility::hash_eigen: boost-style hash_combine over the coefficients */
    size_t operator()(const Vec3i& v) const {
        size_t seed = 0;
        const int c[3] = {v.x, v.y, v.z};
        for (int i = 0; i < 3; ++i)
            seed ^= std::hash<int>()(c[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

struct TSDFVoxel { /* geometry::TSDFVoxel: float tsdf_, float weight_, Vector3d color_ */
    float tsdf = 0.f;
    float weight = 0.f;
    double color[3] = {0, 0, 0};
};

struct Intrinsic {
    int width, height;
    double fx, fy, cx, cy;
};

struct Image { /* float depth + u8 colour of one RGBDImage */
    int width, height;
    const float* depth;   /* [H,W] */
    const uint8_t* color; /* [H,W,3] or null */
};

/* UniformTSDFVolume restricted to what ScalableTSDFVolume uses */
struct Unit {
    double origin[3];
    std::vector<TSDFVoxel> voxels;
};

struct Volume {
    double voxel_length;
    double sdf_trunc;
    int color_type; /* 0 none, 1 RGB8 */
    int resolution; /* volume_unit_resolution_ (16) */
    int stride;     /* depth_sampling_stride_ (4) */
    double unit_length;
    std::unordered_map<Vec3i, std::unique_ptr<Unit>, Vec3iHash> units;
    std::vector<Vec3i> order; /* allocation order, for export */
    int64_t block_updates = 0;
    int num_threads = 0;
    /* result of the last oracle_tsdf_extract_mesh */
    std::vector<double> mesh_vertices, mesh_colors;
    std::vector<int32_t> mesh_triangles, mesh_edge_index;
    int64_t mesh_zero_offset_vertices = 0;
};

inline int IndexOf(int x, int y, int z, int res) { return x * res * res + y * res + z; }

/* Eigen::Matrix4d::inverse() for a general 4x4: cofactor expansion in double.  (Eigen uses
 * a blocked cofactor formula with SSE; results agree to ~1 ulp, which can only matter for a
 * back-projected point within 1e-16 relative of a block boundary.) */
bool invert4(const double* m, double* inv) {
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int i = 0; i < 16; i++) inv[i] = a[i] * det;
    return true;
}

/* UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier */
void IntegrateUnit(const Volume& vol, Unit& unit, const Image& image, const Intrinsic& K,
                   const double* extrinsic /* row-major 4x4 world->cam */,
                   const float* dist_mult /* [H,W] */) {
    const float fx = static_cast<float>(K.fx);
    const float fy = static_cast<float>(K.fy);
    const float cx = static_cast<float>(K.cx);
    const float cy = static_cast<float>(K.cy);
    float E[4][4]; /* extrinsic.cast<float>() */
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) E[r][c] = static_cast<float>(extrinsic[4 * r + c]);
    const float voxel_length_f = static_cast<float>(vol.voxel_length);
    const float half_voxel_length_f = voxel_length_f * 0.5f;
    const float sdf_trunc_f = static_cast<float>(vol.sdf_trunc);
    const float sdf_trunc_inv_f = 1.0f / sdf_trunc_f;
    /* extrinsic_scaled_f = extrinsic_f * voxel_length_f; only column 2, rows 0..2 used */
    const float Es02 = E[0][2] * voxel_length_f;
    const float Es12 = E[1][2] * voxel_length_f;
    const float Es22 = E[2][2] * voxel_length_f;
    const float safe_width_f = K.width - 0.0001f;
    const float safe_height_f = K.height - 0.0001f;
    const int res = vol.resolution;

#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(vol.num_threads > 0 ? vol.num_threads : omp_get_max_threads())
#endif
    for (int x = 0; x < res; x++) {
        for (int y = 0; y < res; y++) {
            /* Eigen::Vector4f pt_3d_homo(float(half + vl*x + origin(0)), ...): the float sum
             * (half + vl*x) is promoted to double by + origin_, then cast to float */
            const float p0 = float(half_voxel_length_f + voxel_length_f * x + unit.origin[0]);
            const float p1 = float(half_voxel_length_f + voxel_length_f * y + unit.origin[1]);
            const float p2 = float(half_voxel_length_f + unit.origin[2]);
            /* pt_camera = extrinsic_f * pt_3d_homo  (Eigen 4x4 * 4x1: sum over j = 0..3 in order) */
            float pc0 = E[0][0] * p0 + E[0][1] * p1 + E[0][2] * p2 + E[0][3] * 1.f;
            float pc1 = E[1][0] * p0 + E[1][1] * p1 + E[1][2] * p2 + E[1][3] * 1.f;
            float pc2 = E[2][0] * p0 + E[2][1] * p1 + E[2][2] * p2 + E[2][3] * 1.f;
            for (int z = 0; z < res; z++, pc0 += Es02, pc1 += Es12, pc2 += Es22) {
                if (pc2 <= 0) continue;
                float u_f = pc0 * fx / pc2 + cx + 0.5f;
                float v_f = pc1 * fy / pc2 + cy + 0.5f;
                if (!(u_f >= 0.0001f && u_f < safe_width_f && v_f >= 0.0001f && v_f < safe_height_f))
                    continue;
                int u = (int)u_f;
                int v = (int)v_f;
                float d = image.depth[(size_t)v * image.width + u];
                if (d <= 0.0f) continue;
                int v_ind = IndexOf(x, y, z, res);
                float sdf = (d - pc2) * dist_mult[(size_t)v * image.width + u];
                if (sdf > -sdf_trunc_f) {
                    float tsdf = std::min(1.0f, sdf * sdf_trunc_inv_f);
                    TSDFVoxel& vx = unit.voxels[v_ind];
                    vx.tsdf = (vx.tsdf * vx.weight + tsdf) / (vx.weight + 1.0f);
                    if (vol.color_type == 1 && image.color) {
                        const uint8_t* rgb = image.color + 3 * ((size_t)v * image.width + u);
                        for (int c = 0; c < 3; ++c)
                            vx.color[c] = (vx.color[c] * vx.weight + (double)rgb[c]) / (vx.weight + 1.0f);
                    }
                    vx.weight += 1.0f;
                }
            }
        }
    }
}

} /* namespace */

extern "C" {

typedef struct Volume oracle_tsdf;

oracle_tsdf* oracle_tsdf_create(double voxel_length, double sdf_trunc, int color_type,
                                int volume_unit_resolution, int depth_sampling_stride) {
    Volume* v = new Volume();
    v->voxel_length = voxel_length;
    v->sdf_trunc = sdf_trunc;
    v->color_type = color_type;
    v->resolution = volume_unit_resolution;
    v->stride = depth_sampling_stride;
    /* ScalableTSDFVolume ctor: volume_unit_length_ = voxel_length * volume_unit_resolution */
    v->unit_length = voxel_length * volume_unit_resolution;
    return v;
}
void oracle_tsdf_destroy(oracle_tsdf* v) { delete v; }
void oracle_tsdf_set_threads(oracle_tsdf* v, int n) { v->num_threads = n; }
int oracle_tsdf_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * RGBDImage::CreateFromColorAndDepth(color, depth, depth_scale, depth_trunc, false):
 * depth_f = depth / (float)depth_scale; if (depth_f >= depth_trunc) depth_f = 0
 * (Image::ConvertDepthToFloatImage; the comparison promotes the float to double).
 */
void oracle_rgbd_convert_depth(const float* depth_in, float* depth_out, int64_t n,
                               double depth_scale, double depth_trunc) {
    for (int64_t i = 0; i < n; ++i) {
        float p = depth_in[i];
        p /= (float)depth_scale;
        if (p >= depth_trunc) p = 0.0f;
        depth_out[i] = p;
    }
}

/* Image::CreateDepthToCameraDistanceMultiplierFloatImage */
void oracle_dist_multiplier(int width, int height, double fx, double fy, double cx, double cy,
                            float* out) {
    float ffl_inv[2] = {1.0f / (float)fx, 1.0f / (float)fy};
    float fpp[2] = {(float)cx, (float)cy};
    std::vector<float> xx(width), yy(height);
    for (int j = 0; j < width; j++) xx[j] = (j - fpp[0]) * ffl_inv[0];
    for (int i = 0; i < height; i++) yy[i] = (i - fpp[1]) * ffl_inv[1];
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++)
            out[(size_t)i * width + j] = sqrtf(xx[j] * xx[j] + yy[i] * yy[i] + 1.0f);
}

/*
 * ScalableTSDFVolume::Integrate(image, intrinsic, extrinsic).  depth is the ALREADY
 * converted float depth (oracle_rgbd_convert_depth); color [H,W,3] u8 or NULL;
 * extrinsic row-major 4x4 double world->camera.  Returns blocks integrated this frame.
 */
int64_t oracle_tsdf_integrate(oracle_tsdf* vol, const float* depth, const uint8_t* color,
                              int width, int height, double fx, double fy, double cx, double cy,
                              const double* extrinsic) {
    Intrinsic K{width, height, fx, fy, cx, cy};
    Image image{width, height, depth, color};
    std::vector<float> dist_mult((size_t)width * height);
    oracle_dist_multiplier(width, height, fx, fy, cx, cy, dist_mult.data());
    /* PointCloud::CreateFromDepthImage(depth, intrinsic, extrinsic, 1000, 1000, stride):
     * float depth -> CreatePointCloudFromFloatDepthImage */
    double pose[16];
    if (!invert4(extrinsic, pose)) return -1;
    std::unordered_set<Vec3i, Vec3iHash> touched;
    int64_t n_int = 0;
    const double L = vol->unit_length;
    const double tr = vol->sdf_trunc;
    for (int i = 0; i < height; i += vol->stride) {
        for (int j = 0; j < width; j += vol->stride) {
            const float p = depth[(size_t)i * width + j];
            if (!(p > 0)) continue;
            double z = (double)p;
            double x = (j - cx) * z / fx;
            double y = (i - cy) * z / fy;
            /* camera_pose * Vector4d(x, y, z, 1.0) */
            double pw[3];
            for (int r = 0; r < 3; ++r)
                pw[r] = pose[4 * r + 0] * x + pose[4 * r + 1] * y + pose[4 * r + 2] * z + pose[4 * r + 3] * 1.0;
            /* LocateVolumeUnit(point -/+ (trunc,trunc,trunc)) = floor(p / unit_length) */
            int lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = (int)std::floor((pw[a] - tr) / L);
                hi[a] = (int)std::floor((pw[a] + tr) / L);
            }
            for (int bx = lo[0]; bx <= hi[0]; bx++)
                for (int by = lo[1]; by <= hi[1]; by++)
                    for (int bz = lo[2]; bz <= hi[2]; bz++) {
                        Vec3i loc{bx, by, bz};
                        if (touched.find(loc) != touched.end()) continue;
                        touched.insert(loc);
                        /* OpenVolumeUnit */
                        auto& slot = vol->units[loc];
                        if (!slot) {
                            slot.reset(new Unit());
                            slot->origin[0] = (double)bx * L;
                            slot->origin[1] = (double)by * L;
                            slot->origin[2] = (double)bz * L;
                            slot->voxels.resize((size_t)vol->resolution * vol->resolution * vol->resolution);
                            vol->order.push_back(loc);
                        }
                        IntegrateUnit(*vol, *slot, image, K, extrinsic, dist_mult.data());
                        n_int++;
                    }
        }
    }
    vol->block_updates += n_int;
    return n_int;
}

int64_t oracle_tsdf_num_blocks(oracle_tsdf* vol) { return (int64_t)vol->order.size(); }
int64_t oracle_tsdf_block_updates(oracle_tsdf* vol) { return vol->block_updates; }

/* Export in allocation order: keys[n,3], tsdf[n,res^3], weight[n,res^3], color[n,res^3,3] (f64) */
void oracle_tsdf_export(oracle_tsdf* vol, int32_t* keys, float* tsdf, float* weight,
                        double* color) {
    const size_t nv = (size_t)vol->resolution * vol->resolution * vol->resolution;
    for (size_t b = 0; b < vol->order.size(); ++b) {
        const Vec3i& k = vol->order[b];
        const Unit& u = *vol->units[k];
        if (keys) {
            keys[3 * b] = k.x;
            keys[3 * b + 1] = k.y;
            keys[3 * b + 2] = k.z;
        }
        for (size_t i = 0; i < nv; ++i) {
            if (tsdf) tsdf[b * nv + i] = u.voxels[i].tsdf;
            if (weight) weight[b * nv + i] = u.voxels[i].weight;
            if (color)
                for (int c = 0; c < 3; ++c) color[(b * nv + i) * 3 + c] = u.voxels[i].color[c];
        }
    }
}

/* ScalableTSDFVolume::ExtractTriangleMesh (Open3D 0.17.0 cpp/open3d/pipelines/integration/ScalableTSDFVolume.cpp, the
 * call of gs2mesh_utils/tsdf_utils.py:108), restated from knowledge of upstream -- PARITY UNPINNED like the rest of this
 * file.  Marching cubes after P. Bourke's polygonise: for every volume unit and every voxel (x, y, z) of it the cube of
 * the voxel and its +1 neighbours (`shift`; a corner outside the unit is looked up in the neighbouring unit, a missing
 * unit gives weight 0) is skipped when any corner has weight 0; corner i is inside when tsdf < 0; for every cut edge
 * (edge_table bit) the vertex is created once, keyed by the edge's global index (unit_index * resolution + (x, y, z) +
 * edge_shift, 4th component = axis):
 *     pt = half_voxel + voxel_length * edge_index.xyz;   pt[axis] += f0 * voxel_length / (f0 + f1)
 * with f0 = |double(f[edge_to_vert[e][0]])|, f1 likewise, colour (f1 c0 + f0 c1) / (f0 + f1), c = voxel colour / 255; the
 * triangles of tri_table[cube_index] are pushed as (e[i], e[i + 2], e[i + 1]).  Upstream walks its unordered_map of units
 * (implementation-defined order): the mesh is defined up to the numbering of vertices and the order of triangles; here the
 * units are walked in allocation order.  `tri_table` [256][16] (-1 terminated rows) is passed in by the caller (the table
 * tools/mc_classic_table.py holds and verifies = MarchingCubesConst.h's); edge_table is derived from it (the edges a row
 * uses are exactly the cut edges of the case -- one of the properties that script checks).
 * Returns the number of triangles; *n_vertices = vertices; *n_zero = vertices that lie exactly on a corner of their edge
 * (tsdf == 0 at one end: several edges then share one position -- upstream keeps them apart). */
int64_t oracle_tsdf_extract_mesh(oracle_tsdf* vol, const int8_t* tri_table, int64_t* n_vertices, int64_t* n_zero) {
    static const int shift[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int edge_shift[12][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1}, {0, 0, 1, 0}, {1, 0, 1, 1},
                                          {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2}};
    static const int edge_to_vert[12][2] = {{0, 1}, {1, 2}, {3, 2}, {0, 3}, {4, 5}, {5, 6}, {7, 6}, {4, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    int edge_table[256];
    for (int c = 0; c < 256; ++c) {
        edge_table[c] = 0;
        for (int i = 0; i < 16 && tri_table[c * 16 + i] >= 0; ++i) edge_table[c] |= 1 << tri_table[c * 16 + i];
    }
    struct Vec4i {
        int v[4];
        bool operator==(const Vec4i& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    };
    struct Vec4iHash {
        size_t operator()(const Vec4i& k) const {
            size_t seed = 0;
            for (int i = 0; i < 4; ++i) seed ^= std::hash<int>()(k.v[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
            return seed;
        }
    };
    std::unordered_map<Vec4i, int, Vec4iHash> edgeindex_to_vertexindex;
    vol->mesh_vertices.clear();
    vol->mesh_colors.clear();
    vol->mesh_triangles.clear();
    vol->mesh_edge_index.clear();
    vol->mesh_zero_offset_vertices = 0;
    const int res = vol->resolution;
    const double half_voxel_length = vol->voxel_length * 0.5;
    int edge_to_index[12];
    for (const Vec3i& index0 : vol->order) {
        const Unit& volume0 = *vol->units[index0];
        for (int x = 0; x < res; x++)
            for (int y = 0; y < res; y++)
                for (int z = 0; z < res; z++) {
                    int cube_index = 0;
                    float w[8], f[8];
                    double c[8][3];
                    for (int i = 0; i < 8; i++) {
                        Vec3i index1 = index0;
                        int idx1[3] = {x + shift[i][0], y + shift[i][1], z + shift[i][2]};
                        const TSDFVoxel* vox = nullptr;
                        if (idx1[0] < res && idx1[1] < res && idx1[2] < res) {
                            vox = &volume0.voxels[IndexOf(idx1[0], idx1[1], idx1[2], res)];
                        } else {
                            int* i1 = &index1.x;
                            for (int j = 0; j < 3; j++)
                                if (idx1[j] >= res) {
                                    idx1[j] -= res;
                                    i1[j] += 1;
                                }
                            auto it = vol->units.find(index1);
                            if (it != vol->units.end()) vox = &it->second->voxels[IndexOf(idx1[0], idx1[1], idx1[2], res)];
                        }
                        if (vox) {
                            w[i] = vox->weight;
                            f[i] = vox->tsdf;
                            for (int k = 0; k < 3; ++k) c[i][k] = vol->color_type == 1 ? vox->color[k] / 255.0 : 0.0;
                        } else {
                            w[i] = 0.0f;
                            f[i] = 0.0f;
                        }
                        if (w[i] == 0.0f) {
                            cube_index = 0;
                            break;
                        } else if (f[i] < 0.0f) {
                            cube_index |= (1 << i);
                        }
                    }
                    if (cube_index == 0 || cube_index == 255) continue;
                    for (int i = 0; i < 12; i++) {
                        if (!(edge_table[cube_index] & (1 << i))) continue;
                        Vec4i edge_index;
                        edge_index.v[0] = index0.x * res + x + edge_shift[i][0];
                        edge_index.v[1] = index0.y * res + y + edge_shift[i][1];
                        edge_index.v[2] = index0.z * res + z + edge_shift[i][2];
                        edge_index.v[3] = edge_shift[i][3];
                        auto found = edgeindex_to_vertexindex.find(edge_index);
                        if (found == edgeindex_to_vertexindex.end()) {
                            const int vi = (int)(vol->mesh_vertices.size() / 3);
                            edge_to_index[i] = vi;
                            edgeindex_to_vertexindex[edge_index] = vi;
                            double pt[3] = {half_voxel_length + vol->voxel_length * edge_index.v[0],
                                            half_voxel_length + vol->voxel_length * edge_index.v[1],
                                            half_voxel_length + vol->voxel_length * edge_index.v[2]};
                            const double f0 = std::abs((double)f[edge_to_vert[i][0]]);
                            const double f1 = std::abs((double)f[edge_to_vert[i][1]]);
                            pt[edge_index.v[3]] += f0 * vol->voxel_length / (f0 + f1);
                            if (f0 == 0.0 || f1 == 0.0) vol->mesh_zero_offset_vertices++;
                            for (int k = 0; k < 4; ++k) vol->mesh_edge_index.push_back(edge_index.v[k]);
                            for (int k = 0; k < 3; ++k) vol->mesh_vertices.push_back(pt[k]);
                            const double* c0 = c[edge_to_vert[i][0]];
                            const double* c1 = c[edge_to_vert[i][1]];
                            for (int k = 0; k < 3; ++k) vol->mesh_colors.push_back((f1 * c0[k] + f0 * c1[k]) / (f0 + f1));
                        } else {
                            edge_to_index[i] = found->second;
                        }
                    }
                    for (int i = 0; i < 16 && tri_table[cube_index * 16 + i] != -1; i += 3) {
                        vol->mesh_triangles.push_back(edge_to_index[tri_table[cube_index * 16 + i]]);
                        vol->mesh_triangles.push_back(edge_to_index[tri_table[cube_index * 16 + i + 2]]);
                        vol->mesh_triangles.push_back(edge_to_index[tri_table[cube_index * 16 + i + 1]]);
                    }
                }
    }
    if (n_vertices) *n_vertices = (int64_t)(vol->mesh_vertices.size() / 3);
    if (n_zero) *n_zero = vol->mesh_zero_offset_vertices;
    return (int64_t)(vol->mesh_triangles.size() / 3);
}

/* vertices [nv,3] f64, colors [nv,3] f64 (zeros without colour), triangles [nt,3] i32, edge_index [nv,4] i32 (the key
 * upstream's edgeindex_to_vertexindex map holds for the vertex) of the last extraction */
void oracle_tsdf_mesh_copy(oracle_tsdf* vol, double* vertices, double* colors, int32_t* triangles, int32_t* edge_index) {
    if (vertices) memcpy(vertices, vol->mesh_vertices.data(), vol->mesh_vertices.size() * sizeof(double));
    if (colors) memcpy(colors, vol->mesh_colors.data(), vol->mesh_colors.size() * sizeof(double));
    if (triangles) memcpy(triangles, vol->mesh_triangles.data(), vol->mesh_triangles.size() * sizeof(int32_t));
    if (edge_index) memcpy(edge_index, vol->mesh_edge_index.data(), vol->mesh_edge_index.size() * sizeof(int32_t));
}

/* Test aid: set the state of n volume units directly (allocating them): keys [n,3], tsdf / weight [n,res^3] in IndexOf
 * order (x * res^2 + y * res + z), color [n,res^3,3] f64 (may be null).  No upstream counterpart. */
void oracle_tsdf_import(oracle_tsdf* vol, int64_t n, const int32_t* keys, const float* tsdf, const float* weight,
                        const double* color) {
    const size_t nv = (size_t)vol->resolution * vol->resolution * vol->resolution;
    for (int64_t b = 0; b < n; ++b) {
        const Vec3i k{keys[3 * b], keys[3 * b + 1], keys[3 * b + 2]};
        auto it = vol->units.find(k);
        if (it == vol->units.end()) {
            std::unique_ptr<Unit> u(new Unit());
            u->origin[0] = k.x * vol->unit_length;
            u->origin[1] = k.y * vol->unit_length;
            u->origin[2] = k.z * vol->unit_length;
            u->voxels.resize(nv);
            it = vol->units.emplace(k, std::move(u)).first;
            vol->order.push_back(k);
        }
        for (size_t i = 0; i < nv; ++i) {
            TSDFVoxel& v = it->second->voxels[i];
            v.tsdf = tsdf[b * nv + i];
            v.weight = weight[b * nv + i];
            for (int c = 0; c < 3; ++c) v.color[c] = color ? color[(b * nv + i) * 3 + c] : 0.0;
        }
    }
}

} /* extern "C" */
