"""Mesh extraction: GPU marching cubes + host welding / cleaning / PLY, on both back-ends.
Open3D is absent (parity unpinned), so the checks are analytic: an injected plane field must come out
as the exact plane, a fused sphere as a consistently oriented 2-manifold on the sphere, and the
TSDF.run -> save_mesh -> clean_mesh sequence of run_single.py:161-174 must produce the reference's files."""
import os
from argparse import Namespace

import numpy as np

from gs2mesh_amd import synthetic
from gs2mesh_amd.integration import (PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume,
                                     TSDFVolumeColorType)
from gs2mesh_amd.mesh import read_triangle_mesh


def edge_counts(tri):
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=0).astype(np.int64)
    e.sort(axis=1)
    key = e[:, 0] * (tri.max() + 2) + e[:, 1]
    _, cnt = np.unique(key, return_counts=True)
    return cnt


def test_injected_plane_field_is_extracted_exactly(backend):
    be = backend
    vl, z0 = 1.0 / 32, 0.4321
    vol = ScalableTSDFVolume(vl, 4 * vl, TSDFVolumeColorType.RGB8, max_blocks=64, lib=be.lib)
    keys = np.array([[bx, by, bz] for bx in range(2) for by in range(2) for bz in range(2)], np.int32)
    n = len(keys)
    wsum = np.zeros((n, 4096), np.float32)
    w = np.ones((n, 4096), np.float32)
    rgb = np.zeros((n, 3, 4096), np.uint32)
    x, y, z = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    vidx = ((z >> 2) * 16 + (x >> 2) * 4 + (y >> 2)) * 64 + (z & 3) * 16 + (x & 3) * 4 + (y & 3)   # device layout
    for b, k in enumerate(keys):
        zc = (k[2] * 16 + z + 0.5) * vl
        f = np.clip((zc - z0) / (4 * vl), -1, 1).astype(np.float32)        # positive above the plane
        wsum[b, vidx.reshape(-1)] = f.reshape(-1)
        rgb[b, 0, :] = 255
        rgb[b, 1, vidx.reshape(-1)] = ((k[0] * 16 + x) * 4).reshape(-1)
    buf = np.concatenate([wsum[:, None], w[:, None], rgb.astype(np.float32)], axis=1)     # [n, 5, 4096] sum form
    vol.unpack_sum(be.dev(keys), be.dev(np.ascontiguousarray(buf)))
    mesh = vol.extract_triangle_mesh()
    assert mesh.triangles.shape[0] == 2 * 31 * 31          # 31 x 31 cubes across 2 x 2 blocks, one quad each
    assert mesh.vertices.shape[0] == 32 * 32
    np.testing.assert_allclose(mesh.vertices[:, 2], z0, atol=1e-6)
    mesh.compute_vertex_normals()
    np.testing.assert_allclose(mesh.triangle_normals, np.tile([0, 0, 1.0], (mesh.triangles.shape[0], 1)), atol=1e-9)
    assert (edge_counts(mesh.triangles) <= 2).all()
    np.testing.assert_allclose(mesh.vertex_colors[:, 0], 1.0, atol=1e-12)
    gx = np.round(mesh.vertices[:, 0] / vl - 0.5)
    np.testing.assert_allclose(mesh.vertex_colors[:, 1], gx * 4 / 255.0, atol=1e-12)


def fused_sphere(be, n_views=10, W=200, H=150, f=210.0, r=0.6, voxel=2.0 / 96, trunc=0.09):
    vol = ScalableTSDFVolume(voxel, trunc, max_blocks=4096, lib=be.lib)
    col = synthetic.color_pattern(W, H)
    intr = PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0)
    for p in synthetic.ring_poses(n_views, 3.5):
        E = np.eye(4)
        E[:3] = p
        d = synthetic.sphere_depth(p, W, H, f, f, W / 2.0, H / 2.0, r)
        vol.integrate(RGBDImage(be.dev(col), be.dev(d)), intr, E)
    return vol, r, voxel


def test_fused_sphere_mesh_is_an_oriented_manifold_on_the_sphere(backend):
    vol, r, voxel = fused_sphere(backend)
    mesh = vol.extract_triangle_mesh()
    nt = mesh.triangles.shape[0]
    assert nt > 5000
    rad = np.linalg.norm(mesh.vertices, axis=1)
    assert np.abs(rad - r).max() < 1.5 * voxel and np.abs(rad - r).mean() < 0.35 * voxel
    mesh.compute_vertex_normals()
    cen = mesh.vertices[mesh.triangles].mean(axis=1)
    outward = (mesh.triangle_normals * cen).sum(1) / np.linalg.norm(cen, axis=1)
    assert (outward > 0.3).mean() > 0.999                   # normals point to the outside (positive tsdf)
    cnt = edge_counts(mesh.triangles)
    assert (cnt <= 2).all()                                 # 2-manifold: no edge shared by 3+ triangles
    assert (cnt == 1).mean() < 0.05                         # open only where the ring of cameras did not see
    assert abs(mesh.vertices.shape[0] - nt / 2) < 0.06 * nt # Euler: V ~ T/2 for a (nearly) closed surface
    assert 0 <= mesh.vertex_colors.min() and mesh.vertex_colors.max() <= 1
    # deterministic
    m2 = vol.extract_triangle_mesh()
    np.testing.assert_array_equal(m2.vertices, mesh.vertices)
    np.testing.assert_array_equal(m2.triangles, mesh.triangles)


def canonical_mesh(vertices, colors, triangles, vertex_key=None):
    """Numbering-independent form: vertices sorted lexicographically (by position, or by the columns of ``vertex_key``),
    every triangle rotated so that its smallest index comes first (winding kept), triangles sorted."""
    k = vertices if vertex_key is None else np.asarray(vertex_key)
    order = np.lexsort(tuple(k[:, c] for c in range(k.shape[1] - 1, -1, -1)))
    rank = np.empty(order.size, np.int64)
    rank[order] = np.arange(order.size)
    t = rank[np.asarray(triangles, np.int64)]
    k = t.argmin(axis=1)
    t = np.stack([t[np.arange(len(t)), (k + j) % 3] for j in range(3)], axis=1)
    t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    return vertices[order], (None if colors is None else colors[order]), t


def test_fused_sphere_mesh_equals_the_restated_open3d_extraction(backend):
    """SURVEY 8f-2 against an oracle: the same frames fused by the restated Open3D 0.17 volume (oracle/tsdf_oracle.cpp,
    bit-identical tsdf / weight: tests/test_tsdf_parity.py) and extracted by its restated ExtractTriangleMesh (edge-keyed
    vertices, classic table, (i, i + 2, i + 1) winding).  Open3D's vertex numbering follows its hash map's iteration order, so
    the meshes are compared in a numbering-independent form: vertex positions bit for bit (float64), triangles exactly,
    colours to 1e-12 (the oracle keeps Open3D's running mean in double, the device integer sums)."""
    import sys
    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import mc_classic_table
    be = backend
    W, H, f, r, voxel, trunc = 200, 150, 210.0, 0.6, 2.0 / 96, 0.09
    vol = ScalableTSDFVolume(voxel, trunc, max_blocks=4096, lib=be.lib)
    ref = oracle.ScalableTSDFVolume(voxel, trunc, 1)
    intr = PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0)
    for k, p in enumerate(synthetic.ring_poses(6, 3.5)):
        E = np.eye(4)
        E[:3] = p
        d = synthetic.sphere_depth(p, W, H, f, f, W / 2.0, H / 2.0, r)
        col = np.roll(synthetic.color_pattern(W, H), 11 * k, axis=1)
        vol.integrate(RGBDImage(be.dev(col), be.dev(d)), intr, E)
        ref.integrate(oracle.ScalableTSDFVolume.convert_depth(d, 1.0, float("inf")), col, W, H, f, f, W / 2.0, H / 2.0, E)
    mesh = vol.extract_triangle_mesh()
    om = ref.extract_triangle_mesh(mc_classic_table.T)
    assert om["triangles"].shape[0] > 5000
    # position welding (ours) == edge-keyed vertices (Open3D) unless a corner value is exactly 0
    assert om["zero_offset_vertices"] == 0
    assert mesh.triangles.shape == om["triangles"].shape and mesh.vertices.shape == om["vertices"].shape
    v, c, t = canonical_mesh(mesh.vertices, mesh.vertex_colors, mesh.triangles)
    ov, oc, ot = canonical_mesh(om["vertices"], om["colors"], om["triangles"])
    np.testing.assert_array_equal(v, ov)
    np.testing.assert_array_equal(t, ot)
    np.testing.assert_allclose(c, oc, rtol=0, atol=1e-12)


def test_exact_zero_tsdf_values_keep_open3ds_vertices_apart(backend):
    """A diagonal plane through voxel centres: tsdf == 0 exactly on those voxels, so the cut edges that end there (up to
    three, one per axis) put their vertices at ONE position.  Open3D keys vertices by edge and keeps them apart; the device
    path emits every vertex with its edge key (gs2m_tsdf_extract_indexed) and welds by it: same vertex count, same triangles
    as the restated ExtractTriangleMesh -- a weld by position would merge them."""
    import sys
    import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import mc_classic_table
    from gs2mesh_amd.mesh import TriangleMesh
    be = backend
    vl = 1.0 / 32
    vol = ScalableTSDFVolume(vl, 4 * vl, TSDFVolumeColorType.RGB8, max_blocks=64, lib=be.lib)
    ref = oracle.ScalableTSDFVolume(vl, 4 * vl, 1)
    keys = np.array([[bx, by, bz] for bx in range(2) for by in range(2) for bz in range(2)], np.int32)
    n = len(keys)
    x, y, z = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
    vidx = ((z >> 2) * 16 + (x >> 2) * 4 + (y >> 2)) * 64 + (z & 3) * 16 + (x & 3) * 4 + (y & 3)   # device layout
    wsum = np.zeros((n, 4096), np.float32)
    w = np.ones((n, 4096), np.float32)
    rgb = np.zeros((n, 3, 4096), np.uint32)
    o_tsdf = np.zeros((n, 4096), np.float32)
    o_col = np.zeros((n, 4096, 3), np.float64)
    for b, k in enumerate(keys):
        g = (k[0] * 16 + x) + (k[1] * 16 + y) + (k[2] * 16 + z)            # integer: exact in fp32
        f = np.clip((g - 40).astype(np.float32) / np.float32(8.0), -1, 1)  # == 0 exactly on the plane x + y + z = 40
        wsum[b, vidx.reshape(-1)] = f.reshape(-1)
        rgb[b, 0, vidx.reshape(-1)] = ((k[0] * 16 + x) * 4).reshape(-1)
        rgb[b, 1, :] = 128
        o_tsdf[b] = f.reshape(-1)                                          # (x, y, z) index order
        o_col[b, :, 0] = ((k[0] * 16 + x) * 4).reshape(-1)
        o_col[b, :, 1] = 128
    buf = np.concatenate([wsum[:, None], w[:, None], rgb.astype(np.float32)], axis=1)
    vol.unpack_sum(be.dev(keys), be.dev(np.ascontiguousarray(buf)))
    ref.import_state(keys, o_tsdf, w, o_col)
    om = ref.extract_triangle_mesh(mc_classic_table.T)
    assert om["zero_offset_vertices"] > 100
    mesh = vol.extract_triangle_mesh()
    assert mesh.vertices.shape == om["vertices"].shape and mesh.triangles.shape == om["triangles"].shape
    # positions alone do not identify the vertices here
    assert len(np.unique(om["vertices"], axis=0)) < len(om["vertices"])
    # numbering-independent comparison: vertices ordered by their edge key on both sides
    ov, oc, ot = canonical_mesh(om["vertices"], om["colors"], om["triangles"], vertex_key=om["edge_index"])
    v, c, t = canonical_mesh(mesh.vertices, mesh.vertex_colors, mesh.triangles, vertex_key=mesh.edge_index)
    np.testing.assert_array_equal(v, ov)
    np.testing.assert_array_equal(t, ot)
    np.testing.assert_allclose(c, oc, rtol=0, atol=1e-12)
    # and a weld by position would have lost vertices
    soup = mesh.vertices[mesh.triangles]
    assert TriangleMesh.from_triangle_soup(soup).vertices.shape[0] < mesh.vertices.shape[0]


class FakeRenderer:
    def __init__(self, root, poses, W, H, f, baseline):
        self.output_dir_root, self.baseline, self.left_cameras = root, baseline, []
        for p in poses:
            E = np.eye(4)
            E[:3] = p
            self.left_cameras.append(dict(width=W, height=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, extrinsic=np.linalg.inv(E)))

    def __len__(self):
        return len(self.left_cameras)

    def render_folder_name(self, i):
        return os.path.join(self.output_dir_root, f"{i:03}")


def test_tsdf_run_save_and_clean_mesh_like_run_single(backend, tmp_path):
    from gs2mesh_amd.tsdf_utils import TSDF
    from test_pipeline_classes import make_args
    W, H, f = 200, 150, 210.0
    poses = synthetic.ring_poses(8, 3.5)
    ren = FakeRenderer(str(tmp_path), poses, W, H, f, 0.245)
    col = synthetic.color_pattern(W, H)

    def frame(i):
        d = synthetic.sphere_depth(poses[i], W, H, f, f, W / 2.0, H / 2.0, 0.6)
        d[:12, :12] = 3.0                                       # a small floating blob: its own component
        return dict(image=col, depth=d)

    args = make_args(TSDF_use_occlusion_mask=False, TSDF_voxel=12, TSDF_sdf_trunc=0.09, TSDF_cleaning_threshold=2000,
                     TSDF_min_depth_baselines=4, TSDF_max_depth_baselines=20)
    t = TSDF(ren, Namespace(model_name="DLNR_Middlebury"), args, "out", frame_source=frame, max_blocks=4096, lib=backend.lib)
    t.run()
    assert t.mesh.triangles.shape[0] > 3000 and t.mesh.has_vertex_normals()
    t.save_mesh()
    t.clean_mesh()
    raw = read_triangle_mesh(os.path.join(str(tmp_path), "out_mesh.ply"))
    clean = read_triangle_mesh(os.path.join(str(tmp_path), "out_cleaned_mesh.ply"))
    assert raw.triangles.shape[0] == t.mesh.triangles.shape[0]
    np.testing.assert_allclose(raw.vertices, t.mesh.vertices, atol=0)
    labels, counts, areas = t.mesh.cluster_connected_triangles()
    assert len(counts) >= 2 and counts.max() > 2000 and counts.min() < 2000
    assert clean.triangles.shape[0] == counts[counts >= 2000].sum()
    assert clean.vertices.shape[0] < raw.vertices.shape[0]
    assert t.clean_mesh is not None and not callable(t.clean_mesh)       # the method rebinds itself (tsdf_utils.py:138)


# ---- device-side welding and clustering (round 5) against their host statements ------------------------------------------------
def extract_soup(vol):
    """the un-welded output of gs2m_tsdf_extract_indexed (what round 4 copied to the host and welded there)"""
    import ctypes as C
    from gs2mesh_amd import _lib
    from gs2mesh_amd.rasterizer import _ptr, _stream_of
    n = C.c_int64(0)
    _lib.check(vol._lib.gs2m_tsdf_extract_count(vol._h, C.c_void_p(0), C.byref(n)), vol._lib)
    nt = int(n.value)
    verts = _lib.MEMORY.zeros((nt, 3, 3), np.float64, vol.device)
    cols = _lib.MEMORY.zeros((nt, 3, 3), np.float64, vol.device)
    eidx = _lib.MEMORY.zeros((nt, 3, 4), np.int32, vol.device)
    got = C.c_int64(0)
    _lib.check(vol._lib.gs2m_tsdf_extract_indexed(vol._h, _stream_of(verts, None), nt, _ptr(verts), _ptr(cols), _ptr(eidx), C.byref(got)),
               vol._lib)
    vol.status()
    return _lib.MEMORY.download(verts), _lib.MEMORY.download(cols), _lib.MEMORY.download(eidx)


def scipy_clusters(mesh):
    """the host statement of ClusterConnectedTriangles (what gs2mesh_amd.mesh did until round 4): scipy connected components of
    the triangle graph whose arcs are shared edges"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    t = mesh.triangles.astype(np.int64)
    n = t.shape[0]
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    owner = np.tile(np.arange(n), 3)
    key = e[:, 0] * (int(mesh.vertices.shape[0]) + 1) + e[:, 1]
    order = np.argsort(key, kind="stable")
    key, owner = key[order], owner[order]
    same = key[1:] == key[:-1]
    g = coo_matrix((np.ones(int(same.sum()), np.int8), (owner[:-1][same], owner[1:][same])), shape=(n, n))
    n_comp, labels = connected_components(g, directed=False)
    return labels.astype(np.int32), np.bincount(labels, minlength=n_comp).astype(np.int64)


def test_device_weld_equals_the_host_weld_of_the_soup(backend):
    """gs2m_tsdf_extract_mesh == TriangleMesh.from_triangle_soup(gs2m_tsdf_extract_indexed): same vertices in the same (first
    appearance) order, same triangle indices, same colours and cut edges -- on the sphere and with a second, detached component."""
    from gs2mesh_amd.mesh import TriangleMesh
    vol, r, voxel = fused_sphere(backend, n_views=6)
    for case in range(2):
        if case == 1:   # a floating blob in front of the first camera: a second component, other blocks
            W, H, f = 200, 150, 210.0
            p = synthetic.ring_poses(6, 3.5)[0]
            E = np.eye(4)
            E[:3] = p
            d = np.zeros((H, W), np.float32)
            d[60:90, 80:120] = 1.7
            vol.integrate(RGBDImage(backend.dev(synthetic.color_pattern(W, H)), backend.dev(d)), PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0), E)
        mesh = vol.extract_triangle_mesh()
        v, c, e = extract_soup(vol)
        ref = TriangleMesh.from_triangle_soup(v, c, edge_index=e)
        assert mesh.triangles.shape[0] == v.shape[0] > 3000
        np.testing.assert_array_equal(mesh.triangles, ref.triangles)
        np.testing.assert_array_equal(mesh.vertices, ref.vertices)
        np.testing.assert_array_equal(mesh.vertex_colors, ref.vertex_colors)
        np.testing.assert_array_equal(mesh.edge_index, ref.edge_index)


def test_device_clustering_equals_scipy_connected_components(backend):
    """gs2m_mesh_cluster: the labels (clusters numbered by their first triangle) and sizes scipy gives, on the extraction's own
    device copy of the triangles, on an uploaded copy after a mask removed triangles, and on hand-made meshes (a bow tie: two
    triangles sharing ONE vertex are not connected; a fan sharing edges is; an isolated triangle)."""
    from gs2mesh_amd.mesh import TriangleMesh
    vol, r, voxel = fused_sphere(backend, n_views=6)
    W, H, f = 200, 150, 210.0
    p = synthetic.ring_poses(6, 3.5)[0]
    E = np.eye(4)
    E[:3] = p
    d = np.zeros((H, W), np.float32)
    d[60:90, 80:120] = 1.7
    d[10:20, 10:30] = 2.2
    vol.integrate(RGBDImage(backend.dev(synthetic.color_pattern(W, H)), backend.dev(d)), PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0), E)
    mesh = vol.extract_triangle_mesh()
    labels, counts, areas = mesh.cluster_connected_triangles()
    ref_labels, ref_counts = scipy_clusters(mesh)
    assert len(counts) >= 3
    np.testing.assert_array_equal(labels, ref_labels)
    np.testing.assert_array_equal(counts, ref_counts)
    assert np.isclose(areas.sum(), 0.5 * np.linalg.norm(np.cross(*(mesh.vertices[mesh.triangles[:, k]] - mesh.vertices[mesh.triangles[:, 0]]
                                                                  for k in (1, 2))), axis=1).sum())
    # after an edit the device copy is stale: the triangles are uploaded again
    mesh.remove_triangles_by_mask(np.arange(mesh.triangles.shape[0]) % 7 == 0)
    l2, c2, _ = mesh.cluster_connected_triangles(lib=backend.lib)
    r2, rc2 = scipy_clusters(mesh)
    np.testing.assert_array_equal(l2, r2)
    np.testing.assert_array_equal(c2, rc2)
    # hand-made: bow tie (0-1-2, 2-3-4), a fan of three around vertex 5 sharing edges, an isolated triangle
    tri = np.array([[0, 1, 2], [2, 3, 4], [5, 6, 7], [5, 7, 8], [5, 8, 9], [10, 11, 12]], np.int32)
    m = TriangleMesh(np.random.default_rng(2).random((13, 3)), tri)
    l3, c3, a3 = m.cluster_connected_triangles(lib=backend.lib)
    assert l3.tolist() == [0, 1, 2, 2, 2, 3] and c3.tolist() == [1, 1, 3, 1] and len(a3) == 4
    assert TriangleMesh().cluster_connected_triangles(lib=backend.lib)[1].shape == (0,)
