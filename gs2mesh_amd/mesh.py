"""Triangle-mesh container with the Open3D calls ``gs2mesh_utils/tsdf_utils.py:108-142`` makes (``scale``,
``compute_vertex_normals``, ``cluster_connected_triangles``, ``remove_triangles_by_mask``, ``remove_unreferenced_vertices``,
``o3d.io.write_triangle_mesh``).  Extraction, vertex welding (``gs2m_tsdf_extract_mesh``) and the connected components
(``gs2m_mesh_cluster``) run on the GPU; what is left here is the container, O(n) numpy bookkeeping on the final arrays
(scale, normals, mask / compaction) and the PLY writer.  ``from_triangle_soup`` is the host statement of the weld that the
device path is tested against."""
from __future__ import annotations

import numpy as np


class TriangleMesh:
    def __init__(self, vertices=None, triangles=None, vertex_colors=None):
        self.vertices = np.zeros((0, 3), np.float64) if vertices is None else np.asarray(vertices, np.float64)
        self.triangles = np.zeros((0, 3), np.int32) if triangles is None else np.asarray(triangles, np.int32)
        self.vertex_colors = (np.zeros((0, 3), np.float64) if vertex_colors is None
                              else np.asarray(vertex_colors, np.float64))
        self.vertex_normals = np.zeros((0, 3), np.float64)
        self.triangle_normals = np.zeros((0, 3), np.float64)
        self.edge_index = np.zeros((0, 4), np.int32)   # per vertex, after extraction: the cut edge (Open3D's vertex key)

    # ---- construction from the un-welded GPU output ------------------------------------------
    @staticmethod
    def from_triangle_soup(verts, cols=None, edge_index=None):
        """verts [n,3,3] float64.  With ``edge_index`` [n,3,4] int32 (the cut edge of every emitted vertex: global voxel
        index of its lower corner + axis) the vertices are welded by that key, which is Open3D's own vertex identity
        (ExtractTriangleMesh's edge -> vertex map); without it by position (vertices shared by neighbouring triangles are
        bit-identical) -- the same mesh unless a tsdf value is exactly 0, where up to three edges share one position."""
        v = np.ascontiguousarray(verts, np.float64).reshape(-1, 3)
        if v.shape[0] == 0:
            return TriangleMesh()
        if edge_index is not None:
            key = np.ascontiguousarray(edge_index, np.int32).reshape(-1, 4).view([("", np.int32)] * 4).reshape(-1)
        else:
            key = v.view([("", np.float64)] * 3).reshape(-1)
        _, first, inv = np.unique(key, return_index=True, return_inverse=True)
        # keep first-appearance order (deterministic, independent of float ordering)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        tri = rank[inv].reshape(-1, 3).astype(np.int32)
        sel = first[order]
        m = TriangleMesh(v[sel], tri, None if cols is None else np.asarray(cols, np.float64).reshape(-1, 3)[sel])
        if edge_index is not None:
            m.edge_index = np.ascontiguousarray(edge_index, np.int32).reshape(-1, 4)[sel]   # [n_vertices, 4]: Open3D's vertex keys
        return m

    # ---- Open3D API subset -----------------------------------------------------------------
    def scale(self, s, center=(0, 0, 0)):
        c = np.asarray(center, np.float64)
        self.vertices = (self.vertices - c) * float(s) + c
        return self

    def compute_triangle_normals(self):
        v = self.vertices
        t = self.triangles
        n = np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]])
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        self.triangle_normals = n / np.where(ln > 0, ln, 1.0)
        return self

    def compute_vertex_normals(self):
        """Open3D ComputeVertexNormals: sum of the (normalised) normals of the adjacent triangles, normalised."""
        self.compute_triangle_normals()
        vn = np.zeros_like(self.vertices)
        for k in range(3):
            np.add.at(vn, self.triangles[:, k], self.triangle_normals)
        ln = np.linalg.norm(vn, axis=1, keepdims=True)
        self.vertex_normals = vn / np.where(ln > 0, ln, 1.0)
        return self

    # ---- device attachment -------------------------------------------------------------------
    def attach_device_triangles(self, tri_dev, lib, device):
        """Keep the device copy of ``triangles`` the extraction left behind (valid while ``self.triangles`` is that array)."""
        self._dev = (tri_dev, lib, int(device), self.triangles)

    def __deepcopy__(self, memo):
        import copy
        m = TriangleMesh(self.vertices.copy(), self.triangles.copy(), self.vertex_colors.copy())
        m.vertex_normals, m.triangle_normals, m.edge_index = (copy.deepcopy(self.vertex_normals, memo),
                                                              copy.deepcopy(self.triangle_normals, memo), self.edge_index.copy())
        return m                                   # the device attachment belongs to the original

    def cluster_connected_triangles(self, lib=None, device=None):
        """-> (triangle_clusters[n_tri], cluster_n_triangles[n_clusters], cluster_area[n_clusters]); triangles are connected
        when they share an edge (Open3D ClusterConnectedTriangles; clusters numbered by their first triangle).  The components
        are found on the GPU (``gs2m_mesh_cluster``: union-find over an edge hash table) -- on the triangle indices the
        extraction left on the device when this mesh still is that mesh, else on an upload of ``triangles`` (12 B each);
        the areas are one numpy bincount over the labels."""
        import ctypes as C
        from . import _lib
        n = int(self.triangles.shape[0])
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.float64)
        dev = getattr(self, "_dev", None)
        if dev is not None and dev[3] is self.triangles and lib in (None, dev[1]):
            tri_dev, lib, device = dev[0], dev[1], dev[2]
        else:
            lib = lib if lib is not None else _lib.get()
            import torch
            if device is None:      # the caller's current device, not device 0 (a rank whose GPU is not 0)
                device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            tri_dev = _lib.MEMORY.upload(np.ascontiguousarray(self.triangles, np.int32), torch.int32, device)
        labels_dev = _lib.MEMORY.zeros((n,), np.int32, device)
        count_dev = _lib.MEMORY.zeros((n,), np.int64, device)
        nc = C.c_int64(0)
        # torch's current stream: the zero fills above were queued on it (the NULL stream is not ordered against a
        # non-blocking stream)
        _lib.check(lib.gs2m_mesh_cluster(int(device), _lib.MEMORY.current_stream(device), n, _lib.MEMORY.ptr(tri_dev),
                                         _lib.MEMORY.ptr(labels_dev), _lib.MEMORY.ptr(count_dev), C.byref(nc)), lib)
        labels = np.asarray(_lib.MEMORY.download(labels_dev), np.int32)
        counts = np.asarray(_lib.MEMORY.download(count_dev), np.int64)[: int(nc.value)].copy()
        # triangle areas: component-wise cross product (np.cross / np.linalg.norm on 0.9 M rows were 60 of the 70 ms this call took
        # on C2 -- bench `cluster_ms` -- next to 0.9 ms of kernels)
        v, t = self.vertices, self.triangles
        p0 = v[t[:, 0]]
        a, b = v[t[:, 1]] - p0, v[t[:, 2]] - p0
        cx = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
        cy = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
        cz = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        area = 0.5 * np.sqrt(cx * cx + cy * cy + cz * cz)
        return labels, counts, np.bincount(labels, weights=area, minlength=int(nc.value))

    def remove_triangles_by_mask(self, mask):
        keep = ~np.asarray(mask, bool)
        self.triangles = self.triangles[keep]
        if self.triangle_normals.shape[0] == keep.shape[0]:
            self.triangle_normals = self.triangle_normals[keep]
        return self

    def remove_unreferenced_vertices(self):
        used = np.zeros(self.vertices.shape[0], bool)
        used[self.triangles.reshape(-1)] = True
        remap = np.cumsum(used) - 1
        self.triangles = remap[self.triangles].astype(np.int32)
        self.vertices = self.vertices[used]
        if self.vertex_colors.shape[0] == used.shape[0]:
            self.vertex_colors = self.vertex_colors[used]
        if self.vertex_normals.shape[0] == used.shape[0]:
            self.vertex_normals = self.vertex_normals[used]
        if self.edge_index.shape[0] == used.shape[0]:
            self.edge_index = self.edge_index[used]
        return self

    def has_vertex_normals(self):
        return self.vertex_normals.shape[0] == self.vertices.shape[0] and self.vertices.shape[0] > 0

    def has_vertex_colors(self):
        return self.vertex_colors.shape[0] == self.vertices.shape[0] and self.vertices.shape[0] > 0


def write_triangle_mesh(path, mesh: TriangleMesh):
    """Binary little-endian PLY in the layout ``o3d.io.write_triangle_mesh`` produces for a TriangleMesh:
    double x y z [double nx ny nz] [uchar red green blue], faces as ``list uchar uint vertex_indices``."""
    nv, nt = mesh.vertices.shape[0], mesh.triangles.shape[0]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    hdr = ["ply", "format binary_little_endian 1.0", "comment Created by gs2mesh_amd", f"element vertex {nv}",
           "property double x", "property double y", "property double z"]
    if mesh.has_vertex_normals():
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
        hdr += ["property double nx", "property double ny", "property double nz"]
    if mesh.has_vertex_colors():
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        hdr += ["property uchar red", "property uchar green", "property uchar blue"]
    hdr += [f"element face {nt}", "property list uchar uint vertex_indices", "end_header"]
    v = np.zeros(nv, dtype=fields)
    v["x"], v["y"], v["z"] = mesh.vertices[:, 0], mesh.vertices[:, 1], mesh.vertices[:, 2]
    if mesh.has_vertex_normals():
        v["nx"], v["ny"], v["nz"] = mesh.vertex_normals[:, 0], mesh.vertex_normals[:, 1], mesh.vertex_normals[:, 2]
    if mesh.has_vertex_colors():
        c = np.clip(np.floor(mesh.vertex_colors * 255.0), 0, 255).astype(np.uint8)   # Open3D: (uint8_t)(c*255) clamped
        v["red"], v["green"], v["blue"] = c[:, 0], c[:, 1], c[:, 2]
    f = np.zeros(nt, dtype=[("n", "u1"), ("i", "<u4", (3,))])
    f["n"] = 3
    f["i"] = mesh.triangles
    with open(path, "wb") as fh:
        fh.write(("\n".join(hdr) + "\n").encode())
        fh.write(v.tobytes())
        fh.write(f.tobytes())
    return True


def read_triangle_mesh(path) -> TriangleMesh:
    """Reader for the files ``write_triangle_mesh`` writes (used by the tests)."""
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"ply"
        props, nv, nt, cur = [], 0, 0, None
        while True:
            tok = fh.readline().strip().split()
            if tok[0] == b"element":
                cur = tok[1]
                if cur == b"vertex":
                    nv = int(tok[2])
                else:
                    nt = int(tok[2])
            elif tok[0] == b"property" and cur == b"vertex":
                props.append((tok[2].decode(), {"double": "<f8", "float": "<f4", "uchar": "u1"}[tok[1].decode()]))
            elif tok[0] == b"end_header":
                break
        v = np.frombuffer(fh.read(nv * np.dtype(props).itemsize), dtype=props, count=nv)
        f = np.frombuffer(fh.read(nt * 13), dtype=[("n", "u1"), ("i", "<u4", (3,))], count=nt)
    m = TriangleMesh(np.stack([v["x"], v["y"], v["z"]], 1), f["i"].astype(np.int32))
    names = [p[0] for p in props]
    if "nx" in names:
        m.vertex_normals = np.stack([v["nx"], v["ny"], v["nz"]], 1)
    if "red" in names:
        m.vertex_colors = np.stack([v["red"], v["green"], v["blue"]], 1) / 255.0
    return m
