"""Inference-time Gaussian store: the part of GS/scene/gaussian_model.py the hot path touches.

  * ``load_ply``  (gaussian_model.py:215-256; attribute order :177-189):
    reads ``point_cloud/iteration_N/point_cloud.ply`` without ``plyfile`` -- one ``vertex`` element of
    float32 properties ``x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3``;
    ``f_rest`` is channel-major on disk (j = c*15 + k-1) and coefficient-major in memory [P,15,3].
  * getters with the reference's activations (gaussian_model.py:95-115) for the operator-level API;
  * ``raw()``: the pre-activation tensors for the fused pipeline-level API (no ``torch.cat`` of
    dc/rest per call, no separate exp / normalize / sigmoid passes).
Importing this module does not need ``simple_knn`` (the reference imports it at module scope,
gaussian_model.py:20) nor a GPU.
"""
from __future__ import annotations

import numpy as np
import torch


def read_gaussian_ply(path):
    """-> dict of float32 numpy arrays: xyz[P,3], f_dc[P,3], f_rest[P,R], opacity[P,1], scale[P,3], rot[P,4]."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.strip().split()
            if not tok:
                continue
            if tok[0] == b"format":
                fmt = tok[1].decode()
            elif tok[0] == b"element":
                in_vertex = tok[1] == b"vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == b"property" and in_vertex:
                if tok[1] == b"list":
                    raise ValueError("list property in vertex element")
                props.append((tok[2].decode(), tok[1].decode()))
            elif tok[0] == b"end_header":
                break
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        ch = "<" if fmt == "binary_little_endian" else ">"
        tmap = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
                "int": "i4", "int32": "i4", "uint": "u4", "short": "i2", "ushort": "u2", "char": "i1"}
        dt = np.dtype([(name, ch + tmap[t]) for name, t in props])
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    col = lambda name: np.asarray(data[name], np.float32)
    names = [p[0] for p in props]
    srt = lambda pre: sorted([k for k in names if k.startswith(pre)], key=lambda x: int(x.split("_")[-1]))
    rest = srt("f_rest_")
    return dict(
        xyz=np.stack([col("x"), col("y"), col("z")], 1),
        f_dc=np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1),
        f_rest=np.stack([col(k) for k in rest], 1) if rest else np.zeros((n, 0), np.float32),
        opacity=col("opacity")[:, None],
        scale=np.stack([col(k) for k in srt("scale_")], 1),
        rot=np.stack([col(k) for k in srt("rot")], 1))


def write_gaussian_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Inverse of ``read_gaussian_ply`` in the layout GaussianModel.save_ply writes
    (gaussian_model.py:191-208): float32, binary little endian, f_rest channel-major."""
    P = xyz.shape[0]
    f_dc = np.asarray(features_dc, np.float32).transpose(0, 2, 1).reshape(P, -1)
    f_rest = np.asarray(features_rest, np.float32).transpose(0, 2, 1).reshape(P, -1)
    cols = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(f_dc.shape[1])] + \
           [f"f_rest_{i}" for i in range(f_rest.shape[1])] + ["opacity"] + \
           [f"scale_{i}" for i in range(scaling.shape[1])] + [f"rot_{i}" for i in range(rotation.shape[1])]
    arr = np.concatenate([np.asarray(xyz, np.float32), np.zeros((P, 3), np.float32), f_dc, f_rest,
                          np.asarray(opacity, np.float32).reshape(P, 1), np.asarray(scaling, np.float32),
                          np.asarray(rotation, np.float32)], axis=1).astype("<f4")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {P}\n".encode())
        for c in cols:
            f.write(f"property float {c}\n".encode())
        f.write(b"end_header\n")
        f.write(arr.tobytes())


class GaussianModel:
    def __init__(self, sh_degree: int, device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.device = device
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e

    # ---- GS/scene/gaussian_model.py:95-115 ---------------------------------------------------
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1.0):
        """GS/scene/gaussian_model.py:117-118 -> build_covariance_from_scaling_rotation (:27-31): Sigma = L L^T with
        L = R(q / |q|) diag(modifier * exp(scaling)), returned as the 6 upper-triangle entries
        [xx, xy, xz, yy, yz, zz] (general_utils.strip_lowerdiag) -- the layout the rasteriser's cov3D_precomp takes."""
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling).unsqueeze(1)      # R @ diag(s)
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)

    def load_ply(self, path):
        d = read_gaussian_ply(path)
        P = d["xyz"].shape[0]
        n_rest = 3 * (self.max_sh_degree + 1) ** 2 - 3
        assert d["f_rest"].shape[1] == n_rest, (d["f_rest"].shape, n_rest)
        f_rest = d["f_rest"].reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1).transpose(0, 2, 1)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
        self._xyz = t(d["xyz"])
        self._features_dc = t(d["f_dc"][:, None, :])
        self._features_rest = t(f_rest)
        self._opacity = t(d["opacity"])
        self._scaling = t(d["scale"])
        self._rotation = t(d["rot"])
        self.active_sh_degree = self.max_sh_degree

    def load_arrays(self, xyz, features_dc, features_rest, scaling, rotation, opacity):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(self.device)
        self._xyz, self._features_dc, self._features_rest = t(xyz), t(features_dc), t(features_rest)
        self._scaling, self._rotation, self._opacity = t(scaling), t(rotation), t(opacity)
        self.active_sh_degree = self.max_sh_degree

    def save_ply(self, path):
        c = lambda x: x.detach().cpu().numpy()
        write_gaussian_ply(path, c(self._xyz), c(self._features_dc), c(self._features_rest), c(self._opacity),
                           c(self._scaling), c(self._rotation))

    def raw(self):
        """Pre-activation tensors for ``Rasterizer.render_views`` (activations fused in-kernel)."""
        return dict(xyz=self._xyz, scaling=self._scaling, rotation=self._rotation, opacity=self._opacity,
                    features_dc=self._features_dc, features_rest=self._features_rest, raw=True,
                    sh_degree=self.active_sh_degree)
