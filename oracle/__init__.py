"""CPU oracle for the render -> fuse hot path (TEST INFRASTRUCTURE ONLY).

ctypes front-end of ``oracle/raster_oracle.c`` (restatement of the reference rasteriser
forward, DGR/cuda_rasterizer/{forward.cu,rasterizer_impl.cu,auxiliary.h}) and
``oracle/tsdf_oracle.cpp`` (restatement of Open3D 0.17 ``ScalableTSDFVolume::Integrate``
as called at gs2mesh_utils/tsdf_utils.py:53-56,88-93,106-107).

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may
import this package, and only as the checker.  The product package ``gs2mesh_amd`` never
does (tests/test_capi_symbols.py greps for it).

Parity status: SH->RGB, Sigma=R S^2 R^T and the camera matrices are pinned by golden
vectors generated from the reference's own Python (tests/golden/make_golden.py).  The whole
rasteriser forward (projection, tile rects, binning order, compositing) is pinned against the
REFERENCE'S OWN KERNELS compiled for the CPU (``oracle/_ref``: oracle/build_ref.py +
oracle/ref_driver.cpp + oracle/ref_shim/; ``ref_forward`` below) and against golden vectors
produced by them (tests/golden/ref_forward.npz).  The TSDF is "parity unpinned" (Open3D absent);
it is covered by analytic known-answer tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the oracle with gcc/g++ (seconds)."""
    srcs = [os.path.join(_HERE, "raster_oracle.c"), os.path.join(_HERE, "tsdf_oracle.cpp")]
    outs = [os.path.join(_BUILD, "liboracle_raster.so"), os.path.join(_BUILD, "liboracle_tsdf.so")]
    fresh = all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(s) for s, o in zip(srcs, outs))
    if fresh and not force:
        return
    subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


_raster = None
_tsdf = None


def _opt(a, dtype):
    """None -> NULL pointer, array -> contiguous array of dtype."""
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def raster_lib():
    global _raster
    if _raster is None:
        build()
        lib = C.CDLL(os.path.join(_BUILD, "liboracle_raster.so"))
        vp = C.c_void_p
        lib.oracle_preprocess.restype = None
        lib.oracle_preprocess.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, vp, vp, vp,
                                          vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                          vp, vp, vp, vp, vp, vp, vp, vp]
        lib.oracle_mark_visible.restype = None
        lib.oracle_mark_visible.argtypes = [C.c_int, vp, vp, vp, vp]
        lib.oracle_bin.restype = C.c_int64
        lib.oracle_bin.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int64]
        lib.oracle_render.restype = None
        lib.oracle_render.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.oracle_rasterize_forward.restype = C.c_int64
        lib.oracle_rasterize_forward.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp,
                                                 vp, C.c_float, vp, vp, vp, vp, vp, C.c_float, C.c_float,
                                                 C.c_int, vp, vp]
        lib.oracle_render_flip_bounds.restype = None
        lib.oracle_render_flip_bounds.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]
        lib.oracle_tile_may_contribute.restype = C.c_int
        lib.oracle_tile_may_contribute.argtypes = [C.c_float] * 6 + [C.c_int, C.c_int]
        _raster = lib
    return _raster


def tsdf_lib():
    global _tsdf
    if _tsdf is None:
        build()
        lib = C.CDLL(os.path.join(_BUILD, "liboracle_tsdf.so"))
        vp = C.c_void_p
        lib.oracle_tsdf_create.restype = vp
        lib.oracle_tsdf_create.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        lib.oracle_tsdf_destroy.argtypes = [vp]
        lib.oracle_tsdf_set_threads.argtypes = [vp, C.c_int]
        lib.oracle_tsdf_max_threads.restype = C.c_int
        lib.oracle_rgbd_convert_depth.argtypes = [vp, vp, C.c_int64, C.c_double, C.c_double]
        lib.oracle_dist_multiplier.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, vp]
        lib.oracle_tsdf_integrate.restype = C.c_int64
        lib.oracle_tsdf_integrate.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                              C.c_double, vp]
        lib.oracle_tsdf_num_blocks.restype = C.c_int64
        lib.oracle_tsdf_num_blocks.argtypes = [vp]
        lib.oracle_tsdf_block_updates.restype = C.c_int64
        lib.oracle_tsdf_block_updates.argtypes = [vp]
        lib.oracle_tsdf_export.argtypes = [vp, vp, vp, vp, vp]
        lib.oracle_tsdf_extract_mesh.restype = C.c_int64
        lib.oracle_tsdf_extract_mesh.argtypes = [vp, vp, vp, vp]
        lib.oracle_tsdf_mesh_copy.argtypes = [vp, vp, vp, vp, vp]
        lib.oracle_tsdf_import.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
        _tsdf = lib
    return _tsdf


# --------------------------------------------------------------------------------------
# rasteriser
# --------------------------------------------------------------------------------------

def preprocess(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H,
               tanfovx, tanfovy, sh_degree=3, scale_modifier=1.0, cov3D_precomp=None,
               colors_precomp=None):
    """preprocessCUDA (forward.cu:155-256) -> dict of per-Gaussian arrays."""
    lib = raster_lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    P = means3D.shape[0]
    scales = _opt(scales, np.float32)
    rotations = _opt(rotations, np.float32)
    opacities = np.ascontiguousarray(opacities, np.float32).reshape(-1)
    shs = _opt(shs, np.float32)
    M = 0 if shs is None else shs.shape[1]
    cov3D_precomp = _opt(cov3D_precomp, np.float32)
    colors_precomp = _opt(colors_precomp, np.float32)
    vm = np.ascontiguousarray(viewmatrix, np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, np.float32).reshape(16)
    cp = np.ascontiguousarray(campos, np.float32).reshape(3)
    out = dict(
        radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
        cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), tiles_touched=np.zeros(P, np.uint32),
        rect=np.zeros((P, 4), np.uint32))
    lib.oracle_preprocess(P, sh_degree, M, _ptr(means3D), _ptr(scales), float(scale_modifier), _ptr(rotations),
                          _ptr(opacities), _ptr(shs), _ptr(cov3D_precomp), _ptr(colors_precomp), _ptr(vm),
                          _ptr(pm), _ptr(cp), int(W), int(H), float(tanfovx), float(tanfovy),
                          _ptr(out["radii"]), _ptr(out["means2D"]), _ptr(out["depths"]), _ptr(out["cov3D"]),
                          _ptr(out["rgb"]), _ptr(out["conic_opacity"]), _ptr(out["tiles_touched"]),
                          _ptr(out["rect"]))
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    lib = raster_lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    P = means3D.shape[0]
    vm = np.ascontiguousarray(viewmatrix, np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, np.float32).reshape(16)
    present = np.zeros(P, np.uint8)
    lib.oracle_mark_visible(P, _ptr(means3D), _ptr(vm), _ptr(pm), _ptr(present))
    return present.astype(bool)


def bin_instances(geom, W, H, exact_cull=False):
    """duplicateWithKeys + stable sort + identifyTileRanges -> (point_list[n], ranges[tiles,2])."""
    lib = raster_lib()
    P = geom["radii"].shape[0]
    cap = int(geom["tiles_touched"].astype(np.int64).sum())
    pl = np.zeros(max(cap, 1), np.uint32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = np.zeros((gx * gy, 2), np.uint32)
    n = lib.oracle_bin(P, W, H, _ptr(geom["radii"]), _ptr(geom["means2D"]), _ptr(geom["depths"]),
                       _ptr(geom["conic_opacity"]), int(exact_cull), _ptr(pl), _ptr(ranges), cap)
    return pl[:n].copy(), ranges


def render(W, H, ranges, point_list, means2D, features, conic_opacity, bg):
    """renderCUDA (forward.cu:261-374) -> (color[3,H,W], final_T[H,W], n_contrib[H,W])."""
    lib = raster_lib()
    out = np.zeros((3, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    pl = np.ascontiguousarray(point_list, np.uint32)
    if pl.size == 0:
        pl = np.zeros(1, np.uint32)
    lib.oracle_render(W, H, _ptr(np.ascontiguousarray(ranges, np.uint32)), _ptr(pl),
                      _ptr(np.ascontiguousarray(means2D, np.float32)),
                      _ptr(np.ascontiguousarray(features, np.float32)),
                      _ptr(np.ascontiguousarray(conic_opacity, np.float32)),
                      _ptr(np.ascontiguousarray(bg, np.float32)), _ptr(out), _ptr(final_T), _ptr(n_contrib))
    return out, final_T, n_contrib


def render_flip_bounds(W, H, ranges, point_list, means2D, conic_opacity, cmax, rel_eps=1e-5):
    """Replay of renderCUDA's three threshold decisions (power > 0, alpha < 1/255, T(1-alpha) < 1e-4): per pixel, the
    number of decisions taken within ``rel_eps`` of their threshold and a bound on the change of the pixel if an
    implementation with different last-ulp rounding takes the other side (oracle_render_flip_bounds).
    -> dict(n_alpha[H,W], n_T[H,W], n_power[H,W] u16, bound[H,W] f32, cond[H,W] f32 = the reference's own rounding noise
    on ill-conditioned evaluations: 2 alpha T cmax x 2^-22 x the largest term of `power`, where that exceeds rel_eps)."""
    lib = raster_lib()
    out = dict(n_alpha=np.zeros((H, W), np.uint16), n_T=np.zeros((H, W), np.uint16), n_power=np.zeros((H, W), np.uint16),
               bound=np.zeros((H, W), np.float32), cond=np.zeros((H, W), np.float32))
    pl = np.ascontiguousarray(point_list, np.uint32)
    if pl.size == 0:
        pl = np.zeros(1, np.uint32)
    lib.oracle_render_flip_bounds(int(W), int(H), _ptr(np.ascontiguousarray(ranges, np.uint32)), _ptr(pl),
                                  _ptr(np.ascontiguousarray(means2D, np.float32)),
                                  _ptr(np.ascontiguousarray(conic_opacity, np.float32)), float(rel_eps), float(cmax),
                                  _ptr(out["n_alpha"]), _ptr(out["n_T"]), _ptr(out["n_power"]), _ptr(out["bound"]), _ptr(out["cond"]))
    return out


def rasterize_forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg,
                      shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                      sh_degree=3, scale_modifier=1.0, exact_cull=False):
    """Rasterizer::forward (rasterizer_impl.cu:198-336) -> (color[3,H,W], radii[P], num_rendered)."""
    lib = raster_lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    P = means3D.shape[0]
    out = np.zeros((3, H, W), np.float32)  # torch::full({3,H,W}, 0.0): rasterize_points.cu:68
    radii = np.zeros(P, np.int32)
    if P == 0:
        return out, radii, 0
    shs = _opt(shs, np.float32)
    M = 0 if shs is None else shs.shape[1]
    colors_precomp = _opt(colors_precomp, np.float32)
    scales = _opt(scales, np.float32)
    rotations = _opt(rotations, np.float32)
    cov3D_precomp = _opt(cov3D_precomp, np.float32)
    opacities = np.ascontiguousarray(opacities, np.float32).reshape(-1)
    vm = np.ascontiguousarray(viewmatrix, np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, np.float32).reshape(16)
    cp = np.ascontiguousarray(campos, np.float32).reshape(3)
    bgc = np.ascontiguousarray(bg, np.float32).reshape(3)
    n = lib.oracle_rasterize_forward(P, sh_degree, M, _ptr(bgc), int(W), int(H), _ptr(means3D), _ptr(shs),
                                     _ptr(colors_precomp), _ptr(opacities), _ptr(scales), float(scale_modifier),
                                     _ptr(rotations), _ptr(cov3D_precomp), _ptr(vm), _ptr(pm), _ptr(cp),
                                     float(tanfovx), float(tanfovy), int(exact_cull), _ptr(out),
                                     _ptr(radii))
    return out, radii, int(n)


def activate(scaling_raw, rotation_raw, opacity_raw):
    """GaussianModel getters (GS/scene/gaussian_model.py:95-115): exp, F.normalize (eps 1e-12),
    sigmoid, all in fp32."""
    s = np.exp(np.asarray(scaling_raw, np.float32)).astype(np.float32)
    q = np.asarray(rotation_raw, np.float32)
    n = np.sqrt((q * q).sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
    q = (q / np.maximum(n, np.float32(1e-12))).astype(np.float32)
    o = np.asarray(opacity_raw, np.float32)
    o = (np.float32(1) / (np.float32(1) + np.exp(-o).astype(np.float32))).astype(np.float32)
    return s, q, o


# --------------------------------------------------------------------------------------
# TSDF
# --------------------------------------------------------------------------------------

class ScalableTSDFVolume:
    """Restated open3d.pipelines.integration.ScalableTSDFVolume (integrate + extract_triangle_mesh)."""

    def __init__(self, voxel_length, sdf_trunc, color_type=1, volume_unit_resolution=16,
                 depth_sampling_stride=4):
        self._lib = tsdf_lib()
        self.res = volume_unit_resolution
        self._h = self._lib.oracle_tsdf_create(float(voxel_length), float(sdf_trunc), int(color_type),
                                               int(volume_unit_resolution), int(depth_sampling_stride))

    def __del__(self):
        try:
            self._lib.oracle_tsdf_destroy(self._h)
        except Exception:
            pass

    def set_threads(self, n):
        self._lib.oracle_tsdf_set_threads(self._h, int(n))

    @staticmethod
    def max_threads():
        return int(tsdf_lib().oracle_tsdf_max_threads())

    @staticmethod
    def convert_depth(depth, depth_scale, depth_trunc):
        d = np.ascontiguousarray(depth, np.float32)
        out = np.empty_like(d)
        tsdf_lib().oracle_rgbd_convert_depth(_ptr(d), _ptr(out), d.size, float(depth_scale), float(depth_trunc))
        return out

    def integrate(self, depth_f, color_u8, width, height, fx, fy, cx, cy, extrinsic_w2c):
        """depth_f: converted float depth [H,W]; color_u8 [H,W,3] or None."""
        d = np.ascontiguousarray(depth_f, np.float32)
        c = None if color_u8 is None else np.ascontiguousarray(color_u8, np.uint8)
        E = np.ascontiguousarray(extrinsic_w2c, np.float64).reshape(16)
        n = self._lib.oracle_tsdf_integrate(self._h, _ptr(d), _ptr(c), int(width), int(height), float(fx),
                                            float(fy), float(cx), float(cy), _ptr(E))
        if n < 0:
            raise RuntimeError("singular extrinsic")
        return int(n)

    @property
    def num_blocks(self):
        return int(self._lib.oracle_tsdf_num_blocks(self._h))

    @property
    def block_updates(self):
        return int(self._lib.oracle_tsdf_block_updates(self._h))

    def export(self):
        n = self.num_blocks
        nv = self.res ** 3
        keys = np.zeros((n, 3), np.int32)
        tsdf = np.zeros((n, nv), np.float32)
        weight = np.zeros((n, nv), np.float32)
        color = np.zeros((n, nv, 3), np.float64)
        self._lib.oracle_tsdf_export(self._h, _ptr(keys), _ptr(tsdf), _ptr(weight), _ptr(color))
        return keys, tsdf, weight, color


    def extract_triangle_mesh(self, tri_table):
        """Restated ``ScalableTSDFVolume::ExtractTriangleMesh``.  ``tri_table``: the 256 rows of the classic marching-cubes
        table (lists of edge ids, e.g. ``tools/mc_classic_table.T``).  -> dict(vertices [nv,3] f64, colors [nv,3] f64,
        triangles [nt,3] i32, zero_offset_vertices)."""
        tab = np.full((256, 16), -1, np.int8)
        for c, row in enumerate(tri_table):
            tab[c, :len(row)] = row
        nv, nz = C.c_int64(0), C.c_int64(0)
        nt = int(self._lib.oracle_tsdf_extract_mesh(self._h, _ptr(tab), C.addressof(nv), C.addressof(nz)))
        v = np.zeros((int(nv.value), 3), np.float64)
        col = np.zeros((int(nv.value), 3), np.float64)
        tri = np.zeros((nt, 3), np.int32)
        eidx = np.zeros((int(nv.value), 4), np.int32)
        self._lib.oracle_tsdf_mesh_copy(self._h, _ptr(v), _ptr(col), _ptr(tri), _ptr(eidx))
        return dict(vertices=v, colors=col, triangles=tri, edge_index=eidx, zero_offset_vertices=int(nz.value))

    def import_state(self, keys, tsdf, weight, color=None):
        """Test aid: set the state of the units ``keys`` [n,3] directly; tsdf / weight [n,res^3] in (x, y, z) index order
        x * res^2 + y * res + z, color [n,res^3,3] float64 or None."""
        k = np.ascontiguousarray(keys, np.int32)
        t = np.ascontiguousarray(tsdf, np.float32)
        w = np.ascontiguousarray(weight, np.float32)
        c = None if color is None else np.ascontiguousarray(color, np.float64)
        self._lib.oracle_tsdf_import(self._h, int(k.shape[0]), _ptr(k), _ptr(t), _ptr(w), _ptr(c))


def dist_multiplier(width, height, fx, fy, cx, cy):
    out = np.zeros((height, width), np.float32)
    tsdf_lib().oracle_dist_multiplier(int(width), int(height), float(fx), float(fy), float(cx), float(cy), _ptr(out))
    return out


# ---- the reference's own kernels on the CPU (oracle/_ref) --------------------------------------
_ref = None


def ref_available(build: bool = True) -> bool:
    """True if oracle/_ref/libref_raster.so exists (or can be built here: needs /root/reference)."""
    from . import build_ref
    if build:
        try:
            return build_ref.build() is not None
        except Exception:
            return False
    return os.path.exists(build_ref.LIB)


def ref_lib():
    global _ref
    if _ref is None:
        from . import build_ref
        path = build_ref.build()
        if path is None:
            raise RuntimeError("oracle/_ref is not built and /root/reference is absent")
        lib = C.CDLL(path)
        lib.ref_forward.restype = C.c_longlong
        lib.ref_forward.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + \
            [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int] + \
            [C.c_void_p] * 9 + [C.c_longlong, C.c_void_p]
        lib.ref_mark_visible.restype = None
        lib.ref_mark_visible.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _ref = lib
    return _ref


def ref_forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg,
                shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                sh_degree=3, scale_modifier=1.0):
    """CudaRasterizer::Rasterizer::forward run by the reference's own kernels (oracle/ref_driver.cpp).
    -> dict(color[3,H,W], radii, means2D, depths, cov3D, rgb, conic_opacity, tiles_touched, point_list, ranges,
    num_rendered)."""
    lib = ref_lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    P = means3D.shape[0]
    shs = _opt(shs, np.float32)
    M = 0 if shs is None else shs.shape[1]
    colors_precomp = _opt(colors_precomp, np.float32)
    scales = _opt(scales, np.float32)
    rotations = _opt(rotations, np.float32)
    cov3D_precomp = _opt(cov3D_precomp, np.float32)
    opacities = np.ascontiguousarray(opacities, np.float32).reshape(-1)
    vm = np.ascontiguousarray(viewmatrix, np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, np.float32).reshape(16)
    cp = np.ascontiguousarray(campos, np.float32).reshape(3)
    bgc = np.ascontiguousarray(bg, np.float32).reshape(3)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    o = dict(color=np.zeros((3, H, W), np.float32), radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), np.float32),
             depths=np.zeros(P, np.float32), cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
             conic_opacity=np.zeros((P, 4), np.float32), tiles_touched=np.zeros(P, np.uint32),
             ranges=np.zeros((gx * gy, 2), np.uint32))
    cap = max(1, 64 * P)
    while True:
        pl = np.zeros(cap, np.uint32)
        n = lib.ref_forward(P, sh_degree, M, _ptr(bgc), int(W), int(H), _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                            _ptr(opacities), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                            _ptr(vm), _ptr(pm), _ptr(cp), float(tanfovx), float(tanfovy), 0, _ptr(o["color"]),
                            _ptr(o["radii"]), _ptr(o["means2D"]), _ptr(o["depths"]), _ptr(o["cov3D"]), _ptr(o["rgb"]),
                            _ptr(o["conic_opacity"]), _ptr(o["tiles_touched"]), _ptr(pl), cap, _ptr(o["ranges"]))
        if n >= 0:
            break
        cap *= 4
    o["point_list"] = pl[:n].copy()
    o["num_rendered"] = int(n)
    return o


def ref_mark_visible(means3D, viewmatrix, projmatrix):
    lib = ref_lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    P = means3D.shape[0]
    vm = np.ascontiguousarray(viewmatrix, np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, np.float32).reshape(16)
    present = np.zeros(P, np.uint8)
    lib.ref_mark_visible(P, _ptr(means3D), _ptr(vm), _ptr(pm), _ptr(present))
    return present.astype(bool)
