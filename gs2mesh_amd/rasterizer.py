"""Thin object wrapper over the rasteriser half of the C ABI (handle + arenas + overflow retry).

This is plumbing between torch tensors (device memory, current stream) and
``libgs2mesh_amd.so``; the reference-shaped APIs are built on it:
``gs2mesh_amd.diff_gaussian_rasterization`` (operator level) and
``gs2mesh_amd.renderer_utils.Renderer`` (pipeline level).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

try:  # torch is plumbing for device memory / streams; the emulator tests run without a device
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _ptr(x, dtype=None, name="tensor"):
    """Pointer for the C ABI: the device pointer of a contiguous torch tensor on a GPU (``_lib.MEMORY``).  None -> NULL."""
    return _lib.MEMORY.ptr(x, dtype, name)


_NP2T = {}
if torch is not None:
    _NP2T = {np.float32: torch.float32, np.int32: torch.int32, np.uint8: torch.uint8, np.uint32: torch.int32,
             np.int64: torch.int64}


def _empty(like, shape, np_dtype):
    if _is_torch(like):
        return torch.empty(shape, dtype=_NP2T[np_dtype], device=like.device)
    return np.empty(shape, np_dtype)


def _stream_of(x, stream):
    if stream is not None:
        return C.c_void_p(int(stream))
    if _is_torch(x) and x.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    return C.c_void_p(0)


def auto_cull_level(P: int) -> int:
    """``GS2M_OPT_EXACT_TILE_CULL`` level of the pipeline-level callers (image-preserving either way): 1 = every rect with
    corners is tested tile by tile; 2 = rects of at most 4 binning tiles keep all their tiles.  Level 2 pays on models of small
    splats, where the test removes ~2 % of the instances at 60 % of the counting kernel's instructions (C3: count 49 -> 28 us per
    pair, rasteriser 478 -> 452) and costs ~1 % on a scene of larger / anisotropic splats (3 % more instances to composite):
    chosen by model size (C2 with the ordered model, round 6: level 2 +0.5 %, inside the run-to-run spread)."""
    return 2 if int(P) >= 1_000_000 else 1


SPATIAL_ORDER_MIN_P = 32768


def auto_spatial_order(P: int) -> bool:
    """``spatial_order="auto"`` of the pipeline-level callers: the Morton-ordered packed copy of the model
    (``Rasterizer.pack_model``: one-off prepare, every output bit-identical) for every model of at least 32 768 Gaussians.
    Rounds 2-5 applied it from 1 M Gaussians, judged by the ISOLATED stage times of a 300 k model (scatter - 4, counting + 4 us
    per pair); round 6 measured the pipelined job: the unordered model's scattered 8-byte key stores (3.9 x write amplification)
    cost the kernels of the other stream as well -- C2 3643 -> 3787 pairs/s with the ordered model, 3842-3858 with four pairs per
    launch on top; C5 3263 -> 3623-3643 (profiles/r6_experiments.txt C20)."""
    return int(P) >= SPATIAL_ORDER_MIN_P


def auto_blend_mode(gaussians: dict, share: float = 0.10) -> int:
    """``GS2M_OPT_BLEND_MODE`` of the pipeline-level callers (same image either way): 2, or 3 when at least ``share`` of the
    model's opacities can reach the reference's alpha cap (opacity > 0.98).  Mode 2 splits a staged batch into software-pipelined
    runs at every such instance (it needs ``min(0.99, alpha)``); mode 3 applies the cap to every instance instead.  Measured on
    MI355X (profiles/r6_experiments.txt B17): a trained-like splat with 30 % of its opacities at the cap composites in 172 instead
    of 190 us per pair with mode 3, `synth_v1` (2 %) in 176 instead of 170 -- break-even near 10 %.  ``gaussians`` = the dict
    ``render_views`` takes (``opacity``: logits when ``raw`` is set, activated values otherwise)."""
    op = gaussians.get("opacity")
    if op is None:
        return 2
    thr = math.log(0.98 / 0.02) if gaussians.get("raw") else 0.98
    if _is_torch(op):
        frac = float((op.reshape(-1) > thr).float().mean().item()) if op.numel() else 0.0
    else:
        a = np.asarray(op).reshape(-1)
        frac = float((a > thr).mean()) if a.size else 0.0
    return 3 if frac >= share else 2


def make_camera(width, height, tanfovx, tanfovy, viewmatrix, projmatrix, campos) -> _lib.Camera:
    """Host ``gs2m_camera`` from numpy-convertible matrices (transposed row-major like the
    reference's world_view_transform / full_proj_transform)."""
    c = _lib.Camera()
    c.width, c.height = int(width), int(height)
    c.tanfovx, c.tanfovy = float(tanfovx), float(tanfovy)
    vm = np.asarray(viewmatrix.detach().cpu() if _is_torch(viewmatrix) else viewmatrix, np.float32).reshape(16)
    pm = np.asarray(projmatrix.detach().cpu() if _is_torch(projmatrix) else projmatrix, np.float32).reshape(16)
    cp = np.asarray(campos.detach().cpu() if _is_torch(campos) else campos, np.float32).reshape(3)
    c.viewmatrix[:] = vm.tolist()
    c.projmatrix[:] = pm.tolist()
    c.campos[:] = cp.tolist()
    return c


def camera_from(cam) -> _lib.Camera:
    """From a ``gs2mesh_amd.graphics.Camera`` (or anything with the 3DGS Camera attributes)."""
    import math
    return make_camera(cam.image_width, cam.image_height, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                       cam.world_view_transform, cam.full_proj_transform, cam.camera_center)


def morton_order(xyz, bits: int = 10):
    """Permutation (int32, position -> Gaussian id) that sorts the Gaussians along the 3-D Morton curve of their centres
    (``bits`` per axis over the bounding box; stable, so equal codes keep their id order).  One-time preparation on the
    device with torch (numpy for the emulator tests); any spatially coherent order serves ``pack_model`` equally."""
    if _is_torch(xyz):
        lo = xyz.amin(dim=0)
        span = (xyz.amax(dim=0) - lo).clamp_min(1e-20)
        q = ((xyz - lo) / span * (2 ** bits - 1)).to(torch.int64).clamp_(0, 2 ** bits - 1)
        code = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
        for b in range(bits):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return torch.argsort(code, stable=True).to(torch.int32).contiguous()
    x = np.asarray(xyz, np.float64)
    lo = x.min(axis=0)
    span = np.maximum(x.max(axis=0) - lo, 1e-20)
    q = np.clip(((x - lo) / span * (2 ** bits - 1)).astype(np.int64), 0, 2 ** bits - 1)
    code = np.zeros(x.shape[0], np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.ascontiguousarray(np.argsort(code, kind="stable").astype(np.int32))


class Rasterizer:
    """Owns one ``gs2m_raster`` handle (persistent arenas).  Not thread-safe; one stream at a time."""

    def __init__(self, device: int = 0, lib=None):
        self._lib = lib or _lib.get()
        h = C.c_void_p()
        _lib.check(self._lib.gs2m_raster_create(C.byref(h), int(device)), self._lib)
        self._h = h
        self.device = device
        self.last_num_rendered = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gs2m_raster_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option: int, value: int):
        _lib.check(self._lib.gs2m_raster_set_option(self._h, option, value), self._lib)

    def reserve(self, P, n_views, W, H, instances):
        _lib.check(self._lib.gs2m_raster_reserve(self._h, int(P), int(n_views), int(W), int(H), int(instances)),
                   self._lib)

    def status(self, n_views=1, stream=None):
        """Synchronises.  -> (num_rendered list, overflow flag, required instances per view)."""
        nr = (C.c_int64 * max(n_views, 1))()
        ov = C.c_int(0)
        req = C.c_int64(0)
        _lib.check(self._lib.gs2m_raster_status(self._h, stream or C.c_void_p(0), n_views, nr, C.byref(ov),
                                                C.byref(req)), self._lib)
        return list(nr)[:n_views], bool(ov.value), int(req.value)

    def stage_times(self, stream=None):
        """With OPT_STAGE_TIMING on: synchronises and returns {stage: (total_ms, launches)} measured with
        hipEvents on the work stream since the previous query."""
        n = len(_lib.RASTER_STAGES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(self._lib.gs2m_raster_stage_times(self._h, stream or C.c_void_p(0), ms, cnt), self._lib)
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.RASTER_STAGES)}

    def blend_cycles(self, stream=None):
        """With OPT_BLEND_PROFILE on: {counter: value} of the compositing kernel's phase stamps since the last query."""
        n = len(_lib.BLEND_PROF_COUNTERS)
        c = (C.c_uint64 * n)()
        _lib.check(self._lib.gs2m_raster_blend_cycles(self._h, stream or C.c_void_p(0), c), self._lib)
        return {name: int(c[i]) for i, name in enumerate(_lib.BLEND_PROF_COUNTERS)}

    # -- operator level -------------------------------------------------------------------------
    def forward(self, means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=3,
                scale_modifier=1.0, prefiltered=False, debug=False, stream=None, sync=True):
        """``gs2m_rasterize_forward``; returns (color[3,H,W], radii[P]).  With ``sync`` (default) the
        instance arena is checked after the call and the pass is repeated once if it overflowed."""
        P = int(means3D.shape[0])
        D = int(sh_degree)
        M = 0 if shs is None or (hasattr(shs, "numel") and shs.numel() == 0) or getattr(shs, "size", 1) == 0 \
            else int(shs.shape[1])
        out = _empty(means3D, (3, int(H), int(W)), np.float32)
        radii = _empty(means3D, (P,), np.int32)
        st = _stream_of(means3D, stream)
        f32 = torch.float32 if _is_torch(means3D) else None

        def call():
            _lib.check(self._lib.gs2m_rasterize_forward(
                self._h, P, D, M, _ptr(bg, f32, "bg"), int(W), int(H), _ptr(means3D, f32, "means3D"),
                _ptr(shs, f32, "shs"), _ptr(colors_precomp, f32, "colors_precomp"),
                _ptr(opacities, f32, "opacities"), _ptr(scales, f32, "scales"), float(scale_modifier),
                _ptr(rotations, f32, "rotations"), _ptr(cov3D_precomp, f32, "cov3D_precomp"),
                _ptr(viewmatrix, f32, "viewmatrix"), _ptr(projmatrix, f32, "projmatrix"),
                _ptr(campos, f32, "campos"), float(tanfovx), float(tanfovy), int(bool(prefiltered)), _ptr(out),
                _ptr(radii), int(bool(debug)), st), self._lib)

        call()
        if sync:
            nr, ov, req = self.status(1, st)
            if ov:
                self.reserve(P, 1, W, H, int(req * 1.25) + 1024)
                call()
                nr, ov, req = self.status(1, st)
                if ov:
                    raise RuntimeError("instance arena overflow persists after growing")
            self.last_num_rendered = nr[0]
        return out, radii

    def mark_visible(self, positions, viewmatrix, projmatrix, stream=None):
        P = int(positions.shape[0])
        present = _empty(positions, (P,), np.uint8)
        st = _stream_of(positions, stream)
        _lib.check(self._lib.gs2m_mark_visible(P, _ptr(positions), _ptr(viewmatrix), _ptr(projmatrix),
                                               _ptr(present), st), self._lib)
        return present

    # -- pipeline level -------------------------------------------------------------------------
    def _gaussians_struct(self, gaussians: dict):
        xyz = gaussians["xyz"]
        g = _lib.Gaussians()
        g.P = int(xyz.shape[0])
        g.sh_degree = int(gaussians.get("sh_degree", 3))
        g.raw = int(bool(gaussians.get("raw", True)))
        f32 = torch.float32 if _is_torch(xyz) else None
        g.xyz = _ptr(xyz, f32, "xyz")
        g.scales = _ptr(gaussians["scaling"], f32, "scaling")
        g.rotations = _ptr(gaussians["rotation"], f32, "rotation")
        g.opacities = _ptr(gaussians["opacity"], f32, "opacity")
        if gaussians.get("features") is not None:
            g.shs = _ptr(gaussians["features"], f32, "features")
            g.shs_rest = None
            g.M = int(gaussians["features"].shape[1])
        else:
            g.shs = _ptr(gaussians["features_dc"], f32, "features_dc")
            g.shs_rest = _ptr(gaussians["features_rest"], f32, "features_rest")
            g.M = 1 + int(gaussians["features_rest"].shape[1])
        return g

    def pack_sh(self, gaussians: dict, stream=None):
        """One-time preparation (``gs2m_raster_pack_sh``): cache a wave-transposed copy of the SH block of
        these Gaussians in the handle; ``render_views`` with the same tensors then reads it."""
        g = self._gaussians_struct(gaussians)
        _lib.check(self._lib.gs2m_raster_pack_sh(self._h, C.byref(g), _stream_of(gaussians["xyz"], stream)), self._lib)

    def pack_model(self, gaussians: dict, order=None, stream=None):
        """One-time preparation (``gs2m_raster_pack_model``, a superset of ``pack_sh``): cache a SPATIALLY ORDERED packed
        copy of these Gaussians in the handle.  ``order`` = int32 permutation (position -> Gaussian id) on the device;
        default: the Morton order of ``xyz`` (``morton_order``).  ``render_views`` with the same tensors then works on
        the copy; images, radii and taps are those of the unordered model."""
        if order is None:
            order = morton_order(gaussians["xyz"])
        g = self._gaussians_struct(gaussians)
        i32 = torch.int32 if _is_torch(order) else None
        if not _is_torch(order):
            order = np.ascontiguousarray(order, np.int32)
        _lib.check(self._lib.gs2m_raster_pack_model(self._h, C.byref(g), _ptr(order, i32, "order"),
                                                    _stream_of(gaussians["xyz"], stream)), self._lib)
        return order

    def invalidate_pack(self):
        """``gs2m_raster_pack_invalidate``: forget the packed copies (they are matched to the caller's tensors by device
        pointer only -- call after an in-place update of the Gaussians, or pack again)."""
        _lib.check(self._lib.gs2m_raster_pack_invalidate(self._h), self._lib)

    def render_views(self, gaussians: dict, cams, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, want_color=True,
                     want_rgb8=False, want_radii=False, out_color=None, out_rgb8=None, stream=None, sync=True):
        """``gs2m_render_views``.  ``gaussians``: dict with xyz[P,3], scaling[P,3], rotation[P,4],
        opacity[P,1|P], and either features[P,M,3] or features_dc[P,1,3] + features_rest[P,M-1,3];
        ``raw`` (default True) = pre-activation GaussianModel parameters; ``sh_degree`` (default 3).
        ``cams``: list of ``_lib.Camera``.  Returns dict(color, rgb8, radii, num_rendered)."""
        xyz = gaussians["xyz"]
        P = int(xyz.shape[0])
        n = len(cams)
        W, H = cams[0].width, cams[0].height
        g = self._gaussians_struct(gaussians)
        cam_arr = (_lib.Camera * n)(*cams)
        bg_arr = (C.c_float * 3)(*[float(b) for b in bg])
        if want_color and out_color is None:
            out_color = _empty(xyz, (n, 3, H, W), np.float32)
        if want_rgb8 and out_rgb8 is None:
            out_rgb8 = _empty(xyz, (n, H, W, 3), np.uint8)
        radii = _empty(xyz, (n, P), np.int32) if want_radii else None
        st = _stream_of(xyz, stream)

        def call():
            _lib.check(self._lib.gs2m_render_views(self._h, C.byref(g), cam_arr, n, bg_arr, float(scale_modifier),
                                                   _ptr(out_color), _ptr(out_rgb8), _ptr(radii), st), self._lib)

        call()
        nr = None
        if sync:
            nr, ov, req = self.status(n, st)
            if ov:
                self.reserve(P, min(n, 2), W, H, int(req * 1.25) + 1024)
                call()
                nr, ov, req = self.status(n, st)
                if ov:
                    raise RuntimeError("instance arena overflow persists after growing")
            self.last_num_rendered = nr
        return dict(color=out_color, rgb8=out_rgb8, radii=radii, num_rendered=nr)

    # -- parity taps ------------------------------------------------------------------------------
    def download_geometry(self, v, P, stream=None):
        out = dict(means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
                   conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
                   rect=np.zeros((P, 4), np.uint16), tiles_touched=np.zeros(P, np.uint32))
        vp = lambda a: C.c_void_p(a.ctypes.data)
        _lib.check(self._lib.gs2m_raster_download_geometry(
            self._h, stream or C.c_void_p(0), int(v), int(P), vp(out["means2D"]), vp(out["depths"]),
            vp(out["conic_opacity"]), vp(out["rgb"]), vp(out["rect"]), vp(out["tiles_touched"])), self._lib)
        return out

    def download_binning(self, v, n, n_tiles, stream=None):
        pl = np.zeros(max(int(n), 1), np.uint32)
        ranges = np.zeros((n_tiles, 2), np.uint32)
        _lib.check(self._lib.gs2m_raster_download_binning(self._h, stream or C.c_void_p(0), int(v), int(n),
                                                          C.c_void_p(pl.ctypes.data), int(n_tiles),
                                                          C.c_void_p(ranges.ctypes.data)), self._lib)
        return pl[:int(n)], ranges
