"""Operator-level drop-in for the reference's ``diff_gaussian_rasterization`` module
(DGR/diff_gaussian_rasterization/__init__.py), forward pass only.

Same public names, argument meaning and error behaviour:

    from gs2mesh_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rasterizer = GaussianRasterizer(raster_settings=GaussianRasterizationSettings(...))
    color, radii = rasterizer(means3D=..., means2D=..., shs=..., colors_precomp=None, opacities=...,
                              scales=..., rotations=..., cov3D_precomp=None)

``GS/gaussian_renderer/__init__.py:14`` imports exactly these two names, so putting this package on
``sys.path`` as ``diff_gaussian_rasterization`` makes the reference's ``render()`` run
on the HIP kernels unchanged.  The hot path runs under ``torch.no_grad()``
(gs2mesh_utils/renderer_utils.py:374); the backward pass (3DGS training) is out of scope and raises.
All tensors must be float32, contiguous and on the HIP device; outputs are freshly allocated.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from ..rasterizer import Rasterizer

_HANDLES = {}


def _handle(device: torch.device) -> Rasterizer:
    """One persistent arena set per device (the reference re-allocates three byte arenas per call,
    DGR/rasterize_points.cu:73-78)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = _HANDLES.get(idx)
    if h is None:
        h = _HANDLES[idx] = Rasterizer(idx)
    return h


class GaussianRasterizationSettings(NamedTuple):
    """DGR/diff_gaussian_rasterization/__init__.py:157-169."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """DGR __init__.py:21-44 / _RasterizeGaussians.forward :46-98 (forward only)."""
    if any(t is not None and isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled()
           for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)):
        raise NotImplementedError(
            "gs2mesh_amd implements the forward rasteriser only (the GS2Mesh hot path runs under "
            "torch.no_grad(), renderer_utils.py:374); 3DGS training gradients are out of scope")
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
    rs = raster_settings
    h = _handle(means3D.device)
    none_if_empty = lambda t: None if t is None or t.numel() == 0 else _f32c(t)
    if rs.debug:
        # the reference's debug mode (DGR __init__.py:83-90): keep a host copy of every argument and, if the rasteriser
        # fails, leave it in snapshot_fw.dump for post-mortem; the C ABI call itself runs with a sync + check per launch
        keep = tuple(t.detach().cpu().clone() if isinstance(t, torch.Tensor) else t for t in (
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
            rs.prefiltered, rs.debug))
        try:
            return _forward(h, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, none_if_empty)
        except Exception as ex:
            torch.save(keep, "snapshot_fw.dump")
            print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise ex
    return _forward(h, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, none_if_empty)


def _forward(h, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, none_if_empty):
    color, radii = h.forward(
        _f32c(means3D), _f32c(opacities).reshape(-1), _f32c(rs.viewmatrix), _f32c(rs.projmatrix), _f32c(rs.campos),
        _f32c(rs.bg), int(rs.image_width), int(rs.image_height), float(rs.tanfovx), float(rs.tanfovy),
        shs=none_if_empty(sh), colors_precomp=none_if_empty(colors_precomp), scales=none_if_empty(scales),
        rotations=none_if_empty(rotations), cov3D_precomp=none_if_empty(cov3Ds_precomp),
        sh_degree=int(rs.sh_degree), scale_modifier=float(rs.scale_modifier), prefiltered=bool(rs.prefiltered),
        debug=bool(rs.debug))
    return color, radii


class GaussianRasterizer(nn.Module):
    """DGR/diff_gaussian_rasterization/__init__.py:171-220."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            rs = self.raster_settings
            h = _handle(positions.device)
            present = h.mark_visible(_f32c(positions), _f32c(rs.viewmatrix), _f32c(rs.projmatrix))
        return present.to(torch.bool)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        # Invoke the HIP rasterization routine
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
