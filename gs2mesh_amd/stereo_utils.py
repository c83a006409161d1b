"""Disparity post-processing of ``gs2mesh_utils.stereo_utils.Stereo`` on the device.

The stereo network itself (DLNR, PyTorch) is outside the hot path; what sits between it and the TSDF is
small data-parallel work the reference does in numpy on the host with four ``np.save`` round trips per view:

  * ``get_occlusion_mask(L2R, R2L, threshold)``   stereo_utils.py:149-179 (left-right consistency)
  * ``depth = fx * baseline / disparity_LR``      stereo_utils.py:133

``depth_and_occlusion`` fuses both into one HIP kernel whose outputs (depth f32, mask u8) are exactly
the ``depth`` / ``mask`` inputs of ``ScalableTSDFVolume.integrate``.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .rasterizer import _empty, _ptr, _stream_of


def depth_and_occlusion(disparity_LR, disparity_RL, fx, baseline, occlusion_threshold=3, want_depth=True,
                        want_mask=True, lib=None, stream=None):
    """-> (depth [H,W] f32 or None, visible_mask [H,W] u8 (1 = visible) or None), device tensors."""
    lib = lib or _lib.get()
    H, W = int(disparity_LR.shape[0]), int(disparity_LR.shape[1])
    depth = _empty(disparity_LR, (H, W), np.float32) if want_depth else None
    mask = _empty(disparity_LR, (H, W), np.uint8) if want_mask else None
    _lib.check(lib.gs2m_stereo_depth_occlusion(_ptr(disparity_LR), _ptr(disparity_RL) if want_mask else None, W, H,
                                               float(fx) * float(baseline), float(occlusion_threshold), _ptr(depth),
                                               _ptr(mask), _stream_of(disparity_LR, stream)), lib)
    return depth, mask


def get_occlusion_mask(L2R_disparity, R2L_disparity, occlusion_threshold, lib=None):
    """Same name / arguments / meaning as Stereo.get_occlusion_mask: boolean array, True = visible."""
    _, mask = depth_and_occlusion(L2R_disparity, R2L_disparity, 1.0, 1.0, occlusion_threshold, want_depth=False, lib=lib)
    return mask.astype(bool) if isinstance(mask, np.ndarray) else mask.to(bool)
