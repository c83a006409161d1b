"""Per-view camera math of the render hot path (host side, numpy, no device work).

Mirrors, without the reference's GPU round trips:
  * GS/utils/graphics_utils.py:38-49  getWorld2View2
  * GS/utils/graphics_utils.py:51-71  getProjectionMatrix
  * GS/utils/graphics_utils.py:73-77  fov2focal / focal2fov
  * GS/scene/cameras.py:18-57         Camera (world_view_transform, projection_matrix,
                                      full_proj_transform, camera_center)
The reference builds these with ``.cuda()``, ``bmm`` and ``inverse`` on the device (several
tiny launches per eye) and uploads a ``torch.rand(3,h,w)`` image only
to carry (h, w) (renderer_utils.py:386).  Here they are 4x4 numpy float32 ops; the result
is handed to the kernels as a ``gs2m_camera`` host struct.
"""
from __future__ import annotations

import math

import numpy as np


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """graphics_utils.py:38-49.  R is the TRANSPOSED world->camera rotation (3DGS convention)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + translate) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """graphics_utils.py:51-71, as float32 numpy [4,4] (P[3,2] = 1, z in [0,1])."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), np.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """Inference part of GS/scene/cameras.py:Camera -- same attribute names, numpy float32.

    ``image`` is not needed: pass ``width``/``height`` (the reference passes a random image
    only for its shape, renderer_utils.py:386).
    """

    def __init__(self, colmap_id, R, T, FoVx, FoVy, width, height, image_name="", uid=0,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0):
        self.uid = uid
        self.colmap_id = colmap_id
        self.R = np.asarray(R)
        self.T = np.asarray(T)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_name = image_name
        self.image_width = int(width)
        self.image_height = int(height)
        self.zfar = 100.0
        self.znear = 0.01
        self.trans = trans
        self.scale = scale
        # cameras.py:54-57 (row-vector convention: matrices are stored transposed)
        self.world_view_transform = np.ascontiguousarray(getWorld2View2(self.R, self.T, trans, scale).T)
        self.projection_matrix = np.ascontiguousarray(
            getProjectionMatrix(self.znear, self.zfar, self.FoVx, self.FoVy).T)
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).astype(np.float32)
        self.camera_center = np.linalg.inv(self.world_view_transform)[3, :3].astype(np.float32)

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)
