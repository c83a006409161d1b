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
    """World -> view matrix of graphics_utils.py:38-49.  R is the TRANSPOSED world->camera rotation (3DGS convention).
    Mathematically this is [ R^T | -R^T (c + translate) scale ] with the camera centre c = -R t (= [ R^T | t ] for the
    defaults); it is evaluated the way the reference does -- invert, move the centre, invert back, in float64 -- because
    the kernels' projected records are compared BIT FOR BIT with the reference's and the two LAPACK inversions leave
    their own last-bit rounding in the float32 result (the closed form differs from it by 1 ulp in ~1 entry of 16)."""
    w2c = np.block([[np.asarray(R, np.float64).T, np.reshape(np.asarray(t, np.float64), (3, 1))],
                    [np.zeros((1, 3)), np.ones((1, 1))]])
    c2w = np.linalg.inv(w2c)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Clip matrix of graphics_utils.py:51-71 (float32 [4,4], w = z_view, z mapped to [0,1]): the frustum is symmetric
    (right = -left = znear tan(fovX / 2), likewise in y), so the off-centre terms vanish and the scales are 1 / tan."""
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1.0 / math.tan(fovX / 2)
    P[1, 1] = 1.0 / math.tan(fovY / 2)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
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
