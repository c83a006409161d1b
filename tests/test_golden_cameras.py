"""Per-view uniforms and stereo poses vs golden vectors produced by the reference's own Python
(transformation_utils + graphics_utils, chained as Renderer / Camera do; see
tests/golden/make_golden.py)."""
import os

import numpy as np

from gs2mesh_amd import synthetic
from gs2mesh_amd.graphics import Camera, getProjectionMatrix


def test_stereo_pair_matches_reference_euler_detour(golden_dir):
    g = np.load(os.path.join(golden_dir, "stereo_cameras.npz"))
    W, H, fx, fy, b = int(g["width"]), int(g["height"]), float(g["fx"]), float(g["fy"]), float(g["baseline"])
    for i, pose in enumerate(g["poses"]):
        left, right = synthetic.stereo_cameras(pose, W, H, fx, fy, b)
        # R_gs = R_w2c^T, T_gs = t, right: t - (b,0,0); reference goes through float32 Euler angles
        np.testing.assert_allclose(left.R, g["R_gs_left"][i], atol=2e-6)
        np.testing.assert_allclose(left.T, g["T_gs_left"][i], atol=1e-5)
        np.testing.assert_allclose(right.R, g["R_gs_right"][i], atol=2e-6)
        np.testing.assert_allclose(right.T, g["T_gs_right"][i], atol=1e-5)
        np.testing.assert_allclose(left.world_view_transform, g["wvt_left"][i], atol=1e-5)
        np.testing.assert_allclose(right.world_view_transform, g["wvt_right"][i], atol=1e-5)
        np.testing.assert_allclose(left.projection_matrix, g["proj"][i], atol=1e-6)
        np.testing.assert_allclose(left.full_proj_transform, g["full_left"][i], atol=2e-5)
        np.testing.assert_allclose(right.full_proj_transform, g["full_right"][i], atol=2e-5)
        np.testing.assert_allclose(left.camera_center, g["center_left"][i], atol=1e-5)
        np.testing.assert_allclose(right.camera_center, g["center_right"][i], atol=1e-5)
        # left_camera['extrinsic'] = inv([R|t]) (camera -> world), renderer_utils.py:192
        E = np.eye(4)
        E[:3] = pose
        np.testing.assert_allclose(np.linalg.inv(E), g["extrinsic"][i], atol=2e-6)


def test_uniforms_from_golden_R_T_are_exact(golden_dir):
    """Feeding the reference's own (R, T) into our Camera reproduces its matrices to fp32 rounding."""
    g = np.load(os.path.join(golden_dir, "stereo_cameras.npz"))
    W, H, fx, fy = int(g["width"]), int(g["height"]), float(g["fx"]), float(g["fy"])
    FoVx = 2 * np.arctan2(W, 2 * fx)
    FoVy = 2 * np.arctan2(H, 2 * fy)
    for i in range(len(g["poses"])):
        cam = Camera(0, g["R_gs_left"][i], g["T_gs_left"][i], FoVx, FoVy, W, H)
        np.testing.assert_array_equal(cam.world_view_transform, g["wvt_left"][i])
        np.testing.assert_array_equal(cam.projection_matrix, g["proj"][i])
        np.testing.assert_allclose(cam.full_proj_transform, g["full_left"][i], rtol=2e-7, atol=1e-7)
        np.testing.assert_allclose(cam.camera_center, g["center_left"][i], rtol=0, atol=2e-6)


def test_projection_matrix_layout():
    P = getProjectionMatrix(0.01, 100.0, 1.0, 0.8)
    assert P[3, 2] == 1.0 and P[2, 3] < 0 and P[0, 2] == 0 and P[1, 2] == 0
