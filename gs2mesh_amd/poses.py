"""Pose helpers the stereo renderer needs (host, numpy).

Same names and results as the functions of gs2mesh_utils/transformation_utils.py that
``Renderer`` calls (eul2rotm :83-112, rotm2eul :114-141, RT_from_rot_pos :23-40,
convert_R_T_to_GS :42-63, intrinsic_from_camera_params :65-81, calculate_right_camera_pose
:207-224), so that ``Renderer.cameras`` holds the same dictionaries (Euler angles in degrees, float32
round trips included).  Checked against golden vectors produced by the reference's own functions
(tests/test_pipeline_classes.py, tests/golden/stereo_cameras.npz).
"""
from __future__ import annotations

import numpy as np

_TINY = 1e-7


def _snap(x):
    """Reference's fix_zero: |x| < 1e-7 -> 0."""
    x = np.asarray(x)
    return np.where(np.abs(x) < _TINY, 0, x)


def eul2rotm(angles_deg):
    """R = Rz @ Ry @ Rx from XYZ Euler angles in degrees, float32 like the reference."""
    ax, ay, az = np.radians(angles_deg)
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float32)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float32)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float32)
    return _snap(Rz @ Ry @ Rx)


def rotm2eul(R):
    R = np.asarray(R, dtype=np.float32)
    sy = np.sqrt(R[0, 0] ** 2 + R[1, 0] ** 2)
    if sy >= 1e-6:
        ang = [np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])]
    else:
        ang = [np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), 0]
    return _snap(np.degrees(ang))


def RT_from_rot_pos(rot, pos):
    """camera -> world 4x4 with the y/z axes flipped back (left_camera['extrinsic'])."""
    M = np.eye(4)
    R = eul2rotm(rot)
    R[:, 1:] *= -1
    M[:3, :3] = R
    M[:3, 3] = np.array(pos)
    return M


def convert_R_T_to_GS(rot, pos):
    """(R, T) in the 3DGS Camera convention: R = world->cam rotation TRANSPOSED, T = translation."""
    M = np.zeros((4, 4))
    M[:3, :3] = eul2rotm(rot)
    M[:3, 3] = np.asarray(pos, dtype=np.float32)
    M[3, 3] = 1.0
    W2C = np.linalg.inv(M)
    T = W2C[:3, 3]
    T[1:] *= -1
    R = W2C[:3, :3].transpose()
    R[:, 1:] *= -1
    return R, T


def intrinsic_from_camera_params(p):
    return np.array([[p['fx'], 0, p['cx']], [0, p['fy'], p['cy']], [0, 0, 1]])


def calculate_right_camera_pose(R_left, T_left, baseline):
    shift = eul2rotm(R_left) @ np.array([baseline, 0, 0], dtype=np.float32)
    T_right = np.array(T_left, dtype=np.float32) + shift
    return tuple(np.asarray(R_left).tolist()), tuple(_snap(T_right).tolist())


def qvec2rotmat(q):
    """COLMAP quaternion (w, x, y, z) -> rotation matrix."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])
