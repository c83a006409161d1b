"""Minimal readers for the COLMAP *text* model the renderer needs (sparse/0/images.txt, cameras.txt).
Replaces the vendored read_write_model.py (gs2mesh_utils/third_party/colmap_runner/utils/
read_write_model.py:101,193) + poses_from_file (gs2mesh_utils/colmap_utils.py:26-42) for this path."""
from __future__ import annotations

from collections import namedtuple

import numpy as np

from .poses import qvec2rotmat

ColmapCamera = namedtuple("ColmapCamera", "id model width height params")


def read_cameras_text(path):
    cams = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            e = line.split()
            cams[int(e[0])] = ColmapCamera(int(e[0]), e[1], int(e[2]), int(e[3]), np.array(list(map(float, e[4:]))))
    return cams


def read_image_poses_text(path):
    """-> (ids sorted ascending, poses [N,3,4] world->camera, names)."""
    recs = {}
    with open(path) as f:
        lines = [ln.strip() for ln in f]
    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        if not ln or ln.startswith("#"):
            continue
        e = ln.split()
        q = np.array(list(map(float, e[1:5])))
        t = np.array(list(map(float, e[5:8])))
        recs[int(e[0])] = (q, t, e[9] if len(e) > 9 else "")
        i += 1          # the 2D-points line
    ids = sorted(recs)
    poses = np.stack([np.concatenate([qvec2rotmat(recs[k][0]), recs[k][1][:, None]], axis=1) for k in ids])
    return ids, poses, [recs[k][2] for k in ids]


def poses_from_file(extrinsic_file):
    """colmap_utils.py:26-42: [N,3,4] world->camera, sorted by image id."""
    return read_image_poses_text(extrinsic_file)[1]
