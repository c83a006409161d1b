"""Diagnostic: full-size trained_like parity, right eye (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import parity
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from
cfg = synthetic.CONFIGS["C2"]
W, H = cfg.width, cfg.height
g = synthetic.trained_like(cfg.P, 4242, cfg.log_s_mu, focal=cfg.focal, ring_radius=cfg.ring_radius)
gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
gd["raw"] = True
pose = synthetic.ring_poses(2, cfg.ring_radius, 3, cfg.n_pairs)[1]
l, r = synthetic.stereo_cameras(pose, W, H, cfg.focal, cfg.focal, cfg.baseline)
for name, cam in (("left", l), ("right", r)):
    ref = parity.oracle_eye(g, cam, W, H)
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    R.set_option(_lib.OPT_TILE_ROWS, 2)
    res = R.render_views(gd, [camera_from(cam)])
    img = res["color"][0].cpu().numpy()
    fa = parity.flip_attribution(g, cam, W, H, img, ref["color"])
    print(name, {k: fa[k] for k in ("flip_pixels", "max_abs_clean", "unexplained_pixels")}, fa["worst_unexplained"], flush=True)
    for wu in fa["worst_unexplained"]:
        y, x = wu["y"], wu["x"]
        print("   gpu", img[:, y, x], "ref", ref["color"][:, y, x])
