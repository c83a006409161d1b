"""Diagnostic: full-size trained_like parity under several rasteriser options (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
import oracle
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
ref = parity.oracle_eye(g, l, W, H)
s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
imgs = {}
for name, opts in [("rows2_cull1_v4", dict(rows=2, cull=1, v=4)), ("rows1_cull0_v4", dict(rows=1, cull=0, v=4)),
                   ("rows1_cull0_v0", dict(rows=1, cull=0, v=0)), ("rows1_cull1_v4", dict(rows=1, cull=1, v=4)),
                   ("rows2_cull0_v4", dict(rows=2, cull=0, v=4)), ("rows2_cull1_v7", dict(rows=2, cull=1, v=7))]:
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, opts["cull"])
    R.set_option(_lib.OPT_TILE_ROWS, opts["rows"])
    R.set_option(_lib.OPT_BLEND_VARIANT, opts["v"])
    res = R.render_views(gd, [camera_from(l)], want_radii=True)
    img = res["color"][0].cpu().numpy()
    imgs[name] = img
    fa = parity.flip_attribution(g, l, W, H, img, ref["color"])
    print(name, res["num_rendered"], {k: fa[k] for k in ("flip_pixels", "max_abs_clean", "unexplained_pixels", "pixels_over_clean_bar")},
          fa["worst_unexplained"][:3], flush=True)
# activated inputs through the operator-level API
R = Rasterizer(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
img, radii = R.forward(dev(g["xyz"]), dev(o), dev(l.world_view_transform), dev(l.full_proj_transform), dev(l.camera_center),
                       dev(np.zeros(3, np.float32)), W, H, l.tanfovx, l.tanfovy, shs=dev(shs), scales=dev(s), rotations=dev(q))
img = img.cpu().numpy()
fa = parity.flip_attribution(g, l, W, H, img, ref["color"])
print("operator_api_activated", {k: fa[k] for k in ("flip_pixels", "max_abs_clean", "unexplained_pixels")}, fa["worst_unexplained"][:3])
print("radii mismatches", int((radii.cpu().numpy() != ref["radii"]).sum()))
a, b = imgs["rows2_cull1_v4"], imgs["rows1_cull0_v0"]
d = np.abs(a - b).max(axis=0)
print("v4 vs v0: max", d.max(), "pixels > 2e-4:", int((d > 2e-4).sum()))
np.save("gpurun_out/diag_img_v4.npy", imgs["rows2_cull1_v4"][:, 700:960, 400:480])
