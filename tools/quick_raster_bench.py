"""Quick rasteriser timing on synthetic configs (development aid; bench.py is the contract)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import json
import time

import numpy as np
import torch

from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--cull", type=int, default=0)
ap.add_argument("--blend", type=int, default=4)
ap.add_argument("--rows", type=int, default=1)
ap.add_argument("--morton", type=int, default=0, help="cells per axis of the Morton pre-sort (0 = original order); order inside a cell stays the original (random) one")
ap.add_argument("--stages", type=int, default=1)
ap.add_argument("--pack", type=int, default=1)
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.config]
g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
if a.morton:
    cells = 1024 if a.morton == 1 else a.morton
    q = np.clip(((g["xyz"] + 1.0) * 0.5 * cells).astype(np.int64), 0, cells - 1)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = np.argsort(code, kind="stable")
    g = {k: np.ascontiguousarray(v[perm]) for k, v in g.items()}
gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
gd["raw"] = True
poses = synthetic.ring_poses(a.pairs, cfg.ring_radius, 0, cfg.n_pairs)
cams = []
for p in poses:
    l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
    cams.append([camera_from(l), camera_from(r)])
R = Rasterizer(0)
R.set_option(_lib.OPT_EXACT_TILE_CULL, a.cull)
R.set_option(_lib.OPT_BLEND_VARIANT, a.blend)
R.set_option(_lib.OPT_TILE_ROWS, a.rows)
if a.pack == 2:
    R.pack_model(gd)
elif a.pack:
    R.pack_sh(gd)
out = torch.empty((2, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
res = R.render_views(gd, cams[0], out_color=out)
print("num_rendered", res["num_rendered"], "mean", float(out.mean()))
torch.cuda.synchronize()
for it in range(a.iters):
    t0 = time.perf_counter()
    for c in cams:
        R.render_views(gd, c, out_color=out, sync=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nr, ov, req = R.status(2)
    if a.stages and it == a.iters - 1:
        R.set_option(_lib.OPT_STAGE_TIMING, 1)
        for c in cams:
            R.render_views(gd, c, out_color=out, sync=False)
        print({k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in R.stage_times().items()})
        R.set_option(_lib.OPT_STAGE_TIMING, 0)
    print(json.dumps(dict(config=a.config, cull=a.cull, blend=a.blend, pairs_per_s=len(cams) / dt,
                          ms_per_pair=1e3 * dt / len(cams), overflow=ov, num_rendered=nr)))
