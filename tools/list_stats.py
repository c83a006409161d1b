"""Instance-list statistics of a configuration at the bench's binning shape (16 x 32 tiles, exact cull as the pipeline picks it):
size histogram of the per-tile lists and the share of lists / keys per sort size class.

    python tools/list_stats.py --configs C2,C3 [--scene trained]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json
import numpy as np
import torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from, auto_cull_level

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3")
ap.add_argument("--scene", default="synth", choices=["synth", "trained"])
a = ap.parse_args()
EDGES = [0, 1, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 1 << 30]
for cname in a.configs.split(","):
    cfg = synthetic.CONFIGS[cname]
    g = (synthetic.trained_like if a.scene == "trained" else synthetic.synth_v1)(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    gd["raw"] = True
    p = synthetic.ring_poses(2, cfg.ring_radius, 0, cfg.n_pairs)[0]
    l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, auto_cull_level(cfg.P))
    R.set_option(_lib.OPT_TILE_ROWS, 2)
    packed = cfg.P >= 32768
    (R.pack_model if packed else R.pack_sh)(gd)
    out = torch.empty((2, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
    res = R.render_views(gd, [camera_from(l), camera_from(r)], out_color=out)
    n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 31) // 32)
    n = int(res["num_rendered"][0])
    pl, ranges = R.download_binning(0, n, n_tiles)
    sz = (ranges[:, 1].astype(np.int64) - ranges[:, 0].astype(np.int64))
    hist_l, _ = np.histogram(sz, bins=EDGES)
    hist_k, _ = np.histogram(sz, bins=EDGES, weights=sz)
    print(json.dumps(dict(config=cname, scene=a.scene, lists=int(len(sz)), keys=int(sz.sum()), mean=float(sz.mean()), max=int(sz.max()),
                          edges=EDGES[:-1], lists_per_bin=hist_l.tolist(), keys_per_bin=[int(x) for x in hist_k])), flush=True)
    R.close()
