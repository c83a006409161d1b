"""Render-only loop of one configuration with the pipeline level's own rules (ordered model from 32 768 Gaussians, four pairs per
launch with it, exact-cull level and compositing loop form by model) -- the command behind the kernel stats / PMC of the
trained-like sub-line:   bash tools/pmc_cmd.sh r6g_pmc_C2T tools/render_only.py --config C2 --scene trained
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from, auto_blend_mode, auto_cull_level, auto_spatial_order

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--scene", default="trained", choices=["synth", "trained"])
ap.add_argument("--groups", type=int, default=6)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.config]
g = (synthetic.trained_like(cfg.P, 4242, cfg.log_s_mu, focal=cfg.focal, ring_radius=cfg.ring_radius) if a.scene == "trained"
     else synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu))
gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
gd["raw"] = True
ordered = auto_spatial_order(cfg.P)
ppl = 4 if ordered else 2
poses = synthetic.ring_poses(ppl * a.groups, cfg.ring_radius, 0, cfg.n_pairs)
cams = []
for p in poses:
    l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
    cams.append([camera_from(l), camera_from(r)])
groups = [[c for pair in cams[i:i + ppl] for c in pair] for i in range(0, len(cams), ppl)]
R = Rasterizer(0)
R.set_option(_lib.OPT_EXACT_TILE_CULL, auto_cull_level(cfg.P))
R.set_option(_lib.OPT_TILE_ROWS, 2)
R.set_option(_lib.OPT_PAIR_BATCH, ppl)
mode = auto_blend_mode(gd)
R.set_option(_lib.OPT_BLEND_MODE, mode)
(R.pack_model if ordered else R.pack_sh)(gd)
out = torch.empty((2 * ppl, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
res = R.render_views(gd, groups[0], out_color=out)
R.reserve(cfg.P, 2 * ppl, cfg.width, cfg.height, int(max(res["num_rendered"]) * 1.5))
R.render_views(gd, groups[0], out_color=out)
best = None
for rnd in range(a.rounds):
    R.set_option(_lib.OPT_STAGE_TIMING, 1)
    for grp in groups:
        R.render_views(gd, grp, out_color=out, sync=False)
    tms = R.stage_times()
    R.set_option(_lib.OPT_STAGE_TIMING, 0)
    cur = {s: 1e3 * tms[s][0] / max(tms[s][1], 1) / ppl for s in tms}
    best = cur if best is None else {s: min(best[s], cur[s]) for s in cur}
print(json.dumps(dict(config=a.config, scene=a.scene, ordered_model=bool(ordered), pairs_per_launch=ppl, blend_mode=mode,
                      num_rendered=[int(x) for x in res["num_rendered"]][:2], us_per_pair={k: round(v, 2) for k, v in best.items()},
                      raster_us_per_pair=round(sum(best.values()), 1))))
