import os, sys
sys.path.insert(0, os.getcwd())
import json, torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from
cfg = synthetic.CONFIGS["C2"]
g = synthetic.trained_like(cfg.P, 4242, cfg.log_s_mu, focal=cfg.focal, ring_radius=cfg.ring_radius)
gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}; gd["raw"] = True
poses = synthetic.ring_poses(12, cfg.ring_radius, 0, cfg.n_pairs)
cams = []
for p in poses:
    l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
    cams.append([camera_from(l), camera_from(r)])
groups = [cams[i] + cams[i + 1] for i in range(0, len(cams), 2)]
out = torch.empty((4, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
ref = None
for mode in (0, 2, 0, 2, 0, 2):
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1); R.set_option(_lib.OPT_TILE_ROWS, 2); R.set_option(_lib.OPT_PAIR_BATCH, 2); R.set_option(_lib.OPT_BLEND_MODE, mode)
    R.pack_sh(gd)
    res = R.render_views(gd, groups[0], out_color=out)
    R.reserve(cfg.P, 4, cfg.width, cfg.height, int(max(res["num_rendered"]) * 1.5))
    img = out.clone()
    if ref is None: ref = img
    best = None
    for rep in range(3):
        R.set_option(_lib.OPT_STAGE_TIMING, 1)
        for grp in groups: R.render_views(gd, grp, out_color=out, sync=False)
        st = R.stage_times(); R.set_option(_lib.OPT_STAGE_TIMING, 0)
        b = 1e3 * st["blend"][0] / max(st["blend"][1], 1) / 2
        best = b if best is None else min(best, b)
    print(json.dumps(dict(scene="trained_like C2 size", mode=mode, blend_us_per_pair=round(best, 1), identical=bool(torch.equal(img, ref)), num_rendered=res["num_rendered"][:2])), flush=True)
    R.close()
