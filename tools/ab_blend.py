"""Development A/B of the compositing kernel's loop forms (GS2M_OPT_BLEND_MODE) at the default launch shape (2 stereo pairs per
launch, 16 x 32 binning tiles, exact tile cull): blend stage time per pair + a checksum of the image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json
import torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3")
ap.add_argument("--modes", default="2,0,2,0", help="GS2M_OPT_BLEND_MODE per pass")
ap.add_argument("--scene", default="synth", choices=["synth", "trained"])
ap.add_argument("--groups", type=int, default=6)
ap.add_argument("--libs", default="", help="comma-separated extra builds of libgs2mesh_amd.so (A/B of two trees in one process); a mode "
                                          "entry `2@1` runs mode 2 on the first of them, `2` / `2@0` on the tree's own build")
a = ap.parse_args()
import ctypes
LIBS = [None] + [_lib.bind(ctypes.CDLL(os.path.abspath(p))) for p in a.libs.split(",") if p]
for cname in a.configs.split(","):
    cfg = synthetic.CONFIGS[cname]
    g = (synthetic.trained_like if a.scene == "trained" else synthetic.synth_v1)(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    gd["raw"] = True
    poses = synthetic.ring_poses(2 * a.groups, cfg.ring_radius, 0, cfg.n_pairs)
    cams = []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
    groups = [cams[i] + cams[i + 1] for i in range(0, len(cams), 2)]
    out = torch.empty((4, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
    ref = None
    torch.cuda.synchronize()
    for mname in a.modes.split(","):
        mode = int(mname.split("@")[0])
        R = Rasterizer(0, lib=LIBS[int(mname.split("@")[1])] if "@" in mname else None)
        R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
        R.set_option(_lib.OPT_TILE_ROWS, 2)
        R.set_option(_lib.OPT_PAIR_BATCH, 2)
        R.set_option(_lib.OPT_BLEND_MODE, mode)
        if cfg.P >= 32768:
            R.pack_model(gd)
        else:
            R.pack_sh(gd)
        res = R.render_views(gd, groups[0], out_color=out)
        R.reserve(cfg.P, 4, cfg.width, cfg.height, int(max(res["num_rendered"]) * 1.5))
        img = out.clone()
        if ref is None:
            ref = img
        same = bool(torch.equal(img, ref))
        dmax = float((img - ref).abs().max())
        best = None
        for rep in range(3):
            R.set_option(_lib.OPT_STAGE_TIMING, 1)
            for grp in groups:
                R.render_views(gd, grp, out_color=out, sync=False)
            st = R.stage_times()
            R.set_option(_lib.OPT_STAGE_TIMING, 0)
            b = 1e3 * st["blend"][0] / max(st["blend"][1], 1) / 2
            best = b if best is None else min(best, b)
        tot = sum(1e3 * ms / max(n, 1) / 2 for ms, n in st.values())
        print(json.dumps(dict(config=cname, scene=a.scene, mode=mname, blend_us_per_pair=round(best, 2), raster_us_per_pair=round(tot, 1),
                              identical_to_first=same, max_abs_vs_first=dmax)), flush=True)
        R.close()
