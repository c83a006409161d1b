"""In-kernel phase profile of the compositing kernel (GS2M_OPT_BLEND_PROFILE: s_memtime stamps per wave, raster_blend.h PROF = 1)
at the bench's launch shape (2 stereo pairs per launch, 16 x 32 binning tiles, exact tile cull) -> one JSON object per config:
shares of the wave cycles per phase, staged / listed instance counts, cycles per staged instance.
    python tools/blend_profile.py --configs C2,C3 > profiles/r5_blend_cycles.json"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json
import torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3")
ap.add_argument("--groups", type=int, default=6)
a = ap.parse_args()
out = {}
for cname in a.configs.split(","):
    cfg = synthetic.CONFIGS[cname]
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    gd["raw"] = True
    poses = synthetic.ring_poses(2 * a.groups, cfg.ring_radius, 0, cfg.n_pairs)
    cams = []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
    groups = [cams[i] + cams[i + 1] for i in range(0, len(cams), 2)]
    img = torch.empty((4, 3, cfg.height, cfg.width), dtype=torch.float32, device="cuda")
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    R.set_option(_lib.OPT_TILE_ROWS, 2)
    R.set_option(_lib.OPT_PAIR_BATCH, 2)
    R.set_option(_lib.OPT_BLEND_MODE, 2)
    if cfg.P >= 32768:
        R.pack_model(gd)
    else:
        R.pack_sh(gd)
    res = R.render_views(gd, groups[0], out_color=img)
    R.reserve(cfg.P, 4, cfg.width, cfg.height, int(max(res["num_rendered"]) * 1.5))
    ref = img.clone()
    # plain kernel: stage time
    R.set_option(_lib.OPT_STAGE_TIMING, 1)
    for grp in groups:
        R.render_views(gd, grp, out_color=img, sync=False)
    st = R.stage_times()
    plain_us = 1e3 * st["blend"][0] / max(st["blend"][1], 1) / 2
    # instrumented kernel
    R.set_option(_lib.OPT_BLEND_PROFILE, 1)
    R.render_views(gd, groups[0], out_color=img)
    same = bool(torch.equal(img, ref))
    R.blend_cycles()
    R.stage_times()
    for grp in groups:
        R.render_views(gd, grp, out_color=img, sync=False)
    st = R.stage_times()
    prof_us = 1e3 * st["blend"][0] / max(st["blend"][1], 1) / 2
    c = R.blend_cycles()
    tot = max(c["wave_cycles"], 1)
    phases = ("dma_wait", "staging", "prefetch_issue", "loop", "epilogue")
    out[cname] = dict(
        blend_us_per_pair_plain=round(plain_us, 2), blend_us_per_pair_instrumented=round(prof_us, 2), image_identical=same,
        waves=c["waves"], batches=c["batches"], listed_instances=c["listed_instances"], staged_instances=c["staged_instances"],
        staged_over_listed=round(c["staged_instances"] / max(c["listed_instances"], 1), 4),
        share={k: round(c[k] / tot, 4) for k in phases},
        unaccounted_share=round(1.0 - sum(c[k] for k in phases) / tot, 4),
        cycles_per_wave=round(tot / max(c["waves"], 1), 1),
        wave_cycles_per_staged_instance={k: round(c[k] / max(c["staged_instances"], 1), 2) for k in ("wave_cycles", "loop", "staging", "dma_wait")},
        note="shader cycles (s_memtime) summed over the waves of %d launches of 2 stereo pairs; a wave shares its SIMD with 6 others, "
             "so wave cycles are residency, not issue time" % len(groups))
    R.close()
print(json.dumps(out, indent=1))
