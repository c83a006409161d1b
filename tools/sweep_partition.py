"""Sweep of the CU partition / stream layout of RenderFusePipeline on the bench workload (development aid; bench.py is
the contract).  One process, inputs built once; every configuration runs the driver's job shape (K steps, warm-up,
barrier + synchronize on both sides), median of the repeats.

    python tools/sweep_partition.py --config C2 --steps 20 --configs "0:all:all:2,224:all:all:2,..."
    configuration = blend_cus:bin_cus:fuse_cus:blend_streams[:inflight[:blend_wg_per_cu[:plain_blend_stream]]]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import json
import statistics
import time

import numpy as np
import torch

from gs2mesh_amd import synthetic
from gs2mesh_amd.integration import PinholeCameraIntrinsic, ScalableTSDFVolume
from gs2mesh_amd.pipeline import RenderFusePipeline
from gs2mesh_amd.rasterizer import camera_from

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--repeats", type=int, default=12)
ap.add_argument("--inflight", type=int, default=6)
ap.add_argument("--no-fuse", action="store_true")
ap.add_argument("--configs", default="0:all:all:2")
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.config]
dev = torch.device("cuda:0")
K, Wm = a.steps, a.warmup
Wd, Ht = cfg.width, cfg.height
g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
gd["raw"] = True
poses = synthetic.ring_poses(K + Wm, cfg.ring_radius, first=0, total=K + Wm)
cams, depths, Es = [], [], []
for p in poses:
    l, r = synthetic.stereo_cameras(p, Wd, Ht, cfg.focal, cfg.focal, cfg.baseline)
    cams.append([camera_from(l), camera_from(r)])
    depths.append(synthetic.sphere_depth_torch(p, Wd, Ht, cfg.focal, cfg.focal, Wd / 2.0, Ht / 2.0, cfg.sphere_radius, dev))
    E = np.eye(4)
    E[:3] = p
    Es.append(E)
intr = PinholeCameraIntrinsic(Wd, Ht, cfg.focal, cfg.focal, Wd / 2.0, Ht / 2.0)
n_sweeps = max(2, -(-K // 32))
fuse_batch = max(1, -(-K // n_sweeps))
vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=(cfg.tsdf_n // 16) ** 3, device=0)
for spec in a.configs.split(","):
    f = spec.split(":")
    blend_cus, bin_cus, fuse_cus, nbs = int(f[0]), f[1], f[2], int(f[3])
    inflight = int(f[4]) if len(f) > 4 else a.inflight
    wgcap = int(f[5]) if len(f) > 5 else 0
    plain = bool(int(f[6])) if len(f) > 6 else False
    layout = f[7] if len(f) > 7 else "per_slot"
    nbin = int(f[8]) if len(f) > 8 else 1
    pipe = RenderFusePipeline(gd, Wd, Ht, vol, intr, inflight=inflight, device=0, exact_tile_cull=1, blend_variant=4, tile_rows=2,
                              fuse_batch=fuse_batch, blend_cus=blend_cus, blend_streams=nbs, bin_cus=bin_cus, fuse_cus=fuse_cus,
                              blend_wg_per_cu=wgcap, blend_stream_plain=plain, layout=layout, bin_streams=nbin)
    pipe.prepare(cams[0], headroom=2.0)

    def step(i):
        if a.no_fuse:
            pipe.submit(cams[i])
        else:
            pipe.submit(cams[i], depths[i], Es[i], depth_scale=1.0, depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4)

    for i in range(Wm):
        step(i)
    pipe.finish()
    dts = []
    for rep in range(a.repeats):
        vol.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
        pipe.drain()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
        pipe.finish()
    print(json.dumps(dict(spec=spec, inflight=inflight, ms_per_step=round(1e3 * statistics.median(dts) / K, 4),
                          min=round(1e3 * min(dts) / K, 4), max=round(1e3 * max(dts) / K, 4))), flush=True)
    pipe.close()
    del pipe
    torch.cuda.synchronize()
