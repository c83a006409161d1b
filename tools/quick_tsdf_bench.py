"""TSDF-only timing (development aid): distinct vs repeated depth frames, with/without colour."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np, torch
from gs2mesh_amd import synthetic
from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume, TSDFVolumeColorType
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2"); ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--same", type=int, default=0); ap.add_argument("--color", type=int, default=1)
ap.add_argument("--batch", type=int, default=0, help="frames per voxel-stationary sweep (0 = frame by frame)")
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.config]; W, H = cfg.width, cfg.height; dev = torch.device("cuda:0")
poses = synthetic.ring_poses(a.frames, cfg.ring_radius, 0, cfg.n_pairs)
deps = [synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev) for p in poses]
Es = []
for p in poses:
    E = np.eye(4); E[:3] = p; Es.append(E)
col = torch.from_numpy(synthetic.color_pattern(W, H)).to(dev)
vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, TSDFVolumeColorType.RGB8 if a.color else TSDFVolumeColorType.NoColor,
                         max_blocks=(cfg.tsdf_n // 16) ** 3)
intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
def run():
    if a.batch:
        for i0 in range(0, a.frames, a.batch):
            idx = [0 if a.same else i for i in range(i0, min(a.frames, i0 + a.batch))]
            vol.integrate_batch([RGBDImage(col, deps[k], depth_scale=1.0, depth_trunc=cfg.baseline * 20) for k in idx], intr,
                                [Es[k] for k in idx], min_depth=cfg.baseline * 4)
        return
    for i in range(a.frames):
        k = 0 if a.same else i
        vol.integrate(RGBDImage(col, deps[k], depth_scale=1.0, depth_trunc=cfg.baseline * 20), intr, Es[k], min_depth=cfg.baseline * 4)
run(); vol.status(); vol.reset(); vol.set_stage_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = vol.stage_times(); nb, bu, _ = vol.status()
print(json.dumps(dict(same=a.same, color=a.color, us_per_frame=1e6 * dt / a.frames, blocks_per_frame=bu / a.frames,
                      stages={k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in st.items()})))
