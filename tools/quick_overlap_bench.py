"""Development aid: does rendering stereo pairs on two handles / two streams overlap on the device?
   python tools/quick_overlap_bench.py --config C2 --pairs 16 --streams 2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, json, time
import numpy as np, torch
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.rasterizer import Rasterizer, camera_from
from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2"); ap.add_argument("--pairs", type=int, default=16)
ap.add_argument("--streams", type=int, default=2); ap.add_argument("--tsdf", type=int, default=1)
ap.add_argument("--blend", type=int, default=4); ap.add_argument("--fuse-prio", type=int, default=0)
ap.add_argument("--rows", type=int, default=1)
a = ap.parse_args()
cfg = synthetic.CONFIGS[a.config]
dev = torch.device("cuda:0")
g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}; gd["raw"] = True
poses = synthetic.ring_poses(a.pairs, cfg.ring_radius, 0, cfg.n_pairs)
Wd, Ht = cfg.width, cfg.height
cams, depths, Es = [], [], []
for p in poses:
    l, r = synthetic.stereo_cameras(p, Wd, Ht, cfg.focal, cfg.focal, cfg.baseline)
    cams.append([camera_from(l), camera_from(r)])
    depths.append(synthetic.sphere_depth_torch(p, Wd, Ht, cfg.focal, cfg.focal, Wd / 2, Ht / 2, cfg.sphere_radius, dev))
    E = np.eye(4); E[:3] = p; Es.append(E)
intr = PinholeCameraIntrinsic(Wd, Ht, cfg.focal, cfg.focal, Wd / 2, Ht / 2)
S = a.streams
Rs, cols, rgbs, streams = [], [], [], []
for j in range(S):
    R = Rasterizer(0); R.set_option(_lib.OPT_EXACT_TILE_CULL, 1); R.set_option(_lib.OPT_BLEND_VARIANT, a.blend); R.set_option(_lib.OPT_TILE_ROWS, a.rows)
    if j == 0: R.pack_sh(gd)
    Rs.append(R)
    cols.append(torch.empty((2, 3, Ht, Wd), dtype=torch.float32, device=dev))
    rgbs.append(torch.empty((2, Ht, Wd, 3), dtype=torch.uint8, device=dev))
    streams.append(torch.cuda.Stream())
fuse_stream = torch.cuda.Stream(priority=a.fuse_prio)
vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=(cfg.tsdf_n // 16) ** 3, device=0)
for j in range(S):
    res = Rs[j].render_views(gd, cams[0], out_color=cols[j], out_rgb8=rgbs[j])
    Rs[j].reserve(cfg.P, 2, Wd, Ht, int(max(res["num_rendered"]) * 1.3))
rendered = [torch.cuda.Event() for _ in range(S)]
fused = [torch.cuda.Event() for _ in range(S)]
def run():
    for i in range(a.pairs):
        j = i % S
        with torch.cuda.stream(streams[j]):
            streams[j].wait_event(fused[j])
            Rs[j].render_views(gd, cams[i], out_color=cols[j], out_rgb8=rgbs[j], sync=False)
            rendered[j].record(streams[j])
        if a.tsdf:
            with torch.cuda.stream(fuse_stream):
                fuse_stream.wait_event(rendered[j])
                vol.integrate(RGBDImage(rgbs[j][0], depths[i], depth_scale=1.0, depth_trunc=cfg.baseline * 20), intr, Es[i],
                              min_depth=cfg.baseline * 4)
                fused[j].record(fuse_stream)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run()
    t_enq = time.perf_counter() - t0     # host time to enqueue everything (no sync inside)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(config=a.config, streams=S, tsdf=a.tsdf, ms_per_pair=1e3 * dt / a.pairs, pairs_per_s=a.pairs / dt,
                          host_enqueue_ms_per_pair=1e3 * t_enq / a.pairs)))
