"""Development aid: cost of parallel.reduce_volume on one GPU (world size 1, collectives forced): the
non-network part of the N-GPU reduction (key union, pack, RCCL calls on one rank, unpack)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from gs2mesh_amd import synthetic
from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
from gs2mesh_amd.parallel import reduce_volume
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
W, H = cfg.width, cfg.height
dev = torch.device("cuda:0")
vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=(cfg.tsdf_n // 16) ** 3, device=0)
intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
img = torch.from_numpy(synthetic.color_pattern(W, H)).to(dev)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 49
full_ring = len(sys.argv) > 3 and sys.argv[3] == "ring"     # views all around the object: the block union of a whole job
for mode, payload, algo in (("reduce_scatter", "f32", "rccl"), ("reduce_scatter", "f32", "rccl"), ("reduce_scatter", "packed", "rccl"),
                            ("reduce_scatter", "packed", "direct"), ("allreduce", "packed", "rccl"), ("allreduce", "f32", "rccl")):
    vol.reset()
    for p in synthetic.ring_poses(n, cfg.ring_radius, 0, n if full_ring else 8 * n):
        d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev)
        E = np.eye(4); E[:3] = p
        vol.integrate(RGBDImage(img, d, depth_trunc=cfg.baseline * 20), intr, E, min_depth=cfg.baseline * 4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = reduce_volume(vol, mode=mode, always_collective=True, payload=payload, algo=algo)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(mode=mode, payload=info["payload"], algo=info["algo"], ms=round(1e3 * dt, 2), blocks=info["n_blocks_union"], MB=round(info["bytes_per_rank"] / 1e6, 1))))
dist.destroy_process_group()
