"""Phase breakdown of gs2mesh_amd.parallel.reduce_volume at world size 1 (RCCL calls issued on one GPU): which part of the
exchange is data movement and which is host round trips.  Run on the GPU box:
    python tools/reduce_breakdown.py [--config C2] [--frames 20] [--repeats 7]
Prints one JSON object: median milliseconds per phase (device drained at every phase boundary) and the un-instrumented time."""
import argparse
import json
import os
import socket
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def install_legacy_keys():
    """the round-3 key exchange (mask select, shift / stack key arithmetic, scalar header writes, two status reads) for an A/B on
    the same box: `--legacy-keys`"""
    import torch
    import torch.distributed as dist
    import gs2mesh_amd.parallel as P

    def pack_keys(keys):
        k = keys.to(torch.int64) + (1 << 20)
        return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]

    def unpack_keys(u):
        out = torch.stack([(u >> 42) & 0x1FFFFF, (u >> 21) & 0x1FFFFF, u & 0x1FFFFF], dim=1) - (1 << 20)
        return out.to(torch.int32).contiguous()

    def lex_unique(keys):
        return keys if keys.shape[0] == 0 else unpack_keys(torch.unique(pack_keys(keys), sorted=True))

    def canonical_keys(volume, group=None, always_collective=False, marks=None):
        keys = P._as_tensor(volume.block_keys(raise_on_overflow=False))
        _, _, ov = volume.status(raise_on_overflow=False)
        world = dist.get_world_size(group)
        K = int(volume.max_blocks)
        n_local = int(keys.shape[0])
        buf = volume.exchange_buffer("keys_send", (K + 2, 3), torch.int32, keys.device)
        buf.fill_(P._SENTINEL)
        buf[:n_local] = keys
        buf[K, 0] = n_local
        buf[K, 1] = int(ov)
        buf[K, 2] = K
        buf[K + 1, 0] = int(volume.frames_local)
        buf[K + 1, 1] = int(volume.frames_base)
        buf[K + 1, 2] = int(bool(volume.replicated))
        gathered = volume.exchange_buffer("keys_recv", (world * (K + 2), 3), torch.int32, keys.device)
        P._mark(marks, "keys: local list + header")
        dist.all_gather_into_tensor(gathered, buf, group=group)
        P._mark(marks, "keys: all_gather")
        g = gathered.view(world, K + 2, 3)
        head = g[:, K:, :].cpu()
        ov_any = 0
        for f in head[:, 0, 1].tolist():
            ov_any |= int(f)
        frames_total = int(head[:, 1, 0].sum()) + int(head[:, 1, 1].max())
        body = g[:, :K, :].reshape(-1, 3)
        valid = body[:, 0] != P._SENTINEL
        return lex_unique(body[valid]), ov_any, 1, frames_total, False

    P.canonical_keys = canonical_keys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys-via", default="map", choices=["map", "gather"], help="key exchange: block map + all_reduce(MAX) (round 5) or gathered key lists (round 4)")
    ap.add_argument("--legacy-keys", action="store_true")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--payload", default="auto")
    ap.add_argument("--algo", default="rccl")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from gs2mesh_amd import synthetic
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from gs2mesh_amd.parallel import reduce_volume

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    real_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    cfg = synthetic.CONFIGS[args.config]
    W, H = cfg.width, cfg.height
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0)
    poses = synthetic.ring_poses(args.frames, cfg.ring_radius, first=0, total=args.frames)
    rng = np.random.default_rng(3)
    col = torch.from_numpy(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).to(dev)
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=(cfg.tsdf_n // 16) ** 3, device=0)

    def fill():
        vol.reset()
        ims, Es = [], []
        for p in poses:
            d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0, cfg.sphere_radius, dev)
            E = np.eye(4)
            E[:3] = p
            ims.append(RGBDImage(col, d, depth_scale=1.0, depth_trunc=cfg.baseline * 20))
            Es.append(E)
        vol.integrate_batch(ims, intr, Es, min_depth=cfg.baseline * 4)
        vol.status()

    if args.legacy_keys:
        install_legacy_keys()
    fill()
    reduce_volume(vol, always_collective=True, payload=args.payload, algo=args.algo, keys_via=args.keys_via)      # communicator + buffers
    phases, plain = {}, []
    info = None
    for it in range(args.repeats):
        fill()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = reduce_volume(vol, always_collective=True, payload=args.payload, algo=args.algo, keys_via=args.keys_via)
        torch.cuda.synchronize()
        plain.append(time.perf_counter() - t0)
        fill()
        marks = []
        reduce_volume(vol, always_collective=True, payload=args.payload, algo=args.algo, marks=marks, keys_via=args.keys_via)
        for (_, ta), (name, tb) in zip(marks[:-1], marks[1:]):
            phases.setdefault(name, []).append(tb - ta)
    out = dict(keys="legacy (round 3)" if args.legacy_keys else ("block map (round 5)" if args.keys_via == "map" else "gathered lists (round 4)"), config=args.config, frames=args.frames, union_blocks=int(info["n_blocks_union"]), bytes_per_rank=int(info["bytes_per_rank"]),
               payload=info["payload"], algo=info["algo"], collectives=int(info["collectives"]),
               plain_ms=round(1e3 * statistics.median(plain), 4),
               phases_ms={k: round(1e3 * statistics.median(v), 4) for k, v in phases.items()},
               note="phases: device drained at every boundary (their sum exceeds plain_ms by the extra synchronisations)")
    print(json.dumps(out), file=real_out)
    real_out.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
