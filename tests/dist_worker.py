"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo group.  Integrates its shard of
the frames (emulator build of the kernels, CPU tensors), runs the production reduction code
(gs2mesh_amd.parallel.reduce_volume) and dumps the resulting volume."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    from backends import make
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from gs2mesh_amd.parallel import exchange_halo, reduce_volume, shard_range
    from test_tsdf_parity import frames

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = make("emu")
    # mode = "<reduce mode>[+mesh][:payload[:algo[:n_frames]]]"
    parts = mode.split(":")
    mode = parts[0]
    payload = parts[1] if len(parts) > 1 else "auto"
    algo = parts[2] if len(parts) > 2 else "rccl"
    n_frames = int(parts[3]) if len(parts) > 3 else 5
    fake = int(parts[4]) if len(parts) > 4 else 0      # pretend every rank integrated this many more frames (weight bound only)
    frs, K = frames(n_frames, 128, 96, 140.0)
    W, H, fx, fy, cx, cy = K
    lo, hi = shard_range(len(frs), rank, world)
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=2048, lib=be.lib)
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    for d, c, E in frs[lo:hi]:
        vol.integrate(RGBDImage(c, d), intr, E)
    again = mode.endswith("+again")
    mode = mode.replace("+again", "")
    twice = mode.endswith("+twice")
    mode = mode.replace("+twice", "")
    mesh_mode = mode.endswith("+mesh")
    vol.frames_local += fake
    total = n_frames + fake * world
    if payload == "packed" and total > 1023:
        # the bound travels in the gathered header: EVERY rank takes the same decision and raises (nobody waits in a collective)
        try:
            reduce_volume(vol, mode=mode.replace("+mesh", ""), payload=payload, algo=algo)
            raise SystemExit("reduce_volume(payload='packed') accepted %d frames" % total)
        except RuntimeError as e:
            assert "1023" in str(e), e
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), refused=1)
        dist.barrier()
        dist.destroy_process_group()
        return
    info = reduce_volume(vol, mode=mode.replace("+mesh", ""), payload=payload, algo=algo)
    assert info["payload"] == ("f32" if payload == "f32" or total > 1023 else "packed") and info["frames_total"] == total
    assert vol.frames_integrated == total
    # a second reduction must be able to re-use the persistent exchange buffers (same objects, no growth)
    ids = {k: v.data_ptr() for k, v in vol._xbuf.items()}
    extra = {}
    if mesh_mode:
        # owner-side finalisation: halo blocks from the other ranks, then this rank's part of the mesh
        owned_keys = vol.download()[0]
        extra["n_halo"] = exchange_halo(vol, info)
        m = vol.extract_triangle_mesh()
        extra["tri_xyz"] = m.vertices[m.triangles] if len(m.triangles) else np.zeros((0, 3, 3))
        extra["owned_keys"] = owned_keys
        assert info["collectives"] == (2 if info["payload"] == "f32" else 3)
        assert all(vol._xbuf[k].data_ptr() == p for k, p in ids.items())
    if twice:
        try:
            reduce_volume(vol, mode=mode, payload=payload, algo=algo)
            raise SystemExit("a replicated volume was summed again")
        except RuntimeError as e:
            assert "replicated" in str(e), e
        extra["refused_second"] = 1
    if again:
        # second reduction on volumes that now hold halo copies of the other rank's blocks
        info2 = reduce_volume(vol, mode="reduce_scatter", payload=payload, algo=algo)
        k2, t2, w2, c2 = vol.download()
        extra.update(keys2=k2, weight2=w2, rgb2=c2)
        assert info2["n_blocks_union"] == info["n_blocks_union"]
    keys, tsdf, weight, rgb = vol.download()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=keys, tsdf=tsdf, weight=weight, rgb=rgb,
             union=info["n_blocks_union"], owned=np.array(info["owned"]), **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
