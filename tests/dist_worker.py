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
    opts = set(filter(None, os.environ.get("GS2M_TEST_OPTS", "").split(",")))
    if "window=small" in opts:
        # a 2^3-block window: most of the scene's blocks lie outside it -> every rank takes the gather path together
        vol.set_exchange_window((-1, -1, -1), (2, 2, 2))
    if "window=mismatch" in opts and rank == 1:
        vol.set_exchange_window((-31, -32, -32), (64, 64, 64))      # same size, other origin
    if "window=mismatch_size" in opts and rank == 1:
        # another SIZE: the block-map buffers would differ in length -- refused by the fixed-size agreement collective before the
        # map all_reduce is entered (mismatched tensors in a collective hang or are undefined under RCCL)
        vol.set_exchange_window((-32, -32, -32), (64, 32, 64))
    keys_via = "gather" if "keys=gather" in opts else "map"
    if "badpack" in opts:
        # state injected through the C API with a LYING frame bound on rank 0 only: one voxel weight of 2000 does not fit the
        # packed form.  Only rank 0's pack kernel sees it; the verdict must reach every rank (the others used to
        # unpack the corrupted sums and walk into the next collective alone).
        import torch
        from gs2mesh_amd import _lib
        if rank == 0:
            k0 = torch.tensor([[0, 0, 0]], dtype=torch.int32)
            b0 = torch.zeros((1, 5, 4096), dtype=torch.float32)
            b0[0, 1, 7] = 2000.0
            vol.unpack(k0, _lib.XFORM_RAW_F32, b0, frames=1)
        try:
            reduce_volume(vol, mode="reduce_scatter", payload="packed", algo=algo)
            raise SystemExit(f"rank {rank}: reduce_volume accepted a voxel that does not fit the packed form")
        except RuntimeError as e:
            assert "packed exchange form" in str(e), e
        dist.barrier()          # every rank got here: nobody is stuck in a collective
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), refused=1)
        dist.destroy_process_group()
        return
    if "checkpoint" in opts:
        # the SAME state injected on every rank (one checkpoint loaded everywhere) with an honest bound of 600 frames: the
        # bounds ADD UP across ranks (2 x 600 > 1023), so "auto" must take the f32 payload -- with max() over the ranks (round 4)
        # it took the packed form and the 10-bit weight field carried into the colour sum
        import torch
        from gs2mesh_amd import _lib
        k0 = torch.tensor([[0, 0, 0]], dtype=torch.int32)
        b0 = torch.zeros((1, 5, 4096), dtype=torch.float32)
        b0[0, 1, 7] = 600.0
        b0[0, 0, 7] = 0.25
        b0[0, 2, 7] = 600.0 * 200
        vol.unpack(k0, _lib.XFORM_RAW_F32, b0, frames=600)
        info = reduce_volume(vol, mode="allreduce", payload="auto", algo=algo)
        assert info["payload"] == "f32" and info["frames_total"] >= 600 * world, info
        keys, tsdf, weight, rgb = vol.download()
        i = [tuple(k) for k in keys.tolist()].index((0, 0, 0))
        flat = weight[i].reshape(-1)
        assert flat.max() >= 600.0 * world and int(rgb[i].reshape(3, -1)[0].max()) >= 600 * 200 * world
        dist.barrier()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), refused=1)
        dist.destroy_process_group()
        return
    if "window=mismatch" in opts or "window=mismatch_size" in opts:
        try:
            reduce_volume(vol, mode="reduce_scatter", payload=payload, algo=algo)
            raise SystemExit("mismatching exchange windows were accepted")
        except RuntimeError as e:
            assert "exchange window" in str(e), e
        dist.barrier()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), refused=1)
        dist.destroy_process_group()
        return
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
    info = reduce_volume(vol, mode=mode.replace("+mesh", ""), payload=payload, algo=algo, keys_via=keys_via)
    if "window=small" in opts:
        assert info["collectives"] >= 3          # bitmap (says: outside) + gather + payload
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
        # bitmap all_reduce + the payload collective(s) + (packed form) the 4-byte verdict on the pack overflow flag
        assert info["collectives"] == (2 if info["payload"] == "f32" else 4) + (1 if "window=small" in opts else 0)
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
