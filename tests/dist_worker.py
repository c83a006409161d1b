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
    frs, K = frames(5, 128, 96, 140.0)
    W, H, fx, fy, cx, cy = K
    lo, hi = shard_range(len(frs), rank, world)
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=2048, lib=be.lib)
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    for d, c, E in frs[lo:hi]:
        vol.integrate(RGBDImage(c, d), intr, E)
    mesh_mode = mode.endswith("+mesh")
    info = reduce_volume(vol, mode=mode.replace("+mesh", ""))
    extra = {}
    if mesh_mode:
        # owner-side finalisation: halo blocks from the other ranks, then this rank's part of the mesh
        owned_keys = vol.download()[0]
        extra["n_halo"] = exchange_halo(vol, info)
        m = vol.extract_triangle_mesh()
        extra["tri_xyz"] = m.vertices[m.triangles] if len(m.triangles) else np.zeros((0, 3, 3))
        extra["owned_keys"] = owned_keys
        assert info["collectives"] == 2
    keys, tsdf, weight, rgb = vol.download()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=keys, tsdf=tsdf, weight=weight, rgb=rgb,
             union=info["n_blocks_union"], owned=np.array(info["owned"]), **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
