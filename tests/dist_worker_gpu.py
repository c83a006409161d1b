"""Worker of the RCCL tests (tests/test_pipeline_overlap.py): one rank of a world_size-N nccl (= RCCL) group on real
GPUs.  Rank r integrates its shard of the frames into its own volume on cuda:r, runs the production reduction
(gs2mesh_amd.parallel.reduce_volume, + exchange_halo and owner-side extraction after a reduce-scatter) and rank 0 checks the
union of the ranks' results against integrating everything on one GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from gs2mesh_amd import synthetic
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from gs2mesh_amd.parallel import exchange_halo, reduce_volume, shard_range

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dev = torch.device(f"cuda:{rank}")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = synthetic.CONFIGS["C1"]
    W, H = cfg.width, cfg.height
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
    poses = synthetic.ring_poses(6, cfg.ring_radius, 0, 6)
    img = torch.from_numpy(synthetic.color_pattern(W, H)).to(dev)

    def fuse(vol, which):
        for p in which:
            d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev)
            E = np.eye(4)
            E[:3] = p
            vol.integrate(RGBDImage(img, d, depth_trunc=cfg.baseline * 20), intr, E)

    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=rank)
    lo, hi = shard_range(len(poses), rank, world)
    fuse(vol, poses[lo:hi])
    torch.cuda.synchronize()
    # mode = "<reduce mode>[:payload[:algo]]"
    parts = mode.split(":")
    mode = parts[0]
    payload = parts[1] if len(parts) > 1 else "auto"
    algo = parts[2] if len(parts) > 2 else "rccl"
    info = reduce_volume(vol, mode=mode, always_collective=True, payload=payload, algo=algo)
    assert info["collectives"] == (2 if info["payload"] == "f32" else 3)
    tri = np.zeros((0, 3, 3))
    if mode == "reduce_scatter":
        exchange_halo(vol, info)
        m = vol.extract_triangle_mesh()
        tri = m.vertices[m.triangles] if len(m.triangles) else tri
    keys, tsdf, weight, rgb = vol.download()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=keys, tsdf=tsdf, weight=weight, rgb=rgb, tri=tri,
             owned=np.array(info["owned"]), union=info["n_blocks_union"])
    dist.barrier()
    if rank == 0:
        ref = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=0)
        fuse(ref, poses)
        kf, tf, wf, cf = ref.download()
        mref = ref.extract_triangle_mesh()
        idx = {tuple(k): i for i, k in enumerate(kf.tolist())}
        seen, tris = set(), []
        for r in range(world):
            z = np.load(os.path.join(out_dir, f"rank{r}.npz"))
            assert int(z["union"]) == len(idx) > 10
            lo_r, hi_r = z["owned"]
            kk = [tuple(k) for k in z["keys"].tolist()]
            sel = np.array([idx[k] for k in kk], dtype=int)
            if mode == "allreduce":
                assert set(kk) == set(idx)
                own = np.ones(len(kk), bool)
            else:
                own = np.array([k not in seen for k in kk])     # halo blocks repeat other ranks' blocks: same content
            seen |= set(kk)
            assert np.array_equal(z["weight"], wf[sel]) and np.array_equal(z["rgb"], cf[sel])
            assert np.abs(z["tsdf"] - tf[sel]).max() <= 2e-6
            tris.append(z["tri"])
        assert seen == set(idx)
        if mode == "reduce_scatter":
            got = np.concatenate(tris, axis=0)
            want = mref.vertices[mref.triangles]
            assert got.shape == want.shape and len(got) > 1000

            # the sharded tsdf differs from the one-GPU one by the order of the sums: triangles matched by nearest neighbour in
            # the 9-D space of their vertices, both ways, every vertex within 2e-6 (as tests/test_dist_gloo.py does)
            from scipy.spatial import cKDTree
            a, b = got.reshape(len(got), 9), want.reshape(len(want), 9)
            _, nn = cKDTree(b).query(a, k=1)
            _, mm = cKDTree(a).query(b, k=1)
            assert np.abs(a - b[nn]).max() < 2e-6 and np.abs(b - a[mm]).max() < 2e-6
        print("RCCL_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
