"""RenderFusePipeline (gs2mesh_amd/pipeline.py): pairs overlapped on separate HIP streams give exactly the
images and the volume of the serial single-stream order."""
import numpy as np
import pytest

from gs2mesh_amd import synthetic

pytestmark = pytest.mark.gpu


def _run(inflight, n_views=6):
    import torch
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, ScalableTSDFVolume
    from gs2mesh_amd.pipeline import RenderFusePipeline
    from gs2mesh_amd.rasterizer import camera_from
    cfg = synthetic.CONFIGS["C1"]
    dev = torch.device("cuda:0")
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    W, H = cfg.width, cfg.height
    poses = synthetic.ring_poses(n_views, cfg.ring_radius, 0, n_views)
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=0)
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=inflight, device=0)
    cams, depths, Es = [], [], []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
        depths.append(synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev))
        E = np.eye(4)
        E[:3] = p
        Es.append(E)
    pipe.prepare(cams[0])
    images = []
    for i in range(n_views):
        slot = pipe.submit(cams[i], depths[i], Es[i], depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4)
        pipe.wait_rendered(slot)
        images.append((pipe.color[slot].cpu().numpy().copy(), pipe.rgb8[slot].cpu().numpy().copy()))
    pipe.finish()
    keys, tsdf, weight, rgb = vol.download()
    order = np.lexsort(keys.T[::-1])
    return images, keys[order], tsdf[order], weight[order], rgb[order]


@pytest.mark.parametrize("inflight", [2, 3])
def test_overlapped_pipeline_equals_serial(inflight):
    ref = _run(1)
    got = _run(inflight)
    for (c0, u0), (c1, u1) in zip(ref[0], got[0]):
        assert np.array_equal(c0, c1) and np.array_equal(u0, u1)
    for a, b in zip(ref[1:], got[1:]):
        assert np.array_equal(a, b)


def test_pipeline_steady_state_without_reading_back():
    """No wait between submits (the bench loop): same volume as the serial order."""
    import torch
    ref = _run(1)
    # same as _run(2) but without wait_rendered between submits
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, ScalableTSDFVolume
    from gs2mesh_amd.pipeline import RenderFusePipeline
    from gs2mesh_amd.rasterizer import camera_from
    cfg = synthetic.CONFIGS["C1"]
    dev = torch.device("cuda:0")
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    W, H = cfg.width, cfg.height
    poses = synthetic.ring_poses(6, cfg.ring_radius, 0, 6)
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=0)
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=2, device=0)
    subs = []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline)
        E = np.eye(4)
        E[:3] = p
        subs.append(([camera_from(l), camera_from(r)],
                     synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev), E))
    pipe.prepare(subs[0][0])
    for c, d, E in subs:
        pipe.submit(c, d, E, depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4)
    pipe.finish()
    keys, tsdf, weight, rgb = vol.download()
    order = np.lexsort(keys.T[::-1])
    for a, b in zip(ref[1:], (keys[order], tsdf[order], weight[order], rgb[order])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["allreduce", "reduce_scatter"])
def test_reduce_volume_over_rccl_single_rank(mode):
    """The RCCL calls of parallel.reduce_volume (all_gather of keys, all_reduce / reduce_scatter of the f32 and
    i32 accumulators) on device tensors, world size 1 (one GPU on the box): the volume must come back unchanged
    up to the tsdf*w/w round trip."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent(f"""
        import os, numpy as np, torch, torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        from gs2mesh_amd import synthetic
        from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
        from gs2mesh_amd.parallel import reduce_volume
        cfg = synthetic.CONFIGS["C1"]; W, H = cfg.width, cfg.height
        dev = torch.device("cuda:0")
        vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=0)
        intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
        for p in synthetic.ring_poses(3, cfg.ring_radius, 0, 3):
            d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev)
            E = np.eye(4); E[:3] = p
            img = torch.from_numpy(synthetic.color_pattern(W, H)).to(dev)
            vol.integrate(RGBDImage(img, d, depth_trunc=cfg.baseline * 20), intr, E)
        k0, t0, w0, c0 = vol.download()
        o0 = np.lexsort(k0.T[::-1])
        info = reduce_volume(vol, mode="{mode}", always_collective=True)
        k1, t1, w1, c1 = vol.download()
        o1 = np.lexsort(k1.T[::-1])
        assert np.array_equal(k0[o0], k1[o1]) and np.array_equal(w0[o0], w1[o1]) and np.array_equal(c0[o0], c1[o1])
        assert np.abs(t0[o0] - t1[o1]).max() <= 2e-6
        assert info["n_blocks_union"] == len(k0) > 10
        dist.destroy_process_group()
        print("RCCL_OK")
    """)
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
