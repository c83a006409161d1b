"""RenderFusePipeline (gs2mesh_amd/pipeline.py): pairs overlapped on separate HIP streams give exactly the
images and the volume of the serial single-stream order."""
import numpy as np
import pytest

from gs2mesh_amd import synthetic

pytestmark = pytest.mark.gpu


def _run(inflight, n_views=6, fuse_batch=1, **pipe_kw):
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
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=inflight, device=0, fuse_batch=fuse_batch, **pipe_kw)
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
    pipe.close()
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


@pytest.mark.parametrize("inflight,fuse_batch", [(1, 4), (3, 4), (2, 2), (3, [3, 2]), (1, [2, 3])])
def test_batched_fusion_equals_view_by_view(inflight, fuse_batch):
    """fuse_batch > 1: the views are integrated by the voxel-stationary batch kernel, 4 (then the remaining 2) at a time,
    from per-view copies of the left image: images and volume are bit-identical to the serial view-by-view order."""
    ref = _run(1)
    got = _run(inflight, fuse_batch=fuse_batch)
    for (c0, u0), (c1, u1) in zip(ref[0], got[0]):
        assert np.array_equal(c0, c1) and np.array_equal(u0, u1)
    for a, b in zip(ref[1:], got[1:]):
        assert np.array_equal(a, b)


def _run_grouped(inflight, n_views, fuse_batch, ppl=2):
    """pairs_per_launch = ppl: the same job as `_run`, `ppl` consecutive stereo pairs per chain of launches."""
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
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=inflight, device=0, fuse_batch=fuse_batch, pairs_per_launch=ppl)
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
        if i % ppl == ppl - 1:                             # the group of views i - ppl + 1 .. i has been launched on `slot`
            pipe.wait_rendered(slot)
            c, u = pipe.color[slot].cpu().numpy(), pipe.rgb8[slot].cpu().numpy()
            images += [(c[2 * k:2 * k + 2].copy(), u[2 * k:2 * k + 2].copy()) for k in range(ppl)]
    pipe.finish()                                          # flushes an incomplete last group
    if n_views % ppl:
        c, u = pipe.color[slot].cpu().numpy(), pipe.rgb8[slot].cpu().numpy()
        images += [(c[2 * k:2 * k + 2].copy(), u[2 * k:2 * k + 2].copy()) for k in range(n_views % ppl)]
    keys, tsdf, weight, rgb = vol.download()
    pipe.close()
    order = np.lexsort(keys.T[::-1])
    assert len(images) == n_views
    return images, keys[order], tsdf[order], weight[order], rgb[order]


@pytest.mark.parametrize("n_views,fuse_batch,ppl", [(6, 6, 2), (7, 4, 2), (6, [4, 2], 2), (8, 8, 4), (10, 8, 4), (9, 6, 3)])
def test_two_pairs_per_launch_pipeline_equals_serial(n_views, fuse_batch, ppl):
    """RenderFusePipeline(pairs_per_launch=2): two consecutive stereo pairs per chain of launches (GS2M_OPT_PAIR_BATCH),
    their u8 pairs rendered into consecutive buffers of the pending TSDF batch; an odd last view is flushed by `finish`.
    Images and volume are bit-identical to the serial order."""
    ref = _run(1, n_views=n_views)
    got = _run_grouped(3 if ppl == 2 else 2, n_views, fuse_batch, ppl)
    for (c0, u0), (c1, u1) in zip(ref[0], got[0]):
        assert np.array_equal(c0, c1) and np.array_equal(u0, u1)
    for a, b in zip(ref[1:], got[1:]):
        assert np.array_equal(a, b)


def test_model_updated_orders_the_render_streams_after_the_callers_stream():
    """The render streams wait for the caller's stream once per slot, not per step (pipeline.py): a model changed in place on
    the caller's stream needs `model_updated()` -- packed copies dropped, every render stream re-armed -- and the next renders
    show the new model, bit for bit what a fresh handle renders."""
    import torch
    from gs2mesh_amd.pipeline import RenderFusePipeline
    from gs2mesh_amd.rasterizer import Rasterizer, camera_from
    cfg = synthetic.CONFIGS["C1"]
    dev = torch.device("cuda:0")
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    W, H = cfg.width, cfg.height
    poses = synthetic.ring_poses(4, cfg.ring_radius, 0, 4)
    cams = []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
    pipe = RenderFusePipeline(gd, W, H, None, None, inflight=3, device=0)
    pipe.prepare(cams[0])
    for i in range(4):
        pipe.submit(cams[i])
    # in-place update on the caller's (current) stream while earlier renders may still be in flight on theirs
    pipe.drain()
    gd["xyz"].mul_(0.9)
    gd["opacity"].add_(0.5)
    gd["features_dc"].mul_(0.5)
    pipe.model_updated()
    got = []
    for i in range(4):
        slot = pipe.submit(cams[i])
        pipe.wait_rendered(slot)
        got.append(pipe.color[slot].cpu().numpy().copy())
    pipe.finish()
    pipe.close()
    fresh = Rasterizer(0)
    for i in range(4):
        res = fresh.render_views(gd, cams[i])
        assert np.array_equal(res["color"].cpu().numpy(), got[i]), i


def test_serial_mode_with_two_pairs_per_launch_equals_view_by_view():
    """inflight = 1 + pairs_per_launch = 2 (what tools/profile_round.sh profiles: the timed launch shapes, serial on one
    stream): same images and volume as one pair per launch."""
    ref = _run(1)
    got = _run_grouped(1, 6, 6)
    for (c0, u0), (c1, u1) in zip(ref[0], got[0]):
        assert np.array_equal(c0, c1) and np.array_equal(u0, u1)
    for a, b in zip(ref[1:], got[1:]):
        assert np.array_equal(a, b)


def test_release_fences_the_slots_next_render_behind_the_consumers_reads():
    """`release(slot)`: reads of a slot's images enqueued on the caller's stream are ordered before the slot's
    re-render `inflight` submits later, without a host synchronisation -- a slow consumer kernel chain on a side stream still
    sees the image of ITS view."""
    import torch
    from gs2mesh_amd.pipeline import RenderFusePipeline
    from gs2mesh_amd.rasterizer import Rasterizer, camera_from
    cfg = synthetic.CONFIGS["C1"]
    dev = torch.device("cuda:0")
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    W, H = cfg.width, cfg.height
    poses = synthetic.ring_poses(6, cfg.ring_radius, 0, 6)
    cams = []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
    pipe = RenderFusePipeline(gd, W, H, inflight=2, device=0)
    pipe.prepare(cams[0])
    consumer = torch.cuda.Stream(device=dev)
    big = torch.zeros((64, 1024, 1024), device=dev)
    copies = []
    for i in range(6):
        slot = pipe.submit(cams[i])
        with torch.cuda.stream(consumer):
            pipe.wait_rendered(slot, stream_only=True)      # no host sync: the consumer stream waits for the render
            for _ in range(4):
                big.add_(1.0)                               # a slow consumer in front of the read
            copies.append(pipe.color[slot].clone())
            pipe.release(slot)
    pipe.finish()
    consumer.synchronize()
    fresh = Rasterizer(0)
    fresh.set_option(1, 1)
    fresh.set_option(5, 2)
    for i in range(6):
        res = fresh.render_views(gd, cams[i])
        assert np.array_equal(res["color"].cpu().numpy(), copies[i].cpu().numpy()), i


def test_fuseless_render_does_not_touch_a_pending_views_image():
    """A submit without depth on a slot whose previous view is still pending in a TSDF batch renders into the slot's own
    buffer, not into the pending view's batch buffer (which the later sweep still reads)."""
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
    poses = synthetic.ring_poses(6, cfg.ring_radius, 0, 6)
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2, H / 2)
    out = []
    for extra in (False, True):
        vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=4096, device=0)
        pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=2, device=0, fuse_batch=4)
        subs = []
        for p in poses:
            l, r = synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline)
            E = np.eye(4)
            E[:3] = p
            subs.append(([camera_from(l), camera_from(r)],
                         synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2, H / 2, cfg.sphere_radius, dev), E))
        pipe.prepare(subs[0][0])
        for i, (c, d, E) in enumerate(subs[:4]):
            pipe.submit(c, d, E, depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4)
            if extra and i < 3:
                # fuse-less renders of OTHER views on both slots while views 0..i are pending
                pipe.submit(subs[5][0])
                pipe.submit(subs[4][0])
        pipe.finish()
        keys, tsdf, weight, rgb = vol.download()
        order = np.lexsort(keys.T[::-1])
        out.append((keys[order], tsdf[order], weight[order], rgb[order]))
    for a, b in zip(*out):
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


def _run_rccl(world, mode, tmp_path):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, PYTHONPATH=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "dist_worker_gpu.py"), str(tmp_path), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and "RCCL_OK" in outs[0], "\n".join(o[-3000:] for o in outs)


@pytest.mark.parametrize("mode", ["allreduce", "reduce_scatter", "reduce_scatter:f32", "reduce_scatter:packed:direct",
                                  "allreduce:f32"])
def test_reduce_volume_over_rccl_single_rank(mode, tmp_path):
    """The RCCL calls of parallel.reduce_volume (fixed-size all_gather of the keys + flags, ONE all_reduce / reduce_scatter
    of the packed fp32 accumulators, the all_to_all of the halo exchange) on device tensors, world size 1 (one GPU on the
    box): the volume must come back unchanged up to the tsdf*w/w round trip, and its mesh with it."""
    _run_rccl(1, mode, tmp_path)


@pytest.mark.parametrize("mode", ["allreduce", "reduce_scatter", "reduce_scatter:packed:direct"])
def test_reduce_volume_over_rccl_two_ranks(mode, tmp_path):
    """The same on two GPUs when the box has them: views sharded over 2 ranks, RCCL over xGMI, the union of the ranks'
    volumes / partial meshes equals the single-GPU result (counts and colour sums exact, tsdf within 2e-6)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (one-GPU boxes run the single-rank RCCL test)")
    _run_rccl(2, mode, tmp_path)
