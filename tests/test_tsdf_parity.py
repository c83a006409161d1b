"""TSDF parity: HIP integrator (through the C ABI / Open3D-shaped front-end) vs the CPU oracle
(restated Open3D 0.17 ScalableTSDFVolume) on identical frames, on both back-ends.

Tolerances (DESIGN.md "Parity"):
  * set of allocated blocks, set of blocks integrated per frame: exact;
  * weight (integer counts stored as f32): exact;
  * tsdf: BIT-EXACT when frames are integrated in the same order (same fp32 running-mean
    update, translation unit compiled without FMA contraction);
  * colour: the HIP path keeps exact integer sums; mean = sum/weight vs upstream's f64 running
    mean: <= 1e-9 on the 0..255 scale.
"""
import numpy as np
import pytest

import oracle
from gs2mesh_amd import synthetic
from gs2mesh_amd.integration import (Image, PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume,
                                     TSDFVolumeColorType)


def frames(n, W, H, f, radius=0.6, ring=3.5, total=16):
    poses = synthetic.ring_poses(n, ring, 0, total)
    cx, cy = W / 2.0, H / 2.0
    out = []
    for k, p in enumerate(poses):
        E = np.eye(4)
        E[:3] = p
        d = synthetic.sphere_depth(p, W, H, f, f, cx, cy, radius)
        c = np.roll(synthetic.color_pattern(W, H), 7 * k, axis=1)
        out.append((d, c, E))
    return out, (W, H, f, f, cx, cy)


def run_both(be, frs, K, voxel, trunc, depth_scale=1.0, depth_trunc=1e9, mask=None, min_depth=0.0, max_blocks=2048,
             color=True):
    W, H, fx, fy, cx, cy = K
    ct = TSDFVolumeColorType.RGB8 if color else TSDFVolumeColorType.NoColor
    vol = ScalableTSDFVolume(voxel, trunc, ct, max_blocks=max_blocks, lib=be.lib)
    ref = oracle.ScalableTSDFVolume(voxel, trunc, int(ct))
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    per_frame = []
    for (d, c, E) in frs:
        rgbd = RGBDImage.create_from_color_and_depth(Image(be.dev(c)), Image(be.dev(d)), depth_scale=depth_scale,
                                                     depth_trunc=depth_trunc, convert_rgb_to_intensity=False)
        before = vol.status()[1]
        vol.integrate(rgbd, intr, E, mask=None if mask is None else be.dev(mask), min_depth=min_depth)
        per_frame.append(vol.status()[1] - before)
        dd = d.copy()
        if mask is not None:
            dd = dd * (mask != 0)
        if min_depth > 0:
            dd = np.where(dd < np.float32(min_depth), 0, dd).astype(np.float32)
        dd = oracle.ScalableTSDFVolume.convert_depth(dd, depth_scale, depth_trunc)
        nref = ref.integrate(dd, c if color else None, W, H, fx, fy, cx, cy, E)
        assert per_frame[-1] == nref, "blocks integrated in this frame"
    return vol, ref


def compare(vol, ref, color=True, tsdf_exact=True):
    keys, tsdf, weight, rgb = vol.download()
    rk, rt, rw, rc = ref.export()
    assert keys.shape[0] == rk.shape[0] > 0
    got = {tuple(k): i for i, k in enumerate(keys.tolist())}
    assert set(got) == set(map(tuple, rk.tolist()))
    order = np.array([got[tuple(k)] for k in rk.tolist()])
    tsdf, weight, rgb = tsdf[order], weight[order], rgb[order]
    np.testing.assert_array_equal(weight, rw)
    if tsdf_exact:
        np.testing.assert_array_equal(tsdf, rt)
    else:
        np.testing.assert_allclose(tsdf, rt, atol=1e-5, rtol=0)
    if color:
        mean = rgb.astype(np.float64) / np.maximum(weight, 1)[..., None]
        np.testing.assert_allclose(mean, rc, atol=1e-9, rtol=0)
    return keys.shape[0]


def test_sphere_frames_match_oracle(backend):
    frs, K = frames(3, 160, 120, 170.0)
    vol, ref = run_both(backend, frs, K, 2.0 / 128, 0.08)
    n = compare(vol, ref)
    assert n > 50
    assert vol.voxel_updates == ref.block_updates * 4096


def test_reference_default_parameters_small_image(backend):
    """voxel 2/512, trunc 0.04 (argument_utils.py:86-87), i.e. the DTU-like setting, on a small frame."""
    frs, K = frames(2, 128, 96, 400.0, radius=0.25)
    vol, ref = run_both(backend, frs, K, 2.0 / 512, 0.04, max_blocks=4096)
    compare(vol, ref)


def test_depth_scale_trunc_mask_and_min_depth_are_fused_identically(backend):
    frs, K = frames(2, 160, 120, 170.0)
    W, H = K[0], K[1]
    mask = np.ones((H, W), np.uint8)
    mask[:, : W // 3] = 0                       # object / occlusion mask (tsdf_utils.py:68-81)
    # TSDF_scale 0.5 -> depth/0.5; depth_trunc cuts the far half of the sphere; min depth cuts the near cap
    vol, ref = run_both(backend, frs, K, 2.0 / 64, 0.16, depth_scale=0.5, depth_trunc=6.9, mask=mask, min_depth=3.0)
    compare(vol, ref)


def test_no_color_volume(backend):
    frs, K = frames(2, 96, 72, 100.0)
    vol, ref = run_both(backend, frs, K, 2.0 / 64, 0.16, color=False)
    compare(vol, ref, color=False)


def test_view_order_only_changes_tsdf_in_the_last_bits(backend):
    frs, K = frames(4, 128, 96, 140.0)
    vol_a, ref = run_both(backend, frs, K, 2.0 / 96, 0.1)
    vol_b, _ = run_both(backend, frs[::-1], K, 2.0 / 96, 0.1)
    ka, ta, wa, ca = vol_a.download()
    kb, tb, wb, cb = vol_b.download()
    ia = {tuple(k): i for i, k in enumerate(ka.tolist())}
    order = np.array([ia[tuple(k)] for k in kb.tolist()])
    np.testing.assert_array_equal(wa[order], wb)           # counts: exact
    np.testing.assert_array_equal(ca[order], cb)           # integer colour sums: exact
    np.testing.assert_allclose(ta[order], tb, atol=5e-7)   # running mean: fp32 reassociation only


def test_empty_depth_and_format_errors(backend):
    vol = ScalableTSDFVolume(1 / 64, 0.06, max_blocks=64, lib=backend.lib)
    intr = PinholeCameraIntrinsic(40, 30, 50, 50, 20, 15)
    d = np.zeros((30, 40), np.float32)
    c = np.zeros((30, 40, 3), np.uint8)
    vol.integrate(RGBDImage(backend.dev(c), backend.dev(d)), intr, np.eye(4))
    assert vol.status() == (0, 0, 0)
    with pytest.raises(RuntimeError, match="Unsupported image format"):
        vol.integrate(RGBDImage(backend.dev(c), backend.dev(d)), PinholeCameraIntrinsic(41, 30, 50, 50, 20, 15),
                      np.eye(4))
    with pytest.raises(RuntimeError, match="Unsupported image format"):
        RGBDImage.create_from_color_and_depth(Image(c[:10]), Image(d), convert_rgb_to_intensity=False)


def test_block_pool_overflow_is_reported(backend):
    frs, K = frames(1, 160, 120, 170.0)
    W, H, fx, fy, cx, cy = K
    vol = ScalableTSDFVolume(2.0 / 128, 0.08, max_blocks=8, lib=backend.lib)
    d, c, E = frs[0]
    vol.integrate(RGBDImage(backend.dev(c), backend.dev(d)), PinholeCameraIntrinsic(W, H, fx, fy, cx, cy), E)
    with pytest.raises(RuntimeError, match="block pool exhausted"):
        vol.status()


def test_pack_unpack_round_trip_and_merge(backend):
    """The multi-GPU exchange primitives: two half-volumes packed on a canonical key list, summed,
    and unpacked equal the volume that integrated all frames (weights / colours exact)."""
    be = backend
    frs, K = frames(4, 128, 96, 140.0)
    voxel, trunc = 2.0 / 96, 0.1
    full, _ = run_both(be, frs, K, voxel, trunc)
    a, _ = run_both(be, frs[:2], K, voxel, trunc)
    b, _ = run_both(be, frs[2:], K, voxel, trunc)
    ka, kb = be.host(a.block_keys()), be.host(b.block_keys())
    canon = np.unique(np.concatenate([ka, kb]), axis=0).astype(np.int32)
    n = canon.shape[0]
    bufs = []
    for v in (a, b):
        buf = be.dev(np.zeros((n, 5, 4096), np.float32))
        v.pack_sum(be.dev(canon), buf)
        be.sync()
        bufs.append(be.host(buf).copy())
    total = bufs[0] + bufs[1]                      # what the RCCL sum-reduction computes
    assert np.array_equal(total[:, 1:], np.rint(total[:, 1:]))   # counts / colour sums: integers, exact in fp32
    merged = ScalableTSDFVolume(voxel, trunc, max_blocks=2048, lib=be.lib)
    merged.unpack_sum(be.dev(canon), be.dev(total))
    km, tm, wm, cm = merged.download()
    kf, tf, wf, cf = full.download()
    im = {tuple(k): i for i, k in enumerate(km.tolist())}
    assert set(im) == set(map(tuple, kf.tolist()))
    order = np.array([im[tuple(k)] for k in kf.tolist()])
    np.testing.assert_array_equal(wm[order], wf)
    np.testing.assert_array_equal(cm[order], cf)
    np.testing.assert_allclose(tm[order], tf, atol=1e-6)


@pytest.mark.parametrize("lo,dim", [((-32, -32, -32), (64, 64, 64)), ((-5, -4, -3), (11, 9, 7)), ((0, -2, -1), (3, 5, 4))])
def test_block_map_key_exchange_primitives(backend, lo, dim):
    """``gs2m_tsdf_block_map`` / ``gs2m_tsdf_map_keys`` (the key exchange of gs2mesh_amd.parallel): the bytewise MAX of two
    ranks' maps gives the sorted set union of their in-window keys in canonical order, the header carries the flags, the frame
    counts per rank and the count; a block outside the window raises the `outside` byte; a too-small key buffer is reported
    through the count, not overrun.  Windows whose cell count is not a multiple of 16 included."""
    be = backend
    frs, K = frames(4, 128, 96, 140.0)
    voxel, trunc = 2.0 / 96, 0.1
    a, _ = run_both(be, frs[:2], K, voxel, trunc)
    b, _ = run_both(be, frs[2:], K, voxel, trunc)
    world = 2
    maps = []
    for r, v in enumerate((a, b)):
        v.set_exchange_window(lo, dim)
        nb = v.map_bytes(world)
        cells = be.dev(np.full(nb, 7, np.uint8))        # garbage in: block_map clears the buffer itself
        v.frames_base = 3 * r
        v.block_map(cells, r, world, flags=2 if r else 1)
        be.sync()
        maps.append(be.host(cells).copy())
    n_cells = (dim[0] * dim[1] * dim[2] + 15) // 16 * 16
    assert maps[0].shape[0] == n_cells + 32 + 8 * world
    red = np.maximum(maps[0], maps[1])                   # what all_reduce(MAX) over uint8 computes
    ka, kb = be.host(a.block_keys()), be.host(b.block_keys())
    allk = np.unique(np.concatenate([ka, kb]), axis=0).astype(np.int64)
    lo_a, dim_a = np.array(lo), np.array(dim)
    inside = np.all((allk >= lo_a) & (allk < lo_a + dim_a), axis=1)
    want = allk[inside].astype(np.int32)                 # np.unique rows: lexicographic (x, y, z) order
    kbuf = be.dev(np.zeros((max(len(want), 1), 3), np.int32))
    head = a.map_keys(be.dev(red), world, kbuf)
    n = int(head[24]) | int(head[25]) << 8 | int(head[26]) << 16 | int(head[27]) << 24
    assert n == len(want)
    assert np.array_equal(be.host(kbuf)[:n], want)
    assert int(head[0]) == int(not inside.all())         # a block outside the window
    assert head[1:5].tolist() == [0, 0, 0, 0] and head[5] == 1 and head[6] == 1      # flags: rank 0 sent bit 0, rank 1 bit 1
    assert all(int(head[8 + i]) + int(head[12 + i]) == 255 for i in range(4))      # max_blocks agrees
    assert all(int(head[16 + i]) + int(head[20 + i]) == 255 for i in range(4))     # the window agrees
    u32 = lambda o: int(head[o]) | int(head[o + 1]) << 8 | int(head[o + 2]) << 16 | int(head[o + 3]) << 24
    assert [u32(32), u32(36), u32(40), u32(44)] == [2, 0, 2, 3]                   # frames_local / frames_base per rank
    if len(want) > 3:
        small = be.dev(np.full((3, 3), -99, np.int32))
        head2 = a.map_keys(be.dev(red), world, small)
        assert (int(head2[24]) | int(head2[25]) << 8) == len(want) & 0xffff and np.array_equal(be.host(small), want[:3])
    # ranks that disagree on the window: the hash bytes no longer complement each other
    b.set_exchange_window((lo[0] + 1, lo[1], lo[2]), dim)
    cells = be.dev(np.zeros(b.map_bytes(world), np.uint8))
    b.block_map(cells, 1, world, flags=0)
    be.sync()
    red2 = np.maximum(maps[0], be.host(cells))
    assert any(int(red2[n_cells + 16 + i]) + int(red2[n_cells + 20 + i]) != 255 for i in range(4))


def test_replace_equals_reset_plus_unpack_and_leaves_clean_slots(backend):
    """``gs2m_tsdf_replace`` (the tail of reduce_volume): same state as reset + unpack, and the slots it did not refill are
    clean -- frames integrated afterwards allocate them and must give the volume a fresh handle gives."""
    from gs2mesh_amd import _lib
    be = backend
    frs, K = frames(6, 128, 96, 140.0)
    voxel, trunc = 2.0 / 96, 0.1
    W, H, fx, fy, cx, cy = K
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    src, _ = run_both(be, frs[:4], K, voxel, trunc)
    keys = np.unique(be.host(src.block_keys()), axis=0).astype(np.int32)
    half = np.ascontiguousarray(keys[: len(keys) // 2])
    buf = be.dev(np.zeros((len(half), 5, 4096), np.float32))
    src.pack(be.dev(half), _lib.XFORM_RAW_F32, buf)
    be.sync()
    vols = []
    for how in ("replace", "reset+unpack"):
        v, _ = run_both(be, frs[:4], K, voxel, trunc)            # a volume with MORE blocks in use than it is about to keep
        if how == "replace":
            v.replace(be.dev(half), _lib.XFORM_RAW_F32, buf, frames=4)
        else:
            v.reset()
            v.unpack(be.dev(half), _lib.XFORM_RAW_F32, buf, frames=4)
        for d, c, E in frs[4:]:
            v.integrate(RGBDImage(be.dev(c), be.dev(d)), intr, E)
        vols.append(v.download())
    (k0, t0, w0, c0), (k1, t1, w1, c1) = vols
    i1 = {tuple(k): i for i, k in enumerate(k1.tolist())}
    assert set(map(tuple, k0.tolist())) == set(i1) and len(k0) > len(half)
    o = np.array([i1[tuple(k)] for k in k0.tolist()])
    assert np.array_equal(t0, t1[o]) and np.array_equal(w0, w1[o]) and np.array_equal(c0, c1[o])


def test_replace_with_a_repeated_key_leaves_no_stale_slot_behind(backend):
    """``gs2m_tsdf_replace`` skips clearing slots [0, n) because its unpack overwrites them -- but a key that repeats
    (or lies outside the key range) takes no slot of its own, so the unpack hands out fewer than n and the slots above its count
    kept the PREVIOUS volume's voxels, which reappeared in the next block allocated there.  The follow-up pass (k_tsdf_clear_gap)
    clears them on the device: frames integrated afterwards must give what reset + unpack of the distinct keys gives."""
    from gs2mesh_amd import _lib
    be = backend
    frs, K = frames(6, 128, 96, 140.0)
    voxel, trunc = 2.0 / 96, 0.1
    W, H, fx, fy, cx, cy = K
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    src, _ = run_both(be, frs[:4], K, voxel, trunc)
    keys = np.unique(be.host(src.block_keys()), axis=0).astype(np.int32)
    half = np.ascontiguousarray(keys[: len(keys) // 2])
    dup = np.ascontiguousarray(np.concatenate([half, half[:3]]))         # three repeated keys: n = len(half) + 3 rows
    buf = be.dev(np.zeros((len(dup), 5, 4096), np.float32))
    src.pack(be.dev(dup), _lib.XFORM_RAW_F32, buf)
    be.sync()
    vols = []
    for how in ("replace-with-repeats", "reset+unpack"):
        v, _ = run_both(be, frs[:4], K, voxel, trunc)                    # more blocks in use than it is about to keep
        if how == "reset+unpack":
            v.reset()
            v.unpack(be.dev(half), _lib.XFORM_RAW_F32, buf[: len(half)], frames=4)
        else:
            v.replace(be.dev(dup), _lib.XFORM_RAW_F32, buf, frames=4)
        assert v.status()[0] == len(half)
        for d, c, E in frs[4:]:
            v.integrate(RGBDImage(be.dev(c), be.dev(d)), intr, E)
        vols.append(v.download())
    (k0, t0, w0, c0), (k1, t1, w1, c1) = vols
    i1 = {tuple(k): i for i, k in enumerate(k1.tolist())}
    assert set(map(tuple, k0.tolist())) == set(i1) and len(k0) > len(half)
    o = np.array([i1[tuple(k)] for k in k0.tolist()])
    assert np.array_equal(t0, t1[o]) and np.array_equal(w0, w1[o]) and np.array_equal(c0, c1[o])
    # chunked injection into ONE volume does not add its bounds up (unpack replaces state)
    v = ScalableTSDFVolume(voxel, trunc, max_blocks=len(keys) + 8, lib=be.lib)
    for part in (half[: len(half) // 2], half[len(half) // 2:]):
        pb = be.dev(np.zeros((len(part), 5, 4096), np.float32))
        v.unpack(be.dev(np.ascontiguousarray(part)), _lib.XFORM_RAW_F32, pb, frames=4)
    assert v.frames_integrated == 4
    # the pool-exhausted case names itself
    small = ScalableTSDFVolume(voxel, trunc, max_blocks=2, lib=be.lib)
    with pytest.raises(RuntimeError, match="block pool exhausted"):
        small.replace(be.dev(half[:3]), _lib.XFORM_RAW_F32, buf[:3], frames=4)


def test_packed_exchange_form_flags_values_that_do_not_fit_its_fields(backend):
    """XFORM_SUM_PACKED holds w in 10 bits and the colour sums in 18: state injected through unpack_sum with larger
    weights (a C-API user, or a volume whose frame bound was lost) must not carry silently between the fields.
    k_tsdf_pack<PACKED> raises status bit 8; `unpack*` without a `frames` bound disqualifies the volume from the packed
    payload (frames_integrated > 1023); with the bound it stays eligible."""
    from gs2mesh_amd import _lib
    be = backend
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=64, lib=be.lib)
    keys = np.array([[0, 0, 0], [1, 0, 0]], np.int32)
    buf = np.zeros((2, 5, 4096), np.float32)
    buf[:, 1] = 5.0          # weights
    buf[:, 0] = 2.5          # wsum
    buf[:, 2:] = 600.0       # colour sums
    vol.unpack_sum(be.dev(keys), be.dev(buf), frames=5)
    assert vol.frames_integrated == 5
    fb, ib = be.dev(np.zeros((2, 1, 4096), np.float32)), be.dev(np.zeros((2, 4096), np.int64))
    vol.pack(be.dev(keys), _lib.XFORM_SUM_PACKED, fb, ib)
    assert vol.status(raise_on_overflow=False)[2] == 0
    p = be.host(ib)
    assert np.all((p & 0x3ff) == 5) and np.all(((p >> 10) & 0x3ffff) == 600) and np.all((p >> 46) == 600)
    buf[1, 1, 7] = 1500.0    # one voxel beyond the 10-bit weight field
    vol.unpack_sum(be.dev(keys), be.dev(buf))            # no bound given
    assert vol.frames_integrated > _lib.XFORM_PACKED_MAX_FRAMES
    vol.pack(be.dev(keys), _lib.XFORM_SUM_PACKED, fb, ib)
    assert vol.status(raise_on_overflow=False)[2] & 8
    with pytest.raises(RuntimeError, match="packed exchange form"):
        vol.status()
    vol.reset()
    assert vol.frames_integrated == 0 and vol.status()[2] == 0


@pytest.mark.parametrize("color,use_mask,n_frames,splits", [(True, False, 7, ((0, 4), (4, 7))), (True, True, 7, ((0, 4), (4, 7))),
                                                            (False, False, 7, ((0, 4), (4, 7))), (True, True, 11, ((0, 11),)),
                                                            (True, False, 11, ((0, 2), (2, 11)))])
def test_batch_integrate_is_bit_identical_to_frame_by_frame(backend, color, use_mask, n_frames, splits):
    """gs2m_tsdf_integrate_batch (voxel-stationary: all frames of the batch in one sweep over the touched blocks) vs the
    per-frame path in the same frame order and vs the oracle: block sets, counts, tsdf and colour sums bit for bit, with
    min-depth, depth scale / truncation and (optionally) per-frame masks; a second batch continues the same volume.  Sweeps of
    11 / 9 frames: blocks touched by more than four frames take several rounds of the frame-lane sweep (k_tsdf_sweep_fl: four
    frames per round, ping-pong LDS buffers), a partial last round included."""
    be = backend
    frs, K = frames(n_frames, 160, 120, 170.0)
    W, H, fx, fy, cx, cy = K
    voxel, trunc = 2.0 / 128, 0.08
    rng = np.random.default_rng(4)
    masks = [(rng.uniform(size=(H, W)) > 0.2).astype(np.uint8) if (use_mask and k % 2 == 0) else None for k in range(len(frs))]
    ct = TSDFVolumeColorType.RGB8 if color else TSDFVolumeColorType.NoColor
    intr = PinholeCameraIntrinsic(W, H, fx, fy, cx, cy)
    kw = dict(depth_scale=1.25, depth_trunc=3.4)
    seq = ScalableTSDFVolume(voxel, trunc, ct, max_blocks=2048, lib=be.lib)
    ref = oracle.ScalableTSDFVolume(voxel, trunc, int(ct))
    for (d, c, E), m in zip(frs, masks):
        seq.integrate(RGBDImage(be.dev(c), be.dev(d), **kw), intr, E, mask=None if m is None else be.dev(m), min_depth=2.9)
        dd = d * (m != 0) if m is not None else d
        dd = np.where(dd < np.float32(2.9), 0, dd).astype(np.float32)
        ref.integrate(oracle.ScalableTSDFVolume.convert_depth(dd, 1.25, 3.4), c if color else None, W, H, fx, fy, cx, cy, E)
    bat = ScalableTSDFVolume(voxel, trunc, ct, max_blocks=2048, lib=be.lib)
    for lo, hi in splits:                           # the second batch continues the running means
        bat.integrate_batch([RGBDImage(be.dev(c), be.dev(d), **kw) for d, c, E in frs[lo:hi]], intr,
                            [E for _, _, E in frs[lo:hi]],
                            masks=[None if m is None else be.dev(m) for m in masks[lo:hi]] if use_mask else None,
                            min_depth=2.9)
    assert bat.status()[1] == seq.status()[1] == ref.block_updates        # blocks integrated, summed over the frames
    ks, ts, ws, cs = seq.download()
    kb, tb, wb, cb = bat.download()
    ib = {tuple(k): i for i, k in enumerate(kb.tolist())}
    assert set(ib) == set(map(tuple, ks.tolist())) and len(ib) > 20
    order = np.array([ib[tuple(k)] for k in ks.tolist()])
    np.testing.assert_array_equal(wb[order], ws)
    np.testing.assert_array_equal(tb[order], ts)
    np.testing.assert_array_equal(cb[order], cs)
    compare(bat, ref, color=color)


def test_round5_entry_points_refuse_bad_arguments(backend):
    """Error behaviour of the new C-ABI functions (block map, replace, device mesh): non-zero return + a message, no crash."""
    import ctypes as C
    from gs2mesh_amd import _lib
    be = backend
    vol = ScalableTSDFVolume(2.0 / 96, 0.1, max_blocks=64, lib=be.lib)
    lib = be.lib
    dim = (C.c_int32 * 3)(4, 4, 4)
    lo = (C.c_int32 * 3)(0, 0, 0)
    bad_dim = (C.c_int32 * 3)(4, 0, 4)
    assert lib.gs2m_tsdf_map_bytes(dim, 2) == 64 + 32 + 16 and lib.gs2m_tsdf_map_bytes(bad_dim, 2) == -1 and lib.gs2m_tsdf_map_bytes(dim, 0) == -1
    cells = be.dev(np.zeros(64 + 32 + 16, np.uint8))
    from gs2mesh_amd.rasterizer import _ptr
    assert lib.gs2m_tsdf_block_map(vol._h, lo, dim, 2, 2, 0, 0, 0, _ptr(cells), None) != 0          # rank >= world
    assert b"block_map" in lib.gs2m_last_error()
    assert lib.gs2m_tsdf_block_map(vol._h, lo, bad_dim, 0, 2, 0, 0, 0, _ptr(cells), None) != 0
    assert lib.gs2m_tsdf_block_map(vol._h, lo, dim, 0, 2, -1, 0, 0, _ptr(cells), None) != 0          # negative frame count
    hdr = (C.c_uint8 * 48)()
    assert lib.gs2m_tsdf_map_keys(vol._h, lo, dim, 2, _ptr(cells), None, 5, hdr, None) != 0           # keys missing
    with pytest.raises(ValueError):
        vol.set_exchange_window((0, 0, 0), (0, 1, 1))
    k = be.dev(np.zeros((1, 3), np.int32))
    assert lib.gs2m_tsdf_replace(vol._h, _ptr(k), 65, _lib.XFORM_RAW_F32, None, None, None) != 0     # more keys than the pool holds
    assert lib.gs2m_tsdf_replace(vol._h, _ptr(k), 1, _lib.XFORM_RAW_F32, None, None, None) != 0      # no buffer
    nv, nt = C.c_int64(-1), C.c_int64(-1)
    assert lib.gs2m_tsdf_extract_mesh(vol._h, None, C.byref(nv), C.byref(nt)) == 0 and nv.value == 0 and nt.value == 0   # empty volume
    assert lib.gs2m_tsdf_extract_mesh(vol._h, None, None, C.byref(nt)) != 0
    assert lib.gs2m_tsdf_mesh_copy(vol._h, None, None, None, None, None) == 0                      # nothing cached: a no-op
    nc = C.c_int64(-1)
    assert lib.gs2m_mesh_cluster(0, None, -1, None, None, None, C.byref(nc)) != 0
    assert lib.gs2m_mesh_cluster(0, None, 0, None, None, None, C.byref(nc)) == 0 and nc.value == 0
    assert lib.gs2m_mesh_cluster(0, None, 3, None, None, None, C.byref(nc)) != 0
