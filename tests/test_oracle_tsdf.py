"""Analytic known-answer tests for the TSDF oracle (restated Open3D 0.17
ScalableTSDFVolume::Integrate).  PARITY UNPINNED against the real Open3D (absent here and
in the reference tree); these pin the restatement to the published algorithm's maths."""
import numpy as np

import oracle
from gs2mesh_amd import synthetic


def _plane_case(D=1.0, W=160, H=120, f=150.0, voxel=1.0 / 64, trunc=0.06):
    depth = np.full((H, W), D, np.float32)
    color = synthetic.color_pattern(W, H)
    E = np.eye(4)
    vol = oracle.ScalableTSDFVolume(voxel, trunc, 1)
    n = vol.integrate(depth, color, W, H, f, f, W / 2 - 0.5, H / 2 - 0.5, E)
    return vol, n, depth, color, (W, H, f, voxel, trunc, D)


def test_plane_tsdf_values():
    vol, n, depth, color, (W, H, f, voxel, trunc, D) = _plane_case()
    keys, tsdf, weight, col = vol.export()
    assert n == keys.shape[0] == vol.num_blocks and n > 0
    L = voxel * 16
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    mult = oracle.dist_multiplier(W, H, f, f, cx, cy)
    checked = 0
    for b in range(keys.shape[0]):
        org = keys[b].astype(np.float64) * L
        x, y, z = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij")
        pw = np.stack([org[0] + (x + .5) * voxel, org[1] + (y + .5) * voxel, org[2] + (z + .5) * voxel], -1)
        pz = pw[..., 2]
        u_f = pw[..., 0] * f / np.where(pz > 0, pz, 1) + cx + 0.5
        v_f = pw[..., 1] * f / np.where(pz > 0, pz, 1) + cy + 0.5
        ok = (pz > 0) & (u_f >= 1e-4) & (u_f < W - 1e-4) & (v_f >= 1e-4) & (v_f < H - 1e-4)
        u = np.clip(u_f.astype(int), 0, W - 1)
        v = np.clip(v_f.astype(int), 0, H - 1)
        sdf = (D - pz) * mult[v, u]
        upd = ok & (sdf > -trunc)
        exp_t = np.where(upd, np.minimum(1.0, sdf / trunc), 0.0).reshape(-1)
        exp_w = upd.astype(np.float32).reshape(-1)
        # voxels within float rounding of a decision boundary may legitimately differ: mask them out
        # (sdf/trunc near -1, or projection within 1e-3 px of an image edge / pixel edge)
        frag = (np.abs(sdf / trunc + 1) < 1e-3) | (np.abs(u_f - np.round(u_f)) < 1e-3) | \
               (np.abs(v_f - np.round(v_f)) < 1e-3)
        m = ~frag.reshape(-1)
        np.testing.assert_array_equal(weight[b][m], exp_w[m])
        np.testing.assert_allclose(tsdf[b][m], exp_t[m], atol=2e-4)
        # colour of updated voxels = colour of the pixel they project to
        cexp = color[v, u].reshape(-1, 3).astype(np.float64)
        mm = m & (exp_w > 0)
        np.testing.assert_array_equal(col[b][mm], cexp[mm])
        checked += int(mm.sum())
    assert checked > 10_000


def test_touched_blocks_are_the_trunc_box_of_strided_points():
    vol, n, depth, color, (W, H, f, voxel, trunc, D) = _plane_case()
    keys, *_ = vol.export()
    L = voxel * 16
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    exp = set()
    for i in range(0, H, 4):
        for j in range(0, W, 4):
            p = np.array([(j - cx) * D / f, (i - cy) * D / f, D])
            lo = np.floor((p - trunc) / L).astype(int)
            hi = np.floor((p + trunc) / L).astype(int)
            for bx in range(lo[0], hi[0] + 1):
                for by in range(lo[1], hi[1] + 1):
                    for bz in range(lo[2], hi[2] + 1):
                        exp.add((bx, by, bz))
    got = set(map(tuple, keys.tolist()))
    assert got == exp


def test_running_mean_and_view_order_invariance():
    """Two frames: weight 2 where both see the voxel; result independent of frame order up to fp32."""
    cfg = synthetic.CONFIGS["C1"]
    W, H, f = 200, 150, 200.0
    cx, cy = W / 2, H / 2
    poses = synthetic.ring_poses(8, cfg.ring_radius)[:2]
    Es, deps = [], []
    for p in poses:
        E = np.eye(4)
        E[:3] = p
        Es.append(E)
        deps.append(synthetic.sphere_depth(p, W, H, f, f, cx, cy, cfg.sphere_radius))
    col = synthetic.color_pattern(W, H)
    res = []
    for order in ((0, 1), (1, 0)):
        vol = oracle.ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, 1)
        for k in order:
            vol.integrate(deps[k], col, W, H, f, f, cx, cy, Es[k])
        keys, tsdf, weight, color = vol.export()
        d = {tuple(k): (tsdf[i], weight[i], color[i]) for i, k in enumerate(keys.tolist())}
        res.append(d)
    assert set(res[0]) == set(res[1])
    wmax = 0
    for k in res[0]:
        t0, w0, c0 = res[0][k]
        t1, w1, c1 = res[1][k]
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_allclose(t0, t1, atol=2e-7)
        np.testing.assert_allclose(c0, c1, atol=1e-9)
        wmax = max(wmax, w0.max())
    assert wmax == 2.0


def test_depth_conversion_scale_and_trunc():
    d = np.array([[0.5, 1.0, 2.0, 3.0]], np.float32)
    out = oracle.ScalableTSDFVolume.convert_depth(d, depth_scale=0.5, depth_trunc=4.0)
    np.testing.assert_array_equal(out, np.array([[1.0, 2.0, 0.0, 0.0]], np.float32))   # >= trunc -> 0


def test_invalid_depth_touches_nothing():
    vol = oracle.ScalableTSDFVolume(1 / 64, 0.06, 1)
    n = vol.integrate(np.zeros((40, 40), np.float32), None, 40, 40, 50, 50, 20, 20, np.eye(4))
    assert n == 0 and vol.num_blocks == 0
