"""Disparity -> depth + occlusion mask kernel against golden vectors produced by
executing the reference's own ``Stereo.get_occlusion_mask`` (tests/golden/make_golden.py) and the
depth formula of stereo_utils.py:133, on both back-ends; plus the hand-off into the TSDF."""
import os

import numpy as np

from gs2mesh_amd import stereo_utils


def test_occlusion_mask_matches_the_reference_function(backend, golden_dir):
    g = np.load(os.path.join(golden_dir, "occlusion_mask.npz"))
    L, R = g["L2R"], g["R2L"]
    for thr in (1, 3):
        got = stereo_utils.get_occlusion_mask(backend.dev(L), backend.dev(R), thr, lib=backend.lib)
        got = backend.host(got).astype(bool)
        ref = g[f"mask_thr{thr}"]
        assert got.shape == ref.shape and 0.2 < ref.mean() < 0.98
        np.testing.assert_array_equal(got, ref)          # exact: integer / float64 decisions


def test_depth_is_the_float32_quotient(backend, golden_dir):
    g = np.load(os.path.join(golden_dir, "occlusion_mask.npz"))
    L, R = g["L2R"], g["R2L"]
    fx, baseline = 2892.33, 0.245
    depth, mask = stereo_utils.depth_and_occlusion(backend.dev(L), backend.dev(R), fx, baseline, 3, lib=backend.lib)
    # stereo_utils.py:133 under the reference's numpy (value-based casting): float32(fx*baseline) / float32 disparity
    ref = np.float32(fx * baseline) / L
    np.testing.assert_array_equal(backend.host(depth), ref)
    assert backend.host(mask).dtype == np.uint8 and set(np.unique(backend.host(mask))) <= {0, 1}


def test_outputs_feed_the_tsdf_directly(backend):
    """depth / mask tensors go straight into ScalableTSDFVolume.integrate (in-memory hand-off)."""
    import oracle
    from gs2mesh_amd import synthetic
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    W, H, f, b = 160, 120, 170.0, 0.245
    pose = synthetic.ring_poses(1, 3.5)[0]
    E = np.eye(4)
    E[:3] = pose
    d = synthetic.sphere_depth(pose, W, H, f, f, W / 2.0, H / 2.0, 0.6)
    disp = np.where(d > 0, np.float32(f * b) / np.maximum(d, 1e-6), np.float32(-1.0)).astype(np.float32)
    rl = disp.copy()
    rl[:, : W // 2] += 8.0                              # left half inconsistent -> masked out
    col = synthetic.color_pattern(W, H)
    depth, mask = stereo_utils.depth_and_occlusion(backend.dev(disp), backend.dev(rl), f, b, 3, lib=backend.lib)
    vol = ScalableTSDFVolume(2.0 / 128, 0.08, max_blocks=1024, lib=backend.lib)
    vol.integrate(RGBDImage(backend.dev(col), depth, 1.0, 1e9), PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0), E,
                  mask=mask)
    keys, tsdf, weight, rgb = vol.download()
    dm = backend.host(depth) * (backend.host(mask) != 0)
    ref = oracle.ScalableTSDFVolume(2.0 / 128, 0.08, 1)
    ref.integrate(oracle.ScalableTSDFVolume.convert_depth(dm.astype(np.float32), 1.0, 1e9), col, W, H, f, f, W / 2.0,
                  H / 2.0, E)
    rk, rt, rw, rc = ref.export()
    got = {tuple(k): i for i, k in enumerate(keys.tolist())}
    assert set(got) == set(map(tuple, rk.tolist())) and len(got) > 5
    order = np.array([got[tuple(k)] for k in rk.tolist()])
    np.testing.assert_array_equal(weight[order], rw)
    np.testing.assert_array_equal(tsdf[order], rt)
