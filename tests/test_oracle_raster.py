"""Pin the rasteriser oracle: golden vectors generated from the reference's own Python
(tests/golden/make_golden.py) + analytic known-answer tests for the stages the reference
has no second implementation of (EWA projection, binning, compositing)."""
import math
import os

import numpy as np
import pytest

import oracle
from gs2mesh_amd import synthetic
from gs2mesh_amd.graphics import Camera


def _cam(width=64, height=48, f=60.0, t=(0.0, 0.0, 4.0)):
    """Identity-rotation camera looking down +z, world origin at camera-space t."""
    FoVx = 2 * np.arctan2(width, 2 * f)
    FoVy = 2 * np.arctan2(height, 2 * f)
    return Camera(0, np.eye(3), np.array(t, float), FoVx, FoVy, width, height)


def _pre(cam, xyz, scales, rots, opac, shs, deg=3, **kw):
    return oracle.preprocess(xyz, scales, rots, opac, shs, cam.world_view_transform, cam.full_proj_transform,
                             cam.camera_center, cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy,
                             sh_degree=deg, **kw)


# ---------------------------------------------------------------------------------------
# golden: SH -> RGB  (reference eval_sh, GS/utils/sh_utils.py:57-112)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_to_rgb_matches_reference_eval_sh(golden_dir, deg):
    g = np.load(os.path.join(golden_dir, "sh_rgb.npz"))
    P = g["xyz"].shape[0]
    # a camera whose centre is the golden campos and that has the points in front of it
    campos = g["campos"].astype(np.float64)
    Rw2c = np.diag([-1.0, 1.0, -1.0])          # look down world -z so the points are in front
    cam = Camera(0, Rw2c.T, -Rw2c @ campos, 1.0, 1.0, 64, 64)
    np.testing.assert_allclose(cam.camera_center, campos, atol=1e-6)
    scales = np.full((P, 3), 0.01, np.float32)
    rots = np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1))
    opac = np.full(P, 0.5, np.float32)
    # camera_center from the fp32 inverse may differ from campos in the last ulp: pass campos itself
    out = oracle.preprocess(g["xyz"], scales, rots, opac, g["shs"], cam.world_view_transform,
                            cam.full_proj_transform, g["campos"], 64, 64, cam.tanfovx, cam.tanfovy, sh_degree=deg)
    vis = out["radii"] > 0
    assert vis.sum() > 20           # points with z_view > 0.2 that hit the image
    ref = g[f"rgb_deg{deg}"]
    np.testing.assert_allclose(out["rgb"][vis], ref[vis], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------
# golden: Sigma = R S^2 R^T  (reference build_covariance_from_scaling_rotation)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mod", [1.0, 0.5])
def test_cov3d_matches_reference_python(golden_dir, mod):
    g = np.load(os.path.join(golden_dir, "cov3d.npz"))
    P = g["scales"].shape[0]
    cam = _cam()
    xyz = np.zeros((P, 3), np.float32)      # all at the origin, 4 units in front of the camera
    opac = np.full(P, 0.5, np.float32)
    shs = np.zeros((P, 16, 3), np.float32)
    out = _pre(cam, xyz, g["scales"], g["rots_normalized"], opac, shs, scale_modifier=mod)
    ref = g[f"cov_mod{mod}"]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    np.testing.assert_allclose(out["cov3D"] / scale, ref / scale, rtol=0, atol=3e-6)


# ---------------------------------------------------------------------------------------
# analytic KATs (parity unpinned stages)
# ---------------------------------------------------------------------------------------
def test_single_isotropic_gaussian_projection_kat():
    W, H, f, z = 64, 48, 60.0, 4.0
    cam = _cam(W, H, f, (0, 0, z))
    s = 0.05
    xyz = np.zeros((1, 3), np.float32)
    out = _pre(cam, xyz, np.full((1, 3), s, np.float32), np.array([[1, 0, 0, 0]], np.float32),
               np.array([0.8], np.float32), np.zeros((1, 16, 3), np.float32))
    # centre of the image: ndc 0 -> pix = (S-1)/2
    np.testing.assert_allclose(out["means2D"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-4)
    np.testing.assert_allclose(out["depths"][0], z, atol=1e-6)
    var = (f * s / z) ** 2 + 0.3                        # EWA on axis: J = diag(f/z), + 0.3 low-pass
    np.testing.assert_allclose(out["conic_opacity"][0], [1 / var, 0.0, 1 / var, 0.8], rtol=1e-5, atol=1e-7)
    lam = var + math.sqrt(0.1)                          # lambda = mid + sqrt(max(0.1, mid^2 - det)), det = mid^2
    assert out["radii"][0] == math.ceil(3 * math.sqrt(lam))
    # SH all zero -> colour 0.5
    np.testing.assert_allclose(out["rgb"][0], 0.5, atol=1e-7)
    r = out["radii"][0]
    x0 = max(0, int(((W - 1) / 2 - r) / 16))
    x1 = min(4, int(((W - 1) / 2 + r + 15) / 16))
    assert out["rect"][0, 0] == x0 and out["rect"][0, 2] == x1
    assert out["tiles_touched"][0] == (out["rect"][0, 2] - x0) * (out["rect"][0, 3] - out["rect"][0, 1])


def test_near_plane_and_offscreen_culls():
    cam = _cam()
    xyz = np.array([[0, 0, -3.9],      # z_view = 0.1  -> near culled
                    [0, 0, -3.8 + 1e-3],  # z_view just above 0.2 -> kept
                    [50, 0, 0]], np.float32)  # far off-screen -> zero-area rect
    P = 3
    out = _pre(cam, xyz, np.full((P, 3), 0.01, np.float32), np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32),
               np.full(P, 0.5, np.float32), np.zeros((P, 16, 3), np.float32))
    assert out["radii"][0] == 0 and out["tiles_touched"][0] == 0
    assert out["radii"][1] > 0
    assert out["radii"][2] == 0
    vis = oracle.mark_visible(xyz, cam.world_view_transform, cam.full_proj_transform)
    assert vis.tolist() == [False, True, True]      # markVisible is the near test only


def test_single_gaussian_compositing_kat():
    W, H, f, z = 64, 48, 60.0, 4.0
    cam = _cam(W, H, f, (0, 0, z))
    s, o = 0.08, 0.7
    rgbc = np.array([[0.9, 0.4, 0.1]], np.float32)
    bg = np.array([0.2, 0.3, 0.4], np.float32)
    img, radii, n = oracle.rasterize_forward(
        np.zeros((1, 3), np.float32), np.array([o], np.float32), cam.world_view_transform,
        cam.full_proj_transform, cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, bg,
        colors_precomp=rgbc, scales=np.full((1, 3), s, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32))
    var = (f * s / z) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs - (W - 1) / 2) ** 2 + (ys - (H - 1) / 2) ** 2
    alpha = np.minimum(0.99, o * np.exp(-0.5 * d2 / var))
    alpha = np.where(alpha < 1 / 255, 0.0, alpha)
    # pixels outside the touched tiles see only background
    r = int(radii[0])
    tx0, tx1 = max(0, int(((W - 1) / 2 - r) / 16)), min(4, int(((W - 1) / 2 + r + 15) / 16))
    ty0, ty1 = max(0, int(((H - 1) / 2 - r) / 16)), min(3, int(((H - 1) / 2 + r + 15) / 16))
    inside = (xs // 16 >= tx0) & (xs // 16 < tx1) & (ys // 16 >= ty0) & (ys // 16 < ty1)
    alpha = np.where(inside, alpha, 0.0)
    exp = rgbc[0][:, None, None] * alpha[None] + (1 - alpha)[None] * bg[:, None, None]
    np.testing.assert_allclose(img, exp, atol=2e-6)
    assert n == (tx1 - tx0) * (ty1 - ty0)


def test_two_gaussians_front_to_back_and_tie_order():
    """Equal depths: ties resolve by ascending Gaussian index (stable sort, rasterizer_impl.cu:303-308)."""
    W, H, f, z = 32, 32, 40.0, 3.0
    cam = _cam(W, H, f, (0, 0, z))
    xyz = np.zeros((2, 3), np.float32)
    rgbc = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    o = np.array([0.6, 0.6], np.float32)
    args = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H, cam.tanfovx, cam.tanfovy,
            np.zeros(3, np.float32))
    kw = dict(scales=np.full((2, 3), 0.2, np.float32), rotations=np.tile([1, 0, 0, 0], (2, 1)).astype(np.float32))
    img, _, _ = oracle.rasterize_forward(xyz, o, *args, colors_precomp=rgbc, **kw)
    c = img[:, H // 2, W // 2]
    assert c[0] > c[1] > 0           # index 0 (red) composited first
    # move red behind: green must now dominate
    xyz2 = xyz.copy()
    xyz2[0, 2] = 0.5
    img2, _, _ = oracle.rasterize_forward(xyz2, o, *args, colors_precomp=rgbc, **kw)
    c2 = img2[:, H // 2, W // 2]
    assert c2[1] > c2[0] > 0


def test_saturation_stops_before_accumulating():
    """T*(1-alpha) < 1e-4 -> the Gaussian that would cross the threshold is NOT accumulated
    (forward.cu:346-351)."""
    W, H, f, z = 16, 16, 40.0, 3.0
    cam = _cam(W, H, f, (0, 0, z))
    n = 4
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 2] = np.arange(n) * 0.1
    rgbc = np.ones((n, 3), np.float32)
    o = np.full(n, 0.999, np.float32)     # alpha clamps to 0.99 -> T: 1, 1e-2, 1e-4 (not < 1e-4), 1e-6 stop
    img, _, _ = oracle.rasterize_forward(
        xyz, o, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H, cam.tanfovx,
        cam.tanfovy, np.zeros(3, np.float32), colors_precomp=rgbc, scales=np.full((n, 3), 1.0, np.float32),
        rotations=np.tile([1, 0, 0, 0], (n, 1)).astype(np.float32))
    c = float(img[0, 8, 8])
    a = np.float32(0.99)
    T1 = np.float32(1) * (np.float32(1) - a)          # T after Gaussian 0 (accumulated with T = 1)
    T2 = T1 * (np.float32(1) - a)                     # test_T of Gaussian 1
    if T2 < np.float32(1e-4):
        expected = float(a)                            # Gaussian 1 crosses the threshold: dropped, pixel done
    else:
        expected = float(a + a * T1)                   # Gaussian 1 accepted, Gaussian 2 (T -> 1e-6) dropped
    assert abs(c - expected) < 1e-6


def test_empty_scene_is_zero_not_background():
    """P == 0 short-circuits before the render kernel: the image is all zeros, not bg
    (rasterize_points.cu:68,81)."""
    cam = _cam()
    img, radii, n = oracle.rasterize_forward(
        np.zeros((0, 3), np.float32), np.zeros(0, np.float32), cam.world_view_transform, cam.full_proj_transform,
        cam.camera_center, 64, 48, cam.tanfovx, cam.tanfovy, np.ones(3, np.float32),
        colors_precomp=np.zeros((0, 3), np.float32), scales=np.zeros((0, 3), np.float32),
        rotations=np.zeros((0, 4), np.float32))
    assert n == 0 and radii.shape == (0,) and not img.any()


def test_c1_synthetic_statistics():
    """BASELINE config C1 inputs behave as SURVEY.md 8(d) characterised them."""
    cfg = synthetic.CONFIGS["C1"]
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.concatenate([g["features_dc"], g["features_rest"]], axis=1)
    pose = synthetic.ring_poses(cfg.n_pairs, cfg.ring_radius)[0]
    left, right = synthetic.stereo_cameras(pose, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
    img, radii, n = oracle.rasterize_forward(
        g["xyz"], o, left.world_view_transform, left.full_proj_transform, left.camera_center, cfg.width, cfg.height,
        left.tanfovx, left.tanfovy, np.zeros(3, np.float32), shs=shs, scales=s, rotations=q)
    assert (radii > 0).sum() > 0.9 * cfg.P
    assert 20_000 < n < 120_000
    assert np.isfinite(img).all() and img.max() > 0.2
    # exact tile culling (extension) must not change the image
    img2, radii2, n2 = oracle.rasterize_forward(
        g["xyz"], o, left.world_view_transform, left.full_proj_transform, left.camera_center, cfg.width, cfg.height,
        left.tanfovx, left.tanfovy, np.zeros(3, np.float32), shs=shs, scales=s, rotations=q, exact_cull=True)
    assert n2 < n
    np.testing.assert_array_equal(img, img2)
    np.testing.assert_array_equal(radii, radii2)
