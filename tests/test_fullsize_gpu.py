"""BASELINE.json full-size configurations on the real GPU: direct oracle comparison where the CPU
oracle finishes in seconds (one C2 eye, three C2 TSDF frames), size-independent properties elsewhere
(determinism, image invariance under exact culling / kernel variant, view-order invariance)."""
import numpy as np
import pytest

import oracle
from gs2mesh_amd import _lib, synthetic

pytestmark = pytest.mark.gpu


def _setup(cfg_name, n_pairs=1):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gs2mesh_amd.rasterizer import Rasterizer, camera_from
    cfg = synthetic.CONFIGS[cfg_name]
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    gd["raw"] = True
    poses = synthetic.ring_poses(n_pairs, cfg.ring_radius, 0, cfg.n_pairs)
    cams = [synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline) for p in poses]
    return cfg, g, gd, poses, cams, Rasterizer, camera_from


def test_c2_stereo_pair_vs_oracle_and_invariants():
    cfg, g, gd, poses, cams, Rasterizer, camera_from = _setup("C2")
    left, right = cams[0]
    pair = [camera_from(left), camera_from(right)]
    R = Rasterizer(0)
    a = R.render_views(gd, pair, want_rgb8=True, want_radii=True)
    n_ref_mode = list(a["num_rendered"])
    img = a["color"].cpu().numpy()
    # determinism (race-freedom): a second run is bitwise identical
    b = R.render_views(gd, pair, want_rgb8=True)
    assert np.array_equal(b["color"].cpu().numpy(), img) and np.array_equal(b["rgb8"].cpu().numpy(), a["rgb8"].cpu().numpy())
    # exact culling: same image, fewer instances
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    c = R.render_views(gd, pair)
    assert np.array_equal(c["color"].cpu().numpy(), img)
    assert max(c["num_rendered"]) < 0.8 * max(n_ref_mode)
    # every compositing kernel variant agrees to rounding
    for variant in (0, 4):
        R.set_option(_lib.OPT_BLEND_VARIANT, variant)
        d = R.render_views(gd, pair)["color"].cpu().numpy()
        diff = np.abs(d - img)
        assert (diff > 1e-5).mean() < 1e-4 and diff.max() < 6e-3, variant
    # u8 hand-off image = cv2's conversion of the float image
    q8 = np.clip(np.rint(img.transpose(0, 2, 3, 1) * 255.0), 0, 255).astype(np.uint8)
    assert np.array_equal(a["rgb8"].cpu().numpy(), q8)
    # left eye against the CPU oracle at full size (~seconds)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.concatenate([g["features_dc"], g["features_rest"]], axis=1)
    ref, ref_radii, ref_n = oracle.rasterize_forward(g["xyz"], o, left.world_view_transform, left.full_proj_transform,
                                                     left.camera_center, cfg.width, cfg.height, left.tanfovx,
                                                     left.tanfovy, np.zeros(3, np.float32), shs=shs, scales=s, rotations=q)
    radii = a["radii"].cpu().numpy()[0]
    assert (radii != ref_radii).mean() < 2e-3
    assert abs(n_ref_mode[0] - ref_n) < 2e-3 * ref_n
    diff = np.abs(img[0] - ref)
    # measured on MI355X: max 2.0e-3 on this eye (profiles/r4_parity_C2_blend4.json: 2.6e-3 worst over the bench's pair);
    # bound = 2x measured, and the outliers are held to the CHECKED flip statement below, not to the global bound alone
    assert (diff > 2e-5).mean() < 2e-3 and diff.max() < 5e-3, float(diff.max())
    from oracle import parity
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 0)
    rr = R.render_views(gd, pair, want_radii=True)
    assert np.array_equal(rr["color"].cpu().numpy(), img)
    fa = parity.compositing_attribution(R.download_geometry(0, cfg.P), rr["radii"].cpu().numpy()[0], cfg.width, cfg.height, img[0])
    assert fa["ok"] and fa["max_abs_clean"] <= 2e-4 and fa["unexplained_pixels"] == 0, {k: fa[k] for k in ("max_abs_clean", "unexplained_pixels", "worst_unexplained")}
    mse = float((diff.astype(np.float64) ** 2).mean())
    psnr = 20 * np.log10(1.0 / np.sqrt(mse)) if mse > 0 else np.inf      # GS/utils/image_utils.py:17-19
    assert psnr > 80.0, psnr
    # the oracle itself against the reference's own kernels (oracle/_ref, prebuilt) at full size: bit-exact
    if oracle.ref_available(build=False):
        rr = oracle.ref_forward(g["xyz"], o, left.world_view_transform, left.full_proj_transform, left.camera_center,
                                cfg.width, cfg.height, left.tanfovx, left.tanfovy, np.zeros(3, np.float32), shs=shs,
                                scales=s, rotations=q)
        assert rr["num_rendered"] == ref_n and np.array_equal(rr["radii"], ref_radii)
        assert np.array_equal(rr["color"], ref)


def test_c3_two_million_gaussians_invariants():
    cfg, g, gd, poses, cams, Rasterizer, camera_from = _setup("C3")
    pair = [camera_from(cams[0][0]), camera_from(cams[0][1])]
    R = Rasterizer(0)
    a = R.render_views(gd, pair)
    img = a["color"].cpu().numpy()
    assert np.isfinite(img).all() and 4e6 < min(a["num_rendered"]) and img.mean() > 0.05
    assert np.array_equal(R.render_views(gd, pair)["color"].cpu().numpy(), img)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    assert np.array_equal(R.render_views(gd, pair)["color"].cpu().numpy(), img)


def test_c2_tsdf_frames_vs_oracle_and_order_invariance():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    cfg = synthetic.CONFIGS["C2"]
    W, H, f = cfg.width, cfg.height, cfg.focal
    poses = synthetic.ring_poses(3, cfg.ring_radius, 0, cfg.n_pairs)
    col = synthetic.color_pattern(W, H)
    intr = PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0)
    trunc, mind = cfg.baseline * 20, cfg.baseline * 4
    frames = []
    for p in poses:
        E = np.eye(4)
        E[:3] = p
        frames.append((synthetic.sphere_depth(p, W, H, f, f, W / 2.0, H / 2.0, cfg.sphere_radius), E))

    def fuse(order):
        vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=8192)
        for k in order:
            d, E = frames[k]
            vol.integrate(RGBDImage(col, d, depth_scale=1.0, depth_trunc=trunc), intr, E, min_depth=mind)
        return vol.download(), vol.status()[1]

    (keys, tsdf, weight, rgb), updates = fuse([0, 1, 2])
    ref = oracle.ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, 1)
    ref.set_threads(16)
    n_ref = 0
    for d, E in frames:
        dd = np.where(d < np.float32(mind), 0, d).astype(np.float32)
        n_ref += ref.integrate(oracle.ScalableTSDFVolume.convert_depth(dd, 1.0, trunc), col, W, H, f, f, W / 2.0, H / 2.0, E)
    assert updates == n_ref and n_ref > 4000
    rk, rt, rw, rc = ref.export()
    got = {tuple(k): i for i, k in enumerate(keys.tolist())}
    assert set(got) == set(map(tuple, rk.tolist()))
    order = np.array([got[tuple(k)] for k in rk.tolist()])
    assert np.array_equal(weight[order], rw)
    assert np.array_equal(tsdf[order], rt)                                   # bit-exact running mean
    mean = rgb[order].astype(np.float64) / np.maximum(weight[order], 1)[..., None]
    assert np.abs(mean - rc).max() < 1e-9
    # view order only reassociates the fp32 running mean
    (k2, t2, w2, c2), _ = fuse([2, 0, 1])
    i2 = {tuple(k): i for i, k in enumerate(k2.tolist())}
    o2 = np.array([i2[tuple(k)] for k in keys.tolist()])
    assert np.array_equal(w2[o2], weight) and np.array_equal(c2[o2], rgb)
    assert np.abs(t2[o2] - tsdf).max() < 1e-6


# ---- the exact configuration bench.py times, at full size, against the oracle --------------
# Measured on MI355X (gpurun_out/parity_<cfg>.json, copied to profiles/r2a_parity_<cfg>.json); every bound below is
# <= 2x the measured worst case over the checked eyes.
BENCH_PARITY_BOUNDS = {
    # all-VALU compositing (blend 4, the default).  Measured r2a/r2i: C2 max 2.0e-3 (2.6e-3 on the bench's pair), mean 4.7e-8,
    # PSNR 115.8 / 114.2 dB, 8.0e-6 of the values off by > 1e-5, 70 u8 pixels off by 1 LSB, 1 radius;  C3 max 3.8e-4,
    # mean 3.0e-8, PSNR 130.8 dB, 1.6e-6, 26 pixels, 1 radius
    ("C2", 4): dict(max_abs=5e-3, mean_abs=1e-7, psnr_db=111.0, frac_gt_1e5=1.6e-5, u8_flipped_pixels=140, radii_mismatches=2),
    ("C3", 4): dict(max_abs=8e-4, mean_abs=6e-8, psnr_db=127.0, frac_gt_1e5=3.2e-6, u8_flipped_pixels=52, radii_mismatches=2),
}


@pytest.mark.parametrize("blend", [4])
@pytest.mark.parametrize("cfg_name", ["C2", "C3"])
def test_bench_configuration_full_size_vs_oracle(cfg_name, blend):
    """RenderFusePipeline(inflight=4) + 16x32 binning tiles + exact tile cull + packed SH + fused raw activations
    + TSDF integration on the fuse stream: the loop of bench.py, both eyes of two of the pipelined views compared
    with the CPU oracle (the reference's own kernels when oracle/_ref is prebuilt) at BASELINE.json's full size."""
    import json
    import os
    import torch
    from oracle import parity
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, ScalableTSDFVolume
    from gs2mesh_amd.pipeline import RenderFusePipeline
    cfg, g, gd, poses, cams, Rasterizer, camera_from = _setup(cfg_name, n_pairs=5)
    W, H = cfg.width, cfg.height
    dev = torch.device("cuda:0")
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0)
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=16384, device=0)
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=4, device=0, raster_options=dict(exact_tile_cull=1, tile_rows=2, blend_variant=blend))
    ccams = [[camera_from(l), camera_from(r)] for l, r in cams]
    first = pipe.prepare(ccams[0], headroom=2.0)
    radii0 = first["radii"].cpu().numpy()
    kept = {}
    for i, p in enumerate(poses):
        d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0, cfg.sphere_radius, dev)
        E = np.eye(4)
        E[:3] = p
        slot = pipe.submit(ccams[i], d, E, depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4)
        kept[slot] = i                      # no wait between submits: views 1..4 are in flight together
    pipe.finish()
    assert vol.status()[0] > 100
    worst = None
    for slot, i in sorted(kept.items())[:2]:                      # two of the four views still held by the slots
        color = pipe.color[slot].cpu().numpy()
        rgb8 = pipe.rgb8[slot].cpu().numpy()
        m = parity.pair_parity(g, cams[i], W, H, color, rgb8, radii0 if i == 0 else None, flips=(blend == 4))
        m["view"] = i
        if worst is None:
            worst = m
        else:
            for k in ("max_abs", "mean_abs", "frac_gt_1e5", "frac_gt_1e4", "u8_flipped_pixels", "u8_flipped_values", "u8_max_lsb",
                      "flip_pixels", "max_abs_clean", "max_abs_flip", "unexplained_pixels", "pixels_over_clean_bar"):
                if k in m:
                    worst[k] = max(worst[k], m[k])
            if "flips_ok" in m:
                worst["flips_ok"] = min(worst["flips_ok"], m["flips_ok"])
            worst["psnr_db"] = min(worst["psnr_db"], m["psnr_db"])
    # radii of the prepare() render of view 0 (fused exp / normalize / sigmoid vs numpy's)
    o0 = [parity.oracle_eye(g, c, W, H) for c in cams[0]]
    worst["radii_mismatches"] = int(max((radii0[v] != o0[v]["radii"]).sum() for v in range(2)))
    worst["num_rendered_reference_lists"] = [o["num_rendered"] for o in o0]
    os.makedirs("gpurun_out", exist_ok=True)
    worst["blend_variant"] = blend
    with open(os.path.join("gpurun_out", f"parity_{cfg_name}_blend{blend}.json"), "w") as fh:
        json.dump(worst, fh, indent=1)
    print("PARITY", cfg_name, "blend", blend, json.dumps(worst))
    b = BENCH_PARITY_BOUNDS[(cfg_name, blend)]
    assert worst["max_abs"] <= b["max_abs"], worst
    assert worst["mean_abs"] <= b["mean_abs"], worst
    assert worst["psnr_db"] >= b["psnr_db"], worst
    assert worst["frac_gt_1e5"] <= b["frac_gt_1e5"], worst
    assert worst["u8_flipped_pixels"] <= b["u8_flipped_pixels"] and worst["u8_max_lsb"] <= 1, worst
    assert worst["radii_mismatches"] <= b["radii_mismatches"], worst
    if blend == 4:
        # the CHECKED form of "the outliers are threshold flips" (oracle/parity.py:flip_attribution): SURVEY.md 8(c)'s
        # max |delta| <= 2e-4 holds on every pixel where no decision of renderCUDA sits within 1e-5 of its threshold, and
        # every other pixel stays within the bound of the contributions that can flip
        assert worst["flips_ok"] == 1 and worst["unexplained_pixels"] == 0 and worst["max_abs_clean"] <= 2e-4, worst


# BASELINE configs C4 (Truck-like: 2.5 M Gaussians, 1920 x 1080, the only config with a 64^3-block key space) and C5
# (MobileBrick-like: 500 k Gaussians, 1920 x 1440 = 10 800 binning tiles with 2 views in the LDS cursors, 14 % baseline),
# through the bench's DEFAULT launch shapes: two stereo pairs per launch.  Bounds <= 2x the values
# measured on MI355X in round 4 (profiles/r4_parity_C4.json / r4_parity_C5.json).
BENCH_PARITY_BOUNDS_PPL2 = {
    # measured r4b: C4 max 3.8e-3 (one flip; clean max 4.2e-7), mean 1.9e-8, PSNR 114.6 dB, 1.8e-6 off by > 1e-5, 31 u8 pixels, 1 radius
    "C4": dict(max_abs=7.6e-3, mean_abs=3.8e-8, psnr_db=111.5, frac_gt_1e5=3.6e-6, u8_flipped_pixels=62, radii_mismatches=2),
    # C5 max 6.2e-4 (clean max 5.5e-7), mean 2.1e-8, PSNR 132.3 dB, 2.2e-6, 52 u8 pixels, 0 radii
    "C5": dict(max_abs=1.3e-3, mean_abs=4.3e-8, psnr_db=129.3, frac_gt_1e5=4.4e-6, u8_flipped_pixels=104, radii_mismatches=0),
}


@pytest.mark.parametrize("cfg_name", ["C4", "C5"])
def test_c4_c5_pair_batched_pipeline_full_size_vs_reference_kernels(cfg_name):
    """One stereo pair of C4 / C5 out of a pair-batched pipelined job (RenderFusePipeline: 16 x 32 binning tiles, exact tile
    cull, fused raw activations, packed -- C4: Morton-ordered -- model, pairs_per_launch = 2, TSDF sweeps on the fuse stream)
    against the reference's own kernels (oracle/_ref) with the checked flip attribution."""
    import json
    import os
    import torch
    from oracle import parity
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, ScalableTSDFVolume
    from gs2mesh_amd.pipeline import RenderFusePipeline
    cfg, g, gd, poses, cams, Rasterizer, camera_from = _setup(cfg_name, n_pairs=4)
    W, H = cfg.width, cfg.height
    dev = torch.device("cuda:0")
    intr = PinholeCameraIntrinsic(W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0)
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=65536, device=0)
    pipe = RenderFusePipeline(gd, W, H, vol, intr, inflight=2, device=0, fuse_batch=4, pairs_per_launch=2)
    assert pipe.spatial_order == (cfg.P >= 32768)
    ccams = [[camera_from(l), camera_from(r)] for l, r in cams]
    first = pipe.prepare(ccams[0], headroom=2.0)
    radii0 = first["radii"].cpu().numpy()
    slots = []
    for i, p in enumerate(poses):
        d = synthetic.sphere_depth_torch(p, W, H, cfg.focal, cfg.focal, W / 2.0, H / 2.0, cfg.sphere_radius, dev)
        E = np.eye(4)
        E[:3] = p
        slots.append(pipe.submit(ccams[i], d, E, depth_trunc=cfg.baseline * 20, min_depth=cfg.baseline * 4))
    pipe.finish()
    assert vol.status()[0] > 100
    # views 2, 3 went through the second launch (slot of view 3); its images are [pair 2 | pair 3]
    slot = slots[3]
    color = pipe.color[slot].cpu().numpy()
    rgb8 = pipe.rgb8[slot].cpu().numpy()
    m = parity.pair_parity(g, cams[3], W, H, color[2:4], rgb8[2:4], None, flips=True)
    m["view"] = 3
    o0 = [parity.oracle_eye(g, c, W, H) for c in cams[0]]
    m["radii_mismatches"] = int(max((radii0[v] != o0[v]["radii"]).sum() for v in range(2)))
    m["num_rendered_reference_lists"] = [o["num_rendered"] for o in o0]
    m["num_rendered"] = [int(x) for x in first["num_rendered"]]
    m["pairs_per_launch"], m["spatial_order"] = 2, int(pipe.spatial_order)
    # the first pair of the same launch against the serial single-pair render of the same handle: bit-identical
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    R.set_option(_lib.OPT_TILE_ROWS, 2)
    R.pack_sh(gd)
    single = R.render_views(gd, ccams[2], want_rgb8=True)
    assert np.array_equal(single["color"].cpu().numpy(), color[0:2]) and np.array_equal(single["rgb8"].cpu().numpy(), rgb8[0:2])
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"parity_{cfg_name}_ppl2.json"), "w") as fh:
        json.dump(m, fh, indent=1, default=str)
    print("PARITY", cfg_name, "ppl2", json.dumps(m, default=str))
    b = BENCH_PARITY_BOUNDS_PPL2[cfg_name]
    assert m["flips_ok"] == 1 and m["unexplained_pixels"] == 0 and m["max_abs_clean"] <= 2e-4, m
    assert m["max_abs"] <= b["max_abs"] and m["mean_abs"] <= b["mean_abs"] and m["psnr_db"] >= b["psnr_db"], m
    assert m["frac_gt_1e5"] <= b["frac_gt_1e5"] and m["u8_flipped_pixels"] <= b["u8_flipped_pixels"] and m["u8_max_lsb"] <= 1, m
    assert m["radii_mismatches"] <= b["radii_mismatches"], m


def test_c4_tsdf_frames_1024_cubed_vs_oracle_and_batch_sweep():
    """C4's fuse half at full size: voxel 4/1024 (a 64^3-block key space), sphere of radius 1.2, 1920 x 1080 frames,
    max_blocks = 64^3.  Three frames view by view against the restated Open3D oracle (block sets, block updates, weights and
    tsdf bit for bit, colour means to 1e-9), and the voxel-stationary batch sweep of the same frames == view by view."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    cfg = synthetic.CONFIGS["C4"]
    W, H, f = cfg.width, cfg.height, cfg.focal
    poses = synthetic.ring_poses(3, cfg.ring_radius, 0, cfg.n_pairs)
    col = synthetic.color_pattern(W, H)
    intr = PinholeCameraIntrinsic(W, H, f, f, W / 2.0, H / 2.0)
    trunc, mind = cfg.baseline * 20, cfg.baseline * 4
    frames = []
    for p in poses:
        E = np.eye(4)
        E[:3] = p
        frames.append((synthetic.sphere_depth(p, W, H, f, f, W / 2.0, H / 2.0, cfg.sphere_radius), E))
    n_dense = (cfg.tsdf_n // 16) ** 3
    assert n_dense == 262144
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=n_dense)
    for d, E in frames:
        vol.integrate(RGBDImage(col, d, depth_scale=1.0, depth_trunc=trunc), intr, E, min_depth=mind)
    updates = vol.status()[1]
    keys, tsdf, weight, rgb = vol.download()
    ref = oracle.ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, 1)
    ref.set_threads(16)
    n_ref = 0
    for d, E in frames:
        dd = np.where(d < np.float32(mind), 0, d).astype(np.float32)
        n_ref += ref.integrate(oracle.ScalableTSDFVolume.convert_depth(dd, 1.0, trunc), col, W, H, f, f, W / 2.0, H / 2.0, E)
    assert updates == n_ref and n_ref > 10000
    rk, rt, rw, rc = ref.export()
    assert np.abs(rk).max() <= 32                      # the blocks of a 4-unit cube: indices inside the 64^3 key space
    got = {tuple(k): i for i, k in enumerate(keys.tolist())}
    assert set(got) == set(map(tuple, rk.tolist()))
    order = np.array([got[tuple(k)] for k in rk.tolist()])
    assert np.array_equal(weight[order], rw)
    assert np.array_equal(tsdf[order], rt)
    mean = rgb[order].astype(np.float64) / np.maximum(weight[order], 1)[..., None]
    assert np.abs(mean - rc).max() < 1e-9
    del tsdf, weight, rgb, mean
    vol.reset()
    vol.integrate_batch([RGBDImage(col, d, depth_scale=1.0, depth_trunc=trunc) for d, _ in frames], intr, [E for _, E in frames],
                        min_depth=mind)
    assert vol.status()[1] == n_ref
    kb, tb, wb, cb = vol.download()
    ib = {tuple(k): i for i, k in enumerate(kb.tolist())}
    assert set(ib) == set(map(tuple, rk.tolist()))
    ob = np.array([ib[tuple(k)] for k in rk.tolist()])
    assert np.array_equal(wb[ob], rw) and np.array_equal(tb[ob], rt)
    assert np.abs(cb[ob].astype(np.float64) / np.maximum(wb[ob], 1)[..., None] - rc).max() < 1e-9


def test_trained_like_splats_full_size_vs_reference_kernels():
    """C2-sized `synthetic.trained_like` model (anisotropy 10-100 : 1, 30 % of the opacities at the 0.99 cap, 0.1 % splats
    wider than 300 px, depth ties) through the bench's configuration -- RenderFusePipeline, 16 x 32 binning tiles, exact
    tile cull, fused raw activations -- against the reference's own kernels (oracle/_ref) at 1600 x 1200, with the flip
    attribution as the image bar.  This is the path the isotropic `synth_v1` scenes do not stress: the flagged `general`
    compositing branch and the > 64-tile wave walk of the binning."""
    import json
    import os
    import torch
    from oracle import parity
    from gs2mesh_amd.pipeline import RenderFusePipeline
    from gs2mesh_amd.rasterizer import camera_from
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = synthetic.CONFIGS["C2"]
    W, H = cfg.width, cfg.height
    g = synthetic.trained_like(cfg.P, 4242, cfg.log_s_mu, focal=cfg.focal, ring_radius=cfg.ring_radius)
    gd = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    gd["raw"] = True
    poses = synthetic.ring_poses(2, cfg.ring_radius, 3, cfg.n_pairs)
    cams = [synthetic.stereo_cameras(p, W, H, cfg.focal, cfg.focal, cfg.baseline) for p in poses]
    pipe = RenderFusePipeline(gd, W, H, None, None, inflight=2, device=0)
    ccams = [[camera_from(l), camera_from(r)] for l, r in cams]
    first = pipe.prepare(ccams[0], headroom=2.0)
    radii0 = first["radii"].cpu().numpy()
    assert radii0.max() > 300                                      # the background splats are there
    slots = [pipe.submit(c) for c in ccams]
    pipe.finish()
    color = pipe.color[slots[1]].cpu().numpy()
    rgb8 = pipe.rgb8[slots[1]].cpu().numpy()
    # (1) against the reference's own kernels: global figures (the raw-parameter path projects with its own exp / sigmoid)
    m = parity.pair_parity(g, cams[1], W, H, color, rgb8, None)
    o0 = [parity.oracle_eye(g, c, W, H) for c in cams[0]]
    m["radii_mismatches"] = int(max((radii0[v] != o0[v]["radii"]).sum() for v in range(2)))
    m["num_rendered_reference_lists"] = [o["num_rendered"] for o in o0]
    m["num_rendered"] = [int(x) for x in first["num_rendered"]]
    # (2) the compositing stage alone, on the record THIS pass projected: the checked flip statement with the 1e-5 band
    R = pipe.rasterizers[slots[1]]
    rr = R.render_views(gd, ccams[1], want_radii=True)          # same handle, same pair: identical image + the radii
    assert np.array_equal(rr["color"].cpu().numpy(), color)
    radii1 = rr["radii"].cpu().numpy()
    comp = []
    for v in range(2):
        rec = R.download_geometry(v, cfg.P)
        fa = parity.compositing_attribution(rec, radii1[v], W, H, color[v])
        comp.append({k: fa[k] for k in ("flip_pixels", "ill_conditioned_pixels", "max_abs_clean", "max_abs_flip", "unexplained_pixels",
                                        "pixels_over_clean_bar", "worst_unexplained", "ok")})
    m["compositing"] = comp
    # (3) record of the raw path vs the oracle's record: relative difference of the conic (anisotropy amplifies 1 ulp of exp)
    s_a, q_a, o_a = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    cam = cams[1][0]
    geom = oracle.preprocess(g["xyz"], s_a, q_a, o_a, shs, cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                             W, H, cam.tanfovx, cam.tanfovy)
    rec = R.download_geometry(0, cfg.P)
    vis = (geom["radii"] > 0) & (radii1[0] > 0) & (rec["tiles_touched"] > 0)     # the exact cull empties some rects: no record kept
    dc = np.abs(rec["conic_opacity"][vis, :3] - geom["conic_opacity"][vis, :3]).max(axis=1)
    rel = dc / np.abs(geom["conic_opacity"][vis, :3]).max(axis=1)
    m["raw_path_conic_rel_diff"] = dict(max=float(rel.max()), p999=float(np.quantile(rel, 0.999)), median=float(np.median(rel)),
                                        exact_fraction=float((dc == 0).mean()))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "parity_C2_trained_like.json"), "w") as fh:
        json.dump(m, fh, indent=1, default=str)
    print("PARITY trained_like", json.dumps(m, default=str))
    assert all(c["ok"] and c["unexplained_pixels"] == 0 and c["max_abs_clean"] <= 2e-4 for c in comp), comp
    assert m["psnr_db"] >= 100.0 and m["u8_max_lsb"] <= 1 and m["radii_mismatches"] <= 4, m
    # measured (profiles/r3_parity_C2_trained_like.json): median 0, p99.9 2.6e-5, max 2.2e-4 -> bounds <= 2x measured
    assert m["raw_path_conic_rel_diff"]["median"] < 1e-6 and m["raw_path_conic_rel_diff"]["p999"] < 5.2e-5, m
    assert m["raw_path_conic_rel_diff"]["max"] < 4.4e-4, m


def test_c2_tile_rows_2_is_bit_identical_to_tile_rows_1():
    """16 x 32 binning tiles only change which list an instance is found in: every pixel composites the same
    instances in the same order with the same arithmetic, so the image is the 16 x 16 image bit for bit."""
    cfg, g, gd, poses, cams, Rasterizer, camera_from = _setup("C2")
    pair = [camera_from(cams[0][0]), camera_from(cams[0][1])]
    R = Rasterizer(0)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    a = R.render_views(gd, pair, want_rgb8=True)
    R2 = Rasterizer(0)
    R2.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    R2.set_option(_lib.OPT_TILE_ROWS, 2)
    R2.pack_sh(gd)
    b = R2.render_views(gd, pair, want_rgb8=True)
    assert max(b["num_rendered"]) < 0.8 * max(a["num_rendered"])
    assert np.array_equal(a["color"].cpu().numpy(), b["color"].cpu().numpy())
    assert np.array_equal(a["rgb8"].cpu().numpy(), b["rgb8"].cpu().numpy())
