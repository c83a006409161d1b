"""Rasteriser parity: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Every test runs on two back-ends (tests/backends.py): `emu` = the kernel sources on the CPU
fiber emulator (build container, `-m "not gpu"`), `gpu` = libgs2mesh_amd.so on an MI355X
(`-m gpu`).

Tolerances (stated here, justified in DESIGN.md "Parity"):
  * projected record (radii, tile rect, mean2D, depth, conic, opacity, rgb): BIT-EXACT --
    the projection translation unit is compiled without FMA contraction and fp32 sqrt/div are
    correctly rounded, so the IEEE sequence equals the oracle's;
  * instance lists (point_list, tile ranges, num_rendered): exact (integers);
  * image: |d| <= 1e-5 on >= 99.99 % of the values, and no value off by more than 6e-3
    (one alpha >= 1/255 threshold flip = one contribution of <= 0.99/255*T*rgb; exp() differs
    in the last ulp between libm and the device).
"""
import math

import numpy as np
import pytest

import oracle
from gs2mesh_amd import _lib, synthetic
from gs2mesh_amd.graphics import Camera
from gs2mesh_amd.rasterizer import Rasterizer, camera_from


def assert_image_close(got, ref, frac_tol=1e-4, small=1e-5, big=6e-3):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.isfinite(got).all()
    bad = (d > small).mean()
    assert bad <= frac_tol, f"{bad:.2e} of values differ by more than {small}"
    assert d.max() <= big, f"max abs diff {d.max():.3e}"


def assert_image_attributed(img, geom, W, H, bg=(0.0, 0.0, 0.0), rgb=None):
    """The CHECKED form of the image bar (oracle/parity.py, as in the full-size tests): max |delta| <= 2e-4 on every pixel where
    no decision of renderCUDA sits within 1e-5 of its threshold in the oracle's own replay of the reference record `geom`, and every
    other pixel within the bound of the contributions that can flip -- zero unexplained pixels.  `rgb`: the colours where they
    were passed precomputed (the oracle's projection then leaves its own rgb empty)."""
    from oracle import parity
    fa = parity.compositing_attribution(dict(means2D=geom["means2D"], depths=geom["depths"], conic_opacity=geom["conic_opacity"],
                                             rgb=geom["rgb"] if rgb is None else rgb), geom["radii"], W, H,
                                        np.asarray(img, np.float32), bg)
    assert fa["ok"] and fa["unexplained_pixels"] == 0 and fa["max_abs_clean"] <= 2e-4, {k: fa[k] for k in (
        "flip_pixels", "max_abs_clean", "max_abs_flip", "unexplained_pixels", "worst_unexplained")}


def scene(P, seed, W, H, f, log_s=math.log(0.03), ring=3.5, az=0.3):
    g = synthetic.synth_v1(P, seed, log_s)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    pose = synthetic.ring_pose(az, ring)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    left, right = synthetic.stereo_cameras(pose, W, H, f, f, 0.245)
    return g, s, q, o, shs, left, right


def run_forward(be, cam, xyz, o, bg, **kw):
    r = Rasterizer(0, lib=be.lib)
    d = be.dev
    out, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform),
                           d(cam.camera_center), d(np.asarray(bg, np.float32)), cam.image_width, cam.image_height,
                           cam.tanfovx, cam.tanfovy, **{k: (d(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    return r, be.host(out), be.host(radii)


def oracle_forward(cam, xyz, o, bg, **kw):
    return oracle.rasterize_forward(xyz, o, cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                                    cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy,
                                    np.asarray(bg, np.float32), **kw)


def test_projection_binning_image_parity(backend):
    W, H, f = 200, 136, 180.0           # not multiples of 16: ragged right/bottom tiles
    g, s, q, o, shs, left, _ = scene(3000, 11, W, H, f)
    bg = [0.1, 0.2, 0.3]
    r, img, radii = run_forward(backend, left, g["xyz"], o, bg, shs=shs, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(left, g["xyz"], o, bg, shs=shs, scales=s, rotations=q)
    geom_ref = oracle.preprocess(g["xyz"], s, q, o, shs, left.world_view_transform, left.full_proj_transform,
                                 left.camera_center, W, H, left.tanfovx, left.tanfovy)
    # -- projected record: bit-exact
    np.testing.assert_array_equal(radii, ref_radii)
    geom = r.download_geometry(0, 3000)
    vis = ref_radii > 0
    assert vis.sum() > 1000
    np.testing.assert_array_equal(geom["rect"].astype(np.uint32), geom_ref["rect"])
    np.testing.assert_array_equal(geom["tiles_touched"], geom_ref["tiles_touched"])
    for k in ("means2D", "depths", "conic_opacity", "rgb"):
        np.testing.assert_array_equal(geom[k][vis], geom_ref[k][vis], err_msg=k)
    # -- instance lists: exact
    assert r.last_num_rendered == ref_n
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl, ranges = r.download_binning(0, ref_n, n_tiles)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    # -- image: the global bar, and the checked flip statement on the (bit-exact) record
    assert_image_close(img, ref_img)
    assert_image_attributed(img, geom_ref, W, H, bg)


@pytest.mark.parametrize("deg,M", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16)])
def test_sh_degrees_and_layouts(backend, deg, M):
    W, H, f = 96, 80, 90.0
    g, s, q, o, shs, left, _ = scene(800, 5, W, H, f)
    shs = np.ascontiguousarray(shs[:, :M])
    bg = [0, 0, 0]
    _, img, radii = run_forward(backend, left, g["xyz"], o, bg, shs=shs, scales=s, rotations=q, sh_degree=deg)
    ref_img, ref_radii, _ = oracle_forward(left, g["xyz"], o, bg, shs=shs, scales=s, rotations=q, sh_degree=deg)
    np.testing.assert_array_equal(radii, ref_radii)
    assert_image_close(img, ref_img)


def test_precomputed_colors_and_covariance(backend):
    W, H, f = 96, 80, 90.0
    g, s, q, o, shs, left, _ = scene(800, 6, W, H, f)
    geom_ref = oracle.preprocess(g["xyz"], s, q, o, shs, left.world_view_transform, left.full_proj_transform,
                                 left.camera_center, W, H, left.tanfovx, left.tanfovy, scale_modifier=0.7)
    cov = geom_ref["cov3D"]
    cols = np.random.default_rng(0).uniform(0, 1, (800, 3)).astype(np.float32)
    bg = [1, 1, 1]
    _, img, radii = run_forward(backend, left, g["xyz"], o, bg, colors_precomp=cols, cov3D_precomp=cov)
    ref_img, ref_radii, _ = oracle_forward(left, g["xyz"], o, bg, colors_precomp=cols, cov3D_precomp=cov)
    np.testing.assert_array_equal(radii, ref_radii)
    assert_image_close(img, ref_img)
    # scale_modifier path computes the same covariance itself
    _, img2, radii2 = run_forward(backend, left, g["xyz"], o, bg, colors_precomp=cols, scales=s, rotations=q,
                                  scale_modifier=0.7)
    np.testing.assert_array_equal(radii2, ref_radii)
    assert_image_close(img2, ref_img)


def test_stereo_pair_fused_raw_parameters(backend):
    """Pipeline-level entry: both eyes in one pass from PRE-activation parameters with the
    activations fused (exp / normalize / sigmoid differ from numpy in the last ulp -> tolerance on
    the record too)."""
    W, H, f = 160, 120, 150.0
    g, s, q, o, shs, left, right = scene(2500, 21, W, H, f)
    be = backend
    r = Rasterizer(0, lib=be.lib)
    gd = dict(xyz=be.dev(g["xyz"]), scaling=be.dev(g["scaling"]), rotation=be.dev(g["rotation"]),
              opacity=be.dev(g["opacity"]), features_dc=be.dev(g["features_dc"]),
              features_rest=be.dev(g["features_rest"]), raw=True, sh_degree=3)
    res = r.render_views(gd, [camera_from(left), camera_from(right)], bg=(0, 0, 0), want_rgb8=True, want_radii=True)
    color = be.host(res["color"])
    rgb8 = be.host(res["rgb8"])
    radii = be.host(res["radii"])
    for v, cam in enumerate((left, right)):
        ref_img, ref_radii, ref_n = oracle_forward(cam, g["xyz"], o, [0, 0, 0], shs=shs, scales=s, rotations=q)
        mism = (radii[v] != ref_radii).mean()
        assert mism <= 2e-3, f"radii mismatch fraction {mism}"
        assert abs(res["num_rendered"][v] - ref_n) <= max(3, 2e-3 * ref_n)
        assert_image_close(color[v], ref_img, frac_tol=2e-3, small=2e-5, big=2e-2)
        q8 = np.clip(np.rint(color[v].transpose(1, 2, 0) * 255.0), 0, 255).astype(np.uint8)
        np.testing.assert_array_equal(rgb8[v], q8)
    # the two eyes differ (baseline shift) but are the same scene
    assert np.abs(color[0] - color[1]).mean() > 1e-4
    # concatenated features give the same result as the dc/rest split
    gd2 = dict(gd)
    gd2.pop("features_dc"), gd2.pop("features_rest")
    gd2["features"] = be.dev(shs)
    res2 = r.render_views(gd2, [camera_from(left), camera_from(right)], bg=(0, 0, 0))
    np.testing.assert_array_equal(be.host(res2["color"]), color)


def test_exact_tile_cull_preserves_the_image(backend):
    W, H, f = 200, 136, 180.0
    g, s, q, o, shs, left, _ = scene(3000, 12, W, H, f)
    be = backend
    d = be.dev
    args = (d(g["xyz"]), d(o), d(left.world_view_transform), d(left.full_proj_transform), d(left.camera_center),
            d(np.zeros(3, np.float32)), W, H, left.tanfovx, left.tanfovy)
    kw = dict(shs=d(shs), scales=d(s), rotations=d(q))
    r = Rasterizer(0, lib=be.lib)
    img0, radii0 = r.forward(*args, **kw)
    n0 = r.last_num_rendered
    r.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
    img1, radii1 = r.forward(*args, **kw)
    n1 = r.last_num_rendered
    np.testing.assert_array_equal(be.host(img0), be.host(img1))
    np.testing.assert_array_equal(be.host(radii0), be.host(radii1))
    _, _, n_ref = oracle_forward(left, g["xyz"], o, [0, 0, 0], shs=shs, scales=s, rotations=q, exact_cull=True)
    assert n1 < 0.8 * n0
    assert abs(n1 - n_ref) <= max(2, 1e-3 * n_ref)
    # level 2 (round 6): rects of at most 4 tiles keep all their tiles -- between the two lists, same image
    r.set_option(_lib.OPT_EXACT_TILE_CULL, 2)
    img2, radii2 = r.forward(*args, **kw)
    n2 = r.last_num_rendered
    np.testing.assert_array_equal(be.host(img0), be.host(img2))
    np.testing.assert_array_equal(be.host(radii0), be.host(radii2))
    _, _, n_ref2 = oracle_forward(left, g["xyz"], o, [0, 0, 0], shs=shs, scales=s, rotations=q, exact_cull=2)
    assert n1 <= n2 < n0 and abs(n2 - n_ref2) <= max(2, 1e-3 * n_ref2)
    with pytest.raises(RuntimeError):
        r.set_option(_lib.OPT_EXACT_TILE_CULL, 3)


def test_empty_and_fully_culled_scenes(backend):
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 1.0, 0.8, 64, 48)
    bg = [0.3, 0.6, 0.9]
    be = backend
    # P == 0: zero image, not background (rasterize_points.cu:68,81)
    r, img, radii = run_forward(be, cam, np.zeros((0, 3), np.float32), np.zeros(0, np.float32), bg,
                                colors_precomp=np.zeros((0, 3), np.float32), scales=np.zeros((0, 3), np.float32),
                                rotations=np.zeros((0, 4), np.float32))
    assert img.shape == (3, 48, 64) and not img.any() and radii.shape == (0,)
    # everything behind the camera: background everywhere, radii 0
    P = 300
    xyz = np.random.default_rng(1).uniform(-1, 1, (P, 3)).astype(np.float32)
    xyz[:, 2] -= 10.0
    r, img, radii = run_forward(be, cam, xyz, np.full(P, 0.5, np.float32), bg,
                                colors_precomp=np.ones((P, 3), np.float32),
                                scales=np.full((P, 3), 0.1, np.float32),
                                rotations=np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32))
    assert not radii.any() and r.last_num_rendered == 0
    for c in range(3):
        np.testing.assert_array_equal(img[c], np.float32(bg[c]))


@pytest.mark.parametrize("P", [9000, 30000])
def test_crowded_tile_uses_the_merge_path_and_saturates(backend, P):
    """> 4096 instances in single tiles (9000: the 8192-key bucket sort; 30000: LDS-sorted runs + rank merges); pixels
    saturate (T < 1e-4) long before the list ends."""
    W, H, f = 48, 32, 60.0
    rng = np.random.default_rng(3)
    xyz = rng.normal(0, 0.02, (P, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-1, 1, P)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    o = rng.uniform(0.02, 0.4, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.05, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r, img, radii = run_forward(backend, cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    assert (ref_ranges[:, 1] - ref_ranges[:, 0]).max() > (4096 if P == 9000 else 8192)
    pl, ranges = r.download_binning(0, ref_n, 3 * 2)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(img, ref_img)


def test_lists_grow_after_a_view_without_large_lists(backend):
    """The size-class kernels of the per-tile sort are replaced by an LDS-free stand-in (sort_class_lists_rank, run by the first workgroups of k_sort_tiles_small) when the previous
    call on the handle found every class empty.  That launch is not a hint-dependent shortcut: a view that suddenly HAS lists
    of 600 ... 9000 instances (all three classes) after a sparse one is still sorted exactly."""
    W, H, f = 48, 32, 60.0
    be = backend
    d = be.dev
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    r = Rasterizer(0, lib=be.lib)

    def render(xyz, o, cols, s, q):
        return r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                         d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols), scales=d(s),
                         rotations=d(q))

    rng = np.random.default_rng(5)
    P0 = 200                                                     # sparse view: no list above 512 instances
    xyz0 = rng.uniform(-0.3, 0.3, (P0, 3)).astype(np.float32)
    s0 = np.full((P0, 3), 0.01, np.float32)
    q0 = np.tile([1, 0, 0, 0], (P0, 1)).astype(np.float32)
    render(xyz0, rng.uniform(0.1, 0.5, P0).astype(np.float32), rng.uniform(0, 1, (P0, 3)).astype(np.float32), s0, q0)
    render(xyz0, rng.uniform(0.1, 0.5, P0).astype(np.float32), rng.uniform(0, 1, (P0, 3)).astype(np.float32), s0, q0)
    # crowded view on the SAME handle and image size: the class hint read back from the sparse views says "all empty"
    P = 12000
    xyz = rng.normal(0, 0.02, (P, 3)).astype(np.float32)
    xyz[:, 1] -= 0.53                                            # ~8300 in the tile above the centre
    xyz[:3000, 0] += 1.07                                        # a second, smaller stack in the tile right of it
    xyz[3000:3700, 0] -= 1.07                                    # and a third of ~700 bottom left
    xyz[3000:3700, 1] += 1.06
    xyz[:, 2] = rng.uniform(-0.2, 0.2, P)
    o = rng.uniform(0.02, 0.4, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.012, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    img, radii = render(xyz, o, cols, s, q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    sizes = (ref_ranges[:, 1] - ref_ranges[:, 0]).astype(np.int64)
    assert sizes.max() > 8192 and ((sizes > 512) & (sizes <= 8192)).any(), sizes
    pl, ranges = r.download_binning(0, ref_n, 3 * 2)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(be.host(img), ref_img)


@pytest.mark.parametrize("depths", ["uniform", "clustered", "equal"])
def test_mid_size_lists_use_the_bucket_sort(backend, depths):
    """512 < n <= 4096 instances per tile: bucket + rank sort (k_sort_tiles_bucket).  `uniform`: depths spread over the
    frustum (0-2 keys per bucket); `clustered`: almost all depths inside a sliver of the range (a bucket overflows
    GS2M_BUCKET_MAX -> the bitonic fallback); `equal`: many exactly equal depths (ties resolved by Gaussian id, like the
    reference's stable radix sort).  The instance lists must equal the oracle's exactly."""
    W, H, f = 64, 48, 70.0
    P = 7000
    rng = np.random.default_rng(17)
    xyz = rng.normal(0, 0.35, (P, 3)).astype(np.float32)
    if depths == "uniform":
        xyz[:, 2] = rng.uniform(-1, 1, P)
    elif depths == "clustered":
        xyz[:, 2] = np.where(rng.uniform(size=P) < 0.97, rng.normal(0.3, 1e-4, P), rng.uniform(-1, 1, P))
    else:
        xyz[:, 2] = np.round(rng.uniform(-1, 1, P) * 8) / 8       # 17 distinct depths: long runs of exact ties
    xyz = xyz.astype(np.float32)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    o = rng.uniform(0.02, 0.3, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.03, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r, img, radii = run_forward(backend, cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    sizes = (ref_ranges[:, 1] - ref_ranges[:, 0]).astype(np.int64)
    assert ((sizes > 512) & (sizes <= 4096)).sum() >= 4, sizes
    pl, ranges = r.download_binning(0, ref_n, 4 * 3)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(img, ref_img)


@pytest.mark.parametrize("depths", ["two_shells", "shell_and_ties", "one_sliver"])
def test_wave_sized_lists_with_clustered_depths_take_the_refined_buckets(backend, depths):
    """64 < n <= 512 instances per tile with depths that CLUSTER (a trained splat sits on surfaces): the equal-width buckets of
    the wave sort overflow and the list is refined (round 6: bucket b with c_b keys split into c_b sub-buckets, second counting
    pass; raster_sort.h) instead of falling back to the bitonic network.  `two_shells`: two thin depth bands + a sparse rest;
    `shell_and_ties`: a band + runs of exactly equal depths (ties resolved by Gaussian id, the fine buckets cannot split them);
    `one_sliver`: 97 % of the keys within 1e-4 of one depth (refinement of a bucket that holds nearly everything).  The instance
    lists must equal the oracle's exactly."""
    W, H, f = 160, 96, 300.0
    P = 5000
    rng = np.random.default_rng(29)
    xyz = rng.uniform(-1.0, 1.0, (P, 3)).astype(np.float32)
    xyz[:, 1] *= 0.6
    u = rng.uniform(size=P)
    if depths == "two_shells":
        xyz[:, 2] = np.where(u < 0.45, rng.normal(-0.4, 2e-3, P), np.where(u < 0.9, rng.normal(0.5, 2e-3, P), rng.uniform(-1, 1, P)))
    elif depths == "shell_and_ties":
        xyz[:, 2] = np.where(u < 0.5, rng.normal(0.2, 3e-3, P), np.round(rng.uniform(-1, 1, P) * 4) / 4)
    else:
        xyz[:, 2] = np.where(u < 0.97, rng.normal(0.3, 1e-4, P), rng.uniform(-1, 1, P))
    xyz = xyz.astype(np.float32)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    o = rng.uniform(0.02, 0.3, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.03, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r, img, radii = run_forward(backend, cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    sizes = (ref_ranges[:, 1] - ref_ranges[:, 0]).astype(np.int64)
    assert ((sizes > 64) & (sizes <= 512)).sum() >= 8 and ((sizes > 128) & (sizes <= 512)).sum() >= 2, sizes
    pl, ranges = r.download_binning(0, ref_n, 10 * 6)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(img, ref_img)


def test_dense_lists_use_the_large_bucket_class(backend):
    """4096 < n <= 8192 instances in a list on a FRESH handle (no class hint: the size-class kernels run, not the LDS-free
    stand-in): the <8192> bucket + rank sort with its capped bucket count (up to 2 keys per bucket on average), next to
    lists of the <4096> class and of <= 512 instances (wave bucket sort).  Instance lists equal the oracle's exactly."""
    W, H, f = 48, 32, 60.0
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    rng = np.random.default_rng(23)
    P = 9400
    xyz = rng.normal(0, 0.02, (P, 3)).astype(np.float32)
    xyz[:, 1] -= 0.53                                            # ~6000 in the tile above the centre
    xyz[6000:9000, 0] += 1.07                                    # ~3000 in the tile right of it
    xyz[9000:, 0] -= 1.07                                        # ~400 bottom left
    xyz[9000:, 1] += 1.06
    xyz[:, 2] = rng.uniform(-0.2, 0.2, P)
    o = rng.uniform(0.02, 0.4, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.012, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r, img, radii = run_forward(backend, cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    sizes = (ref_ranges[:, 1] - ref_ranges[:, 0]).astype(np.int64)
    assert ((sizes > 4096) & (sizes <= 8192)).any() and ((sizes > 512) & (sizes <= 4096)).any() and ((sizes > 64) & (sizes <= 512)).any(), sizes
    pl, ranges = r.download_binning(0, ref_n, 3 * 2)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(img, ref_img)


@pytest.mark.parametrize("n", [63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193])
@pytest.mark.parametrize("ties", [False, True])
def test_list_sizes_at_the_boundaries_of_the_sort_paths(backend, n, ties):
    """One list of exactly n instances, n on both sides of every hand-over of the per-tile sort: register network of 1 / 2 /
    4 / 8 keys per lane and the wave bucket sort (64, 128, 256, 512), size-class kernels (512 | 513, 4096 | 4097), run +
    merge sort (8192 | 8193).  `ties`: depths from 9 values (long runs of equal depth: order by Gaussian id, the bucket
    sort's overflow fallback).  The instance list equals the oracle's exactly."""
    W, H, f = 48, 32, 60.0
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    rng = np.random.default_rng(1000 + n)
    xyz = rng.normal(0, 0.004, (n, 3)).astype(np.float32)
    xyz[:, 1] -= 0.53                                             # all in the tile above the centre
    z = rng.uniform(-0.2, 0.2, n)
    xyz[:, 2] = np.round(z * 20) / 20 if ties else z
    xyz = xyz.astype(np.float32)
    o = rng.uniform(0.02, 0.3, n).astype(np.float32)
    cols = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    s = np.full((n, 3), 0.003, np.float32)
    q = np.tile([1, 0, 0, 0], (n, 1)).astype(np.float32)
    r, img, radii = run_forward(backend, cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform,
                                 cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
    sizes = (ref_ranges[:, 1] - ref_ranges[:, 0]).astype(np.int64)
    assert sizes.max() == n and (sizes > 0).sum() == 1, sizes     # exactly one list, of exactly n instances
    pl, ranges = r.download_binning(0, ref_n, 3 * 2)
    np.testing.assert_array_equal(ranges, ref_ranges)
    np.testing.assert_array_equal(pl, ref_pl)
    assert_image_close(img, ref_img)


@pytest.mark.parametrize("rows,cull,packed,n_views,ppl", [(1, 0, False, 4, 2), (2, 1, False, 4, 2), (2, 1, True, 4, 2), (2, 1, False, 5, 2),
                                                          (2, 1, False, 6, 2), (2, 1, False, 8, 2), (2, 1, False, 8, 4), (2, 1, True, 8, 4),
                                                          (2, 1, False, 7, 4), (1, 0, False, 8, 3)])
def test_two_pairs_per_launch_give_the_results_of_one_pair_per_launch(backend, rows, cull, packed, n_views, ppl):
    """GS2M_OPT_PAIR_BATCH: the projection / counting / scatter kernels take up to `ppl` (2 .. 4) stereo pairs per launch
    (blockIdx.y picks the pair, 1 / pairs of the workgroups per pair), scans / per-tile sort / compositing all the views in one
    grid.  Everything a call returns or leaves behind is identical to one pair per launch: images, u8 images, radii, instance
    counts, and -- per view -- the projected records and the instance lists.  ppl 2: 5 views = one batched pass + one single
    view; 6: a batched pass + a plain pair; 8: two batched passes.  ppl 4: 8 views = one pass of four pairs; 7 = a pass of three
    pairs + a single view.  ppl 3, 8 views: a pass of three pairs + a plain pair.  GS2M_OPT_PROJECT_SHARED_READ (round 6): the
    batched launches once with one grid row per pair (2 = never shared) and once with the pairs walked by the thread that owns
    the Gaussian (1 = always: one model read per launch; the default only does that for models of >= 1 M Gaussians)."""
    W, H, f = 176, 112, 150.0
    g, s, q, o, shs, left, right = scene(2600, 33, W, H, f, log_s=math.log(0.05))
    *_rest, left2, right2 = scene(10, 34, W, H, f, az=0.9)       # a second pair of cameras
    be = backend
    P = g["xyz"].shape[0]
    cams = [camera_from(c) for c in (left, right, left2, right2, right, left2, right2, left)][:n_views]
    gd = dict(xyz=be.dev(g["xyz"]), scaling=be.dev(g["scaling"]), rotation=be.dev(g["rotation"]),
              opacity=be.dev(g["opacity"]), features_dc=be.dev(g["features_dc"]), features_rest=be.dev(g["features_rest"]),
              raw=True, sh_degree=3)
    tiles = ((W + 15) // 16) * (((H + 15) // 16 + rows - 1) // rows)
    outs = []
    for batch, shared in ((0, 0), (ppl, 2), (ppl, 1), (ppl, 0)):     # 0 = auto: shared with a packed model, two groups per thread at ppl 4
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        r.set_option(_lib.OPT_TILE_ROWS, rows)
        r.set_option(_lib.OPT_PAIR_BATCH, batch)
        r.set_option(_lib.OPT_PROJECT_SHARED_READ, shared)
        if packed:
            r.pack_model(gd)
        for rep in range(2):          # second call: class hints of the sort come from the first
            res = r.render_views(gd, cams, want_radii=True, want_rgb8=True)
        nr = list(res["num_rendered"])
        outs.append((be.host(res["color"]).copy(), be.host(res["rgb8"]).copy(), be.host(res["radii"]).copy(), nr, r))
    a = outs[0]
    for b in outs[1:]:
        assert a[3] == b[3] and min(a[3]) > 1000
        np.testing.assert_array_equal(a[2], b[2])
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    for b in (outs[1:] if n_views == 4 and ppl == 2 else []):
        # one pair per launch leaves views 2, 3 in the arenas (as views 0, 1 of its last pass); two pairs per launch all four
        ra, rb = a[4], b[4]
        for v in range(2):
            ga, gb = ra.download_geometry(v, P), rb.download_geometry(2 + v, P)
            for k in ga:
                np.testing.assert_array_equal(ga[k], gb[k], err_msg=k)
            la, lb = ra.download_binning(v, a[3][2 + v], tiles), rb.download_binning(2 + v, b[3][2 + v], tiles)
            np.testing.assert_array_equal(la[0], lb[0])
            np.testing.assert_array_equal(la[1], lb[1])


def test_arena_overflow_is_detected_and_retried(backend):
    W, H, f = 96, 80, 90.0
    g, s, q, o, shs, left, _ = scene(2000, 8, W, H, f, log_s=math.log(0.08))
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    r.reserve(2000, 1, W, H, 1024)          # far too small on purpose
    args = (d(g["xyz"]), d(o), d(left.world_view_transform), d(left.full_proj_transform), d(left.camera_center),
            d(np.zeros(3, np.float32)), W, H, left.tanfovx, left.tanfovy)
    kw = dict(shs=d(shs), scales=d(s), rotations=d(q))
    img_nosync, _ = r.forward(*args, sync=False, **kw)
    nr, ov, req = r.status(1)
    assert ov and req > 1024
    img, radii = r.forward(*args, **kw)      # sync=True: grows and repeats
    ref_img, ref_radii, ref_n = oracle_forward(left, g["xyz"], o, [0, 0, 0], shs=shs, scales=s, rotations=q)
    assert r.last_num_rendered == ref_n
    assert_image_close(be.host(img), ref_img)


def test_overflow_in_an_earlier_call_is_not_erased_by_a_later_one(backend):
    """the overflow word is sticky on the device.  Call 1 overflows, call 2 on the same handle (a view that
    sees nothing of the scene) fits; the status query after both must still report the overflow and what call 1 needed."""
    W, H, f = 96, 80, 90.0
    g, s, q, o, shs, left, _ = scene(2000, 8, W, H, f, log_s=math.log(0.08))
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    r.reserve(2000, 1, W, H, 1024)
    kw = dict(shs=d(shs), scales=d(s), rotations=d(q))

    def args(cam):
        return (d(g["xyz"]), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy)

    away = Camera(0, left.R, np.asarray(left.T) + np.array([0.0, 0.0, -40.0]), left.FoVx, left.FoVy, W, H)  # scene behind
    r.forward(*args(left), sync=False, **kw)
    r.forward(*args(away), sync=False, **kw)
    nr, ov, req = r.status(1)
    assert nr[0] == 0                     # the last call rendered nothing ...
    assert ov and req > 1024              # ... and the earlier overflow is still reported
    nr, ov, req = r.status(1)
    assert not ov                         # consumed by the query
    r.forward(*args(away), sync=False, **kw)
    assert not r.status(1)[1]


def test_mark_visible(backend):
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 1.0, 0.8, 64, 48)
    xyz = np.random.default_rng(2).uniform(-6, 6, (1000, 3)).astype(np.float32)
    r = Rasterizer(0, lib=backend.lib)
    got = backend.host(r.mark_visible(backend.dev(xyz), backend.dev(cam.world_view_transform),
                                      backend.dev(cam.full_proj_transform))).astype(bool)
    np.testing.assert_array_equal(got, oracle.mark_visible(xyz, cam.world_view_transform, cam.full_proj_transform))


def test_argument_validation_errors(backend):
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 1.0, 0.8, 64, 48)
    d = backend.dev
    r = Rasterizer(0, lib=backend.lib)
    xyz = np.zeros((4, 3), np.float32)
    base = (d(xyz), d(np.full(4, .5, np.float32)), d(cam.world_view_transform), d(cam.full_proj_transform),
            d(cam.camera_center), d(np.zeros(3, np.float32)), 64, 48, cam.tanfovx, cam.tanfovy)
    with pytest.raises(RuntimeError, match="excatly one of either SHs or precomputed colors"):
        r.forward(*base, scales=d(np.ones((4, 3), np.float32)), rotations=d(np.ones((4, 4), np.float32)))
    with pytest.raises(RuntimeError, match="exactly one of either scale/rotation pair"):
        r.forward(*base, colors_precomp=d(np.ones((4, 3), np.float32)))


@pytest.mark.parametrize("variant", [0, 4])
def test_blend_variants_match_the_oracle(backend, variant):
    """Every compositing kernel variant (GS2M_OPT_BLEND_VARIANT) on a ragged image, a crowded
    saturating tile stack and with exact culling on."""
    be = backend
    d = be.dev
    # ragged image, SH colours, non-zero background
    W, H, f = 200, 136, 180.0
    g, s, q, o, shs, left, _ = scene(3000, 11, W, H, f)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    for cull in (0, 1):
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_BLEND_VARIANT, variant)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        img, _ = r.forward(d(g["xyz"]), d(o), d(left.world_view_transform), d(left.full_proj_transform),
                           d(left.camera_center), d(bg), W, H, left.tanfovx, left.tanfovy, shs=d(shs), scales=d(s),
                           rotations=d(q))
        ref_img, _, _ = oracle_forward(left, g["xyz"], o, bg, shs=shs, scales=s, rotations=q)
        assert_image_close(be.host(img), ref_img)
    # crowded tiles (early saturation, > 64-instance batches, merge-path sort)
    W, H, f = 48, 32, 60.0
    P = 9000
    rng = np.random.default_rng(3)
    xyz = rng.normal(0, 0.02, (P, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-1, 1, P)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    oo = rng.uniform(0.02, 0.4, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    ss = np.full((P, 3), 0.05, np.float32)
    qq = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r = Rasterizer(0, lib=be.lib)
    r.set_option(_lib.OPT_BLEND_VARIANT, variant)
    img, _ = r.forward(d(xyz), d(oo), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                       d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols),
                       scales=d(ss), rotations=d(qq))
    ref_img, _, _ = oracle_forward(cam, xyz, oo, [0, 0, 0], colors_precomp=cols, scales=ss, rotations=qq)
    assert_image_close(be.host(img), ref_img)


@pytest.mark.parametrize("rows", [1, 2])
def test_thin_rects_take_the_per_lane_path(backend, rows):
    """Rects that are one binning tile wide or one high and at most 4 tiles (k_count_tiles / k_scatter: counted and scattered by
    their own lane, no staging, no tile mask -- round 4) next to 2 x 2 and larger ones in the same waves: needles along x and along
    y (1 x n, n x 1 tiles), dots (1 x 1), and isotropic blobs.  Reference instance lists exactly (16 x 16 tiles, no cull); with
    16 x 32 tiles and the exact cull on: the same image and the oracle's culled instance count."""
    W, H, f = 256, 192, 200.0
    rng = np.random.default_rng(404)
    P = 1500
    xyz = rng.uniform(-1.2, 1.2, (P, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-0.3, 0.3, P)
    s = np.full((P, 3), 0.004, np.float32)                       # dots: ~1 px sigma -> mostly 1 x 1 rects
    kind = rng.integers(0, 4, P)
    s[kind == 1, 0] = rng.uniform(0.03, 0.08, (kind == 1).sum())  # needles along x (the exact cull makes their rects 1 tile high)
    s[kind == 2, 1] = rng.uniform(0.03, 0.08, (kind == 2).sum())  # needles along y
    s[kind == 3] = rng.uniform(0.05, 0.15, ((kind == 3).sum(), 1)).astype(np.float32)   # blobs: 3 x 3 tiles and more
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    o = rng.uniform(0.05, 0.9, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    be = backend
    d = be.dev
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                                 cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    tt = geom_ref["tiles_touched"]
    assert (tt == 1).sum() > 100 and (tt == 2).sum() > 100 and (tt > 4).sum() > 100, np.bincount(np.minimum(tt, 9))
    for cull in (0, 1):
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        r.set_option(_lib.OPT_TILE_ROWS, rows)
        img, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                               d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols), scales=d(s),
                               rotations=d(q))
        ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q,
                                                   exact_cull=bool(cull))
        np.testing.assert_array_equal(be.host(radii), ref_radii)
        assert_image_close(be.host(img), ref_img)
        assert_image_attributed(be.host(img), geom_ref, W, H, rgb=cols)
        if rows == 1 and cull == 0:
            assert r.last_num_rendered == ref_n
            ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
            pl, ranges = r.download_binning(0, ref_n, ((W + 15) // 16) * ((H + 15) // 16))
            np.testing.assert_array_equal(ranges, ref_ranges)
            np.testing.assert_array_equal(pl, ref_pl)
        elif rows == 1:
            assert abs(r.last_num_rendered - ref_n) <= max(2, 2e-3 * ref_n)


@pytest.mark.parametrize("rows", [1, 2])
def test_lane_tiles_settings_give_identical_lists(backend, rows):
    """GS2M_OPT_BIN_LANE_TILES (round 6): rects of at most that many binning tiles are walked by their own lane -- in k_count_tiles
    when they are not tested, in k_scatter always (it replays the mask) -- instead of through the staged walk.  A tuning
    option: instance lists, tile ranges and image must be the same bit for bit at 0 (round 5's thin rects only), the default 4,
    6 and 16, at every level of the exact cull (0, 1, 2 = rects of at most 4 tiles untested) -- on dots, 2 x 2 blobs, needles,
    3 x 3 .. 6 x 6 blobs and needles longer than 64 tiles (one tile high after the cull: the whole-wave walk tests every tile of
    those and needs the record's geometry part, which the counting pass now only loads for rects that are tested)."""
    W, H, f = 1184, 96, 300.0
    rng = np.random.default_rng(606)
    P = 2600
    xyz = np.zeros((P, 3), np.float32)
    xyz[:, 0] = rng.uniform(-7.5, 7.5, P)
    xyz[:, 1] = rng.uniform(-0.55, 0.55, P)
    xyz[:, 2] = rng.uniform(-0.2, 0.2, P)
    s = np.full((P, 3), 0.004, np.float32)                        # dots
    kind = rng.integers(0, 6, P)
    s[kind == 1] = 0.018                                          # ~2 x 2 tiles
    s[kind == 2, 0] = rng.uniform(0.03, 0.09, (kind == 2).sum())  # needles along x
    s[kind == 3, 1] = rng.uniform(0.03, 0.09, (kind == 3).sum())  # needles along y
    s[kind == 4] = rng.uniform(0.03, 0.12, ((kind == 4).sum(), 1)).astype(np.float32)   # blobs
    long_ones = np.nonzero(kind == 5)[0][:6]
    s[long_ones, 0] = rng.uniform(2.5, 4.0, len(long_ones))       # needles of > 64 tiles (3 sigma ~ 1000 px), 1 tile high after the cull
    s[long_ones, 1] = 0.002
    xyz[long_ones, 0] = rng.uniform(-1.0, 1.0, len(long_ones))
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    o = rng.uniform(0.05, 0.9, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    be = backend
    d = be.dev
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                                 cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    tt = geom_ref["tiles_touched"]
    assert (tt == 1).sum() > 50 and (tt == 4).sum() > 50 and (tt > 64).sum() >= 4, np.bincount(np.minimum(tt, 70))
    n_tiles = ((W + 15) // 16) * ((H + 16 * rows - 1) // (16 * rows))
    for cull in (0, 1, 2):
        base = None
        for lane_tiles in (0, 4, 6, 16):
            r = Rasterizer(0, lib=be.lib)
            r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
            r.set_option(_lib.OPT_TILE_ROWS, rows)
            r.set_option(_lib.OPT_BIN_LANE_TILES, lane_tiles)
            img, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                                   d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols), scales=d(s),
                                   rotations=d(q))
            n = r.last_num_rendered
            pl, ranges = r.download_binning(0, n, n_tiles)
            got = (n, pl.copy(), ranges.copy(), be.host(img).copy())
            if base is None:
                base = got
                if rows == 1:
                    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q,
                                                               exact_cull=cull)
                    np.testing.assert_array_equal(be.host(radii), ref_radii)
                    assert n == ref_n
                    if cull == 0:
                        ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
                        np.testing.assert_array_equal(ranges, ref_ranges)
                        np.testing.assert_array_equal(pl, ref_pl)
                    assert_image_close(be.host(img), ref_img)
            else:
                assert got[0] == base[0], (lane_tiles, cull)
                np.testing.assert_array_equal(got[2], base[2])
                np.testing.assert_array_equal(got[1], base[1])
                np.testing.assert_array_equal(got[3], base[3])
    with pytest.raises(RuntimeError):
        Rasterizer(0, lib=be.lib).set_option(_lib.OPT_BIN_LANE_TILES, 17)
    with pytest.raises(RuntimeError):
        Rasterizer(0, lib=be.lib).set_option(_lib.OPT_PROJECT_SHARED_READ, 3)


@pytest.mark.parametrize("cull", [0, 1])
def test_huge_and_tiny_gaussians_mixed(backend, cull):
    """Rect areas from 1 tile to the whole 13 x 9 tile grid in one wave: exercises the balanced tile
    expansion, the <= 64-tile masks and the > 64-tile re-test path of the scatter."""
    W, H, f = 200, 136, 180.0
    rng = np.random.default_rng(77)
    P = 600
    xyz = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    s = np.exp(rng.normal(math.log(0.02), 0.5, (P, 3))).astype(np.float32)
    big = rng.choice(P, 12, replace=False)
    s[big] = rng.uniform(0.3, 1.5, (12, 3)).astype(np.float32)          # screen-filling, anisotropic
    q = rng.normal(0, 1, (P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    o = rng.uniform(0.01, 0.9, P).astype(np.float32)
    o[big[:3]] = 0.003                                                   # below 1/255: never contributes
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    pose = synthetic.ring_pose(0.7, 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W, H, f, f, 0.2)
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
    r.set_option(_lib.OPT_BLEND_VARIANT, 4)
    img, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform),
                           d(cam.camera_center), d(np.array([0.2, 0.1, 0.0], np.float32)), W, H, cam.tanfovx,
                           cam.tanfovy, colors_precomp=d(cols), scales=d(s), rotations=d(q))
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0.2, 0.1, 0.0], colors_precomp=cols, scales=s,
                                               rotations=q, exact_cull=bool(cull))
    np.testing.assert_array_equal(be.host(radii), ref_radii)
    if cull:
        assert abs(r.last_num_rendered - ref_n) <= max(2, 2e-3 * ref_n)
    else:
        assert r.last_num_rendered == ref_n
    assert ref_n > 2000
    assert_image_close(be.host(img), ref_img)
    geom_ref = oracle.preprocess(xyz, s, q, o, None, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                                 cam.tanfovx, cam.tanfovy, colors_precomp=cols)
    assert_image_attributed(be.host(img), geom_ref, W, H, bg=(0.2, 0.1, 0.0), rgb=cols)


def test_packed_sh_layout_gives_identical_results(backend):
    """gs2m_raster_pack_sh (one-time wave-transposed SH copy) must not change a bit of the output,
    for both source layouts (concatenated features, dc + rest) and a P that is not a multiple of 64."""
    W, H, f = 160, 120, 150.0
    g, s, q, o, shs, left, right = scene(2500 + 37, 23, W, H, f)
    be = backend
    cams = [camera_from(left), camera_from(right)]
    base = dict(xyz=be.dev(g["xyz"]), scaling=be.dev(g["scaling"]), rotation=be.dev(g["rotation"]),
                opacity=be.dev(g["opacity"]), raw=True, sh_degree=3)
    for layout in ("split", "cat"):
        gd = dict(base)
        if layout == "split":
            gd["features_dc"], gd["features_rest"] = be.dev(g["features_dc"]), be.dev(g["features_rest"])
        else:
            gd["features"] = be.dev(shs)
        r = Rasterizer(0, lib=be.lib)
        a = be.host(r.render_views(gd, cams, want_radii=True)["color"]).copy()
        r.pack_sh(gd)
        res = r.render_views(gd, cams, want_radii=True)
        np.testing.assert_array_equal(be.host(res["color"]), a)
        geom = r.download_geometry(1, 2537)
        r2 = Rasterizer(0, lib=be.lib)
        r2.render_views(gd, cams)
        np.testing.assert_array_equal(geom["rgb"], r2.download_geometry(1, 2537)["rgb"])


@pytest.mark.parametrize("rows", [1, 2])
def test_spatially_ordered_packed_model_gives_identical_results(backend, rows):
    """gs2m_raster_pack_model (Morton-ordered packed copy of the model) must not change a bit: image, radii, the
    per-Gaussian taps and the instance lists are those of the unordered model -- including the order of Gaussians at
    EXACTLY equal depth (duplicated centres with different colours: the keys carry ids, so ties resolve in id order
    as the reference's stable sort does).  Also checks that a non-permutation is refused."""
    from gs2mesh_amd.rasterizer import morton_order
    W, H, f = 176, 136, 160.0
    P = 3000 + 21
    g, s, q, o, shs, left, right = scene(P, 29, W, H, f)
    rng = np.random.default_rng(5)
    # exact depth ties: 300 Gaussians share the centre of another one (different shape / colour / opacity)
    src = rng.choice(P, 300, replace=False)
    dst = rng.choice(P, 300, replace=False)
    g["xyz"][dst] = g["xyz"][src]
    be = backend
    cams = [camera_from(left), camera_from(right)]
    gd = dict(xyz=be.dev(g["xyz"]), scaling=be.dev(g["scaling"]), rotation=be.dev(g["rotation"]),
              opacity=be.dev(g["opacity"]), features_dc=be.dev(g["features_dc"]), features_rest=be.dev(g["features_rest"]),
              raw=True, sh_degree=3)
    tiles = ((W + 15) // 16) * (((H + 15) // 16 + rows - 1) // rows)
    outs = []
    for packed in (False, True):
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, 1)
        r.set_option(_lib.OPT_TILE_ROWS, rows)
        if packed:
            order = r.pack_model(gd)
            assert sorted(np.asarray(be.host(order)).tolist()) == list(range(P))
            assert not np.array_equal(np.asarray(be.host(order)), np.arange(P))
        res = r.render_views(gd, cams, want_radii=True, want_rgb8=True)
        geom = r.download_geometry(1, P)
        lists = [r.download_binning(v, res["num_rendered"][v], tiles) for v in range(2)]
        outs.append((be.host(res["color"]).copy(), be.host(res["rgb8"]).copy(), be.host(res["radii"]).copy(), geom, lists,
                     list(res["num_rendered"])))
    a, b = outs
    assert a[5] == b[5] and min(a[5]) > 3000
    np.testing.assert_array_equal(a[2], b[2])
    for k in a[3]:
        np.testing.assert_array_equal(a[3][k], b[3][k], err_msg=k)
    for v in range(2):
        np.testing.assert_array_equal(a[4][v][0], b[4][v][0])   # point lists (ids), ties included
        np.testing.assert_array_equal(a[4][v][1], b[4][v][1])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # ties really occur inside the lists
    geom = a[3]
    pl = a[4][1][0]
    d = geom["depths"][pl]
    assert (d[1:] == d[:-1]).sum() > 20
    # a non-permutation is refused
    bad = np.arange(P, dtype=np.int32)
    bad[7] = 8
    r = Rasterizer(0, lib=be.lib)
    with pytest.raises(RuntimeError, match="permutation"):
        r.pack_model(gd, order=be.dev(bad))


def test_hip_path_against_reference_golden(backend):
    """The HIP rasteriser against outputs of the REFERENCE'S OWN KERNELS (tests/golden/ref_forward.npz, produced by
    oracle/_ref = forward.cu / rasterizer_impl.cu kernels compiled for the CPU, tests/golden/make_golden.py):
    projected records and instance lists bit-exact, image within the stated tolerance."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_forward.npz"))
    W, H, P = int(z["W"]), int(z["H"]), z["xyz"].shape[0]
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    img, radii = r.forward(d(z["xyz"]), d(z["opacity"]), d(z["viewmatrix"]), d(z["projmatrix"]), d(z["campos"]), d(z["bg"]),
                           W, H, float(z["tanfovx"]), float(z["tanfovy"]), shs=d(z["shs"]), scales=d(z["scales"]),
                           rotations=d(z["rotations"]))
    radii = be.host(radii)
    np.testing.assert_array_equal(radii, z["out_radii"])
    vis = radii > 0
    geom = r.download_geometry(0, P)
    np.testing.assert_array_equal(geom["tiles_touched"], z["out_tiles_touched"])
    for k in ("means2D", "depths", "conic_opacity", "rgb"):
        np.testing.assert_array_equal(geom[k][vis], z["out_" + k][vis], err_msg=k)
    n = int(z["out_num_rendered"])
    assert r.last_num_rendered == n
    pl, ranges = r.download_binning(0, n, ((W + 15) // 16) * ((H + 15) // 16))
    np.testing.assert_array_equal(pl, z["out_point_list"])
    np.testing.assert_array_equal(ranges, z["out_ranges"])
    assert_image_close(be.host(img), z["out_color"])


@pytest.mark.parametrize("opts", [{_lib.OPT_BIN_WORKGROUPS: 2, _lib.OPT_BIN_WG_THREADS: 128},
                                  {_lib.OPT_BIN_WORKGROUPS: 3, _lib.OPT_BIN_WG_THREADS: 1024},
                                  {_lib.OPT_BIN_WORKGROUPS: 64, _lib.OPT_BIN_WG_THREADS: 64},
                                  {_lib.OPT_BLEND_MODE: 0}, {_lib.OPT_BLEND_MODE: 2}])
def test_tuning_options_do_not_change_results(backend, opts):
    """The tuning options of gs2m_raster_set_option: k_count_tiles / k_scatter with few large workgroup chunks (several loop
    iterations per workgroup, partial last iteration) and with small workgroups; GS2M_OPT_BLEND_MODE 0 = the compositing loop
    with lane masks in scalar registers (raster_blend.h MODE 0).  Same records, instance lists and
    image as the reference golden."""
    import os
    be = backend
    d = be.dev
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_forward.npz"))
    W, H = int(z["W"]), int(z["H"])
    for cull in (0, 1):
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        for k, v in opts.items():
            r.set_option(k, v)
        img, radii = r.forward(d(z["xyz"]), d(z["opacity"]), d(z["viewmatrix"]), d(z["projmatrix"]), d(z["campos"]), d(z["bg"]),
                               W, H, float(z["tanfovx"]), float(z["tanfovy"]), shs=d(z["shs"]), scales=d(z["scales"]),
                               rotations=d(z["rotations"]))
        assert np.array_equal(be.host(radii), z["out_radii"])
        dd = np.abs(be.host(img) - z["out_color"])
        assert (dd > 1e-5).mean() <= 1e-4 and dd.max() <= 6e-3
        if cull == 0:
            n = int(z["out_num_rendered"])
            assert r.last_num_rendered == n
            pl, ranges = r.download_binning(0, n, ((W + 15) // 16) * ((H + 15) // 16))
            assert np.array_equal(pl, z["out_point_list"]) and np.array_equal(ranges, z["out_ranges"])
    with pytest.raises(RuntimeError):
        Rasterizer(0, lib=be.lib).set_option(_lib.OPT_BIN_WG_THREADS, 100)     # not a multiple of 64


@pytest.mark.parametrize("cull", [0, 1])
def test_tile_rows_2_same_image_fewer_instances(backend, cull):
    """GS2M_OPT_TILE_ROWS 2 (16 x 32 binning tiles, one wave composites 8 pixels per lane): the image still matches
    the oracle -- the reference's 16 x 16 tile rect bounds every contribution -- with fewer (Gaussian, tile)
    instances.  Scenes: reference golden (odd number of tile rows: 8.5), huge/tiny mix, crowded saturating stack, fused
    stereo pair from raw parameters."""
    import os
    be = backend
    d = be.dev

    def ras():
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        r.set_option(_lib.OPT_TILE_ROWS, 2)
        return r

    # 1. reference golden
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_forward.npz"))
    W, H = int(z["W"]), int(z["H"])
    r = ras()
    img, radii = r.forward(d(z["xyz"]), d(z["opacity"]), d(z["viewmatrix"]), d(z["projmatrix"]), d(z["campos"]), d(z["bg"]),
                           W, H, float(z["tanfovx"]), float(z["tanfovy"]), shs=d(z["shs"]), scales=d(z["scales"]),
                           rotations=d(z["rotations"]))
    np.testing.assert_array_equal(be.host(radii), z["out_radii"])
    assert r.last_num_rendered < 0.9 * int(z["out_num_rendered"])
    assert_image_close(be.host(img), z["out_color"])
    # the binning taps describe the 16 x 32 tiles: sizes are consistent
    n_tiles2 = ((W + 15) // 16) * ((H + 31) // 32)
    pl, ranges = r.download_binning(0, r.last_num_rendered, n_tiles2)
    assert (ranges[:, 1] - ranges[:, 0]).sum() == r.last_num_rendered and pl.max() < z["xyz"].shape[0]
    # 2. huge and tiny splats (rects from 1 to all tiles, opacity below 1/255, anisotropic)
    W, H, f = 200, 136, 180.0
    rng = np.random.default_rng(77)
    P = 600
    xyz = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    s = np.exp(rng.normal(math.log(0.02), 0.5, (P, 3))).astype(np.float32)
    big = rng.choice(P, 12, replace=False)
    s[big] = rng.uniform(0.3, 1.5, (12, 3)).astype(np.float32)
    q = rng.normal(0, 1, (P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    o = rng.uniform(0.01, 0.9, P).astype(np.float32)
    o[big[:3]] = 0.003
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    pose = synthetic.ring_pose(0.7, 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W, H, f, f, 0.2)
    r = ras()
    img, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                           d(np.array([0.2, 0.1, 0.0], np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols),
                           scales=d(s), rotations=d(q))
    ref_img, ref_radii, _ = oracle_forward(cam, xyz, o, [0.2, 0.1, 0.0], colors_precomp=cols, scales=s, rotations=q)
    np.testing.assert_array_equal(be.host(radii), ref_radii)
    assert_image_close(be.host(img), ref_img)
    # 3. crowded tiles (early saturation, several 64-instance batches, workgroup sort path), image 48 x 32 = 3 x 1 tiles
    W, H, f = 48, 32, 60.0
    P = 9000
    rng = np.random.default_rng(3)
    xyz = rng.normal(0, 0.02, (P, 3)).astype(np.float32)
    xyz[:, 2] = rng.uniform(-1, 1, P)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * f), 2 * math.atan2(H, 2 * f), W, H)
    oo = rng.uniform(0.02, 0.4, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    ss = np.full((P, 3), 0.05, np.float32)
    qq = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    r = ras()
    img, _ = r.forward(d(xyz), d(oo), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                       d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols), scales=d(ss),
                       rotations=d(qq))
    ref_img, _, _ = oracle_forward(cam, xyz, oo, [0, 0, 0], colors_precomp=cols, scales=ss, rotations=qq)
    assert_image_close(be.host(img), ref_img)
    # 4. fused stereo pair from raw parameters == the same handle with 16 x 16 tiles
    W, H, f = 160, 120, 150.0
    g, s, q, o, shs, left, right = scene(2500, 21, W, H, f)
    gd = dict(xyz=d(g["xyz"]), scaling=d(g["scaling"]), rotation=d(g["rotation"]), opacity=d(g["opacity"]),
              features_dc=d(g["features_dc"]), features_rest=d(g["features_rest"]), raw=True, sh_degree=3)
    r2 = ras()
    a = r2.render_views(gd, [camera_from(left), camera_from(right)], want_rgb8=True)
    r1 = Rasterizer(0, lib=be.lib)
    r1.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
    b = r1.render_views(gd, [camera_from(left), camera_from(right)], want_rgb8=True)
    assert max(a["num_rendered"]) < max(b["num_rendered"])
    diff = np.abs(be.host(a["color"]) - be.host(b["color"]))
    assert (diff > 1e-5).mean() < 1e-4 and diff.max() < 6e-3


@pytest.mark.parametrize("rows", [1, 2])
def test_4k_image_tile_cursors_fit_the_lds(backend, rows):
    """3840 x 2160 = 32 400 reference tiles: the per-workgroup tile cursors take 130 KB of the 160 KiB LDS, so the
    counting / scatter workgroups run with fewer threads (less wave staging) instead of failing."""
    W, H = 3840, 2160
    rng = np.random.default_rng(1)
    P = 300
    xyz = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    s = np.full((P, 3), 0.01, np.float32)
    q = np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32)
    o = rng.uniform(0.2, 0.9, P).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    cam = Camera(0, np.eye(3), np.array([0, 0, 4.0]), 2 * math.atan2(W, 2 * 3000.0), 2 * math.atan2(H, 2 * 3000.0), W, H)
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    r.set_option(_lib.OPT_TILE_ROWS, rows)
    img, radii = r.forward(d(xyz), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                           d(np.zeros(3, np.float32)), W, H, cam.tanfovx, cam.tanfovy, colors_precomp=d(cols), scales=d(s),
                           rotations=d(q))
    ref_img, ref_radii, ref_n = oracle_forward(cam, xyz, o, [0, 0, 0], colors_precomp=cols, scales=s, rotations=q)
    np.testing.assert_array_equal(be.host(radii), ref_radii)
    if rows == 1:
        assert r.last_num_rendered == ref_n
    assert_image_close(be.host(img), ref_img)


@pytest.mark.parametrize("rows,cull", [(1, 0), (2, 1)])
def test_trained_like_splats_with_flip_attribution(backend, rows, cull):
    """`synthetic.trained_like`: anisotropy 10-100 : 1, 30 % of the opacities at the 0.99 alpha cap, background splats
    hundreds of tiles wide, exact depth ties.  Projected record / instance lists bit-exact as for the isotropic scenes;
    the image is held to the CHECKED statement (oracle/parity.py:flip_attribution): |delta| <= 2e-4 wherever no
    threshold decision of renderCUDA (forward.cu:336-350) sits within 1e-5 of its threshold, and within the bound of
    the flipped contribution elsewhere."""
    from oracle import parity
    W, H, f = 200, 136, 180.0
    P = 5000
    g = synthetic.trained_like(P, 23, math.log(0.03), focal=f)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    assert (o >= 0.985).mean() > 0.25 and (s.max(axis=1) / s.min(axis=1) > 10).mean() > 0.9
    pose = synthetic.ring_pose(0.3, 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W, H, f, f, 0.245)
    bg = [0.2, 0.1, 0.3]
    be = backend
    d = be.dev
    r = Rasterizer(0, lib=be.lib)
    r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
    r.set_option(_lib.OPT_TILE_ROWS, rows)
    img, radii = r.forward(d(g["xyz"]), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                           d(np.asarray(bg, np.float32)), W, H, cam.tanfovx, cam.tanfovy, shs=d(shs), scales=d(s), rotations=d(q))
    img, radii = be.host(img), be.host(radii)
    ref_img, ref_radii, ref_n = oracle_forward(cam, g["xyz"], o, bg, shs=shs, scales=s, rotations=q)
    np.testing.assert_array_equal(radii, ref_radii)
    assert ref_radii.max() > 150 and (ref_radii > 0).sum() > 3000          # background splats: hundreds of tiles
    if rows == 1 and cull == 0:
        assert r.last_num_rendered == ref_n
        n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        geom_ref = oracle.preprocess(g["xyz"], s, q, o, shs, cam.world_view_transform, cam.full_proj_transform,
                                     cam.camera_center, W, H, cam.tanfovx, cam.tanfovy)
        pl, ranges = r.download_binning(0, ref_n, n_tiles)
        ref_pl, ref_ranges = oracle.bin_instances(geom_ref, W, H)
        np.testing.assert_array_equal(ranges, ref_ranges)
        np.testing.assert_array_equal(pl, ref_pl)                           # depth ties resolved in id order
    graw = dict(g)
    fa = parity.flip_attribution(graw, cam, W, H, img, ref_img, bg)
    assert fa["ok"], fa
    assert fa["max_abs_clean"] <= 2e-4, fa
    assert_image_close(img, ref_img, frac_tol=3e-4)



@pytest.mark.parametrize("rows,cull,log_s", [(1, 0, math.log(0.03)), (2, 1, math.log(0.03)), (2, 1, math.log(0.006))])
def test_blend_loop_forms_are_bit_identical(backend, rows, cull, log_s):
    """GS2M_OPT_BLEND_MODE 0 / 2 / 3 are loop forms of the same arithmetic (raster_blend.h): lane masks in scalar registers,
    "all four quadrants + flag-free runs" (round 5; mode 1, the execution-mask form, was removed in round 6), and mode 2 with the
    alpha cap applied to every instance instead of a run split at every capped one (round 6).  On `synthetic.trained_like` --
    30 % of the opacities at the alpha cap (the general path of every mode), saturating pixels, near-singular conics, lists
    longer than one staging batch -- the images must be equal bit for bit, at both binning tile sizes."""
    W, H, f = 200, 136, 180.0
    g = synthetic.trained_like(5000, 23, log_s, focal=f)
    s, q, o = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    pose = synthetic.ring_pose(0.3, 3.5)
    pose = np.concatenate([pose[0], pose[1][:, None]], axis=1)
    cam, _ = synthetic.stereo_cameras(pose, W, H, f, f, 0.245)
    bg = np.asarray([0.2, 0.1, 0.3], np.float32)
    be = backend
    d = be.dev
    imgs = []
    for mode in (0, 2, 3, "2p"):     # 3 = mode 2 with the alpha cap for every instance (round 6); 2p = mode 2 through the instrumented build
        r = Rasterizer(0, lib=be.lib)
        r.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
        r.set_option(_lib.OPT_TILE_ROWS, rows)
        r.set_option(_lib.OPT_BLEND_MODE, 2 if mode == "2p" else mode)
        if mode == "2p":
            r.set_option(_lib.OPT_BLEND_PROFILE, 1)
        img, _ = r.forward(d(g["xyz"]), d(o), d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center),
                           d(bg), W, H, cam.tanfovx, cam.tanfovy, shs=d(shs), scales=d(s), rotations=d(q))
        imgs.append(np.array(be.host(img)))
        if mode == "2p":
            c = r.blend_cycles()
    assert np.array_equal(imgs[0], imgs[1])
    assert np.array_equal(imgs[0], imgs[2])
    assert np.array_equal(imgs[0], imgs[3])
    from gs2mesh_amd.rasterizer import auto_blend_mode
    assert auto_blend_mode(dict(opacity=g["opacity"], raw=True)) == 3 and auto_blend_mode(dict(opacity=o)) == 3     # 30 % at the cap
    assert auto_blend_mode(dict(opacity=np.full(100, 0.5, np.float32))) == 2 and auto_blend_mode({}) == 2
    assert c["batches"] > 0 and 0 < c["staged_instances"] <= c["listed_instances"] * (2 if rows == 2 else 1)
    assert (imgs[0] != bg[:, None, None]).any()
    with pytest.raises(RuntimeError):
        Rasterizer(0, lib=be.lib).set_option(_lib.OPT_BLEND_MODE, 1)


def test_flip_bounds_count_the_decisions_at_their_thresholds():
    """oracle_render_flip_bounds on hand-made cases: an alpha exactly at 1/255, a transmittance product exactly at 1e-4,
    and a clean pixel."""
    W = H = 16
    ranges = np.array([[0, 2]], np.uint32)
    pl = np.array([0, 1], np.uint32)
    means = np.array([[3.0, 3.0], [9.0, 9.0]], np.float32)
    # instance 0: isotropic, opacity 1/255 at its centre pixel -> alpha == 1/255 at (3,3) only (falls off elsewhere)
    # instance 1: opacity 0.99 at (9,9): test_T = 0.01 -> far from 1e-4
    co = np.array([[0.5, 0.0, 0.5, 1.0 / 255.0], [0.5, 0.0, 0.5, 0.99]], np.float32)
    fb = oracle.render_flip_bounds(W, H, ranges, pl, means, co, cmax=1.0, rel_eps=1e-5)
    assert fb["n_alpha"][3, 3] == 1 and fb["n_alpha"].sum() == 1
    assert fb["bound"][3, 3] > 0 and fb["bound"][9, 9] == 0 and fb["n_T"].sum() == 0
    # a stack whose running transmittance hits 1e-4 exactly: two layers of alpha 0.99 give T = 1e-4 (within rounding)
    pl2 = np.array([0, 1, 2], np.uint32)
    means2 = np.array([[5.0, 5.0]] * 3, np.float32)
    co2 = np.array([[0.5, 0.0, 0.5, 0.99]] * 3, np.float32)
    fb2 = oracle.render_flip_bounds(W, H, np.array([[0, 3]], np.uint32), pl2, means2, co2, cmax=1.0, rel_eps=1e-5)
    assert fb2["n_T"][5, 5] >= 1 and fb2["bound"][5, 5] >= 0.009            # T = 0.01 in front of the deciding layer
