"""Parity measurements of a rendered image against the CPU oracle (TEST INFRASTRUCTURE ONLY).

Used by ``tests/`` and by the ``cpu_baseline`` leg of ``bench.py`` (the oracle as the checker, never as the
thing measured).  The numbers reported here are the ones DESIGN.md section 2 quotes and the ones the
full-size tests bound (tolerances = at most 2x the measured values).
"""
from __future__ import annotations

import numpy as np

from . import activate, bin_instances, preprocess, rasterize_forward, ref_available, ref_forward, render, render_flip_bounds


def quantize_u8(img_chw: np.ndarray) -> np.ndarray:
    """cv2.imwrite's float -> u8 conversion of ``render * 255`` (renderer_utils.py:389-390): HWC u8."""
    return np.clip(np.rint(np.asarray(img_chw, np.float32).transpose(1, 2, 0) * np.float32(255.0)), 0, 255).astype(np.uint8)


def image_parity(img: np.ndarray, ref: np.ndarray, rgb8: np.ndarray | None = None) -> dict:
    """``img`` / ``ref``: [3,H,W] f32.  ``rgb8``: the device-quantised [H,W,3] u8 image, if produced."""
    d = np.abs(np.asarray(img, np.float64) - np.asarray(ref, np.float64))
    mse = float((d * d).mean())
    out = dict(max_abs=float(d.max()), mean_abs=float(d.mean()),
               psnr_db=(float(20.0 * np.log10(1.0 / np.sqrt(mse))) if mse > 0 else float("inf")),   # GS/utils/image_utils.py:17-19
               frac_gt_1e5=float((d > 1e-5).mean()), frac_gt_1e4=float((d > 1e-4).mean()),
               n_values=int(d.size))
    q_ref = quantize_u8(ref)
    q_img = quantize_u8(img) if rgb8 is None else np.asarray(rgb8)
    dq = np.abs(q_img.astype(np.int16) - q_ref.astype(np.int16))
    out["u8_flipped_values"] = int((dq > 0).sum())          # values whose 8-bit code differs from the oracle's
    out["u8_flipped_pixels"] = int((dq.max(axis=2) > 0).sum())
    out["u8_max_lsb"] = int(dq.max())
    return out


CLEAN_BAR = 2e-4     # max |delta| of the fp32 image where no threshold decision can flip


def _attribute(geom: dict, W: int, H: int, img: np.ndarray, ref: np.ndarray | None, bg, rel_eps: float) -> dict:
    pl, ranges = bin_instances(geom, W, H)
    bgv = np.asarray(bg, np.float32)
    if ref is None:
        ref = render(W, H, ranges, pl, geom["means2D"], geom["rgb"], geom["conic_opacity"], bgv)[0]
    cmax = float(max(1.0, float(geom["rgb"].max(initial=0.0)), float(np.max(bg))))
    fb = render_flip_bounds(W, H, ranges, pl, geom["means2D"], geom["conic_opacity"], cmax, rel_eps)
    d = np.abs(np.asarray(img, np.float64) - np.asarray(ref, np.float64)).max(axis=0)     # [H,W], worst channel
    cand = (fb["bound"] > 0) | (fb["cond"] > CLEAN_BAR / 4)
    clean_max = float(d[~cand].max()) if (~cand).any() else 0.0
    excess = d - (CLEAN_BAR + 1.01 * (fb["bound"].astype(np.float64) + fb["cond"].astype(np.float64)))
    excess_nocond = d - (CLEAN_BAR + 1.01 * fb["bound"].astype(np.float64))
    worst = np.argsort(excess.ravel())[::-1][:4]
    detail = [dict(y=int(i // W), x=int(i % W), delta=float(d.ravel()[i]), bound=float(fb["bound"].ravel()[i]),
                   cond=float(fb["cond"].ravel()[i])) for i in worst if excess.ravel()[i] > 0]
    return dict(flip_pixels=int(cand.sum()), ill_conditioned_pixels=int((fb["cond"] > CLEAN_BAR / 4).sum()),
                unexplained_without_conditioning=int((excess_nocond > 0).sum()), worst_unexplained=detail,
                flip_pixels_alpha=int((fb["n_alpha"] > 0).sum()),
                flip_pixels_T=int((fb["n_T"] > 0).sum()), flip_pixels_power=int((fb["n_power"] > 0).sum()),
                max_abs_clean=clean_max, max_abs_flip=float(d[cand].max()) if cand.any() else 0.0,
                pixels_over_clean_bar=int((d > CLEAN_BAR).sum()), unexplained_pixels=int((excess > 0).sum()),
                worst_excess=float(excess.max()), rel_eps=rel_eps, clean_bar=CLEAN_BAR,
                ok=bool(clean_max <= CLEAN_BAR and (excess <= 0).all()))


def flip_attribution(g: dict, cam, W: int, H: int, img: np.ndarray, ref: np.ndarray, bg=(0.0, 0.0, 0.0),
                     rel_eps: float = 1e-5) -> dict:
    """Turns "max |delta| is one threshold flip" into a checked statement.  From the oracle's own projection and
    instance lists (the reference's, bit for bit) `render_flip_bounds` finds every (pixel, instance) decision of
    renderCUDA taken within ``rel_eps`` of its threshold and bounds the change a flip can cause.  Returns the split
    figures; ``ok`` = max |delta| <= CLEAN_BAR on pixels without a candidate AND |delta| <= CLEAN_BAR + bound elsewhere,
    where bound = flips + the reference's own rounding noise on ill-conditioned evaluations (`cond`, see the C source).
    Meaningful when the implementation under test projected the SAME record (activated inputs: bit-exact); for the
    fused raw-parameter path use `compositing_attribution`."""
    s, q, o = activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    geom = preprocess(g["xyz"], s, q, o, shs, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                      cam.tanfovx, cam.tanfovy)
    return _attribute(geom, W, H, img, ref, bg, rel_eps)


def compositing_attribution(record: dict, radii: np.ndarray, W: int, H: int, img: np.ndarray, bg=(0.0, 0.0, 0.0),
                            rel_eps: float = 1e-5) -> dict:
    """The same statement for the COMPOSITING stage alone: the oracle bins and composites the record the implementation
    itself projected (``record`` = Rasterizer.download_geometry: means2D, depths, conic_opacity, rgb; ``radii`` from the
    render) with the reference's rules, so both sides start from identical inputs.  Needed on the fused raw-parameter path:
    exp / sigmoid / normalize differ from numpy's in the last ulp, and the inverse of a 100 : 1 anisotropic covariance
    amplifies that to ~1e-4 of `power` -- a projection-side difference the reference would show against itself with another
    libm, and far outside the 1e-5 band of a compositing flip."""
    geom = dict(radii=np.ascontiguousarray(radii, np.int32), means2D=np.ascontiguousarray(record["means2D"], np.float32),
                depths=np.ascontiguousarray(record["depths"], np.float32),
                conic_opacity=np.ascontiguousarray(record["conic_opacity"], np.float32),
                rgb=np.ascontiguousarray(record["rgb"], np.float32))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    # capacity of the instance list: an upper bound of getRect's tile count (auxiliary.h:46-56) per Gaussian
    r = geom["radii"].astype(np.int64)
    span = 2 * r // 16 + 2
    geom["tiles_touched"] = np.where(r > 0, np.minimum(span, gx) * np.minimum(span, gy), 0).astype(np.uint32)
    return _attribute(geom, W, H, img, None, bg, rel_eps)


def oracle_eye(g: dict, cam, W: int, H: int, bg=(0.0, 0.0, 0.0), prefer_reference: bool = True) -> dict:
    """Render one eye of pre-activation Gaussians ``g`` (numpy, GaussianModel layout) on the CPU: with the
    reference's own kernels (oracle/_ref, when prebuilt) or with the restated oracle (bit-identical, slower).
    -> dict(color, radii, num_rendered, kind)."""
    s, q, o = activate(g["scaling"], g["rotation"], g["opacity"])
    shs = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
    bgv = np.asarray(bg, np.float32)
    if prefer_reference and ref_available(build=False):
        rr = ref_forward(g["xyz"], o, cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                         cam.tanfovx, cam.tanfovy, bgv, shs=shs, scales=s, rotations=q)
        return dict(color=rr["color"], radii=rr["radii"], num_rendered=int(rr["num_rendered"]), kind="reference")
    ref, radii, n = rasterize_forward(g["xyz"], o, cam.world_view_transform, cam.full_proj_transform,
                                      cam.camera_center, W, H, cam.tanfovx, cam.tanfovy, bgv, shs=shs, scales=s,
                                      rotations=q)
    return dict(color=ref, radii=radii, num_rendered=int(n), kind="port")


def pair_parity(g: dict, cams, W: int, H: int, color: np.ndarray, rgb8: np.ndarray | None = None,
                radii: np.ndarray | None = None, bg=(0.0, 0.0, 0.0), flips: bool = False) -> dict:
    """Parity of one rendered stereo pair.  ``cams`` = (left, right) graphics.Camera, ``color`` [2,3,H,W],
    ``rgb8`` [2,H,W,3] or None, ``radii`` [2,P] or None.  Worst case over the two eyes for every figure."""
    eyes = []
    for v, cam in enumerate(cams):
        o = oracle_eye(g, cam, W, H, bg)
        e = image_parity(color[v], o["color"], None if rgb8 is None else rgb8[v])
        if radii is not None:
            e["radii_mismatches"] = int((np.asarray(radii[v]) != o["radii"]).sum())
        if flips:
            fa = flip_attribution(g, cam, W, H, color[v], o["color"], bg)
            for k in ("flip_pixels", "ill_conditioned_pixels", "max_abs_clean", "max_abs_flip", "unexplained_pixels",
                      "unexplained_without_conditioning", "pixels_over_clean_bar"):
                e[k] = fa[k]
            if fa["worst_unexplained"]:
                e["worst_unexplained"] = str(fa["worst_unexplained"])
            e["flips_ok"] = int(fa["ok"])
        e["oracle"] = o["kind"]
        e["oracle_num_rendered"] = o["num_rendered"]
        eyes.append(e)
    worst = dict(eyes[0])
    for e in eyes[1:]:
        for k, val in e.items():
            if k == "psnr_db":
                worst[k] = min(worst[k], val)
            elif k in ("oracle", "n_values", "worst_unexplained"):
                continue
            elif k == "flips_ok":
                worst[k] = min(worst[k], val)
            elif k == "oracle_num_rendered":
                worst[k] = [eyes[0][k], val]
            else:
                worst[k] = max(worst[k], val)
    worst["eyes"] = len(eyes)
    return worst
