"""Real-data readiness (north_star: "rendered stereo PSNR within 0.1 dB of the reference on DTU scan24").

    python tools/psnr_vs_ref.py <colmap_dir> <point_cloud.ply> [--pairs 0,5,10 | --max-pairs 8] [--baseline-percentage 7]
                                [--dataset DTU] [--white-background] [--json out.json]

Renders every requested stereo pair of a trained splat twice -- through ``gs2mesh_amd.renderer_utils.Renderer`` (the HIP path on
cuda:0, exactly as run_single.py drives it: same poses, baseline, 16 x 32 binning tiles, exact tile cull, fused activations) and
through the REFERENCE'S OWN rasteriser kernels compiled for the CPU (oracle/_ref, built by oracle/build_ref.py from
/root/reference; the restated oracle when it is absent) -- and reports per eye
  * PSNR of the HIP image against the reference image as third_party/gaussian-splatting/utils/image_utils.py:17-19 defines it
    (20 log10(1 / sqrt(mse)) over the float image), and of the 8-bit images the pipeline writes;
  * max / mean |delta|, the 8-bit pixels that differ, the radii that differ;
  * the checked flip statement (oracle/parity.py): pixels with a renderCUDA decision within 1e-5 of its threshold, the bound on
    clean pixels (2e-4), unexplained pixels (must be 0).
Against a GROUND-TRUTH photo (--gt-dir with <i>.png per view) it also prints PSNR(HIP, gt) and PSNR(reference, gt) and their
difference: the 0.1 dB claim of the north star is |that difference|.
No dataset ships with this repository; tests/test_pipeline_classes.py exercises the tool on a synthetic COLMAP directory."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def psnr(a, b):
    mse = float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean())     # GS/utils/image_utils.py:17-19
    return float("inf") if mse == 0 else 20.0 * np.log10(1.0 / np.sqrt(mse))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("colmap_dir")
    ap.add_argument("ply")
    ap.add_argument("--pairs", default=None, help="comma-separated view indices (default: the first --max-pairs)")
    ap.add_argument("--max-pairs", type=int, default=4)
    ap.add_argument("--baseline-percentage", type=float, default=7.0)
    ap.add_argument("--baseline-absolute", type=float, default=None)
    ap.add_argument("--dataset", default="custom")
    ap.add_argument("--not-360", action="store_true")
    ap.add_argument("--white-background", action="store_true")
    ap.add_argument("--gt-dir", default=None)
    ap.add_argument("--no-flips", action="store_true", help="skip the flip attribution (it replays the compositing on the CPU)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args(argv)
    import torch
    import oracle
    from oracle import parity
    from gs2mesh_amd.graphics import Camera
    from gs2mesh_amd.poses import convert_R_T_to_GS
    from gs2mesh_amd.renderer_utils import Renderer
    if not torch.cuda.is_available():
        raise SystemExit("psnr_vs_ref needs a GPU for the HIP side (there is no CPU path)")
    args = argparse.Namespace(colmap_name="scene", dataset_name=a.dataset, GS_white_background=a.white_background, GS_iterations=0,
                              renderer_baseline_absolute=a.baseline_absolute, renderer_baseline_percentage=a.baseline_percentage,
                              renderer_scene_360=not a.not_360, renderer_save_json=False, renderer_sort_cameras=False)
    out_root = os.path.join("/tmp", "psnr_vs_ref_out")
    r = Renderer(os.path.dirname(os.path.abspath(a.ply)), a.colmap_dir, out_root, args)
    r.splatting_ply_file_path = a.ply
    r.prepare_renderer()
    raw = r.gaussians.raw()
    g = {k: v.detach().cpu().numpy() for k, v in raw.items() if hasattr(v, "detach")}
    bg = (1.0, 1.0, 1.0) if a.white_background else (0.0, 0.0, 0.0)
    ids = [int(x) for x in a.pairs.split(",")] if a.pairs else list(range(min(len(r), a.max_pairs)))
    rows = []
    for i in ids:
        res = r._raster.render_views(raw, r._pair(i), bg=bg, want_color=True, want_rgb8=True, want_radii=True)
        color, rgb8, radii = res["color"].cpu().numpy(), res["rgb8"].cpu().numpy(), res["radii"].cpu().numpy()
        for k, eye in enumerate(("left", "right")):
            c = r.cameras[i][eye]
            R, T = convert_R_T_to_GS(tuple(c["rot"]), tuple(c["pos"]))
            W, H = c["width"], c["height"]
            cam = Camera(0, R, T, 2 * np.arctan2(W, 2 * c["fx"]), 2 * np.arctan2(H, 2 * c["fy"]), W, H)
            o = parity.oracle_eye(g, cam, W, H, bg)
            e = parity.image_parity(color[k], o["color"], rgb8[k])
            row = dict(view=i, eye=eye, reference=o["kind"], psnr_db_vs_reference=psnr(color[k], o["color"]),
                       psnr_db_u8_vs_reference=psnr(rgb8[k].astype(np.float64) / 255.0, parity.quantize_u8(o["color"]).astype(np.float64) / 255.0),
                       max_abs=e["max_abs"], mean_abs=e["mean_abs"], u8_flipped_pixels=e["u8_flipped_pixels"],
                       radii_mismatches=int((radii[k] != o["radii"]).sum()), num_rendered_reference=o["num_rendered"])
            if not a.no_flips:
                fa = parity.compositing_attribution(r._raster.download_geometry(k, int(g["xyz"].shape[0])), radii[k], W, H, color[k], bg)
                row.update(flip_pixels=fa["flip_pixels"], max_abs_clean=fa["max_abs_clean"], unexplained_pixels=fa["unexplained_pixels"],
                           flips_ok=bool(fa["ok"]))
            if a.gt_dir:
                from PIL import Image
                p = os.path.join(a.gt_dir, f"{i}.png")
                if eye == "left" and os.path.exists(p):
                    gt = np.asarray(Image.open(p).convert("RGB").resize((W, H)), np.float64).transpose(2, 0, 1) / 255.0
                    row.update(psnr_db_hip_vs_gt=psnr(np.clip(color[k], 0, 1), gt), psnr_db_reference_vs_gt=psnr(np.clip(o["color"], 0, 1), gt))
                    row["psnr_db_gap_to_reference"] = row["psnr_db_hip_vs_gt"] - row["psnr_db_reference_vs_gt"]
            rows.append(row)
            print(json.dumps(row), flush=True)
    finite = [x["psnr_db_vs_reference"] for x in rows if np.isfinite(x["psnr_db_vs_reference"])]
    summary = dict(pairs=len(ids), eyes=len(rows), min_psnr_db_vs_reference=(min(finite) if finite else float("inf")),
                   max_abs=max(x["max_abs"] for x in rows), unexplained_pixels=sum(x.get("unexplained_pixels", 0) for x in rows),
                   max_gap_db_vs_gt=max((abs(x["psnr_db_gap_to_reference"]) for x in rows if "psnr_db_gap_to_reference" in x), default=None))
    print(json.dumps(dict(summary=summary)))
    if a.json:
        json.dump(dict(rows=rows, summary=summary), open(a.json, "w"), indent=1)
    return rows, summary


if __name__ == "__main__":
    main()
