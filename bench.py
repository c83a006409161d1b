#!/usr/bin/env python
"""bench.py -- the render -> fuse hot path on synthetic data, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--config C2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

A step = one pass of the hot path over one stereo view: render the L/R pair of the view from the
(pre-activation) Gaussians, then integrate the view's depth frame + the rendered left image (u8, on
device) into the block-sparse TSDF.  The workload at N=1 is BASELINE.json configs[1] (C2: 300k Gaussians,
1600x1200, TSDF 512^3, reference defaults voxel 2/512 / trunc 0.04); depth = analytic sphere (the
reference's depth comes from the DLNR stereo network, which is outside the path).  All inputs are
resident in HBM before the timed region.  N>1: weak scaling -- every rank renders+fuses its own K
views with no data-path collective, then ONE RCCL reduce-scatter of the TSDF accumulators inside the
timed region (gs2mesh_amd/parallel.py).

Timing: the K-step job (+ the reduction when N > 1) is timed EXACTLY as the driver contract says (barrier +
synchronize on both sides, MAX over ranks) and REPEATED (volume reset between repeats, outside the bracket) until at
least 5 repeats and >= 1 s of timed region have accumulated; `ms_per_step` / `value` are the MEDIAN repeat.

Prints ONE JSON line (rank 0): value = whole-job stereo pairs rendered+fused per second.
Extra objects: "tsdf" (Mvoxel-updates/s), "stages" (hipEvent time per kernel launch), "roofline" (dominant kernel),
"raster_roofline" (whole rasteriser vs SURVEY.md 8d's B_pair, with this build's instance count and with the
reference's), "parity" (the first timed pair against the CPU oracle), "cpu_baseline" (oracle = restated Open3D 0.17
integrate on the host cores, bounded sample), "c3" (render-only sub-measurement of the HBM-bound 2 M-Gaussian config).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md); measured float4-copy peak 6.29e12


def alg_bytes(cfg, p_vis, p_vis_union, N_eye, U_frame=0.0):
    """SURVEY.md 8(d) algorithmic bytes per kernel launch (= per stereo pair / per frame)."""
    Wd, Ht = cfg.width, cfg.height
    img = 12.0 * Wd * Ht
    alg = {
        "project": 44.0 * cfg.P + 192.0 * p_vis_union + 40.0 * sum(p_vis),
        "count_tiles": 16.0 * 2 * cfg.P + 32.0 * sum(p_vis),   # rect of every Gaussian + geometry half of the visible ones
        "scatter": 12.0 * sum(N_eye),
        "sort_tiles": 24.0 * sum(N_eye),
        "blend": 40.0 * sum(N_eye) + 2 * img,
        "hist_colscan": 0.0, "tile_scan": 0.0,
        "tsdf_touch": 4.0 * Wd * Ht / 16.0,
        "tsdf_integrate": 40.0 * U_frame + 7.0 * Wd * Ht,
    }
    B_pair = 44.0 * cfg.P + 192.0 * p_vis_union + sum(40.0 * pv + 76.0 * n + img for pv, n in zip(p_vis, N_eye))
    return alg, B_pair


def committed_traffic(cfg_name):
    """HBM bytes per launch and stage from the committed PMC passes (profiles/pmc_traffic.json; not measured in this run)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(cfg_name, {})
    except Exception:
        return {}


def spatial_order_for(args, P):
    """--spatial-order -1 (auto): the pipeline's rule (gs2mesh_amd.rasterizer.auto_spatial_order)."""
    from gs2mesh_amd.rasterizer import auto_spatial_order
    return bool(args.spatial_order) if args.spatial_order >= 0 else auto_spatial_order(P)


def pairs_per_launch_for(args, P):
    """--pairs-per-launch 0 (auto): 4 stereo pairs per launch with a spatially ordered model, else 2 (the argument's help says why)."""
    a = int(getattr(args, "ppl_arg", args.pairs_per_launch))
    return a if a > 0 else (4 if spatial_order_for(args, P) else 2)


def raster_only(args, cfg_name, dev, local_rank, pairs=12, scene="synth_v1", with_parity=False):
    """Render-only sub-measurement (no TSDF, serial on one stream, hipEvents per launch, the default launch shape:
    `--pairs-per-launch` stereo pairs per chain of launches): the C3 sub-line and (scene "trained_like": anisotropic, saturated
    opacities, screen-filling background splats, depth ties; `with_parity`: first pair against the CPU oracle with the flip
    attribution of the compositing stage) the realistic-splat sub-line."""
    import torch
    from gs2mesh_amd import _lib, synthetic
    from gs2mesh_amd.rasterizer import Rasterizer, camera_from
    from gs2mesh_amd.rasterizer import auto_cull_level
    cfg = synthetic.CONFIGS[cfg_name]
    cull = auto_cull_level(cfg.P) if args.cull_arg < 0 else int(args.cull_arg)     # this configuration's own level
    if scene == "trained_like":
        g = synthetic.trained_like(cfg.P, 4242, cfg.log_s_mu, focal=cfg.focal, ring_radius=cfg.ring_radius)
    else:
        g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    g_par = g if with_parity else None        # host copy of the model: only the parity leg needs it
    del g
    poses = synthetic.ring_poses(pairs, cfg.ring_radius, 0, cfg.n_pairs)
    cams, cams_np0 = [], None
    for p in poses:
        l, r = synthetic.stereo_cameras(p, cfg.width, cfg.height, cfg.focal, cfg.focal, cfg.baseline)
        if cams_np0 is None:
            cams_np0 = (l, r)
        cams.append([camera_from(l), camera_from(r)])
    R = Rasterizer(local_rank)
    R.set_option(_lib.OPT_EXACT_TILE_CULL, cull)
    R.set_option(_lib.OPT_TILE_ROWS, int(args.tile_rows))
    R.set_option(_lib.OPT_BLEND_VARIANT, int(args.blend))
    from gs2mesh_amd.rasterizer import auto_blend_mode
    blend_mode = auto_blend_mode(gd)          # what RenderFusePipeline / Renderer.prepare_renderer pick for this model
    R.set_option(_lib.OPT_BLEND_MODE, blend_mode)
    ppl = pairs_per_launch_for(args, cfg.P)
    if ppl > 1:
        R.set_option(_lib.OPT_PAIR_BATCH, ppl)
    if spatial_order_for(args, cfg.P):
        R.pack_model(gd)
    else:
        R.pack_sh(gd)
    color = torch.empty((2 * ppl, 3, cfg.height, cfg.width), dtype=torch.float32, device=dev)
    rgb8 = torch.empty((2 * ppl, cfg.height, cfg.width, 3), dtype=torch.uint8, device=dev)
    res = R.render_views(gd, cams[0], out_color=color[:2], out_rgb8=rgb8[:2], want_radii=True)
    R.reserve(cfg.P, 2 * ppl, cfg.width, cfg.height, int(max(res["num_rendered"]) * 1.5))
    par = None
    if with_parity:
        res = R.render_views(gd, cams[0], out_color=color[:2], out_rgb8=rgb8[:2], want_radii=True)   # (the reserve above may have moved the arenas the taps read)
        # the first pair against the reference's own kernels on the CPU (global figures) + the checked flip statement of the
        # compositing stage on the record THIS pass projected (the fused activations move the record of a 100 : 1 anisotropic
        # splat by rounding: tests/test_fullsize_gpu.py::test_trained_like_splats_full_size_vs_reference_kernels)
        try:
            from oracle import parity
            c_h, u_h, r_h = color[:2].cpu().numpy(), rgb8[:2].cpu().numpy(), res["radii"].cpu().numpy()
            par = parity.pair_parity(g_par, cams_np0, cfg.width, cfg.height, c_h, u_h, r_h)
            comp = []
            for v in range(2):
                fa = parity.compositing_attribution(R.download_geometry(v, cfg.P), r_h[v], cfg.width, cfg.height, c_h[v])
                comp.append({k: fa[k] for k in ("flip_pixels", "ill_conditioned_pixels", "max_abs_clean", "max_abs_flip",
                                                "unexplained_pixels", "pixels_over_clean_bar", "ok")})
            par["compositing"] = comp
            par["what"] = ("first pair, both eyes (worst case): global figures vs the reference's own kernels on the CPU; `compositing` = "
                           "flip attribution of the compositing stage on the record this pass projected (bar: max_abs_clean <= 2e-4, "
                           "0 unexplained pixels)")
        except Exception as e:
            par = dict(error=str(e)[:300])
        g_par = None
    # groups of ppl consecutive pairs = one chain of launches each
    cams = [[c for pair in cams[i:i + ppl] for c in pair] for i in range(0, len(cams) - len(cams) % ppl, ppl)]
    n_pairs = len(cams) * ppl
    radii0 = res["radii"]
    p_vis = [(radii0[v] > 0).sum().item() for v in range(2)]
    p_vis_union = ((radii0[0] > 0) | (radii0[1] > 0)).sum().item()
    N_eye = [float(x) for x in res["num_rendered"]]
    # warm-up: this sub-measurement starts on a GPU that sat idle through the host-side legs (parity, CPU baseline): the first
    # kernels on an idle chip run ~8 % slow (profiles/r4_experiments.txt) -- >= 0.5 s of the same launches first
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.5:
        for c in cams[:2]:
            R.render_views(gd, c, out_color=color, out_rgb8=rgb8, sync=False)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in cams:
        R.render_views(gd, c, out_color=color, out_rgb8=rgb8, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_pairs
    R.set_option(_lib.OPT_STAGE_TIMING, 1)
    for c in cams:
        R.render_views(gd, c, out_color=color, out_rgb8=rgb8, sync=False)
    st = R.stage_times()
    R.set_option(_lib.OPT_STAGE_TIMING, 0)
    _, ov, _ = R.status(2)
    alg, B_pair = alg_bytes(cfg, p_vis, p_vis_union, N_eye)
    tr = committed_traffic(cfg_name)
    st = {k: (ms / ppl, c) for k, (ms, c) in st.items()}     # per stereo pair (a launch covers ppl pairs)
    stages = {k: dict(avg_us=round(1e3 * ms / max(c, 1), 2), frac_hbm=round(alg[k] / max(1e-9, 1e-3 * ms / max(c, 1)) / HBM_PEAK, 4),
                      alg_bytes=int(alg[k]),
                      traffic=(int(tr[k]["hbm_bytes_per_launch"] / max(1, int(tr[k].get("pairs_per_launch", 1))))   # per stereo pair
                               if tr.get(k, {}).get("cull") == cull and tr[k].get("hbm_bytes_per_launch") is not None else None))
              for k, (ms, c) in st.items()}
    if ppl > 1 and spatial_order_for(args, cfg.P) and "project" in stages:
        # round 6 (GS2M_OPT_PROJECT_SHARED_READ, auto for >= 1 M Gaussians): the `ppl` pairs of a launch share ONE read of the model
        # (44 B + the 192-B SH row per Gaussian).  `frac_hbm` stays SURVEY 8(d)'s per-pair figure / time -- it can exceed what one
        # pair alone could reach; `launch_bytes_per_pair` is what a launch actually has to move per pair.
        shared = (44.0 * cfg.P + 192.0 * p_vis_union) * (1.0 - 1.0 / ppl)
        e = stages["project"]
        e["launch_bytes_per_pair"] = int(alg["project"] - shared)
        e["frac_hbm_launch_bytes"] = round((alg["project"] - shared) / max(1e-9, e["avg_us"] * 1e-6) / HBM_PEAK, 4)
        e["note"] = f"the {ppl} stereo pairs of a launch share one read of the model (round 6)"
    t_raster = sum(v["avg_us"] for v in stages.values()) * 1e-6
    return dict(workload=f"{cfg_name}: {cfg.P} {scene} Gaussians, {cfg.width}x{cfg.height}, render only, {n_pairs} pairs, "
                         f"serial on one stream, {ppl} stereo pair(s) per launch (stage times per pair)", pairs_per_launch=ppl,
                exact_tile_cull=cull, blend_mode=blend_mode, parity=par,
                num_rendered_per_eye=[int(x) for x in N_eye], p_visible_per_eye=p_vis, overflow=bool(ov),
                ms_per_pair_wall=round(1e3 * dt, 4), stages=stages,
                traffic_source="profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2 * FETCH + WRITE) KiB per launch)",
                binning_us=round(sum(stages[k]["avg_us"] for k in ("count_tiles", "hist_colscan", "tile_scan", "scatter", "sort_tiles")), 1),
                raster_roofline=dict(B_pair_bytes=int(B_pair), t_pair_us=round(t_raster * 1e6, 1),
                                     achieved_GBps=round(B_pair / t_raster / 1e9, 1),
                                     frac_of_8TBps=round(B_pair / t_raster / HBM_PEAK, 4)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # = the driver's command: `python bench.py` reproduces its number
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--cull", type=int, default=int(os.environ.get("GS2M_BENCH_CULL", "-1")),
                    help="exact tile culling (image-preserving): 0 = reference instance lists, 1 / 2 = GS2M_OPT_EXACT_TILE_CULL levels, "
                         "-1 (default) = auto by model size (gs2mesh_amd.rasterizer.auto_cull_level: the pipeline's own default)")
    ap.add_argument("--blend", type=int, default=int(os.environ.get("GS2M_BENCH_BLEND", "4")))
    ap.add_argument("--reduce", default="reduce_scatter", choices=["allreduce", "reduce_scatter"])
    ap.add_argument("--payload", default="auto", choices=["auto", "packed", "f32"],
                    help="N > 1: exchange payload of the TSDF reduction (packed = fp32 wsum + one int64 of integer lanes per voxel, "
                         "12 B; f32 = five fp32 planes, 20 B; auto = packed while the ranks integrated <= 1023 frames in total)")
    ap.add_argument("--reduce-algo", default="rccl", choices=["rccl", "direct"],
                    help="reduce_scatter as RCCL's collective, or as one all_to_all of the 1/N slices + a local sum")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the driver's contract): --steps views per GPU; strong: --steps views IN TOTAL, sharded "
                         "contiguously over the GPUs (BASELINE config C4: --config C4 --steps 300 --scaling strong)")
    ap.add_argument("--always-collective", action="store_true",
                    help="issue the key all_gather + the RCCL reduction of the TSDF accumulators (inside the timed region) even at "
                         "world size 1: times the local part of the exchange (key union, pack, RCCL self-copy, unpack) on one GPU")
    ap.add_argument("--tile-rows", type=int, default=int(os.environ.get("GS2M_BENCH_TILE_ROWS", "2")), choices=[1, 2],
                    help="binning tile = 16 x (16*rows) pixels (GS2M_OPT_TILE_ROWS); 1 = the reference's tiles")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("GS2M_BENCH_INFLIGHT", "2")),
                    help="launch chains in flight on separate HIP streams (1 = everything serial on one stream).  Round 4: 2 (x 2 "
                         "stereo pairs per launch = 4 pairs in flight) -- with two pairs per launch more slots only add contention and "
                         "pipeline fill / drain: C2 20-step job 0.292 (2 slots) / 0.307 (3) / 0.309 (4) / 0.301 (5) / 0.307 (6) ms per "
                         "step, profiles/r4_pipeline_sweeps.txt")
    ap.add_argument("--fuse-batch", type=int, default=int(os.environ.get("GS2M_BENCH_FUSE_BATCH", "0")),
                    help="views integrated per voxel-stationary TSDF batch sweep (1 = view by view; same volume either way); "
                         "0 (default) = the K views of a job in equal sweeps of at most 32 views")
    ap.add_argument("--spatial-order", type=int, default=int(os.environ.get("GS2M_BENCH_SPATIAL_ORDER", "-1")),
                    help="1 = Morton-ordered packed copy of the model in the handles (gs2m_raster_pack_model, one-time prepare, same "
                         "results); 0 = SH packing only; -1 (default) = the pipeline's rule (rasterizer.auto_spatial_order: >= 32 768 Gaussians)")
    ap.add_argument("--pairs-per-launch", type=int, default=int(os.environ.get("GS2M_BENCH_PAIRS_PER_LAUNCH", "0")), choices=[0, 1, 2, 4],
                    help="consecutive stereo pairs that share every launch of the binning chain and the compositing (GS2M_OPT_PAIR_BATCH; "
                         "same images and volume).  0 (default) = 4 with a spatially ordered model (the pipeline's default from 32 768 "
                         "Gaussians), else 2: the ordered model reads its parameters once per launch and its longer compositing grids "
                         "lose less to ramp and tail (round 6: C2 3787 -> 3842, C3 2546 -> 2627, C5 3540 -> 3643 pairs/s), while an "
                         "UNORDERED model pays for eight open key arrays in the scatter (C2 3643 -> 3547); 1 = one pair per launch")
    ap.add_argument("--min-repeats", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=1.0, help="accumulated timed region to reach")
    ap.add_argument("--max-repeats", type=int, default=400)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 render-only sub-measurement")
    ap.add_argument("--no-trained-like", action="store_true", help="skip the trained-like (C2-sized) render-only sub-measurement")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the 2K-step job of the steady-state probe")
    ap.add_argument("--volume-check", action="store_true", help="add an order-independent fingerprint of the fused volume (summed over the ranks) to the line")
    ap.add_argument("--raster-opts", default=os.environ.get("GS2M_BENCH_RASTER_OPTS", ""),
                    help="development sweeps: extra RenderFusePipeline raster_options of the timed pass, e.g. bin_workgroups=512,bin_wg_threads=512")
    ap.add_argument("--rehearsal", action="store_true", default=bool(int(os.environ.get("GS2M_BENCH_REHEARSAL", "0"))),
                    help="run the whole --gpus N control flow (self-spawn, barriers, dt all-reduce, reduce_volume, rank-0-only line) "
                         "with all N ranks sharing GPU 0 over the gloo backend (RCCL refuses two ranks on one device; the exchange "
                         "buffers are staged through host memory): a dress rehearsal of the driver's multi-GPU run on a one-GPU "
                         "box -- the value it prints is NOT a scaling number")
    args = ap.parse_args()
    extra_raster_opts = {k: int(v) for k, v in (kv.split("=") for kv in args.raster_opts.split(",") if kv)}
    args.cull_arg = args.cull                      # as given (-1 = auto); args.cull = the level the timed configuration runs at
    if args.cull < 0:
        from gs2mesh_amd import synthetic as _syn
        from gs2mesh_amd.rasterizer import auto_cull_level
        args.cull = auto_cull_level(_syn.CONFIGS[args.config].P)

    # --gpus N is the number of ranks of the job.  Under the driver's launcher (torch.distributed.run) WORLD_SIZE says the same;
    # started plainly with --gpus N > 1 the script re-executes itself under that launcher (one rank per GPU, rendezvous on
    # 127.0.0.1); anything else -- fewer visible GPUs than ranks, a launcher world that differs from --gpus -- is an error, not
    # a silent one-GPU run that reports n_gpus: 1.
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if world_env is None and args.gpus > 1:
        import socket
        import torch
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus and not args.rehearsal:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_vis} GPU(s) are visible on this node")
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if int(world_env or "1") != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env or 1} rank(s); "
                         "they must agree (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)")

    import torch
    import torch.distributed as dist
    from gs2mesh_amd import _lib, synthetic
    from gs2mesh_amd.integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from gs2mesh_amd.parallel import reduce_volume
    from gs2mesh_amd.rasterizer import camera_from
    from gs2mesh_amd.pipeline import RenderFusePipeline

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    line_out = sys.stdout
    if world > 1 or args.always_collective:
        # RCCL 2.26 prints a version banner through C stdio on STDOUT (seen on the GPU box: five lines, flushed at exit, i.e.
        # AFTER the JSON line, from every rank).  The contract is ONE line on stdout: keep a private handle on the real stdout
        # for that line and point descriptor 1 at stderr for everything a library may print.
        sys.stdout.flush()
        line_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if args.rehearsal:
        local_rank = 0                                  # every rank on GPU 0
    if world > 1 and args.rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(f"cuda:{local_rank}"))
    elif args.always_collective:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device(f"cuda:{local_rank}"))
    exchange = world > 1 or args.always_collective      # the TSDF reduction is part of the timed job
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    K_total = args.steps * world if args.scaling == "weak" else args.steps
    if args.scaling == "strong":
        from gs2mesh_amd.parallel import shard_range
        lo_k, hi_k = shard_range(args.steps, rank, world)
        K = hi_k - lo_k                         # this rank's share of the job (contiguous views)
    else:
        lo_k, K = rank * args.steps, args.steps
    Wm = args.warmup
    fuse_plan = None
    if args.fuse_batch <= 0:
        # the K views of a job in equal sweeps of at most 32 views.  Measured in round 3: the pipelined step is additive over
        # the kernels that fill the chip, so a sweep gains nothing from running next to the renders -- what counts is its own
        # time, and that falls with its size (the voxel state is read and written once per sweep): one sweep of 20 for the
        # driver's 20-step job 28.4 us per frame and 0.3224 ms per step, two of 10 34.3 us and 0.3254 ms; sweeps of
        # decreasing size (10, 5, 3, 2: only a small sweep after the last render) 0.356 vs 0.348 ms.
        n_sweeps = max(1, -(-K // 32))
        args.fuse_batch = max(1, -(-K // n_sweeps))
    args.ppl_arg = int(args.pairs_per_launch)            # as given (0 = auto): the sub-lines resolve it for THEIR model
    args.pairs_per_launch = pairs_per_launch_for(args, synthetic.CONFIGS[args.config].P)
    if args.pairs_per_launch > 1:
        # two pairs per launch render into consecutive buffers of the pending sweep: even sweep sizes (an odd last view is
        # flushed on its own); nothing to pair up in a one-view job or without the pipeline
        if args.fuse_batch < 2 or K < 2:
            args.pairs_per_launch = 1
        else:
            args.fuse_batch += (-args.fuse_batch) % args.pairs_per_launch
    cfg = synthetic.CONFIGS[args.config]
    Wd, Ht = cfg.width, cfg.height
    cx, cy = Wd / 2.0, Ht / 2.0

    # ---- inputs, resident in HBM --------------------------------------------------------------
    g = synthetic.synth_v1(cfg.P, cfg.seed, cfg.log_s_mu)
    gd = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    gd["raw"] = True
    # timed views of this rank: [lo_k, lo_k + K) of the job's K_total views; the warm-up views precede them on the ring; at
    # N = 1 another K views follow for the 2K-step job of the steady-state probe (slope between the K- and the 2K-step job)
    probe = world == 1 and not args.no_steady_state
    n_local = K + Wm + (K if probe else 0)
    poses = synthetic.ring_poses(n_local, cfg.ring_radius, first=lo_k - Wm, total=max(K_total, 1))
    cams, cams_np, depths, Es = [], [], [], []
    for p in poses:
        l, r = synthetic.stereo_cameras(p, Wd, Ht, cfg.focal, cfg.focal, cfg.baseline)
        cams.append([camera_from(l), camera_from(r)])
        cams_np.append((l, r))
        depths.append(synthetic.sphere_depth_torch(p, Wd, Ht, cfg.focal, cfg.focal, cx, cy, cfg.sphere_radius, dev))
        E = np.eye(4)
        E[:3] = p
        Es.append(E)
    intr = PinholeCameraIntrinsic(Wd, Ht, cfg.focal, cfg.focal, cx, cy)
    depth_trunc = cfg.baseline * 20          # TSDF_max_depth_baselines (argument_utils.py:37)
    min_depth = cfg.baseline * 4             # TSDF_min_depth_baselines
    n_blocks_dense = (cfg.tsdf_n // 16) ** 3
    vol = ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, max_blocks=n_blocks_dense, device=local_rank)
    # `inflight` stereo pairs in flight on separate streams (own rasteriser handle + images each), integration
    # in view order on a third stream (gs2mesh_amd/pipeline.py); inflight = 1 is the serial single-stream order
    pipe = RenderFusePipeline(gd, Wd, Ht, vol, intr, inflight=args.inflight, device=local_rank,
                              raster_options=dict(exact_tile_cull=args.cull, blend_variant=args.blend, tile_rows=args.tile_rows, **extra_raster_opts),
                              fuse_batch=(fuse_plan if fuse_plan and len(fuse_plan) > 1 else args.fuse_batch),
                              spatial_order=("auto" if args.spatial_order < 0 else bool(args.spatial_order)),
                              pairs_per_launch=args.pairs_per_launch)
    spatial_order_used = int(pipe.spatial_order)
    pairs_per_launch_used = int(pipe.ppl)     # (the pipeline object is gone by the time the line is assembled)
    R = pipe.rasterizers[0]
    color, rgb8 = pipe.color[0][:2], pipe._own8[0][:2]     # one pair's worth of slot 0's buffers: the serial / parity passes

    def step(i):
        pipe.submit(cams[i], depths[i], Es[i], depth_scale=1.0, depth_trunc=depth_trunc, min_depth=min_depth)

    serial_imgs = ([torch.empty((Ht, Wd, 3), dtype=torch.uint8, device=dev) for _ in range(args.fuse_batch)]
                   if args.fuse_batch > 1 else [])
    serial_pending = []

    def serial_flush():
        if serial_pending:
            vol.integrate_batch([RGBDImage(serial_imgs[k], depths[i], depth_scale=1.0, depth_trunc=depth_trunc)
                                 for k, i in enumerate(serial_pending)], intr, [Es[i] for i in serial_pending],
                                min_depth=min_depth)
            serial_pending.clear()

    def step_serial(i0, n):
        """instrumented pass: views i0 .. i0 + n - 1 through ONE chain of launches (n = pairs_per_launch: the launch shapes of
        the timed pass), serial on the current stream"""
        R.render_views(gd, [c for i in range(i0, i0 + n) for c in cams[i]], out_color=pipe.color[0][:2 * n],
                       out_rgb8=pipe._own8[0][:2 * n], sync=False)
        for k in range(n):
            left8 = pipe._own8[0][2 * k]
            if args.fuse_batch > 1:
                serial_imgs[len(serial_pending)].copy_(left8, non_blocking=True)
                serial_pending.append(i0 + k)
                if len(serial_pending) == args.fuse_batch:
                    serial_flush()
            else:
                vol.integrate(RGBDImage(left8, depths[i0 + k], depth_scale=1.0, depth_trunc=depth_trunc), intr, Es[i0 + k],
                              min_depth=min_depth)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up (also sizes the instance arena: grow + retry happens here, not in the timed loop)
    res = pipe.prepare(cams[0], headroom=2.0)   # SH re-layout + arena sizing (prepare_renderer stage)
    num_rendered0 = res["num_rendered"]
    radii0 = res["radii"]
    p_vis = [(radii0[v] > 0).sum().item() for v in range(2)]
    p_vis_union = ((radii0[0] > 0) | (radii0[1] > 0)).sum().item()
    for i in range(Wm):
        step(i)
    pipe.finish()                                 # drain the pipeline's streams before touching the volume
    if exchange:
        reduce_volume(vol, mode=args.reduce, payload=args.payload, algo=args.reduce_algo,
                      always_collective=args.always_collective)      # warm the RCCL communicator + size the exchange buffers
    vol.status()
    vol.reset()
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps (+ the volume reduction when N > 1), repeated ------------
    dts, reds, enq = [], [], []
    red = None
    while True:
        vol.reset()
        barrier()
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            step(i)
        enq.append(time.perf_counter() - t0)      # host time to enqueue the K steps (no synchronisation yet)
        pipe.drain()                              # integrates the last (partial) TSDF batch, then waits for the pipeline's streams
        if exchange:
            t_red0 = time.perf_counter()
            red = reduce_volume(vol, mode=args.reduce, payload=args.payload, algo=args.reduce_algo,
                                always_collective=args.always_collective)
            torch.cuda.synchronize()
            reds.append(time.perf_counter() - t_red0)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=("cpu" if args.rehearsal else dev))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())                 # identical on every rank -> identical stopping decision
        pipe.finish()    # raises if an instance arena overflowed in ANY pair of the repeat (sticky device flag)
        dts.append(dt)
        if (len(dts) >= args.min_repeats and sum(dts) >= args.min_seconds) or len(dts) >= args.max_repeats:
            break
    dt = statistics.median(dts)
    t_red = statistics.median(reds) if reds else None
    volume_check = None
    if args.volume_check:
        # order-independent fingerprint of the fused volume of the LAST timed job, summed over the ranks' parts (after a
        # reduce-scatter every rank holds its share of the blocks): block count, sum of a per-key hash, sum of the weights and of
        # the colour sums (integers: exact whatever the sharding), sum of tsdf x weight (fp32 reassociation only).  The rehearsal
        # test compares a 2-rank strong-scaling job with the 1-rank job over the same views.
        keys_h, tsdf_h, w_h, rgb_h = vol.download()
        kk = keys_h.astype(np.int64)
        khash = int(((kk[:, 0] * 73856093) ^ (kk[:, 1] * 19349663) ^ (kk[:, 2] * 83492791)).sum() & ((1 << 62) - 1)) if len(kk) else 0
        loc = np.array([len(kk), khash, float(w_h.astype(np.float64).sum()), float(rgb_h.astype(np.float64).sum()),
                        float((tsdf_h.astype(np.float64) * w_h).sum())], np.float64)
        if world > 1:
            tt = torch.from_numpy(loc.copy()) if args.rehearsal else torch.from_numpy(loc.copy()).to(dev)
            th = torch.tensor([khash], dtype=torch.int64, device=tt.device)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            dist.all_reduce(th, op=dist.ReduceOp.SUM)
            loc = tt.cpu().numpy()
            khash = int(th.item()) & ((1 << 62) - 1)
        volume_check = dict(blocks=int(loc[0]), key_hash=int(khash), weight_sum=float(loc[2]), rgb_sum=float(loc[3]),
                            tsdf_weight_sum=float(loc[4]), views=int(K_total), scaling=args.scaling)
    halo_blocks = None
    if red is not None and world == 1:
        # owner-side finalisation step that follows a reduce-scatter (outside the timed job: mesh extraction is not part of the
        # metric): at world size 1 (--always-collective) this is its empty path.  Not run at N > 1: the scaling run is the one
        # chance to time the reduction on 8 GPUs and must not depend on a step the metric does not contain.
        from gs2mesh_amd.parallel import exchange_halo
        try:
            halo_blocks = int(exchange_halo(vol, red))
        except Exception as e:      # never lose the line to a step that is not part of the metric
            halo_blocks = f"error: {str(e)[:200]}"

    # ---- steady state vs fill / drain: the same job with 2K steps, timed the same way; slope between the two job lengths
    steady = None
    if probe:
        d2 = []
        for _ in range(max(3, min(7, len(dts)))):
            vol.reset()
            barrier()
            t0 = time.perf_counter()
            for i in range(Wm, Wm + 2 * K):
                step(i)
            pipe.drain()
            barrier()
            d2.append(time.perf_counter() - t0)
            pipe.finish()
        dt2 = statistics.median(d2)
        t_ss = (dt2 - dt) / K
        steady = dict(steady_state_ms_per_step=round(1e3 * t_ss, 4), fill_drain_ms=round(1e3 * (dt - K * t_ss), 4),
                      ms_per_step_2K_job=round(1e3 * dt2 / (2 * K), 4),
                      note=f"slope between the {K}-step and the {2 * K}-step job (both timed as the contract says): steady state = "
                           f"(T({2 * K}) - T({K})) / {K}; fill_drain = T({K}) - {K} x steady state (pipeline fill, last TSDF sweep, final sync)")

    # Everything from here to the JSON line is DIAGNOSTICS of the timed job above (instrumented pass, byte models, parity, CPU baseline,
    # sub-lines): rank 0 only -- the other ranks of a multi-GPU job go straight to the final barrier -- and behind one try / except:
    # the 8-rank rehearsal (round 6) lost a rank to a division by zero in here; a diagnostic must never cost the driver its line.
    diag_error = None
    N_eye = [float(x) for x in num_rendered0]
    tsdf, per_kernel, roofline, raster_roofline, par, cpu, c3, trained, dt_instr = None, None, None, None, None, None, None, None, 0.0
    if rank == 0:
        try:
            # ---- instrumented pass (hipEvents around every kernel launch, on the work stream) ---------
            vol.reset()
            vol.status()
            R.set_option(_lib.OPT_STAGE_TIMING, 1)
            vol.set_stage_timing(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ppl = pairs_per_launch_used
            for i in range(Wm, Wm + K, ppl):
                step_serial(i, min(ppl, Wm + K - i))
            serial_flush()
            torch.cuda.synchronize()
            dt_instr = time.perf_counter() - t1
            st_r = R.stage_times()
            st_t = vol.stage_times()
            R.set_option(_lib.OPT_STAGE_TIMING, 0)
            vol.set_stage_timing(False)
            n_blocks, block_updates, _ = vol.status()
            # raster stages: a launch covers `ppl` stereo pairs -> avg_us is quoted per PAIR (the unit of the algorithmic bytes) next to
            # the launch duration itself; TSDF stages: a sweep counts as its frames (gs2m_tsdf_stage_times), avg_us is per frame
            stages = {k: dict(avg_us=1e3 * ms / max(c, 1) / ppl, launch_us=1e3 * ms / max(c, 1), launches=int(c)) for k, (ms, c) in st_r.items()}
            stages.update({k: dict(avg_us=1e3 * ms / max(c, 1), launch_us=None, launches=int(c)) for k, (ms, c) in st_t.items()})
            # voxels that actually updated (pass sdf > -trunc): every update adds 1 to the voxel's weight
            keys_h, tsdf_h, weight_h, _ = vol.download()
            U_total = float(weight_h.sum())
            U_frame = U_total / K
            blocks_frame = block_updates / K

            # ---- algorithmic bytes, per kernel launch = per stereo pair ------------------
            N_eye = [float(x) for x in num_rendered0]           # instances produced by this build (culling mode)
            alg, B_pair = alg_bytes(cfg, p_vis, p_vis_union, N_eye, U_frame)
            t_raster = sum(stages[k]["avg_us"] for k in _lib.RASTER_STAGES) * 1e-6
            t_tsdf = sum(stages[k]["avg_us"] for k in _lib.TSDF_STAGES) * 1e-6
            # the dominant kernel = the stage with the most GPU time among those that move algorithmic bytes (the scans carry none: under
            # contention -- the 8-rank rehearsal on one GPU -- a scan's launch latency once topped the list and the byte ratio below divided
            # by zero on one rank: a diagnostic must never take a rank of the driver's multi-GPU run down)
            dom = max((k for k in stages if alg.get(k, 0.0) > 0.0),
                      key=lambda k: stages[k]["avg_us"] * stages[k]["launches"] * (ppl if stages[k]["launch_us"] else 1))
            upl = ppl if stages[dom]["launch_us"] else 1        # units (stereo pairs) one launch of the dominant kernel covers
            t_launch = (stages[dom]["launch_us"] or stages[dom]["avg_us"]) * 1e-6
            achieved = alg[dom] * upl / t_launch               # algorithmic bytes per launch / live launch duration
            traffic, traffic_source, valu = None, None, None
            prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(prof):
                try:
                    tr = json.load(open(prof))
                    ent = tr.get(args.config, {}).get(dom)
                    if ent and ent.get("cull") == args.cull and ent.get("blend_variant", 4) == args.blend:
                        # the committed counters are per launch of the PROFILED command (its launches cover `pairs_per_launch` pairs,
                        # 1 in the entries of round 3): scaled to the launches of this run
                        scale = upl / max(1, int(ent.get("pairs_per_launch", 1)))
                        traffic = int(ent["hbm_bytes_per_launch"] * scale)
                        traffic_source = (f"{ent.get('source')} ({ent.get('date', 'round 2')}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                          f"passes of tools/profile_round.sh, ({ent.get('read_factor', 2.0)} * FETCH_SIZE + WRITE_SIZE) KiB per launch of {ent.get('kernel')} "
                                          f"[read factor per access pattern: streams 2, gathers 1, profiles/r5_gather_fetch_calibration.json] "
                                          f"({ent.get('pairs_per_launch', 1)} stereo pair(s) per launch there, {upl} here); "
                                          "committed profile, not measured in this run")
                        if ent.get("valu_insts_per_launch"):
                            # the dominant kernel is VALU-bound: wave64 VALU instructions (SQ_INSTS_VALU of the committed PMC
                            # pass) / live launch time, against 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
                            n_valu = ent["valu_insts_per_launch"] * scale
                            rate = n_valu / t_launch
                            valu = dict(insts_per_launch=int(n_valu), achieved_Ginst_s=round(rate / 1e9, 1),
                                        peak_Ginst_s=1228.8, frac=round(rate / 1228.8e9, 4),
                                        measured_plain_fp32_Ginst_s=1010.0,
                                        note="tools/ubench/valu_rates.hip on this chip, 7 waves/SIMD: plain fp32 VALU op 2.4 cycles per "
                                             "wave64 instruction per SIMD (1010 G inst/s), v_exp_f32 8.1, v_cmp / v_cndmask through a lane "
                                             "mask ~4, v_pk_fma_f32 7.0")
                            for name in ("salu_insts_per_launch", "branch_insts_per_launch"):
                                if ent.get(name):
                                    valu[name] = int(ent[name] * scale)
                            if ent.get("salu_insts_per_launch") and ent.get("branch_insts_per_launch"):
                                valu["scalar_plus_branch_per_valu"] = round((ent["salu_insts_per_launch"] + ent["branch_insts_per_launch"]) /
                                                                            ent["valu_insts_per_launch"], 3)
                            if ent.get("valu_trans_per_launch"):
                                # lower bound of the VALU issue time of this instruction mix: transcendental ops at 8.1 cycles,
                                # everything else priced as a plain op (the lane-mask compares / selects cost more)
                                tr_n = ent["valu_trans_per_launch"] * scale
                                cyc = 8.1 * tr_n + 2.4 * (n_valu - tr_n)
                                valu["trans_insts_per_launch"] = int(tr_n)
                                valu["issue_bound_us_at_2p4GHz"] = round(cyc / 1024 / 2.4e3, 1)
                                valu["frac_of_issue_bound"] = round(valu["issue_bound_us_at_2p4GHz"] / (t_launch * 1e6), 3)
                except Exception:
                    traffic = None
            # `bound` names what limits the kernel; achieved / peak / frac stay the HBM figures the contract asks for
            roofline = dict(kernel=dom, bound=("valu" if dom == "blend" else "hbm"), frac_basis="hbm", achieved=round(achieved / 1e9, 2), peak=HBM_PEAK / 1e9, unit="GB/s",
                            frac=round(achieved / HBM_PEAK, 4), traffic=traffic, traffic_source=traffic_source,
                            traffic_over_algorithmic=(round(traffic / (alg[dom] * upl), 3) if traffic and alg[dom] * upl > 0 else None),
                            algorithmic_bytes_per_launch=int(alg[dom] * upl), avg_launch_us=round(t_launch * 1e6, 2),
                            stereo_pairs_per_launch=upl, valu=valu,
                            note="blend is VALU-bound (exp + 10 VALU ops per contributing pixel x instance, data served from LDS); "
                                 "HBM fraction reported as the contract asks, see DESIGN.md")
            tr_all = committed_traffic(args.config)
            def stage_traffic(k):
                """committed PMC traffic of stage k, per unit of `avg_us` (a stereo pair; a FRAME for the TSDF stages, whose launches
                cover `frames_per_launch` frames)"""
                ent = tr_all.get(k, {})
                if ent.get("cull") != args.cull or ent.get("hbm_bytes_per_launch") is None:
                    return None
                return int(ent["hbm_bytes_per_launch"] / max(1, int(ent.get("frames_per_launch", ent.get("pairs_per_launch", 1)))))

            per_kernel = {k: dict(avg_us=round(v["avg_us"], 2), launch_us=(round(v["launch_us"], 2) if v["launch_us"] else None), launches=v["launches"],
                                  alg_GBps=round(alg[k] / max(v["avg_us"], 1e-9) / 1e3, 1),
                                  frac_hbm=round(alg[k] / max(v["avg_us"], 1e-9) * 1e6 / HBM_PEAK, 4),
                                  traffic=stage_traffic(k))
                          for k, v in stages.items()}
            if "tsdf_integrate" in per_kernel:
                # SURVEY.md 8(d) asks for BOTH byte models of the TSDF: the per-frame streaming model (what Open3D does: state read +
                # written per frame, 40 B x updated voxels + 7 B x pixels) and the job-level lower bound (state of every touched voxel
                # read + written ONCE per job, 20 B x voxels of the touched blocks, + the 7 B x pixels of every frame).  The
                # voxel-stationary sweep moves the state once per sweep, so it is priced against the second; the first says what the
                # frame-by-frame algorithm would have to move.
                e = per_kernel["tsdf_integrate"]
                t_frame = max(e["avg_us"], 1e-9) * 1e-6
                B_frame = alg["tsdf_integrate"]
                B_job_frame = (20.0 * n_blocks * 4096 + 7.0 * Wd * Ht * K) / max(K, 1)
                e.update(frac_hbm_frame_model=round(B_frame / t_frame / HBM_PEAK, 4), alg_bytes_frame_model=int(B_frame),
                         frac_hbm_job_bound=round(B_job_frame / t_frame / HBM_PEAK, 4), alg_bytes_job_bound_per_frame=int(B_job_frame),
                         traffic_over_job_bound=(round(e["traffic"] / B_job_frame, 2) if e.get("traffic") else None),
                         note="per FRAME: frame model = 40 B x updated voxels + 7 B x pixels (Open3D's per-frame streaming); job bound = "
                              "(20 B x voxels of the touched blocks + 7 B x pixels x frames) / frames; traffic = committed PMC bytes of "
                              "a sweep / its frames")
            raster_roofline = dict(B_pair_bytes=int(B_pair), t_pair_us=round(t_raster * 1e6, 1),
                                   achieved_GBps=round(B_pair / t_raster / 1e9, 1), frac_of_8TBps=round(B_pair / t_raster / HBM_PEAK, 4),
                                   frac_of_6p29TBps=round(B_pair / t_raster / 6.29e12, 4),
                                   render_only_pairs_per_s=round(1.0 / t_raster, 1))
            tsdf = dict(mvoxel_updates_per_s_job=round(K_total * blocks_frame * 4096 / dt / 1e6, 1),
                        mvoxel_updates_per_s_kernels=round(blocks_frame * 4096 / t_tsdf / 1e6, 1),
                        blocks_per_frame=round(blocks_frame, 1), updated_voxels_per_frame=round(U_frame, 1),
                        allocated_blocks=int(n_blocks), voxel_length=cfg.voxel_length, sdf_trunc=cfg.sdf_trunc,
                        unit="Mvoxel-updates/s (4096 per touched 16^3 block per frame)")
            if rank == 0 and world == 1:
                # mesh extraction of the fused volume, OUTSIDE the timed metric (once per scene, tsdf_utils.py:108 + :133): marching cubes
                # + vertex welding on the device, indexed mesh over PCIe, connected components on the device
                try:
                    torch.cuda.synchronize()
                    tm0 = time.perf_counter()
                    mesh = vol.extract_triangle_mesh()
                    tm1 = time.perf_counter()
                    m_labels, m_counts, _ = mesh.cluster_connected_triangles()
                    tm2 = time.perf_counter()
                    tsdf["mesh"] = dict(triangles=int(mesh.triangles.shape[0]), vertices=int(mesh.vertices.shape[0]),
                                        clusters=int(len(m_counts)), largest_cluster=int(m_counts.max()) if len(m_counts) else 0,
                                        extract_weld_ms=round(1e3 * (tm1 - tm0), 2), cluster_ms=round(1e3 * (tm2 - tm1), 2),
                                        bytes_over_pcie=int(mesh.vertices.nbytes + mesh.vertex_colors.nbytes + mesh.edge_index.nbytes +
                                                            mesh.triangles.nbytes + m_labels.nbytes + m_counts.nbytes),
                                        soup_bytes_round4=int(mesh.triangles.shape[0]) * (2 * 72 + 48),
                                        note="gs2m_tsdf_extract_mesh + gs2m_tsdf_mesh_copy, gs2m_mesh_cluster; wall times incl. the "
                                             "host allocations and copies; not part of `value`")
                    del mesh
                except Exception as e:   # the mesh is an extra of the line, never a reason to lose it
                    tsdf["mesh"] = dict(error=str(e)[:200])
            if red is not None:
                tsdf["reduce"] = dict(mode=args.reduce, seconds=round(t_red, 6), frac_of_timed_region=round(t_red / dt, 4),
                                      union_blocks=int(red["n_blocks_union"]), bytes_per_rank=int(red["bytes_per_rank"]),
                                      collectives=int(red.get("collectives", 0)), payload=red.get("payload"), algo=red.get("algo"),
                                      frames_total=int(red.get("frames_total", 0)), halo_blocks_after=halo_blocks,
                                      world=world, always_collective=bool(args.always_collective),
                                      note="bytes_per_rank = packed union blocks one rank contributes (12 B / voxel packed, 20 B / voxel f32); "
                                           "a reduce-scatter moves (N-1)/N of it over xGMI")

            # ---- oracle legs (rank 0, N = 1 only): parity of the first timed pair, CPU baseline --------------------------
            cpu, par = None, None
            if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_parity):
                import oracle
            if rank == 0 and world == 1 and not args.no_parity:
                # the first timed pair as the timed configuration renders it (this handle: tile_rows, cull, packed SH, fused
                # activations), against the CPU oracle (the reference's own kernels when oracle/_ref is prebuilt)
                from oracle import parity
                rr = R.render_views(gd, cams[Wm], out_color=color, out_rgb8=rgb8, want_radii=True)
                par = parity.pair_parity(g, cams_np[Wm], Wd, Ht, color.cpu().numpy(), rgb8.cpu().numpy(), rr["radii"].cpu().numpy(),
                                         flips=True)
                par["what"] = (f"first timed {args.config} pair (both eyes, worst case), fp32 image on [0,1] vs the CPU oracle; "
                               "radii: fused exp/normalize/sigmoid vs numpy's; flip_pixels = pixels where a decision of renderCUDA "
                               "(power > 0, alpha < 1/255, T(1-alpha) < 1e-4) sits within 1e-5 of its threshold in the oracle's own "
                               "replay; max_abs_clean = max |delta| on all other pixels (bar 2e-4); unexplained_pixels = pixels beyond "
                               "the bound of the contributions that can flip")
                N_ref = par.get("oracle_num_rendered")
                if isinstance(N_ref, list):
                    _, B_pair_ref = alg_bytes(cfg, p_vis, p_vis_union, [float(x) for x in N_ref])
                    raster_roofline["with_reference_instance_count"] = dict(
                        num_rendered_per_eye=N_ref, B_pair_bytes=int(B_pair_ref),
                        frac_of_8TBps=round(B_pair_ref / t_raster / HBM_PEAK, 4),
                        note="same measured time, B_pair evaluated with the reference's num_rendered (16x16 tiles, no culling)")
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                cores = os.cpu_count() or 1
                left_u8 = rgb8[0].cpu().numpy()

                def prep(i):
                    d = depths[i].cpu().numpy()
                    d = np.where(d < np.float32(min_depth), 0, d).astype(np.float32)
                    return oracle.ScalableTSDFVolume.convert_depth(d, 1.0, depth_trunc)

                # upstream runs `#pragma omp parallel for` over the 16 x-slices of one block with all host threads;
                # on a many-core host that oversubscribes (C4 frames took minutes with 256 threads), so probe {16, min(64, cores)} on
                # one frame and keep the faster
                best_threads, best_t = None, None
                d0 = prep(Wm)
                probes = {}

                def probe_one(nt):
                    probe = oracle.ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, 1)
                    probe.set_threads(nt)
                    tc = time.perf_counter()
                    nb = probe.integrate(d0, left_u8, Wd, Ht, cfg.focal, cfg.focal, cx, cy, Es[Wm])
                    tp = time.perf_counter() - tc
                    probes[nt] = dict(threads=nt, value=round(nb * 4096 / tp / 1e6, 2), unit="Mvoxel-updates/s", sample=f"1 frame, {tp:.2f} s")
                    return tp

                for nt in sorted({min(16, cores), min(64, cores)}):
                    tp = probe_one(nt)
                    if best_t is None or tp < best_t:
                        best_threads, best_t = nt, tp
                # "threads = all host cores" -- reported next to the faster setting (one frame; skipped when the
                # 16-thread frame already takes seconds: 256 threads on 16 x-slices only oversubscribe)
                if cores not in probes:
                    if cores <= 64 and best_t < 2.0:
                        probe_one(cores)
                    else:
                        # measured once in round 5 (profiles/r5_bench_C2_with_all_cores_probe.json): 256 OpenMP threads over the 16
                        # x-slices of a block = 218 s for ONE C2 frame (0.03 Mvoxel-updates/s; 16 threads: 0.07 s, 64: 0.30 s) --
                        # the team spins at ~1800 barriers per frame.  Not repeated in every bench run.
                        probes[cores] = dict(threads=cores, skipped="oversubscribed: upstream parallelises the 16 x-slices of one block; "
                                             "measured 0.03 Mvoxel-updates/s (218 s per C2 frame) at 256 threads, "
                                             "profiles/r5_bench_C2_with_all_cores_probe.json")
                ref = oracle.ScalableTSDFVolume(cfg.voxel_length, cfg.sdf_trunc, 1)
                ref.set_threads(best_threads)
                t_cpu, n_cpu, blocks_cpu = 0.0, 0, 0
                for i in range(Wm, Wm + K):
                    d = prep(i)
                    tc = time.perf_counter()
                    blocks_cpu += ref.integrate(d, left_u8, Wd, Ht, cfg.focal, cfg.focal, cx, cy, Es[i])
                    t_cpu += time.perf_counter() - tc
                    n_cpu += 1
                    if t_cpu > args.cpu_seconds:
                        break
                cpu = dict(value=round(blocks_cpu * 4096 / t_cpu / 1e6, 2), unit="Mvoxel-updates/s", cores=best_threads,
                           host_cores=cores, kind="port", all_cores=probes.get(cores), by_threads=[probes[k] for k in sorted(probes)],
                           label="restated Open3D 0.17 ScalableTSDFVolume::Integrate (oracle/tsdf_oracle.cpp, OpenMP over the "
                                 "16 x-slices of a block like upstream; thread count = faster of {16, min(64, host cores)})",
                           sample=f"{n_cpu} of the {K} timed {args.config} frames ({Wd}x{Ht}), integrate() only, {t_cpu:.1f} s")
                # render half: the REFERENCE'S OWN rasteriser kernels compiled for the CPU (oracle/_ref, built in the dev
                # container from /root/reference; the prebuilt library travels to the GPU box), one eye of the first timed pair
                try:
                    if oracle.ref_available(build=False):
                        s_a, q_a, o_a = oracle.activate(g["scaling"], g["rotation"], g["opacity"])
                        shs_a = np.ascontiguousarray(np.concatenate([g["features_dc"], g["features_rest"]], axis=1))
                        lcam = cams_np[Wm][0]
                        tc = time.perf_counter()
                        rr = oracle.ref_forward(g["xyz"], o_a, lcam.world_view_transform, lcam.full_proj_transform,
                                                lcam.camera_center, Wd, Ht, lcam.tanfovx, lcam.tanfovy, np.zeros(3, np.float32),
                                                shs=shs_a, scales=s_a, rotations=q_a)
                        t_eye = time.perf_counter() - tc
                        cpu["raster_reference"] = dict(
                            value=round(0.5 / t_eye, 4), unit="stereo-pairs/s (render only)", cores=cores, kind="reference",
                            label="diff-gaussian-rasterization forward.cu + rasterizer_impl.cu kernels compiled for the CPU "
                                  "(oracle/build_ref.py: CUDA execution shim, blocks over OpenMP threads)",
                            sample=f"left eye of the first timed {args.config} pair ({int(rr['num_rendered'])} instances), {t_eye:.1f} s")
                except Exception as e:  # the reference build is optional test infrastructure
                    cpu["raster_reference"] = dict(error=str(e)[:200])

            # ---- C3 sub-line: the HBM-bound 2 M-Gaussian configuration, render only ----------------------------------------
            c3 = None
            if rank == 0 and world == 1 and not args.no_c3 and args.config != "C3":
                try:
                    del pipe
                    vol.close()
                except Exception:
                    pass
                torch.cuda.empty_cache()
                try:
                    c3 = raster_only(args, "C3", dev, local_rank)
                except Exception as e:   # never lose the main line to the sub-measurement
                    c3 = dict(error=str(e)[:300])

            # ---- trained-like sub-line: the headline's size (C2) with the statistics of a TRAINED splat -- strongly
            # anisotropic scales, 30 % of the opacities at the 0.99 cap, screen-filling background splats, depth ties -- render only
            trained = None
            if rank == 0 and world == 1 and not args.no_trained_like and args.config == "C2":
                try:
                    del pipe
                    vol.close()
                except Exception:
                    pass
                torch.cuda.empty_cache()
                try:
                    trained = raster_only(args, "C2", dev, local_rank, scene="trained_like", with_parity=not args.no_parity)
                    t_syn = sum(v["avg_us"] for k, v in per_kernel.items() if not k.startswith("tsdf"))
                    t_tr = trained["raster_roofline"]["t_pair_us"]
                    n_syn, n_tr = sum(N_eye), sum(trained["num_rendered_per_eye"])
                    trained["vs_synth_v1"] = dict(raster_us_per_pair=round(t_tr, 1), synth_v1_raster_us_per_pair=round(t_syn, 1),
                                                  time_ratio=round(t_tr / t_syn, 3), num_rendered_ratio=round(n_tr / n_syn, 3),
                                                  time_per_instance_ratio=round((t_tr / n_tr) / (t_syn / n_syn), 3),
                                                  note="bar (VERDICT r5 item 3): <= 1.25 x the synth_v1 C2 time per pair at equal "
                                                       "num_rendered scale; DESIGN.md section 6 says where the rest goes")
                except Exception as e:   # never lose the main line to the sub-measurement
                    trained = dict(error=str(e)[:300])

        except Exception as e:
            import traceback
            diag_error = "".join(traceback.format_exception_only(type(e), e)).strip()[:400]
            sys.stderr.write("bench.py: diagnostics failed (the timed value stands):\n" + traceback.format_exc() + "\n")
            roofline = roofline or dict(error=diag_error)
            cpu = cpu or dict(error=diag_error)
            if tsdf is None:
                tsdf = dict(error=diag_error)
                if red is not None:
                    tsdf["reduce"] = dict(mode=args.reduce, seconds=round(t_red, 6), frac_of_timed_region=round(t_red / dt, 4),
                                          union_blocks=int(red["n_blocks_union"]), bytes_per_rank=int(red["bytes_per_rank"]),
                                          payload=red.get("payload"), algo=red.get("algo"), world=world)

    if rank == 0:
        out = dict(
            metric="stereo-pair renders/sec + TSDF Mvoxel-updates/sec",
            value=round(K_total / dt, 2), unit="stereo-pairs/s (rendered L+R and fused)",
            n_gpus=world, steps=args.steps, warmup=Wm, ms_per_step=round(1e3 * dt / max(K, 1), 4), higher_is_better=True,
            scaling=args.scaling, vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=f"{args.config}: {cfg.P} synth_v1 Gaussians (SH deg 3), {K} stereo pairs/GPU at "
                                 f"{Wd}x{Ht}, TSDF {cfg.tsdf_n}^3 (voxel {cfg.voxel_length:g}, trunc {cfg.sdf_trunc}), "
                                 f"sphere depth", gaussians=cfg.P, width=Wd, height=Ht, pairs_per_gpu=K,
                        exact_tile_cull=args.cull, blend_variant=args.blend, tile_rows=args.tile_rows, spatial_order=spatial_order_used, pairs_in_flight=args.inflight, pairs_per_launch=pairs_per_launch_used,
                        tsdf_fuse_batch=(fuse_plan if fuse_plan else args.fuse_batch),
                        parallelism=("1 GPU" if world == 1 else (f"REHEARSAL: {world} ranks on one GPU over gloo (host-staged exchange), not a scaling number"
                                                                    if args.rehearsal else f"views sharded over {world} GPUs + RCCL {args.reduce} of the TSDF"))),
            steady_state=steady,
            timing=dict(repeats=len(dts), timed_region_s=round(sum(dts), 4), statistic="median over repeats of the K-step job",
                        ms_per_step_min=round(1e3 * min(dts) / K, 4), ms_per_step_max=round(1e3 * max(dts) / K, 4),
                        host_enqueue_ms_per_step=round(1e3 * statistics.median(enq) / K, 4)),   # host side of the K steps: launches only, no sync
            num_rendered_per_eye=[int(x) for x in N_eye], p_visible_per_eye=p_vis,
            tsdf=tsdf, stages=per_kernel, roofline=roofline, raster_roofline=raster_roofline, parity=par, cpu_baseline=cpu,
            c3=c3, trained_like=trained, volume_check=volume_check, diagnostics_error=diag_error, instrumented_ms_per_step=round(1e3 * dt_instr / K, 4),
            note_stages="`stages` / `roofline` / `instrumented_ms_per_step`: second pass, serial on one stream with hipEvents "
                        "around every launch, `pairs_per_launch` stereo pairs per chain of launches = the launch shapes of the "
                        "timed pass (kernels in isolation; raster `avg_us` = launch_us / pairs_per_launch, TSDF `avg_us` per frame); "
                        "`value`: timed pass with `pairs_in_flight` slots overlapped on separate streams")
        print(json.dumps(out), file=line_out)
        line_out.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
