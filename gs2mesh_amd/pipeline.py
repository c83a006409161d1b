"""Render -> fuse pipelining over HIP streams (host-side scheduler of the hot path).

The reference walks the views one at a time: render the stereo pair, (estimate depth), integrate the view
(run_single.py:86-160: Renderer.render_image_pair -> Stereo.run -> TSDF.run).  The rasteriser's binning
kernels (project / scans / scatter / per-tile sort) are latency-bound and leave the vector ALUs idle, its
compositing kernel is ALU-bound, the TSDF kernels are short: back to back on one stream they under-use the
chip.  `RenderFusePipeline` keeps ``inflight`` stereo pairs in flight on separate streams, each with its own
rasteriser handle (own scratch arenas and output images), and integrates on a third stream in view order:

    render stream j = i % inflight :  [wait fused(i - inflight)]  render pair i      -> event rendered(i)
    fuse stream                    :  [wait rendered(i)]          integrate view i   -> event fused(i)

so pair i+1's binning overlaps pair i's compositing and view i-1's integration.  Results are identical to the
serial order: every view is rendered by the same kernels on the same inputs and the volume is updated in view
order on one stream (tests/test_pipeline_overlap.py).  No host synchronisation inside `submit`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .integration import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
from .rasterizer import Rasterizer


class RenderFusePipeline:
    """``raster_options``: ``gs2m_raster_set_option`` values of every slot's handle, by name -- ``exact_tile_cull`` (default
    "auto" = ``rasterizer.auto_cull_level``: 1, or 2 for models of >= 1 M Gaussians), ``tile_rows`` (default 2: 16 x 32 binning
    tiles, same image, fewer instances), ``blend_mode`` (default "auto" = ``rasterizer.auto_blend_mode``: 2, or 3 for models with
    >= 10 % of their opacities at the alpha cap), ``blend_variant``, ``bin_lane_tiles``, ``project_shared_read``."""

    RASTER_OPTION_IDS = dict(exact_tile_cull=_lib.OPT_EXACT_TILE_CULL, tile_rows=_lib.OPT_TILE_ROWS,
                             blend_variant=_lib.OPT_BLEND_VARIANT, blend_mode=_lib.OPT_BLEND_MODE,
                             bin_workgroups=_lib.OPT_BIN_WORKGROUPS, bin_wg_threads=_lib.OPT_BIN_WG_THREADS,
                             bin_lane_tiles=_lib.OPT_BIN_LANE_TILES, project_shared_read=_lib.OPT_PROJECT_SHARED_READ)

    def __init__(self, gaussians: dict, width: int, height: int, volume: ScalableTSDFVolume | None = None,
                 intrinsic: PinholeCameraIntrinsic | None = None, inflight: int = 2, device: int = 0, *, fuse_batch=1,
                 pairs_per_launch: int = 1, spatial_order="auto", bg=(0.0, 0.0, 0.0), raster_options: dict | None = None):
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        self.g = gaussians
        # Morton-ordered packed copy of the model in every handle (same results).  "auto": when the model is large.  The
        # ordering pays once the keys one XCD scatters per pass no longer fit its 4 MiB L2 (C3, 2 M Gaussians: scatter
        # 115 -> 62 us); below that the XCD-contiguous rows already merge the key stores and the ordered model only adds
        # LDS-atomic conflicts to the counting kernel (C2, 300 k Gaussians: count 29 -> 35 us, scatter 26 -> 25 us).
        if spatial_order == "auto":
            from .rasterizer import auto_spatial_order
            spatial_order = auto_spatial_order(int(gaussians["xyz"].shape[0]))
        self.spatial_order = bool(spatial_order)
        self.W, self.H = int(width), int(height)
        self.volume, self.intrinsic = volume, intrinsic
        self.inflight = int(inflight)
        # pairs_per_launch = 2 (GS2M_OPT_PAIR_BATCH): two consecutive stereo pairs go through ONE chain of launches (`submit`
        # buffers the first, the second triggers the launch; `drain` flushes an odd one).  Batched fusion only.
        self.ppl = int(pairs_per_launch)
        if self.ppl not in (1, 2, 3, 4):
            raise ValueError("pairs_per_launch must be 1 .. 4")
        self._group = []
        self.device = int(device)
        self.bg = bg
        opts = dict(exact_tile_cull="auto", tile_rows=2)
        opts.update(raster_options or {})
        if opts.get("exact_tile_cull") in ("auto", -1):
            from .rasterizer import auto_cull_level
            opts["exact_tile_cull"] = auto_cull_level(int(gaussians["xyz"].shape[0]))
        self.exact_tile_cull = int(opts["exact_tile_cull"] or 0)
        if opts.get("blend_mode", "auto") in ("auto", -1):
            from .rasterizer import auto_blend_mode
            opts["blend_mode"] = auto_blend_mode(gaussians)
        self.blend_mode = int(opts["blend_mode"])
        unknown = set(opts) - set(self.RASTER_OPTION_IDS)
        if unknown:
            raise ValueError(f"unknown raster_options {sorted(unknown)}; known: {sorted(self.RASTER_OPTION_IDS)}")
        dev = torch.device(f"cuda:{device}")
        self.rasterizers, self.color, self.rgb8, self._own8 = [], [], [], []
        for j in range(self.inflight):
            r = Rasterizer(device)
            for name, value in opts.items():
                if value is not None:
                    r.set_option(self.RASTER_OPTION_IDS[name], int(value))
            if self.ppl > 1:
                r.set_option(_lib.OPT_PAIR_BATCH, self.ppl)
            self.rasterizers.append(r)
            self.color.append(torch.empty((2 * self.ppl, 3, self.H, self.W), dtype=torch.float32, device=dev))
            # rgb8[j] = where slot j's latest u8 pair lives (the slot's own buffer, or the pending view's batch buffer)
            self._own8.append(torch.empty((2 * self.ppl, self.H, self.W, 3), dtype=torch.uint8, device=dev))
            self.rgb8.append(self._own8[j])
        if self.inflight == 1:
            # serial mode: everything on the caller's current stream
            self.render_streams, self.fuse_stream = [None], None
        else:
            # plain streams: every variant that gave the binning / TSDF kernels CUs, hardware queues or CU-internal room of
            # their own (CU-masked streams, a separate compositing stream, workgroup caps) was measured slower in round 3
            # (profiles/r3_partition_sweeps.txt, r3_experiments.txt) and removed in round 4
            self.render_streams = [torch.cuda.Stream(device=dev) for _ in range(self.inflight)]
            self.fuse_stream = torch.cuda.Stream(device=dev)
        # fuse_batch > 1: the views are integrated `fuse_batch` at a time by the voxel-stationary batch kernel
        # (gs2m_tsdf_integrate_batch: same result as view by view, one voxel-state read + write per batch).  The left
        # image of every pending view is kept in its own buffer (a slot's image is re-rendered before the batch runs).
        # a list / tuple = a PLAN of batch sizes, cycled (e.g. 10, 5, 3, 2 for a 20-view job: the sweep that runs after the
        # last render -- nothing left to overlap it with -- is the smallest one)
        plan = [int(b) for b in fuse_batch] if isinstance(fuse_batch, (list, tuple)) else [int(fuse_batch)]
        self._plan = [max(1, min(b, 64)) for b in plan] or [1]
        if len(self._plan) > 1 and min(self._plan) < 2:
            self._plan = [max(2, b) for b in self._plan]       # inside a plan every sweep goes through the batch kernel
        self._plan_i = 0
        self.fuse_batch = max(self._plan)
        self._pending = []
        # two sets of image buffers: batch b + 1 is collected while batch b is still being integrated
        # (the u8 pair of a pending view is rendered straight into its buffer: no copy out of the slot)
        nb = self.fuse_batch if self.fuse_batch > 1 else 0
        # one flat buffer per set: the pairs of consecutive pending views are contiguous (pairs_per_launch = 2 renders two at once)
        self._bflat = [torch.empty((2 * nb, self.H, self.W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self._bpair = [[self._bflat[b][2 * k:2 * k + 2] for k in range(nb)] for b in range(2)]
        if self.ppl > 1 and (self.fuse_batch < 2 or any(b % self.ppl for b in self._plan)):
            raise ValueError("pairs_per_launch > 1 needs batched fusion with sweep sizes (fuse_batch) that are multiples of it")
        self._bcopied = [[torch.cuda.Event() for _ in range(nb)] for _ in range(2)]
        self._batch_done = [None, None]
        self._bset = 0
        self._rendered = [torch.cuda.Event() for _ in range(self.inflight)]
        self._fused = [torch.cuda.Event() for _ in range(self.inflight)]
        self._fused_on = [None] * self.inflight        # the stream _fused[j] was last recorded on
        self._model_seen = [False] * self.inflight     # slot's render stream has waited for the stream the model was produced on
        self._released = [None] * self.inflight        # `release(slot)`: event after the caller's last read of the slot's images
        self._n = 0

    # -- set-up ------------------------------------------------------------------------------------
    def prepare(self, cams, headroom: float = 1.3):
        """Sizes every handle's instance arena from one synchronous render of ``cams`` (grow + retry
        happens here, not in the pipelined loop) and packs the SH block once.  Returns the render."""
        first = None
        order = None
        for j, r in enumerate(self.rasterizers):
            # one-time re-layout, cached per handle: Morton-ordered packed copy of the model (+ wave-transposed SH);
            # the order is computed once and shared by the handles
            if self.spatial_order:
                order = r.pack_model(self.g, order=order)
            else:
                r.pack_sh(self.g)
            res = r.render_views(self.g, cams, bg=self.bg, out_color=self.color[j][:2], out_rgb8=self.rgb8[j][:2],
                                 want_radii=(j == 0))
            # every view of a pass gets its own key range: reserve for the 2 * pairs_per_launch views of a launch, and run
            # one launch of that shape now (the first batched call would otherwise grow keys / records / histogram rows with
            # hipMalloc + hipFree -- a device-wide sync -- inside the pipelined loop)
            r.reserve(int(self.g["xyz"].shape[0]), 2 * self.ppl, self.W, self.H, int(max(res["num_rendered"]) * headroom))
            if self.ppl > 1:
                r.render_views(self.g, list(cams) * self.ppl, bg=self.bg, out_color=self.color[j], out_rgb8=self._own8[j],
                               sync=False)
            if j == 0:
                first = res
        torch.cuda.synchronize(self.device)
        return first

    def model_updated(self):
        """The Gaussians ``self.g`` were changed (in place or re-bound) on the caller's current stream: the packed copies
        are dropped and every render stream waits for that stream before its next render."""
        for r in self.rasterizers:
            r.invalidate_pack()
        self._model_seen = [False] * self.inflight

    # -- steady state ------------------------------------------------------------------------------
    def submit(self, cams, depth=None, extrinsic=None, depth_scale=1.0, depth_trunc=float("inf"), min_depth=0.0,
               mask=None):
        """Enqueue: render the stereo pair ``cams`` and (if ``depth`` is given) integrate the LEFT image with
        ``depth`` [H,W] f32 (device) under ``extrinsic`` (world -> camera, 4x4).  Returns the slot index whose
        ``color[slot]`` / ``rgb8[slot]`` will hold the pair (valid after `wait_rendered(slot)` / `finish`)."""
        if self.ppl > 1:
            if depth is None:
                raise ValueError("pairs_per_launch > 1: every view is fused (depth required)")
            self._group.append((cams, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth))
            j = self._n % self.inflight
            if len(self._group) == self.ppl:
                self._launch_group()
            return j
        j = self._n % self.inflight
        self._n += 1
        r = self.rasterizers[j]
        if self.inflight == 1:
            # a fuse-less render (or view-by-view fusion) writes the slot's OWN buffer, never a pending view's batch buffer
            self.rgb8[0] = self._own8[0]
            if depth is not None and self.fuse_batch > 1:
                k = len(self._pending)
                self.rgb8[0] = self._bpair[self._bset][k]     # rendered straight into the pending view's buffer
            r.render_views(self.g, cams, bg=self.bg, out_color=self.color[0], out_rgb8=self.rgb8[0], sync=False)
            if depth is not None and self.fuse_batch > 1:
                self._pending.append((k, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth))
                if len(self._pending) == self._plan[self._plan_i % len(self._plan)]:
                    self._flush_batch()
            elif depth is not None:
                self.volume.integrate(RGBDImage(self.rgb8[0][0], depth, depth_scale=depth_scale, depth_trunc=depth_trunc),
                                      self.intrinsic, extrinsic, mask=mask, min_depth=min_depth)
            return 0
        # Inputs (Gaussians, depth, mask) were produced on the caller's current stream (e.g. by the stereo network):
        # both private streams wait for it, and the caching allocator is told the fuse stream still reads depth / mask
        # after the caller drops them.
        cur = torch.cuda.current_stream(self.device)
        rs = self.render_streams[j]
        # The render stream reads the model and the cameras only: it waits for the caller's stream ONCE per slot after
        # construction / `prepare` / `model_updated` (the model upload), not at every step.  A per-step `wait_stream(cur)`
        # puts a marker into the caller stream's hardware queue, which that stream shares with a render stream (HIP
        # multiplexes all streams onto 4 hardware queues): the marker sits behind a whole chain + compositing kernel of
        # another slot and holds this slot's chain back (C2: 0.323 -> 0.318 ms per step).  Depth and mask are read by
        # the fuse stream, which does wait for the caller's stream (below / `_flush_batch`).
        if not self._model_seen[j]:
            rs.wait_stream(cur)
            self._model_seen[j] = True
        with torch.cuda.stream(rs):
            if self._fused_on[j] is not rs:
                rs.wait_event(self._fused[j])      # the view that last used this slot's images is integrated (same-stream order otherwise)
            self._wait_released(j, rs)
            batched = depth is not None and self.fuse_batch > 1
            self.rgb8[j] = self._own8[j]   # fuse-less / view-by-view: the slot's own buffer (ordered by _fused[j] above)
            if batched:
                # the u8 pair goes straight into the pending view's buffer (no copy out of the slot): the batch that last
                # read this buffer set (two batches ago) must be done
                k = len(self._pending)
                if self._batch_done[self._bset] is not None:
                    rs.wait_event(self._batch_done[self._bset])
                self.rgb8[j] = self._bpair[self._bset][k]
            r.render_views(self.g, cams, bg=self.bg, out_color=self.color[j], out_rgb8=self.rgb8[j], sync=False)
            self._rendered[j].record(rs)
            if batched:
                self._bcopied[self._bset][k].record(rs)
        if depth is not None and self.fuse_batch > 1:
            for t in (depth, mask):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.fuse_stream)
            self._pending.append((len(self._pending), depth, extrinsic, mask, depth_scale, depth_trunc, min_depth))
            self._fused[j].record(rs)                      # the slot's image has been copied out: free to re-render
            self._fused_on[j] = rs
            if len(self._pending) == self._plan[self._plan_i % len(self._plan)]:
                self._flush_batch(cur)
        elif depth is not None:
            fs = self.fuse_stream
            fs.wait_stream(cur)
            for t in (depth, mask):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(fs)
            with torch.cuda.stream(fs):
                fs.wait_event(self._rendered[j])
                self.volume.integrate(RGBDImage(self.rgb8[j][0], depth, depth_scale=depth_scale, depth_trunc=depth_trunc),
                                      self.intrinsic, extrinsic, mask=mask, min_depth=min_depth)
                self._fused[j].record(fs)
                self._fused_on[j] = fs
        return j

    def _launch_group(self):
        """pairs_per_launch = 2: the buffered views (2, or 1 when `drain` flushes an odd one) through one chain of launches
        on the next slot; their u8 pairs go straight into consecutive buffers of the pending batch."""
        views, self._group = self._group, []
        n = len(views)
        j = self._n % self.inflight
        self._n += 1
        r = self.rasterizers[j]
        if self.inflight == 1:
            # serial mode: the same launches on the caller's current stream
            k0 = len(self._pending)
            self.rgb8[0] = self._bflat[self._bset][2 * k0:2 * (k0 + n)]
            r.render_views(self.g, [c for v in views for c in v[0]], bg=self.bg, out_color=self.color[0][:2 * n],
                           out_rgb8=self.rgb8[0], sync=False)
            for i, (_, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth) in enumerate(views):
                self._pending.append((k0 + i, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth))
            if len(self._pending) >= self._plan[self._plan_i % len(self._plan)]:
                self._flush_batch()
            return
        cur = torch.cuda.current_stream(self.device)
        rs = self.render_streams[j]
        if not self._model_seen[j]:
            rs.wait_stream(cur)
            self._model_seen[j] = True
        k0 = len(self._pending)
        with torch.cuda.stream(rs):
            if self._fused_on[j] is not rs:
                rs.wait_event(self._fused[j])
            self._wait_released(j, rs)
            if self._batch_done[self._bset] is not None:
                rs.wait_event(self._batch_done[self._bset])
            self.rgb8[j] = self._bflat[self._bset][2 * k0:2 * (k0 + n)]
            r.render_views(self.g, [c for v in views for c in v[0]], bg=self.bg, out_color=self.color[j][:2 * n],
                           out_rgb8=self.rgb8[j], sync=False)
            self._rendered[j].record(rs)
            for i in range(n):
                self._bcopied[self._bset][k0 + i].record(rs)
        for i, (_, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth) in enumerate(views):
            for t in (depth, mask):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.fuse_stream)
            self._pending.append((k0 + i, depth, extrinsic, mask, depth_scale, depth_trunc, min_depth))
        self._fused[j].record(rs)
        self._fused_on[j] = rs
        if len(self._pending) >= self._plan[self._plan_i % len(self._plan)]:
            self._flush_batch(cur)

    def _flush_batch(self, cur=None):
        """Integrate the pending views (in submission order) with one batch sweep."""
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        self._plan_i += 1
        p0 = pend[0]
        if any((p[4], p[5], p[6]) != (p0[4], p0[5], p0[6]) for p in pend):
            raise ValueError("fuse_batch: the views of a batch must share depth_scale / depth_trunc / min_depth")
        bset = self._bset
        self._bset ^= 1
        images = [RGBDImage(self._bpair[bset][k][0], d, depth_scale=p0[4], depth_trunc=p0[5]) for k, d, *_ in pend]
        masks = [p[3] for p in pend]
        args = (images, self.intrinsic, [p[2] for p in pend])
        kw = dict(masks=masks if any(m is not None for m in masks) else None, min_depth=p0[6])
        if self.inflight == 1:
            self.volume.integrate_batch(*args, **kw)
            return
        fs = self.fuse_stream
        if cur is not None:
            fs.wait_stream(cur)
        with torch.cuda.stream(fs):
            for k, *_ in pend:
                fs.wait_event(self._bcopied[bset][k])
            self.volume.integrate_batch(*args, **kw)
            self._batch_done[bset] = torch.cuda.Event()
            self._batch_done[bset].record(fs)

    def wait_rendered(self, slot: int, stream_only: bool = False):
        """The slot's images are complete: the HOST waits (default), or -- ``stream_only`` -- only the caller's current
        stream does (work enqueued on it afterwards reads finished images; no host synchronisation)."""
        if self.inflight > 1:
            if stream_only:
                torch.cuda.current_stream(self.device).wait_event(self._rendered[slot])
            else:
                self._rendered[slot].synchronize()
        elif not stream_only:
            torch.cuda.current_stream(self.device).synchronize()

    def release(self, slot: int):
        """Consumer fence of a slot's output buffers: call on the stream that READ ``color[slot]`` / ``rgb8[slot]`` (e.g. the
        stereo network consuming the rendered pair) after enqueueing those reads.  The slot is not re-rendered before that
        stream reaches this point.  Without it the contract is the one INTEGRATION.md states: reads of a slot's images must
        have completed (host-side) before the slot is submitted again, ``inflight`` submits later."""
        if self.inflight > 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._released[slot] = ev

    def _wait_released(self, j, rs):
        if self._released[j] is not None:
            rs.wait_event(self._released[j])
            self._released[j] = None

    def drain(self):
        """Host waits for everything submitted so far (render + fuse streams; a partial batch is integrated first); no
        status query, no device-wide sync.  The batch plan starts over."""
        if self._group:
            self._launch_group()
        self._flush_batch(torch.cuda.current_stream(self.device) if self.inflight > 1 else None)
        self._plan_i = 0
        if self.inflight > 1:
            for s in self.render_streams:
                s.synchronize()
            self.fuse_stream.synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()

    def finish(self):
        """Drain every stream; raises if an instance arena overflowed in ANY pair submitted since the last
        `finish` (the handle's overflow word is sticky on the device: a later pair that fits does not erase it).
        The images of an overflowing pair were composited from a truncated instance list, so the views integrated
        from them are invalid: grow the arenas (`prepare(..., headroom=)`) and redo the loop."""
        self.drain()
        for j, r in enumerate(self.rasterizers):
            nr, ov, req = r.status(2)
            if ov:
                raise RuntimeError(f"instance arena overflow inside the pipelined loop (slot {j}: views "
                                   f"{j}, {j + self.inflight}, ... of the {self._n} submitted; need {req} instances "
                                   f"per view)")

    def close(self):
        """Drain the pipeline's streams."""
        self.drain()

    def set_stage_timing(self, enable: bool):
        for r in self.rasterizers:
            r.set_option(_lib.OPT_STAGE_TIMING, int(bool(enable)))
        if self.volume is not None:
            self.volume.set_stage_timing(enable)
