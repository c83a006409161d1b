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
    def __init__(self, gaussians: dict, width: int, height: int, volume: ScalableTSDFVolume | None,
                 intrinsic: PinholeCameraIntrinsic | None = None, inflight: int = 2, device: int = 0,
                 exact_tile_cull: int = 1, blend_variant: int | None = None, tile_rows: int = 2, bg=(0.0, 0.0, 0.0),
                 lib=None):
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        self.g = gaussians
        self.W, self.H = int(width), int(height)
        self.volume, self.intrinsic = volume, intrinsic
        self.inflight = int(inflight)
        self.device = int(device)
        self.bg = bg
        dev = torch.device(f"cuda:{device}")
        self.rasterizers, self.color, self.rgb8 = [], [], []
        for j in range(self.inflight):
            r = Rasterizer(device, lib=lib)
            r.set_option(_lib.OPT_EXACT_TILE_CULL, int(exact_tile_cull))
            r.set_option(_lib.OPT_TILE_ROWS, int(tile_rows))   # 2 = 16 x 32 binning tiles (same image, fewer instances)
            if blend_variant is not None:
                r.set_option(_lib.OPT_BLEND_VARIANT, int(blend_variant))
            self.rasterizers.append(r)
            self.color.append(torch.empty((2, 3, self.H, self.W), dtype=torch.float32, device=dev))
            self.rgb8.append(torch.empty((2, self.H, self.W, 3), dtype=torch.uint8, device=dev))
        if self.inflight == 1:
            # serial mode: everything on the caller's current stream
            self.render_streams, self.fuse_stream = [None], None
        else:
            self.render_streams = [torch.cuda.Stream(device=dev) for _ in range(self.inflight)]
            self.fuse_stream = torch.cuda.Stream(device=dev)
        self._rendered = [torch.cuda.Event() for _ in range(self.inflight)]
        self._fused = [torch.cuda.Event() for _ in range(self.inflight)]
        self._n = 0

    # -- set-up ------------------------------------------------------------------------------------
    def prepare(self, cams, headroom: float = 1.3):
        """Sizes every handle's instance arena from one synchronous render of ``cams`` (grow + retry
        happens here, not in the pipelined loop) and packs the SH block once.  Returns the render."""
        first = None
        for j, r in enumerate(self.rasterizers):
            r.pack_sh(self.g)      # the wave-transposed SH copy is cached per handle
            res = r.render_views(self.g, cams, bg=self.bg, out_color=self.color[j], out_rgb8=self.rgb8[j],
                                 want_radii=(j == 0))
            r.reserve(int(self.g["xyz"].shape[0]), 2, self.W, self.H, int(max(res["num_rendered"]) * headroom))
            if j == 0:
                first = res
        torch.cuda.synchronize(self.device)
        return first

    # -- steady state ------------------------------------------------------------------------------
    def submit(self, cams, depth=None, extrinsic=None, depth_scale=1.0, depth_trunc=float("inf"), min_depth=0.0,
               mask=None):
        """Enqueue: render the stereo pair ``cams`` and (if ``depth`` is given) integrate the LEFT image with
        ``depth`` [H,W] f32 (device) under ``extrinsic`` (world -> camera, 4x4).  Returns the slot index whose
        ``color[slot]`` / ``rgb8[slot]`` will hold the pair (valid after `wait_rendered(slot)` / `finish`)."""
        j = self._n % self.inflight
        self._n += 1
        r = self.rasterizers[j]
        if self.inflight == 1:
            r.render_views(self.g, cams, bg=self.bg, out_color=self.color[0], out_rgb8=self.rgb8[0], sync=False)
            if depth is not None:
                self.volume.integrate(RGBDImage(self.rgb8[0][0], depth, depth_scale=depth_scale, depth_trunc=depth_trunc),
                                      self.intrinsic, extrinsic, mask=mask, min_depth=min_depth)
            return 0
        # Inputs (Gaussians, depth, mask) were produced on the caller's current stream (e.g. by the stereo network):
        # both private streams wait for it, and the caching allocator is told the fuse stream still reads depth / mask
        # after the caller drops them.
        cur = torch.cuda.current_stream(self.device)
        rs = self.render_streams[j]
        rs.wait_stream(cur)
        with torch.cuda.stream(rs):
            rs.wait_event(self._fused[j])          # the view that last used this slot's images is integrated
            r.render_views(self.g, cams, bg=self.bg, out_color=self.color[j], out_rgb8=self.rgb8[j], sync=False)
            self._rendered[j].record(rs)
        if depth is not None:
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
        return j

    def wait_rendered(self, slot: int):
        if self.inflight > 1:
            self._rendered[slot].synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()

    def drain(self):
        """Host waits for everything submitted so far (render + fuse streams); no status query, no device-wide sync."""
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
        if self.inflight > 1:
            for s in self.render_streams:
                s.synchronize()
            self.fuse_stream.synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()
        for j, r in enumerate(self.rasterizers):
            nr, ov, req = r.status(2)
            if ov:
                raise RuntimeError(f"instance arena overflow inside the pipelined loop (slot {j}: views "
                                   f"{j}, {j + self.inflight}, ... of the {self._n} submitted; need {req} instances "
                                   f"per view)")

    def set_stage_timing(self, enable: bool):
        for r in self.rasterizers:
            r.set_option(_lib.OPT_STAGE_TIMING, int(bool(enable)))
        if self.volume is not None:
            self.volume.set_stage_timing(enable)
