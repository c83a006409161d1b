"""ctypes binding of the C ABI (include/gs2mesh_amd.h).

``get()`` loads ``gs2mesh_amd/libgs2mesh_amd.so`` -- the HIP/gfx950 build -- and raises if it is
missing: there is no CPU fallback in the product.  ``bind(cdll)`` only attaches prototypes; the
test-suite also uses it on the CPU emulator build of the same kernel sources (tests/emu).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgs2mesh_amd.so")

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float
f64 = C.c_double


class Camera(C.Structure):
    """gs2m_camera (host struct)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tanfovx", f32), ("tanfovy", f32),
                ("viewmatrix", f32 * 16), ("projmatrix", f32 * 16), ("campos", f32 * 3)]


class Gaussians(C.Structure):
    """gs2m_gaussians (host struct of device pointers)."""
    _fields_ = [("P", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32), ("raw", C.c_int32),
                ("xyz", vp), ("scales", vp), ("rotations", vp), ("opacities", vp), ("shs", vp), ("shs_rest", vp)]


ABI_VERSION = 600   # GS2M_VERSION of include/gs2mesh_amd.h this binding was written against
OPT_EXACT_TILE_CULL = 1
OPT_BLEND_VARIANT = 2
OPT_TILE_ROWS = 5
OPT_PAIR_BATCH = 8
OPT_BIN_WORKGROUPS = 9      # tuning options: results never change
OPT_BIN_WG_THREADS = 10
OPT_BLEND_MODE = 11
TSDF_MAP_HEADER_BYTES = 32
OPT_BLEND_PROFILE = 12
OPT_BIN_LANE_TILES = 13
OPT_PROJECT_SHARED_READ = 14
BLEND_PROF_COUNTERS = ("waves", "wave_cycles", "dma_wait", "staging", "prefetch_issue", "loop", "epilogue", "batches",
                       "staged_instances", "listed_instances")
XFORM_SUM_F32, XFORM_RAW_F32, XFORM_SUM_PACKED = 0, 1, 2
XFORM_PACKED_MAX_FRAMES = 1023
OPT_DEBUG_SYNC = 3
OPT_STAGE_TIMING = 4
RASTER_STAGES = ("project", "hist_colscan", "tile_scan", "scatter", "sort_tiles", "blend", "count_tiles")
TSDF_STAGES = ("tsdf_touch", "tsdf_integrate")

_PROTOS = {
    "gs2m_version": (i32, []),
    "gs2m_last_error": (C.c_char_p, []),
    "gs2m_raster_create": (i32, [C.POINTER(vp), i32]),
    "gs2m_raster_destroy": (i32, [vp]),
    "gs2m_raster_set_option": (i32, [vp, i32, i32]),
    "gs2m_raster_reserve": (i32, [vp, i32, i32, i32, i32, i64]),
    "gs2m_rasterize_forward": (i32, [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp,
                                     f32, f32, i32, vp, vp, i32, vp]),
    "gs2m_mark_visible": (i32, [i32, vp, vp, vp, vp, vp]),
    "gs2m_render_views": (i32, [vp, C.POINTER(Gaussians), C.POINTER(Camera), i32, C.POINTER(f32), f32, vp, vp, vp,
                                vp]),
    "gs2m_raster_pack_sh": (i32, [vp, C.POINTER(Gaussians), vp]),
    "gs2m_raster_pack_model": (i32, [vp, C.POINTER(Gaussians), vp, vp]),
    "gs2m_raster_pack_invalidate": (i32, [vp]),
    "gs2m_raster_status": (i32, [vp, vp, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i64)]),
    "gs2m_raster_stage_times": (i32, [vp, vp, C.POINTER(f64), C.POINTER(i64)]),
    "gs2m_raster_blend_cycles": (i32, [vp, vp, C.POINTER(C.c_uint64)]),
    "gs2m_raster_download_geometry": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "gs2m_raster_download_binning": (i32, [vp, vp, i32, i64, vp, C.c_int32, vp]),
    "gs2m_tsdf_extract_count": (i32, [vp, vp, C.POINTER(i64)]),
    "gs2m_tsdf_extract": (i32, [vp, vp, i64, vp, vp, C.POINTER(i64)]),
    "gs2m_tsdf_extract_indexed": (i32, [vp, vp, i64, vp, vp, vp, C.POINTER(i64)]),
    "gs2m_stereo_depth_occlusion": (i32, [vp, vp, i32, i32, f64, f64, vp, vp, vp]),
    "gs2m_tsdf_create": (i32, [C.POINTER(vp), f64, f64, i32, i32, i32, i64, i32]),
    "gs2m_tsdf_destroy": (i32, [vp]),
    "gs2m_tsdf_reset": (i32, [vp, vp]),
    "gs2m_tsdf_integrate": (i32, [vp, vp, vp, vp, i32, i32, f64, f64, f64, f64, C.POINTER(f64), f64, f64, f64, vp]),
    "gs2m_tsdf_integrate_batch": (i32, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i32, f64, f64, f64, f64,
                                        C.POINTER(f64), f64, f64, f64, vp]),
    "gs2m_tsdf_set_stage_timing": (i32, [vp, i32]),
    "gs2m_tsdf_stage_times": (i32, [vp, vp, C.POINTER(f64), C.POINTER(i64)]),
    "gs2m_tsdf_status": (i32, [vp, vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]),
    "gs2m_tsdf_download": (i32, [vp, vp, i64, vp, vp, vp, vp]),
    "gs2m_tsdf_block_keys": (i32, [vp, i64, vp, vp]),
    "gs2m_tsdf_flags_device": (i32, [vp, vp, vp]),
    "gs2m_tsdf_pack_sum": (i32, [vp, vp, i64, vp, vp]),
    "gs2m_tsdf_unpack_sum": (i32, [vp, vp, i64, vp, i32, vp]),
    "gs2m_tsdf_pack": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "gs2m_tsdf_extract_mesh": (i32, [vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "gs2m_tsdf_mesh_copy": (i32, [vp, vp, vp, vp, vp, vp]),
    "gs2m_mesh_cluster": (i32, [i32, vp, i64, vp, vp, vp, C.POINTER(i64)]),
    "gs2m_tsdf_replace": (i32, [vp, vp, i64, i32, vp, vp, vp]),
    "gs2m_tsdf_map_bytes": (i64, [C.POINTER(C.c_int32), i32]),
    "gs2m_tsdf_block_map": (i32, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32, i32, i64, i64, i32, vp, vp]),
    "gs2m_tsdf_map_keys": (i32, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32, vp, vp, i64, C.POINTER(C.c_uint8), vp]),
    "gs2m_tsdf_unpack": (i32, [vp, vp, i64, i32, vp, vp, i32, vp]),
}

SYMBOLS = tuple(_PROTOS)


def bind(lib: C.CDLL, require_all: bool = True) -> C.CDLL:
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if require_all:
                raise
            continue
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


class DeviceMemory:
    """Where the wrappers' buffers live and how a buffer becomes a pointer for the C ABI: HIP device memory, nothing
    else (there is no CPU path).  The wrappers go through the module-level ``MEMORY`` object; the emulator test
    harness substitutes its own subclass from the test tree (tests/backends.py) -- no switch for it exists here."""

    def ptr(self, x, dtype=None, name="tensor"):
        """Device pointer of a contiguous torch tensor.  None / empty -> NULL."""
        import torch
        if x is None:
            return None
        if not isinstance(x, torch.Tensor):
            raise RuntimeError(f"{name}: the C ABI takes HIP device memory (a torch tensor on a GPU), got {type(x).__name__}")
        if x.numel() == 0:
            return None
        if not x.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
        if dtype is not None and x.dtype != dtype:
            raise TypeError(f"{name} must be {dtype}, got {x.dtype}")
        if not x.is_cuda:
            raise RuntimeError(f"{name} must live on a HIP device (there is no CPU path)")
        return C.c_void_p(x.data_ptr())

    def buffer_device(self, device):
        import torch
        return torch.device(f"cuda:{int(device)}")

    def zeros(self, shape, np_dtype, device):
        """zero-filled scratch / result buffer on ``device`` (numpy dtype names the element type)"""
        import torch
        tdt = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32, "uint32": torch.int32,
               "int64": torch.int64, "uint8": torch.uint8}[np.dtype(np_dtype).name]
        return torch.zeros(tuple(shape), dtype=tdt, device=self.buffer_device(device))

    def upload(self, a, torch_dtype, device):
        """host array or tensor -> contiguous device tensor of ``torch_dtype``"""
        import torch
        if isinstance(a, torch.Tensor):
            if a.dtype != torch_dtype:
                a = a.to(torch_dtype)
            return a.contiguous() if a.is_cuda else a.contiguous().to(self.buffer_device(device))
        return torch.from_numpy(np.ascontiguousarray(a)).to(torch_dtype).to(self.buffer_device(device))

    def download(self, t):
        """device buffer -> numpy"""
        return t.detach().cpu().numpy()

    def current_stream(self, device):
        """the stream torch is enqueueing on for ``device``: calls of the C ABI that touch buffers torch just created (zero
        fills) or that must follow work the caller queued go on it, not on the NULL stream"""
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.buffer_device(device)).cuda_stream)


MEMORY = DeviceMemory()


def get() -> C.CDLL:
    """The HIP library, loaded once.  Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -m gs2mesh_amd.build`; there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever
        # is loaded first serves both; torch must be that one, or the device torch allocated on is invisible to the
        # runtime this library would otherwise bind to ("no ROCm-capable device is detected").
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is the memory/stream plumbing; the C ABI itself does not need it
            pass
        lib = bind(C.CDLL(LIB_PATH))
        if lib.gs2m_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} is ABI version {lib.gs2m_version()}, this binding needs {ABI_VERSION}: "
                               "rebuild it (python -m gs2mesh_amd.build)")
        _LIB = lib
    return _LIB


def check(rc: int, lib: C.CDLL | None = None):
    if rc != 0:
        lib = lib or get()
        raise RuntimeError("gs2mesh_amd: " + lib.gs2m_last_error().decode(errors="replace"))
