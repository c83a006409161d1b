"""Test back-ends for the C ABI.

  * ``gpu``  -- the product: libgs2mesh_amd.so on a real MI355X, torch tensors on cuda:0.
  * ``emu``  -- the SAME kernel sources compiled against the CPU fiber emulator
                (tests/emu), numpy arrays.  Test infrastructure only: lets kernel logic be
                checked against the oracle in the GPU-less build container.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

from gs2mesh_amd import _lib  # noqa: E402


class HostMemory(_lib.DeviceMemory):
    """Memory policy of the EMULATOR harness: the emulator build of the kernels runs on the CPU, so its "device"
    pointers are host pointers (numpy arrays, CPU torch tensors).  Lives in the test tree; the product's policy
    (``_lib.DeviceMemory``) refuses both."""

    def ptr(self, x, dtype=None, name="tensor"):
        import torch
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            if x.numel() == 0:
                return None
            if not x.is_contiguous():
                raise ValueError(f"{name} must be contiguous")
            if dtype is not None and x.dtype != dtype:
                raise TypeError(f"{name} must be {dtype}, got {x.dtype}")
            return C.c_void_p(x.data_ptr())
        if isinstance(x, np.ndarray):
            if x.size == 0:
                return None
            if not x.flags["C_CONTIGUOUS"]:
                raise ValueError(f"{name} must be contiguous")
            return C.c_void_p(x.ctypes.data)
        raise TypeError(f"{name}: unsupported type {type(x)}")

    def buffer_device(self, device):
        import torch
        return torch.device("cpu")

    def zeros(self, shape, np_dtype, device):
        return np.zeros(tuple(shape), np_dtype)

    def upload(self, a, torch_dtype, device):
        import torch
        if isinstance(a, torch.Tensor):
            return (a if a.dtype == torch_dtype else a.to(torch_dtype)).contiguous()
        np_dtype = {torch.float32: np.float32, torch.uint8: np.uint8, torch.int32: np.int32, torch.int64: np.int64}[torch_dtype]
        return np.ascontiguousarray(a).astype(np_dtype, copy=False)

    def download(self, t):
        return np.asarray(t)

    def current_stream(self, device):
        return C.c_void_p(0)


def use_host_memory(on: bool):
    _lib.MEMORY = HostMemory() if on else _lib.DeviceMemory()


class EmuBackend:
    name = "emu"
    _cached = None

    def __init__(self):
        if EmuBackend._cached is None:
            import build_emu
            EmuBackend._cached = _lib.bind(C.CDLL(build_emu.build()), require_all=False)
        self.lib = EmuBackend._cached
        use_host_memory(True)

    def dev(self, a):
        return None if a is None else np.ascontiguousarray(a)

    def host(self, a):
        return None if a is None else np.asarray(a)

    def sync(self):
        pass


class GpuBackend:
    name = "gpu"

    def __init__(self):
        import torch
        if not torch.cuda.is_available():
            pytest.skip("no GPU visible")
        self.torch = torch
        self.lib = _lib.get()
        use_host_memory(False)

    def dev(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        return self.torch.from_numpy(a).cuda()

    def host(self, a):
        return None if a is None else a.detach().cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


def make(name):
    return EmuBackend() if name == "emu" else GpuBackend()


BACKENDS = [pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)]
