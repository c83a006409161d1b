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


class EmuBackend:
    name = "emu"
    _cached = None

    def __init__(self):
        if EmuBackend._cached is None:
            import build_emu
            EmuBackend._cached = _lib.bind(C.CDLL(build_emu.build()), require_all=False)
        self.lib = EmuBackend._cached
        _lib.ALLOW_HOST_POINTERS = True

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
        _lib.ALLOW_HOST_POINTERS = False

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
