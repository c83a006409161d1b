"""HIP streams restricted to a subset of the compute units (``gs2m_stream_create``, hipExtStreamCreateWithCUMask).

The compositing kernel of the rasteriser is VALU-bound and its grid refills every slot a retiring workgroup frees; the
binning and TSDF kernels of the following views are latency-bound one-workgroup-per-CU kernels with large LDS footprints
that then only get placed in its tail.  Keeping the compositing grid off a few CUs (an XCD- and shader-engine-symmetric
set) leaves those kernels CUs of their own.  torch only enters as the wrapper (`ExternalStream`) that lets tensors,
events and `torch.cuda.stream(...)` contexts use the stream.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def cu_mask_words(first: int, count: int, total: int = 256):
    """Bits [first, first + count) of a ``total``-CU mask as uint32 words.  The driver interleaves the XCDs over the bit
    index (bit i -> XCD i % 8, then round-robin over its shader engines), so a run of bits whose start and length are
    multiples of 32 is symmetric over the 8 XCDs x 4 shader engines of an MI355X."""
    if count <= 0 or first < 0 or first + count > total:
        raise ValueError(f"CU range [{first}, {first + count}) outside 0..{total}")
    words = [0] * ((total + 31) // 32)
    for i in range(first, first + count):
        words[i // 32] |= 1 << (i % 32)
    return words


class MaskedStream:
    """A HIP stream created through the C ABI: ``cus=(first, count)`` -> CU-masked, ``cus=None`` -> plain non-blocking
    stream.  ``.handle`` is the hipStream_t as int, ``.torch`` the torch view of it."""

    def __init__(self, device: int = 0, cus=None, total_cus: int | None = None, lib=None):
        self._lib = lib or _lib.get()
        self.device = int(device)
        h = C.c_void_p()
        if cus is None:
            _lib.check(self._lib.gs2m_stream_create(C.byref(h), self.device, None, 0), self._lib)
        else:
            if total_cus is None:
                total_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            w = cu_mask_words(int(cus[0]), int(cus[1]), int(total_cus))
            arr = (C.c_uint32 * len(w))(*w)
            _lib.check(self._lib.gs2m_stream_create(C.byref(h), self.device, arr, len(w)), self._lib)
        self.cus = cus
        self.handle = int(h.value)
        self.torch = torch.cuda.ExternalStream(self.handle, device=torch.device(f"cuda:{self.device}")) if torch is not None else None

    def close(self):
        """Drain the stream and hand it back to the per-process pool (`acquire`).  Streams are never destroyed while the
        process lives: torch's caching allocator may still hold tensors whose `record_stream` named this stream, and it
        records an event on it when they are freed."""
        if getattr(self, "handle", 0) and not self._pooled:
            if self.torch is not None:
                self.torch.synchronize()
            self._pooled = True
            _POOL.setdefault((self.device, self.cus), []).append(self)

    _pooled = False


_POOL: dict = {}


def acquire(device: int = 0, cus=None, total_cus: int | None = None, lib=None) -> MaskedStream:
    """A (possibly recycled) stream for ``cus`` (None = plain); release it with ``.close()``."""
    free = _POOL.get((int(device), cus))
    if free:
        m = free.pop()
        m._pooled = False
        return m
    return MaskedStream(device, cus, total_cus, lib=lib)
