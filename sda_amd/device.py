"""Minimal device-buffer helper over the C ABI's sda_dev_* functions, so callers (tests, bench, a
host language without its own HIP binding) can keep share matrices resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import check


class DeviceBuffer:
    """`count` int64 elements of HBM."""

    def __init__(self, count: int):
        self._lib = capi.load()
        self.count = int(count)
        self._p = C.c_void_p()
        check(self._lib.sda_dev_malloc(C.byref(self._p), max(self.count, 1) * 8))

    @property
    def ptr(self) -> int:
        return self._p.value

    def at(self, element_offset: int) -> int:
        return self._p.value + 8 * int(element_offset)

    @classmethod
    def from_numpy(cls, a) -> "DeviceBuffer":
        a = np.ascontiguousarray(a, dtype=np.int64)
        b = cls(a.size)
        check(b._lib.sda_dev_upload(b._p, a.ctypes.data_as(C.c_void_p), a.size * 8))
        return b

    def zero(self) -> "DeviceBuffer":
        check(self._lib.sda_dev_memset(self._p, 0, self.count * 8))
        return self

    def to_numpy(self, count: int | None = None, offset: int = 0) -> np.ndarray:
        n = self.count - offset if count is None else count
        out = np.empty(n, dtype=np.int64)
        check(self._lib.sda_dev_synchronize())
        check(self._lib.sda_dev_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.at(offset)), n * 8))
        return out

    def free(self):
        if self._p is not None and self._p.value:
            self._lib.sda_dev_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBytes:
    """`nbytes` raw bytes of HBM (wire-format buffers: containers, base64 text, sealed boxes)."""

    def __init__(self, nbytes: int):
        self._lib = capi.load()
        self.nbytes = int(nbytes)
        self._p = C.c_void_p()
        check(self._lib.sda_dev_malloc(C.byref(self._p), max(self.nbytes, 16)))

    @property
    def ptr(self) -> int:
        return self._p.value

    @classmethod
    def from_bytes(cls, raw) -> "DeviceBytes":
        a = np.frombuffer(bytes(raw), dtype=np.uint8)
        b = cls(a.size)
        if a.size:
            check(b._lib.sda_dev_upload(b._p, a.ctypes.data_as(C.c_void_p), a.size))
        return b

    def zero(self) -> "DeviceBytes":
        check(self._lib.sda_dev_memset(self._p, 0, max(self.nbytes, 1)))
        return self

    def to_bytes(self, count: int | None = None, offset: int = 0) -> bytes:
        n = self.nbytes - offset if count is None else count
        out = np.empty(max(n, 1), dtype=np.uint8)
        check(self._lib.sda_dev_synchronize())
        if n:
            check(self._lib.sda_dev_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(self._p.value + offset), n))
        return out[:n].tobytes()

    def free(self):
        if self._p is not None and self._p.value:
            self._lib.sda_dev_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synchronize():
    check(capi.load().sda_dev_synchronize())
