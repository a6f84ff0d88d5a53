"""DeviceArray: a C-contiguous tile living in MI355X HBM.

It quacks enough like an ndarray (``shape``, ``dtype``, ``ndim``, ``size``,
``__array__``) for dict-of-arrays raster responses to flow between ``process``
calls without leaving the device; ``np.asarray(x)`` / ``x.to_numpy()`` materialise.
Buffers come from the library's caching allocator and return to it when the Python
object dies (all work of a device is ordered on one stream, so reuse is safe).
"""
import ctypes

import numpy as np

from . import runtime
from .dtypes import code_of


class DeviceArray:
    __slots__ = ("ptr", "shape", "dtype", "device", "_owner", "__weakref__")

    def __init__(self, ptr, shape, dtype, device, owner=None):
        self.ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = device
        self._owner = owner  # keeps a parent allocation alive for views

    # -- construction ---------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype):
        runtime.require_device()
        dtype = np.dtype(dtype)
        code_of(dtype)
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = runtime.malloc(max(nbytes, 1))
        return cls(ptr, shape, dtype, runtime.get_device())

    @classmethod
    def from_numpy(cls, array):
        array = np.ascontiguousarray(array)
        out = cls.empty(array.shape, array.dtype)
        if array.nbytes:
            runtime.check(runtime.lib().gm_h2d(ctypes.c_void_p(out.ptr),
                                               array.ctypes.data_as(ctypes.c_void_p),
                                               array.nbytes), "gm_h2d")
        return out

    # -- ndarray-ish surface ----------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def to_numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            runtime.check(runtime.lib().gm_d2h(out.ctypes.data_as(ctypes.c_void_p),
                                               ctypes.c_void_p(self.ptr), out.nbytes), "gm_d2h")
        return out

    def __array__(self, dtype=None, copy=None):
        arr = self.to_numpy()
        return arr if dtype is None else arr.astype(dtype, copy=False)

    def band_slice(self, start, stop):
        """View of bands [start, stop) of a (bands, H, W) tile (no copy)."""
        start, stop = int(start), int(stop)
        plane = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        return DeviceArray(self.ptr + start * plane, (stop - start,) + self.shape[1:], self.dtype,
                           self.device, owner=self)

    def reshape(self, *shape):
        if len(shape) == 1 and hasattr(shape[0], "__iter__"):
            shape = tuple(shape[0])
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError("cannot reshape %s into %s" % (self.shape, shape))
        return DeviceArray(self.ptr, shape, self.dtype, self.device, owner=self)

    def checksum(self):
        out = ctypes.c_uint64()
        runtime.check(runtime.lib().gm_checksum(ctypes.c_void_p(self.ptr), self.nbytes,
                                                ctypes.byref(out)), "gm_checksum")
        return out.value

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%d)" % (self.shape, self.dtype, self.device)

    def __del__(self):
        try:
            if self._owner is None and self.ptr:
                runtime.free(self.ptr)
        except Exception:
            pass


def as_device(values):
    """ndarray -> DeviceArray upload; DeviceArray passes through."""
    if isinstance(values, DeviceArray):
        return values
    return DeviceArray.from_numpy(np.asarray(values))
