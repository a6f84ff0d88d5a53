"""DeviceArray: a tile living in MI355X HBM -- C-contiguous, or a zero-copy window of a larger resident raster:
rows ``pitch`` cells apart, and for a window of several bands (round 4: the frames of a series) bands ``band_pitch``
cells apart.

It quacks enough like an ndarray (``shape``, ``dtype``, ``ndim``, ``size``,
``__array__``) for dict-of-arrays raster responses to flow between ``process``
calls without leaving the device; ``np.asarray(x)`` / ``x.to_numpy()`` materialise.
Buffers come from the library's caching allocator and return to it when the Python
object dies (all work of a device is ordered on one stream, so reuse is safe).
"""
import ctypes

import numpy as np
from math import prod as _prod

from . import runtime
from .dtypes import code_of


class DeviceArray:
    __slots__ = ("ptr", "shape", "dtype", "device", "_owner", "pitch", "band_pitch", "resident", "__weakref__")

    def __init__(self, ptr, shape, dtype, device, owner=None, pitch=None, band_pitch=None):
        self.ptr = ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = device
        self._owner = owner  # keeps a parent allocation alive for views
        # a source's answer that IS resident data (a source raster on the device, a zero-copy window of one, a kept procedural
        # window): the same bytes for the same request every time, so a kept evaluation (core/graphs.py PreparedGraph) may keep
        # the answer instead of asking again.  Never set on anything a kernel computed for the request.
        self.resident = False
        # row pitch in elements of a window into a wider raster; None = C-contiguous
        self.pitch = None if pitch is None or int(pitch) == self.shape[-1] else int(pitch)
        # cells between the bands of a window of several bands (the parent's plane); None: one band, or C-contiguous
        self.band_pitch = int(band_pitch) if band_pitch is not None and self.shape[0] > 1 and (
            self.pitch is not None or int(band_pitch) != _prod(self.shape[1:])) else None
        if self.band_pitch is not None and self.pitch is None:
            self.pitch = self.shape[-1]  # (whole rows, but planes apart: still a window)

    # -- construction ---------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype):
        runtime.require_device()
        dtype = np.dtype(dtype)
        code_of(dtype)
        shape = tuple(int(s) for s in shape)
        nbytes = _prod(shape) * dtype.itemsize
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
        return _prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def is_contiguous(self):
        return self.pitch is None

    def window(self, row0, col0, height, width):
        """Zero-copy view of rows [row0, row0+height) x columns [col0, col0+width) of a (bands, H, W) tile: same allocation,
        row pitch = the parent's; several bands stay a plane of the parent apart (``band_pitch``)."""
        if len(self.shape) != 3:
            raise ValueError("window() needs a (bands, H, W) tile, got %s" % (self.shape,))
        bands, H, W = self.shape
        if bands != 1 and self.pitch is not None:
            raise ValueError("window() of a window of several bands")
        row0, col0, height, width = int(row0), int(col0), int(height), int(width)
        if not (0 <= row0 and 0 <= col0 and row0 + height <= H and col0 + width <= W):
            raise ValueError("window outside the tile")
        pitch = self.pitch or W
        ptr = self.ptr + (row0 * pitch + col0) * self.dtype.itemsize
        return DeviceArray(ptr, (bands, height, width), self.dtype, self.device, owner=self, pitch=pitch,
                           band_pitch=H * W if bands > 1 else None)

    def room_right(self):
        """Columns of the parent raster to the right of this zero-copy window, in the window's rows (0 for a tile that
        is not a window): how far a kernel may read past the window's last column."""
        if self.pitch is None:
            return 0
        root = self
        while root._owner is not None and isinstance(root._owner, DeviceArray):
            root = root._owner
        if root.pitch is not None or len(root.shape) != 3 or root.shape[-1] != self.pitch:
            return 0
        first = (self.ptr - root.ptr) // self.dtype.itemsize
        return root.shape[-1] - (first % root.shape[-1]) - self.shape[-1]

    def contiguous(self):
        """This tile if it is C-contiguous, else a compacted copy (one 2-D device copy)."""
        if self.pitch is None:
            return self
        out = DeviceArray.empty(self.shape, self.dtype)
        item = self.dtype.itemsize
        bands, h, w = self.shape
        L = runtime.lib()
        L.gm_copy_2d.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_size_t, ctypes.c_size_t]
        if self.band_pitch is None:
            runtime.check(L.gm_copy_2d(ctypes.c_void_p(out.ptr), w * item, ctypes.c_void_p(self.ptr), self.pitch * item,
                                       w * item, h), "gm_copy_2d")
        else:  # (the frames of a series inside a tile of its raster: planes of band_pitch / pitch rows, one copy)
            L.gm_copy_3d.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
            runtime.check(L.gm_copy_3d(ctypes.c_void_p(out.ptr), w * item, h, ctypes.c_void_p(self.ptr), self.pitch * item,
                                       self.band_pitch // self.pitch, w * item, h, bands), "gm_copy_3d")
        return out

    def to_numpy(self):
        if self.pitch is not None:
            return self.contiguous().to_numpy()
        # results of 1 MiB and more land in pinned host memory: 22 ms -> 1.4 ms for a 4096^2 float32 tile
        out = (runtime.pinned_empty(self.shape, self.dtype) if self.nbytes >= (1 << 20)
               else np.empty(self.shape, dtype=self.dtype))
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
        if not 0 <= start <= stop <= self.shape[0]:
            raise IndexError("bands [%d, %d) of a tile of %d" % (start, stop, self.shape[0]))
        if self.pitch is not None:
            if self.shape[0] == 1:
                if (start, stop) == (0, 1):
                    return self
                raise ValueError("band_slice() of a strided window")
            # bands of a window of several: windows themselves
            ptr = self.ptr + start * self.band_pitch * self.dtype.itemsize
            return DeviceArray(ptr, (stop - start,) + self.shape[1:], self.dtype, self.device, owner=self, pitch=self.pitch,
                               band_pitch=self.band_pitch)
        plane = _prod(self.shape[1:]) * self.dtype.itemsize
        return DeviceArray(self.ptr + start * plane, (stop - start,) + self.shape[1:], self.dtype,
                           self.device, owner=self)

    def reshape(self, *shape):
        if len(shape) == 1 and hasattr(shape[0], "__iter__"):
            shape = tuple(shape[0])
        if _prod(shape) != self.size:
            raise ValueError("cannot reshape %s into %s" % (self.shape, shape))
        if self.pitch is not None:
            return self.contiguous().reshape(shape)
        return DeviceArray(self.ptr, shape, self.dtype, self.device, owner=self)

    def checksum(self):
        if self.pitch is not None:
            return self.contiguous().checksum()
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
    """ndarray -> DeviceArray upload; a DeviceArray passes through, compacted if it is a strided
    window (what every kernel except the element-wise program kernel needs)."""
    if isinstance(values, DeviceArray):
        return values.contiguous()
    return DeviceArray.from_numpy(np.asarray(values))


def as_device_view(values):
    """Like as_device, but strided windows stay zero-copy (for pitch-aware consumers)."""
    if isinstance(values, DeviceArray):
        return values
    return DeviceArray.from_numpy(np.asarray(values))
