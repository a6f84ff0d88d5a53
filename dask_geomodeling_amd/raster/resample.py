"""Host wrappers of the resampling kernels K6/K7 (csrc/gm_resample.hip)."""
import ctypes

import numpy as np

from ..hip import runtime
from ..hip.array import DeviceArray, as_device, as_device_view
from ..hip.dtypes import code_of

_declared = False


def _lib():
    global _declared
    L = runtime.lib()
    if not _declared:
        c = ctypes
        dp = c.POINTER(c.c_double)
        L.gm_resample_nearest.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int,
                                          c.c_int, c.c_int, dp, dp, c.c_double, c.c_uint64, c.c_int]
        L.gm_resample_nearest.restype = c.c_int
        L.gm_resample_bilinear.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int,
                                           c.c_int, c.c_int, c.c_uint64]
        L.gm_resample_bilinear.restype = c.c_int
        L.gm_resample_bilinear_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int,
                                                  c.c_int, c.c_int, c.c_uint64]
        L.gm_resample_bilinear_window.restype = c.c_int
        _declared = True
    return L


def _params(values):
    return (ctypes.c_double * 4)(*[float(v) for v in values])


def nearest_warp(source, src_geo_transform, bbox, height, width, fill, finite_to_fill=True):
    """Nearest-neighbour warp of a (bands, H, W) device tile onto the grid
    ``GeoTransform.from_bbox(bbox, height, width)`` -- the 'vals' branch of
    RasterSourceBase.process (raster/sources.py:119-149) for same-CRS requests."""
    src = as_device(source)
    bands, hs, ws = src.shape
    p, a, _, q, _, d = src_geo_transform
    x1, y1, x2, y2 = bbox
    ta = (x2 - x1) / width       # target pixel width  (utils.py:228-235)
    td = (y1 - y2) / height      # target pixel height (negative: rows run north -> south)
    out = DeviceArray.empty((bands, height, width), src.dtype)
    runtime.check(_lib().gm_resample_nearest(
        ctypes.c_void_p(src.ptr), code_of(src.dtype), bands, hs, ws, ctypes.c_void_p(out.ptr), height, width, 0,
        _params((y2, td, q, d)), _params((x1, ta, p, a)), 1e-10,
        runtime.scalar_bits(fill, src.dtype), 1 if (finite_to_fill and src.dtype.kind == "f") else 0),
        "gm_resample_nearest")
    return out


def affine_nearest(source, zoom_yx, offset_yx, cval=0.0):
    """scipy.ndimage.affine_transform(order=0, matrix=diag(1, zy, zx), offset=(0, oy, ox)) with
    the input's shape as output shape (Smooth's zoom branch, raster/spatial.py:296-305)."""
    src = as_device(source)
    bands, hs, ws = src.shape
    out = DeviceArray.empty(src.shape, src.dtype)
    runtime.check(_lib().gm_resample_nearest(
        ctypes.c_void_p(src.ptr), code_of(src.dtype), bands, hs, ws, ctypes.c_void_p(out.ptr), hs, ws, 1,
        _params((zoom_yx[0], offset_yx[0], 0, 0)), _params((zoom_yx[1], offset_yx[1], 0, 0)), 0.0,
        runtime.scalar_bits(cval, src.dtype), 0), "gm_resample_nearest")
    return out


def bilinear(source, out_height, out_width, no_data_value):
    """Bilinear resize of a float (bands, H, W) tile (our spec)."""
    src = as_device_view(source)  # (a window of a resident raster is read where it lies)
    if src.band_pitch is not None:
        src = src.contiguous()
    if src.dtype.kind != "f":
        raise TypeError("bilinear resampling needs a float32/float64 tile, got %s" % src.dtype)
    bands, hs, ws = src.shape
    out = DeviceArray.empty((bands, int(out_height), int(out_width)), src.dtype)
    has = no_data_value is not None
    ndv = src.dtype.type(no_data_value) if has else 0.0
    runtime.check(_lib().gm_resample_bilinear_window(
        ctypes.c_void_p(src.ptr), code_of(src.dtype), bands, hs, ws, src.pitch or ws, ctypes.c_void_p(out.ptr),
        int(out_height), int(out_width), 1 if has else 0, runtime.scalar_bits(ndv, src.dtype)),
        "gm_resample_bilinear")
    return out
