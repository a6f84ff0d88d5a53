"""Host wrappers of the stencil kernels (csrc/gm_stencil.hip): the device bodies of
Dilate.process, MovingMax.process and HillShade.process (raster/spatial.py)."""
import ctypes
import functools

import numpy as np

from ..hip import runtime
from ..hip.array import DeviceArray, as_device
from ..hip.array import as_device_view as _any_view
from ..hip.dtypes import code_of

_declared = False


def as_device_view(values):
    """The tile as the kernels of this file read it: a single-band window where it lies, anything of several bands compact."""
    view = _any_view(values)
    return view.contiguous() if view.band_pitch is not None else view


def _lib():
    global _declared
    L = runtime.lib()
    if not _declared:
        c = ctypes
        L.gm_dilate.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_uint64), c.c_int, c.c_void_p]
        L.gm_dilate.restype = c.c_int
        L.gm_moving_max.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_void_p]
        L.gm_moving_max.restype = c.c_int
        L.gm_hillshade.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_uint64,
                                   c.c_double, c.c_double, c.c_double, c.c_double, c.c_void_p]
        L.gm_hillshade.restype = c.c_int
        L.gm_gauss_axis.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double),
                                    c.c_int, c.c_double, c.c_int, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int,
                                    c.c_int, c.c_void_p]
        L.gm_gauss_axis.restype = c.c_int
        L.gm_gauss_axis_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double),
                                           c.c_int, c.c_double, c.c_int, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int,
                                           c.c_int, c.c_void_p]
        L.gm_gauss_axis_window.restype = c.c_int
        L.gm_gauss_2d.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double), c.POINTER(c.c_double),
                                  c.c_int, c.c_double, c.c_int, c.c_uint64, c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_int,
                                  c.c_void_p]
        L.gm_gauss_2d.restype = c.c_int
        L.gm_dilate_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_uint64), c.c_int,
                                       c.c_void_p]
        L.gm_moving_max_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_uint64,
                                           c.c_void_p]
        L.gm_hillshade_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_uint64, c.c_uint64,
                                          c.c_double, c.c_double, c.c_double, c.c_double, c.c_void_p]
        L.gm_gauss_2d_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double),
                                         c.POINTER(c.c_double), c.c_int, c.c_double, c.c_int, c.c_uint64, c.c_uint64, c.c_int,
                                         c.c_int, c.c_int, c.c_int, c.c_void_p]
        L.gm_dilate_into.argtypes = L.gm_dilate_window.argtypes + [c.c_int]
        L.gm_moving_max_into.argtypes = L.gm_moving_max_window.argtypes + [c.c_int]
        L.gm_gauss_2d_into.argtypes = L.gm_gauss_2d_window.argtypes + [c.c_int]
        for fn in (L.gm_dilate_window, L.gm_moving_max_window, L.gm_hillshade_window, L.gm_gauss_2d_window, L.gm_dilate_into,
                   L.gm_moving_max_into, L.gm_gauss_2d_into):
            fn.restype = c.c_int
        _declared = True
    return L


MAX_DILATE_VALUES = 16
MAX_FUSED_RADIUS = 16  # gm_gauss_2d (both axes of a float32 Smooth in one launch)


def _ndv_bits(dtype, ndv):
    """(has_ndv, payload): can a cell of a ``dtype`` tile ever equal ``ndv`` (NumPy's ==)?"""
    from ..hip.program import effective_ndv
    eff = effective_ndv(dtype, ndv)
    if eff is None:
        return 0, 0
    return 1, runtime.scalar_bits(eff, dtype)


GM_ERR_UNSUPPORTED = 9002


def _into(out, shape, dtype):
    """``out`` when it can take the result as it is -- a single band of that shape and dtype, possibly a window of a
    mosaic (row pitch) -- else None."""
    if out is None or tuple(out.shape) != tuple(shape) or out.dtype != np.dtype(dtype) or shape[0] != 1:
        return None
    return out


def _launch_into(src, out, plain, into, what):
    """The kernel on ``src`` (a tile or zero-copy window) with the result in ``out``: ``into(tile, pitch, out_pitch)``
    when ``out`` has a row pitch of its own and the library's route for this tile can write one, else
    ``plain(tile, pitch)`` into a compact tile (-> that tile: the caller's ``out`` was not used)."""
    if out.pitch is not None:
        rc = into(src, src.pitch or src.shape[-1], out.pitch)
        if rc != GM_ERR_UNSUPPORTED:
            runtime.check(rc, what)
            return out
        out = DeviceArray.empty(out.shape, out.dtype)  # this route writes compact tiles only
    _on_window(src, lambda tile, pitch: plain(tile, pitch, out), what)
    return out


def _on_window(src, launch, what):
    """``launch(tile, pitch)`` on a zero-copy window as it is; when the library wants a compact tile for this
    dtype / shape (GM_ERR_UNSUPPORTED), on a compacted copy.  Compact tiles go straight through."""
    rc = launch(src, src.pitch or src.shape[-1])
    if rc == GM_ERR_UNSUPPORTED and src.pitch is not None:
        src = src.contiguous()
        rc = launch(src, src.shape[-1])
    runtime.check(rc, what)


def _cast_tile(src, dtype):
    """The tile in another dtype (a COPY program: NumPy's astype)."""
    from ..hip import program as P
    values, _ = P.evaluate(P.Op("copy", [P.Leaf(None, src.dtype, None, values=src)], np.dtype(dtype), None))
    return values


def dilate(values, dilate_values, out=None):
    """(bands, h, w) tile with a 1-pixel margin -> dilated (bands, h-2, w-2) tile.  ``out``: where the result may be
    written (e.g. the tile's window of a mosaic); what is returned is the tile that holds the result."""
    src = as_device_view(values)
    bands, h, w = src.shape
    vals = np.asarray(dilate_values, dtype=src.dtype)  # the reference's cast (spatial.py:151)
    if h < 3 or w < 3 or bands == 0:  # (no frame in the requested window: an empty stack goes in and comes out)
        return DeviceArray.empty((bands, max(h - 2, 0), max(w - 2, 0)), src.dtype)
    shape = (bands, h - 2, w - 2)
    out = _into(out, shape, src.dtype) or DeviceArray.empty(shape, src.dtype)
    todo = list(vals)
    # (any number of values, spatial.py:147-154: beyond what one launch carries the library applies them group by
    #  group, every group testing the original tile)
    bits = (ctypes.c_uint64 * max(len(todo), 1))(*[runtime.scalar_bits(v, src.dtype) for v in todo])
    L = _lib()
    code = code_of(src.dtype)
    return _launch_into(
        src, out,
        lambda tile, pitch, o: L.gm_dilate_window(ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, bits, len(todo),
                                                  ctypes.c_void_p(o.ptr)),
        lambda tile, pitch, opitch: L.gm_dilate_into(ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, bits, len(todo),
                                                     ctypes.c_void_p(out.ptr), opitch), "gm_dilate")


def moving_max(values, no_data_value, size, out=None):
    src = as_device_view(values)
    bands, h, w = src.shape
    radius = int(size // 2)
    if src.dtype == np.dtype("bool"):
        # the reference fails here too: utils.get_dtype_min has no minimum for bool (spatial.py:198)
        raise ValueError("Invalid integer data type 'b'.")
    shape = (bands, max(h - 2 * radius, 0), max(w - 2 * radius, 0))
    if 0 in shape:  # (no frame in the window, or a tile that is all margin -- a tile of no columns at the edge of a tiled request:
        return DeviceArray.empty(shape, src.dtype)  # the reference's filter + crop leaves an empty array too; round-5 fuzzing)
    out = _into(out, shape, src.dtype) or DeviceArray.empty(shape, src.dtype)
    has, bits = _ndv_bits(src.dtype, no_data_value)
    L = _lib()
    code = code_of(src.dtype)
    return _launch_into(
        src, out,
        lambda tile, pitch, o: L.gm_moving_max_window(ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, int(size), has, bits,
                                                      ctypes.c_void_p(o.ptr)),
        lambda tile, pitch, opitch: L.gm_moving_max_into(ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, int(size), has,
                                                         bits, ctypes.c_void_p(out.ptr), opitch), "gm_moving_max")


def hillshade(values, no_data_value, fill, resolution, altitude, azimuth):
    src = as_device_view(values)
    bands, h, w = src.shape
    with np.errstate(all="ignore"):
        fill_cast = np.array(fill).astype(src.dtype)[()]  # array[mask] = fill casts like this (spatial.py:364)
    if src.dtype == np.dtype("bool"):
        # `2 * array` of a bool array is int64, and so are the Sobel sums (spatial.py:384-404): with cells of 0 / 1
        # they lie in [-4, 4], so the int16 kernel computes the very same integers (no wrap-around to reproduce)
        src = _cast_tile(src, "i2")
        fill_cast = np.int16(bool(fill_cast))
        if no_data_value is not None:
            no_data_value = np.int16(bool(no_data_value))
    out = DeviceArray.empty((bands, max(h - 2, 0), max(w - 2, 0)), "u1")
    if 0 in out.shape:  # (as in moving_max)
        return out
    has, bits = _ndv_bits(src.dtype, no_data_value)
    L = _lib()
    fill_bits = runtime.scalar_bits(fill_cast, src.dtype)
    _on_window(src, lambda tile, pitch: L.gm_hillshade_window(ctypes.c_void_p(tile.ptr), code_of(src.dtype), bands, h, w, pitch, has,
                                                             bits, fill_bits, float(resolution[0]), float(resolution[1]),
                                                             float(altitude), float(azimuth), ctypes.c_void_p(out.ptr)),
               "gm_hillshade")
    return out


def hillshade_batch(tiles, no_data_value, fill, resolution, altitude, azimuth, dest=None):
    """HillShade of the tiles of one mosaic in one launch per tile shape (gm_hillshade_batch; a ragged mosaic has up to
    four shapes: interior, last column, top row, corner): ``tiles`` = single-band device tiles of one dtype (zero-copy
    windows welcome); ``dest(i, shape, dtype)`` may hand out the DeviceArray window the shades of tile i are to be written
    into.  -> list of DeviceArrays, or None when the library wants the tiles one by one (dtype / resolution route)."""
    if not tiles:
        return []
    srcs = [as_device_view(t) for t in tiles]
    first = srcs[0]
    if first.dtype == np.dtype("bool") or any(x.dtype != first.dtype or x.shape[0] != 1 or x.shape[1] < 3 or x.shape[2] < 3
                                                for x in srcs):
        return None
    with np.errstate(all="ignore"):
        fill_cast = np.array(fill).astype(first.dtype)[()]
    has, bits = _ndv_bits(first.dtype, no_data_value)
    fill_bits = runtime.scalar_bits(fill_cast, first.dtype)
    L = _lib()
    c = ctypes
    L.gm_hillshade_batch.argtypes = [c.c_int, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_uint64, c.c_uint64,
                                     c.c_double, c.c_double, c.c_double, c.c_double, c.c_void_p, c.c_void_p]
    L.gm_hillshade_batch.restype = c.c_int
    groups = {}
    for i, x in enumerate(srcs):
        groups.setdefault(x.shape, []).append(i)
    outs = [None] * len(srcs)
    for (_, h, w), members in groups.items():
        shape = (1, h - 2, w - 2)
        for i in members:
            out = dest(i, shape, np.dtype("u1")) if dest is not None else None
            outs[i] = out if out is not None else DeviceArray.empty(shape, "u1")
        n = len(members)
        rc = L.gm_hillshade_batch(n, (c.c_void_p * n)(*[srcs[i].ptr for i in members]), code_of(first.dtype), h, w,
                                  (c.c_int * n)(*[srcs[i].pitch or w for i in members]), has, bits, fill_bits,
                                  float(resolution[0]), float(resolution[1]), float(altitude), float(azimuth),
                                  (c.c_void_p * n)(*[outs[i].ptr for i in members]),
                                  (c.c_int * n)(*[outs[i].pitch or (w - 2) for i in members]))
        if rc == GM_ERR_UNSUPPORTED:
            return None
        runtime.check(rc, "gm_hillshade_batch")
    return outs


_IDENTITY_WEIGHTS = (np.ones(1, dtype=np.float64), 0)


@functools.lru_cache(maxsize=256)
def gaussian_weights(sigma, truncate=4.0):
    """SciPy's Gaussian kernel (scipy.ndimage._filters._gaussian_kernel1d, order 0): weights by
    distance 0..radius, radius = int(truncate * sigma + 0.5); normalised over the full kernel."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    weights = np.ascontiguousarray(phi[radius:], dtype=np.float64)
    weights.setflags(write=False)  # (remembered per sigma: a Smooth block asks for the same kernel all day)
    return weights, radius


def gaussian_smooth(values, no_data_value, fill, sigma_yx, crop_yx=(0, 0), fused=True, out=None):
    """values[values == ndv] = fill, then ndimage.gaussian_filter(values, (0, sy, sx), mode="constant",
    cval=fill) axis by axis (the second axis reads the first one's result in the tile's dtype), the
    result cropped by ``crop_yx`` pixels on every side."""
    src = as_device_view(values)
    bands, h, w = src.shape
    has, ndv_bits = _ndv_bits(src.dtype, no_data_value)
    with np.errstate(all="ignore"):
        fill_cast = np.array(fill).astype(src.dtype)[()]
    fill_bits = runtime.scalar_bits(fill_cast, src.dtype)
    axes = [(axis, sigma) for axis, sigma in ((1, sigma_yx[0]), (2, sigma_yx[1])) if sigma > 1e-15]
    if not axes:  # nothing to smooth: one pass with the identity kernel does the no-data replacement and the crop
        axes = [(2, None)]
    my, mx = crop_yx
    if bands == 0 or h - 2 * my <= 0 or w - 2 * mx <= 0:  # (as in moving_max)
        return DeviceArray.empty((bands, max(h - 2 * my, 0), max(w - 2 * mx, 0)), src.dtype)
    L = _lib()
    if fused and len(axes) == 2 and src.dtype == np.dtype("f4"):
        (wy, ry), (wx, rx) = gaussian_weights(axes[0][1]), gaussian_weights(axes[1][1])
        if ry == rx and 1 <= ry <= MAX_FUSED_RADIUS:  # both axes in one launch (csrc/gm_stencil.hip gauss2d_kernel)
            shape = (bands, h - 2 * my, w - 2 * mx)
            out = _into(out, shape, src.dtype) or DeviceArray.empty(shape, src.dtype)
            as_doubles = ctypes.POINTER(ctypes.c_double)
            code = code_of(src.dtype)
            return _launch_into(
                src, out,
                lambda tile, pitch, o: L.gm_gauss_2d_window(
                    ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, wy.ctypes.data_as(as_doubles),
                    wx.ctypes.data_as(as_doubles), ry, float(fill), 1 if has else 0, ndv_bits, fill_bits, my, mx, h - 2 * my,
                    w - 2 * mx, ctypes.c_void_p(o.ptr)),
                lambda tile, pitch, opitch: L.gm_gauss_2d_into(
                    ctypes.c_void_p(tile.ptr), code, bands, h, w, pitch, wy.ctypes.data_as(as_doubles),
                    wx.ctypes.data_as(as_doubles), ry, float(fill), 1 if has else 0, ndv_bits, fill_bits, my, mx, h - 2 * my,
                    w - 2 * mx, ctypes.c_void_p(out.ptr), opitch), "gm_gauss_2d")
    current, replace = (src if bands == 1 else src.contiguous()), has  # (a single-band window is read where it lies)
    for k, (axis, sigma) in enumerate(axes):
        weights, radius = gaussian_weights(sigma) if sigma is not None else _IDENTITY_WEIGHTS
        last = k == len(axes) - 1
        y0, x0, oh, ow = (my, mx, h - 2 * my, w - 2 * mx) if last else (0, 0, h, w)
        out = DeviceArray.empty((bands, oh, ow), src.dtype)
        runtime.check(L.gm_gauss_axis_window(ctypes.c_void_p(current.ptr), code_of(src.dtype), bands, h, w, current.pitch or w,
                                             axis, weights.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), radius, float(fill),
                                             1 if replace else 0, ndv_bits, fill_bits, y0, x0, oh, ow,
                                             ctypes.c_void_p(out.ptr)), "gm_gauss_axis")
        current, replace = out, 0
    return current
