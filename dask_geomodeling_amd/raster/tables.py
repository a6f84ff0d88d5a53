"""Device implementations of Classify.process and Reclassify.process (K4/K5).

Host side = the small table preparation of the reference (sorting the [from, to] pairs,
merging the no-data key, choosing the comparison dtype NumPy would promote to); the
per-pixel work is one HIP launch with the table staged in LDS (csrc/gm_tables.hip).
"""
import ctypes

import numpy as np

from .. import utils
from ..hip import program as P
from ..hip import runtime
from ..hip.array import DeviceArray, as_device
from ..hip.dtypes import code_of

_declared = False


def _lib():
    global _declared
    L = runtime.lib()
    if not _declared:
        c = ctypes
        L.gm_classify2.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_uint64, c.c_void_p, c.c_int, c.c_int,
                                   c.c_int, c.c_void_p, c.c_int, c.c_uint64, c.c_size_t]
        L.gm_classify2.restype = c.c_int
        L.gm_reclassify.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_uint64,
                                    c.c_void_p, c.c_int, c.c_size_t]
        L.gm_reclassify.restype = c.c_int
        _declared = True
    return L


def classify_process(data, bins, right):
    """Drop-in for Classify.process (raster/misc.py:387-399)."""
    if data is None or "values" not in data:
        return data
    runtime.require_device()
    values = as_device(data["values"])
    out_dtype = utils.get_uint_dtype(len(bins) + 2)
    fill = utils.get_dtype_max(out_dtype)
    edges = np.asarray(bins)
    # np.digitize -> searchsorted compares in result_type(tile, edges): integers stay integers
    tile_is_int = values.dtype.kind in "iub" and values.dtype != np.dtype("u8")
    if tile_is_int and edges.dtype.kind in "iub":
        payload, are_int = np.ascontiguousarray(edges, dtype=np.int64), 1
    else:
        payload, are_int = np.ascontiguousarray(edges, dtype=np.float64), 0
    eff = P.effective_ndv(values.dtype, data["no_data_value"]) if values.dtype != np.dtype("bool") else None
    if values.dtype == np.dtype("bool") and data["no_data_value"] is not None:
        eff = P.effective_ndv(values.dtype, data["no_data_value"])
    out = DeviceArray.empty(values.shape, out_dtype)
    runtime.check(_lib().gm_classify2(
        ctypes.c_void_p(values.ptr), code_of(values.dtype), 1 if eff is not None else 0,
        runtime.scalar_bits(eff if eff is not None else 0, values.dtype),
        payload.ctypes.data_as(ctypes.c_void_p), int(payload.size), are_int, 1 if right else 0,
        ctypes.c_void_p(out.ptr), code_of(out_dtype), int(fill), values.size), "gm_classify")
    return {"values": out, "no_data_value": fill}


def reclassify_table(process_kwargs, no_data_value):
    """Sorted keys + targets with the no-data key merged in (raster/misc.py:489-502)."""
    source, target = zip(*process_kwargs["data"])
    source, target = np.asarray(source), np.asarray(target)
    fill = process_kwargs["fillvalue"]
    if no_data_value is not None and no_data_value not in source:
        source = np.append(source, no_data_value)
        target = np.append(target, fill)
    order = np.argsort(source)
    return source[order], target[order]


def reclassify_process(store_data, process_kwargs):
    """Drop-in for Reclassify.process (raster/misc.py:482-515)."""
    if store_data is None or "values" not in store_data:
        return store_data
    runtime.require_device()
    values = as_device(store_data["values"])
    if values.dtype.kind not in "iub" or values.dtype == np.dtype("u8"):
        raise NotImplementedError("Reclassify on %s tiles is not supported on the device" % values.dtype)
    dtype = np.dtype(process_kwargs["dtype"])
    fill = process_kwargs["fillvalue"]
    keys, targets = reclassify_table(process_kwargs, store_data["no_data_value"])
    if dtype not in (np.dtype("i8"), np.dtype("f8")):
        raise NotImplementedError("Reclassify target dtype %s" % dtype)
    keys64 = np.ascontiguousarray(keys.astype(np.int64))
    tgt = np.ascontiguousarray(targets.astype(dtype))
    out = DeviceArray.empty(values.shape, dtype)
    runtime.check(_lib().gm_reclassify(
        ctypes.c_void_p(values.ptr), code_of(values.dtype),
        keys64.ctypes.data_as(ctypes.c_void_p), tgt.ctypes.data_as(ctypes.c_void_p), int(keys64.size),
        1 if process_kwargs["select"] else 0, runtime.scalar_bits(fill, dtype),
        ctypes.c_void_p(out.ptr), code_of(dtype), values.size), "gm_reclassify")
    return {"values": out, "no_data_value": fill}
