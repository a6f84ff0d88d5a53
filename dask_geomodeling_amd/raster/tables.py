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
        L.gm_classify2_window.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_uint64, c.c_void_p, c.c_int, c.c_int,
                                          c.c_int, c.c_void_p, c.c_int, c.c_uint64, c.c_int, c.c_int, c.c_int]
        L.gm_classify2_window.restype = c.c_int
        L.gm_reclassify.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_uint64,
                                    c.c_void_p, c.c_int, c.c_size_t]
        L.gm_reclassify.restype = c.c_int
        _declared = True
    return L


def classify_node(store, bins, right):
    """Classify.process numerics (raster/misc.py:391-398) as a program node (GM_OP_DIGITIZE).
    Raises FusionLimit when the lookup has to run as its own kernel (many edges, bool / uint64
    tiles, more classes than uint16 holds)."""
    out_dtype = utils.get_uint_dtype(len(bins) + 2)
    if out_dtype.itemsize > 2 or len(bins) > P.GM_MAX_LINEAR_EDGES:
        raise P.FusionLimit("Classify with %d edges runs as its own kernel" % len(bins))
    edges = np.asarray(bins)
    # np.digitize -> searchsorted compares in result_type(tile, edges): integers stay integers
    tile_is_int = store.dtype.kind in "iu"
    if tile_is_int and edges.dtype.kind in "iub":
        payload = np.ascontiguousarray(edges, dtype=np.int64)
    else:
        payload = np.ascontiguousarray(edges, dtype=np.float64)
    fill = utils.get_dtype_max(out_dtype)
    return P.Op("digitize", [store], out_dtype, fill, extra={"edges": payload, "right": bool(right)})


def reclassify_node(store, process_kwargs):
    """Reclassify.process numerics (raster/misc.py:489-514) as a program node (GM_OP_LOOKUP): a dense
    window of pre-resolved targets + at most 8 keys outside it (typically the merged-in no-data key).
    Raises FusionLimit when the table is too sparse for that (it then runs as its own kernel)."""
    dtype = np.dtype(process_kwargs["dtype"])
    if dtype not in (np.dtype("i8"), np.dtype("f8")):
        raise P.FusionLimit("Reclassify target dtype %s" % dtype)
    if store.dtype.kind not in "iub" or store.dtype == np.dtype("u8"):
        raise P.FusionLimit("Reclassify of %s tiles" % store.dtype)
    fill = process_kwargs["fillvalue"]
    select = bool(process_kwargs["select"])
    keys, targets = reclassify_table(process_kwargs, store.ndv)
    keys = keys.astype(np.int64)
    targets = targets.astype(dtype)
    n = len(keys)
    # densest run of keys spanning < LOOKUP_MAX_WINDOW slots
    best_lo, best_cnt, lo = 0, 0, 0
    for hi in range(n):
        while int(keys[hi]) - int(keys[lo]) >= P.LOOKUP_MAX_WINDOW:
            lo += 1
        if hi - lo + 1 > best_cnt:
            best_lo, best_cnt = lo, hi - lo + 1
    if n == 0 or n - best_cnt > P.LOOKUP_MAX_EXCEPTIONS:
        raise P.FusionLimit("Reclassify table too sparse for an in-program lookup")
    w_lo, w_hi = best_lo, best_lo + best_cnt - 1
    kmin = int(keys[w_lo])
    slots = int(keys[w_hi]) - kmin + 1
    if select:
        lut = np.full(slots, fill, dtype=dtype)
    else:  # unmapped values pass through, cast to the output dtype (misc.py:508)
        lut = (np.arange(slots, dtype=np.int64) + kmin).astype(dtype)
    lut[keys[w_lo:w_hi + 1] - kmin] = targets[w_lo:w_hi + 1]
    outside = [i for i in range(n) if i < w_lo or i > w_hi]
    extra = {"kmin": kmin, "lut": lut, "exc_keys": keys[outside], "exc_vals": targets[outside], "select": select}
    return P.Op("lookup", [store], dtype, fill, extra=extra)


# A lone Classify / Reclassify (nothing to fuse with) runs fastest in its dedicated LDS kernel
# (measured 8192^2: Classify 60 vs 111 us, Reclassify u8 -> i64 92 vs 220 us); the program ops exist
# for fusion.  Tests flip this switch to pin the program route to the golden vectors as well.
# 1-byte tiles always take the program route: a program over one 1-byte tile runs as a 256-entry table
# (MEASUREMENTS K1b; Classify uint8 -> uint8 24 us, Reclassify uint8 -> int64 86 us on 8192^2).
_PREFER_PROGRAM = False


def _by_table(dtype):
    return np.dtype(dtype).itemsize == 1


def _try_program(build, data, key):
    """Evaluate the single table node ``build(leaf)`` over the tile as a program (the staged plan kept per ``key`` and
    kind of tile: program.run_expr); None when it has to run as its own kernel."""
    try:
        hash(key)
    except TypeError:
        key = None
    try:
        values, ndv = P.run_expr(lambda leaves: build(leaves[0]), [(data["values"], data["no_data_value"])], key)
    except (P.FusionLimit, NotImplementedError):
        return None
    return {"values": values, "no_data_value": ndv}


def classify_process(data, bins, right):
    """Drop-in for Classify.process (raster/misc.py:387-399)."""
    if data is None or "values" not in data:
        return data
    runtime.require_device()
    if (_PREFER_PROGRAM or _by_table(data["values"].dtype)) and data["values"].dtype != np.dtype("bool"):
        try:  # few edges: one program step (the same one a fused chain uses)
            key = ("classify", tuple(P._scalar_key(b) for b in bins), bool(right))
            hit = _try_program(lambda leaf: classify_node(leaf, bins, right), data, key)
        except P.FusionLimit:
            hit = None
        if hit is not None:
            return hit
    values = data["values"]
    # (a window of a resident raster is read where it lies by the kernel of float32 tiles with many edges; one band)
    windowed = (isinstance(values, DeviceArray) and values.pitch is not None and values.band_pitch is None
                and values.dtype == np.dtype("f4") and len(values.shape) == 3 and values.shape[0] == 1)
    if not windowed:
        values = as_device(values)
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
    if windowed:
        # whole groups of 4 cells per row: a width that is no multiple of 4 runs over the next one where the resident raster has
        # the cells to read (the classes of the extra cells are cropped away: the result is a window of the padded tile)
        pad = -values.shape[2] % 4
        if pad == 0 or values.room_right() >= pad:
            wide = values.shape[2] + pad
            out = DeviceArray.empty((1, values.shape[1], wide), out_dtype)
            rc = _lib().gm_classify2_window(
                ctypes.c_void_p(values.ptr), code_of(values.dtype), 1 if eff is not None else 0,
                runtime.scalar_bits(eff if eff is not None else 0, values.dtype),
                payload.ctypes.data_as(ctypes.c_void_p), int(payload.size), are_int, 1 if right else 0,
                ctypes.c_void_p(out.ptr), code_of(out_dtype), int(fill), values.shape[1], wide, values.pitch)
            if rc == 0:
                return {"values": out.window(0, 0, values.shape[1], values.shape[2]) if pad else out, "no_data_value": fill}
            if rc != 9002:  # GM_ERR_UNSUPPORTED: these edges / this shape want the compact tile
                runtime.check(rc, "gm_classify")
        values = values.contiguous()
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
    in_dtype = store_data["values"].dtype
    if in_dtype == np.dtype("u8"):
        # (the sums of an unsigned series, temporal.py:548-602: as int32 when they lie within 2^24 -- the same keys are hit)
        narrow = P._narrowed(as_device(store_data["values"]), store_data["no_data_value"])
        if narrow is not None:
            return reclassify_process({"values": narrow[0], "no_data_value": narrow[1]}, process_kwargs)
    if in_dtype.kind not in "iub" or in_dtype == np.dtype("u8"):
        raise NotImplementedError("Reclassify on %s tiles is not supported on the device" % in_dtype)
    dtype = np.dtype(process_kwargs["dtype"])
    fill = process_kwargs["fillvalue"]
    hit = None
    if _PREFER_PROGRAM or _by_table(in_dtype):
        try:  # dense table: one program step (the same one a fused chain uses)
            key = ("reclassify", str(process_kwargs["dtype"]), P._scalar_key(process_kwargs["fillvalue"]),
                   bool(process_kwargs["select"]), P._scalar_key(process_kwargs["data"]))
            hit = _try_program(lambda leaf: reclassify_node(leaf, process_kwargs), store_data, key)
        except P.FusionLimit:
            hit = None
    if hit is not None:
        return hit
    values = as_device(store_data["values"])
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
