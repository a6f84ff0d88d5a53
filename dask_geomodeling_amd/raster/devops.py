"""Device implementations of the hot ``process`` bodies.

Each ``*_process`` function has the positional signature of the reference's
``process`` staticmethod it stands in for and keeps its control flow for ``None`` /
``time`` / ``meta`` responses; the numerics are handed to the HIP library as an
expression (one launch).  The ``*_node`` builders are shared with the fusion pass
(dask_geomodeling_amd/fuse.py), so a fused chain and the op-by-op path run the very
same per-op program steps.

There is no NumPy compute path here: without libgmhip / an MI355X these functions
raise ``GmHipError``.
"""
import ctypes

import numpy as np

from ..hip import program as P
from ..hip import runtime
from ..hip.array import DeviceArray, as_device
from ..hip.dtypes import code_of
from .. import utils

BOOL = np.dtype("bool")


# ---------------------------------------------------------------------------
# expression builders (shared with fuse.py)
# ---------------------------------------------------------------------------

def math_node(op, process_kwargs, nodes, extra=None):
    """Node of the no-data aware ufunc wrapper (raster/elemwise.py:243-297)."""
    dtype = np.dtype(process_kwargs["dtype"])
    fill = process_kwargs["fillvalue"]
    if dtype == BOOL:  # elemwise.py:278-284
        return P.Op(op, nodes, BOOL, None, fill=(op == "not_equal"))
    return P.Op(op, nodes, dtype, fill, fill=fill, extra=extra)


def any_negative(values):
    """Does an integer tile hold a negative value anywhere (no-data cells included)?"""
    arr = as_device(values)
    if arr.dtype.kind != "i" or arr.size == 0:
        return False
    below = P.evaluate(P.Op("less", [P.Leaf(None, arr.dtype, None, values=arr), P.Lit(0)], BOOL, None, fill=False))[0]
    return not _all_false(below)


def _all_false(bool_tile):
    out = ctypes.c_int(0)
    runtime.check(runtime.lib().gm_all_equal(ctypes.c_void_p(bool_tile.ptr), code_of(bool_tile.dtype),
                                             runtime.scalar_bits(0, bool_tile.dtype), bool_tile.size,
                                             ctypes.byref(out)), "gm_all_equal")
    return bool(out.value)


def clip_node(store, source):
    """Clip.process numerics (raster/misc.py:115-123)."""
    return P.Op("clip", [store, source], store.dtype, store.ndv)


def mask_dtype(value):
    """Output dtype of Mask (raster/misc.py:195-206)."""
    if isinstance(value, float):
        return np.dtype("float32")
    if value >= 0:
        return utils.get_uint_dtype(value)
    return utils.get_int_dtype(value)


def isclose_params(tile_dtype, ndv):
    """Parameters of ``~np.isclose(tile, ndv)`` as utils.get_index evaluates it for a float
    tile (utils.py:61-64) under NumPy's own scalar-promotion rules:
    (evaluate_in_f64, ndv as compared, tolerance, ndv_is_finite)."""
    x = np.zeros(1, dtype=tile_dtype)
    with np.errstate(all="ignore"):
        y = ndv if isinstance(ndv, (int, float, complex)) else np.asanyarray(ndv)
        if getattr(y, "dtype", None) is not None and y.dtype.kind != "m":
            y = np.asanyarray(y, dtype=np.result_type(y, 1.0))
        elif isinstance(y, int):
            y = float(y)
        diff = np.abs(x - y)
        tol = 1e-8 + 1e-5 * np.abs(y)
        in_f64 = diff.dtype == np.dtype("f8")
        ycmp = np.array(y).astype(diff.dtype)[()]
        tolcmp = np.array(tol).astype(diff.dtype)[()]
        finite = bool(np.isfinite(y))
    return in_f64, float(ycmp), float(tolcmp), finite


def _is_data(values, ndv):
    """utils.get_index (utils.py:61-64) on a small host array (table entries)."""
    values = np.asarray(values)
    if values.dtype.kind == "f":
        with np.errstate(all="ignore"):
            return ~np.isclose(values, ndv)
    return values != ndv


def mask_node(store, value):
    """Mask.process numerics (raster/misc.py:213-221)."""
    dtype = mask_dtype(value)
    fill = 1 if value == 0 else 0
    # NumPy refuses out-of-range python ints on assignment (e.g. value 256 -> uint8): same error
    np.zeros(1, dtype=dtype)[0:1] = value
    if isinstance(store, P.Op) and store.op == "lookup" and "miss" not in store.extra:
        # Mask(Reclassify(x)): "is the looked-up target data?" is a property of the table entry, so
        # the Mask is folded into the table on the host: the lookup then yields `value` / `fill`
        # directly -- entries small enough for every register class, and one step less per pixel.
        # Unmapped values: select -> the lookup's fill -> no data; otherwise the value itself, cast to
        # the table's dtype, which can never be that dtype's maximum (the key would be in the table).
        ex = store.extra
        data = _is_data(ex["lut"], store.ndv)
        lut = np.where(data, value, fill).astype(dtype)
        exc_vals = np.where(_is_data(np.asarray(ex["exc_vals"], dtype=ex["lut"].dtype), store.ndv), value, fill).astype(dtype)
        miss = fill if ex["select"] else value
        extra = {"kmin": ex["kmin"], "lut": lut, "exc_keys": ex["exc_keys"], "exc_vals": exc_vals,
                 "select": True, "miss": dtype.type(miss)}
        return P.Op("lookup", [store.args[0]], dtype, fill, extra=extra)
    if store.dtype.kind == "f":
        in_f64, y, tol, finite = isclose_params(store.dtype, store.ndv)
        flags = (1 if in_f64 else 0) | (2 if finite else 0)
    else:
        flags, y, tol = 4, 0.0, 0.0
    extra = {"value": value, "fill": fill, "flags": flags, "tol": tol, "y": y}
    return P.Op("mask", [store], dtype, fill, extra=extra)


def reduce_max_node(process_kwargs, nodes):
    """reduce_rasters(stack, "max", fillvalue, dtype) (raster/reduction.py:38-119): every raster
    with its no-data (utils.get_index) turned into NaN, np.fmax from left to right (what np.nanmax
    reduces with), NaN -> fill, cast to the output dtype."""
    dtype = np.dtype(process_kwargs["dtype"])
    fill = process_kwargs["fillvalue"]
    work = np.result_type(dtype, np.float16)
    work = np.dtype("f4") if work.itemsize < 4 else work
    acc = None
    half = np.result_type(dtype, np.float16).itemsize < 4
    for node in nodes:
        # (64-bit integer rasters go through float64 like in the reference, reduction.py:99-107: values
        #  beyond 2^53 round on the way, the int64 fill value comes back as INT64_MAX)
        if node.dtype.kind == "f":
            in_f64, y, tol, finite = isclose_params(node.dtype, node.ndv)
            extra = {"flags": (1 if in_f64 else 0) | (2 if finite else 0), "tol": tol, "y": y, "work": work}
        else:
            extra = {"flags": 4, "tol": 0.0, "y": 0.0, "work": work}
        if half and node.dtype.itemsize != 1:
            extra.update(half_round_extra(node.dtype, extra["flags"]))
        coded = P.Op("nanifnodata", [node], work, None, extra=extra)
        acc = coded if acc is None else P.Op("fmax", [acc, coded], work, None, extra={"work": work})
    if half and any(node.dtype.itemsize != 1 for node in nodes):
        # a float16 maximum -- possibly inf: a value beyond 65504 -- assigned into the 1-byte raster
        # (reduction.py:115) converts like NumPy on x86
        filled = P.Op("nantofill", [acc], work, None, extra={"work": work, "fill": fill})
        return P.Op("copy", [filled], dtype, fill, extra={"indefinite": True})
    return P.Op("nantofill", [acc], dtype, fill, extra={"work": work})


def half_round_extra(dtype, flags):
    """NANIFNODATA extras of a raster wider than a byte that enters a float16 stack (a 1-byte output dtype works in
    float16, reduction.py:99-101): the value is rounded to float16 as the reference's assignment rounds it -- once,
    from a register class that holds the raster exactly -- and values beyond 65504 become inf."""
    dtype = np.dtype(dtype)
    wide = dtype.itemsize > 4 or (dtype.kind in "iu" and dtype.itemsize == 4)
    return {"flags": flags | 8, "work": np.dtype("f8") if wide else np.dtype("f4")}


def reduce_max_process(process_kwargs, *args):
    """Drop-in for Max.process = wrap_reduction_function("max") (raster/reduction.py:189-212)."""
    stack = []
    for arg in args:
        if "time" in arg or "meta" in arg:  # (a None argument raises TypeError here, as in the reference)
            return arg
        if arg is None:
            continue
        stack.append(arg)
    if len(stack) == 0:
        return None
    tiles = _without_empty_indices([(a["values"], a["no_data_value"]) for a in stack])
    try:
        key = ("reduce_max", str(process_kwargs["dtype"]), P._scalar_key(process_kwargs["fillvalue"]), len(tiles))
        hash(key)
    except (TypeError, KeyError):
        key = None
    values, ndv = P.run_expr(lambda leaves: reduce_max_node(process_kwargs, leaves), tiles, key)
    return {"values": values, "no_data_value": ndv}


reduce_max_process.gm_op = "reduce_max"


def compare_dtype(tile_dtype, value):
    """dtype in which NumPy carries out ``tile < value`` for a python scalar ``value``."""
    try:
        dt = np.result_type(tile_dtype, value)
    except OverflowError:  # python int outside the tile dtype: the comparison is still exact
        dt = np.dtype("i8")
    if dt.kind == "f" and dt.itemsize < 4:
        dt = np.dtype("f4")
    return dt


def _integer_threshold(store_dtype, value):
    """A 64-bit integer tile against a python float: NumPy compares (double)x with the float, the tile itself
    must stay exact (int64 class).  For integer x and any real v: x < v <=> x < ceil(v) and x > v <=> x >= ceil(v)
    unless v is itself an integer -- so the comparison is carried out on integers.  -> (threshold, exact) with
    `exact` telling whether x == v is possible at all; None when the rewrite does not apply."""
    import math
    dt = np.dtype(store_dtype)
    if not (dt.kind in "iu" and dt.itemsize == 8 and isinstance(value, (float, np.floating))):
        return None
    if not math.isfinite(value) or abs(value) >= 2.0 ** 53:
        return None
    return (int(value), True) if float(value).is_integer() else (math.ceil(value), False)


def maskbelow_node(store, value):
    """MaskBelow.process numerics (raster/misc.py:249-250)."""
    rewritten = _integer_threshold(store.dtype, value)
    if rewritten is not None:
        value = rewritten[0]
    extra = {"value": value, "compare_dtype": compare_dtype(store.dtype, value)}
    return P.Op("maskbelow", [store], store.dtype, store.ndv, extra=extra)


def step_node(store, left, right, location, at):
    """Step.process numerics (raster/misc.py:314-326)."""
    rewritten = _integer_threshold(store.dtype, location)
    if rewritten is not None:
        location, exact = rewritten
        if not exact:
            at = right  # x == ceil(v) means x > v: the value "at" the location can never be taken
    extra = {"value": location, "compare_dtype": compare_dtype(store.dtype, location),
             "values": (left, at, right)}
    return P.Op("step", [store], store.dtype, store.ndv, extra=extra)


def fillnodata_node(prev, tile, dtype, fill):
    """One round of FillNoData's loop (raster/elemwise.py:749-755): ``tile`` where it has data
    (utils.get_index: isclose for float tiles), else ``prev``."""
    if tile.dtype.kind == "f":
        in_f64, y, tol, finite = isclose_params(tile.dtype, tile.ndv)
        flags = (1 if in_f64 else 0) | (2 if finite else 0)
    else:
        flags, y, tol = 4, 0.0, 0.0
    return P.Op("fillnodata", [prev, tile], dtype, fill, fill=fill, extra={"flags": flags, "tol": tol, "y": y})


def invert_node(x):
    return P.Op("invert", [x], BOOL, None)


def isdata_node(x):
    return P.Op("isdata", [x], BOOL, None)


def isnodata_node(x):
    return P.Op("isnodata", [x], BOOL, None)


# ---------------------------------------------------------------------------
# process bodies
# ---------------------------------------------------------------------------

def constant_operand(array):
    """An ndarray operand of an element-wise Block as the device takes it: float16 constants (tests/test_raster.py:
    343-367 pins float16 operand -> float32 result) are widened to float32 -- exact, and what NumPy's own cast to
    the float32 compute dtype does; HBM tiles have no 2-byte float type."""
    return array.astype(np.float32) if array.dtype == np.float16 else array


def is_tile(data):
    return isinstance(data, dict) and "values" in data


def math_process(op, process_kwargs, *args):
    """Drop-in for ``wrap_math_process_func(np.<op>)`` (raster/elemwise.py:243-297)."""
    if not args:
        return None
    tiles, slots = [], []
    for data in args:  # same early-outs, same order as the reference
        if data is None:
            return None
        if not isinstance(data, dict):
            if isinstance(data, np.ndarray):
                slots.append(("tile", len(tiles)))
                tiles.append((constant_operand(data), None))
            else:
                slots.append(("lit", data))
        elif "time" in data or "meta" in data:
            return data
        elif "values" in data:
            slots.append(("tile", len(tiles)))
            tiles.append((data["values"], data.get("no_data_value")))
        else:
            raise TypeError("Cannot apply math function to value {}".format(data))

    extra = None
    if (op == "power" and np.dtype(process_kwargs["dtype"]).kind in "iu" and len(slots) > 1 and
            slots[1][0] == "tile"):
        # np.power(integers, integer array): "Integers to negative integer powers are not allowed" as soon as ONE
        # exponent is negative (elemwise.py:383-406 only guards scalar exponents) -- checked before the launch
        if any_negative(tiles[slots[1][1]][0]):
            raise ValueError("Integers to negative integer powers are not allowed.")
        extra = {"exponent_checked": True}

    def build(leaves):
        nodes = [leaves[ref] if kind == "tile" else P.Lit(ref) for kind, ref in slots]
        return math_node(op, process_kwargs, nodes, extra)

    if not tiles:
        raise TypeError("at least one raster operand is required")
    try:
        key = ("math", op, str(process_kwargs["dtype"]), P._scalar_key(process_kwargs["fillvalue"]), extra is not None,
               tuple(ref if kind == "tile" else P._scalar_key(ref) for kind, ref in slots))
        hash(key)
    except (TypeError, KeyError):
        key = None
    values, ndv = P.run_expr(build, tiles, key)
    return {"no_data_value": ndv, "values": values}


def make_math_process(op):
    def process(process_kwargs, *args):
        return math_process(op, process_kwargs, *args)

    process.__name__ = op
    process.gm_op = op  # recognised by the fusion pass
    return process


def _size(values):
    return int(np.prod(values.shape, dtype=np.int64))


def _without_empty_indices(tiles):
    """The rasters of a FillNoData / reduce_rasters loop less those that cannot change the result: ``out[index] = ...`` with
    an EMPTY index (a raster without a frame in the window, next to a first raster that has some) is no error in NumPy
    and assigns nothing.  (An empty FIRST raster, or frames that differ otherwise, are NumPy's IndexError: raised by
    hip/program.py _check_indexed_shapes.)"""
    first = tuple(tiles[0][0].shape)
    return [tiles[0]] + [t for t in tiles[1:] if not (_size(t[0]) == 0 and tuple(t[0].shape) != first)]


def all_equal(values, ndv):
    """``np.all(values == ndv)`` on the device (Clip early-out, misc.py:108)."""
    arr = as_device(values)
    eff = P.effective_ndv(arr.dtype, ndv)
    if eff is None:
        return arr.size == 0
    out = ctypes.c_int(0)
    runtime.check(runtime.lib().gm_all_equal(ctypes.c_void_p(arr.ptr), code_of(arr.dtype),
                                             runtime.scalar_bits(eff, arr.dtype), arr.size,
                                             ctypes.byref(out)), "gm_all_equal")
    return bool(out.value)


def clip_process(data, source_data):
    """Drop-in for Clip.process (raster/misc.py:98-123)."""
    if data is None:
        return None
    if "values" not in data:
        return data
    if source_data is None:
        # misc.py:108-113: an all-no-data store is handed back untouched, else None
        return data if all_equal(data["values"], data["no_data_value"]) else None
    tiles = [(data["values"], data["no_data_value"]),
             (source_data["values"], source_data.get("no_data_value"))]
    if tuple(tiles[0][0].shape) != tuple(tiles[1][0].shape):
        if all_equal(data["values"], data["no_data_value"]):
            return data  # (misc.py:106-108 looks at the store first; the indexing that refuses the shapes comes after)
        if _size(tiles[1][0]) == 0:
            # values[mask] with an EMPTY mask (no frame of the source in the window) is no error in NumPy whatever the
            # shapes: nothing is masked.  Any other mismatch is NumPy's IndexError (hip/program.py _check_indexed_shapes)
            return {"values": data["values"], "no_data_value": data["no_data_value"]}
    try:
        values, ndv = P.run_expr(lambda leaves: clip_node(leaves[0], leaves[1]), tiles, ("clip",))
    except (NotImplementedError, P.FusionLimit):
        # no register class holds both tiles exactly (e.g. int64 store, float mask source):
        # reduce the source to its boolean "has data" tile first; Clip only needs that
        if np.dtype(tiles[1][0].dtype) == BOOL:
            raise
        has_data, _ = P.run_expr(lambda leaves: isdata_node(leaves[0]), tiles[1:])
        values, ndv = P.run_expr(lambda leaves: clip_node(leaves[0], leaves[1]),
                                 [tiles[0], (has_data, None)])
    return {"values": values, "no_data_value": data["no_data_value"]}


def mask_process(data, value):
    """Drop-in for Mask.process (raster/misc.py:208-222)."""
    if data is None or "values" not in data:
        return data
    tiles = [(data["values"], data["no_data_value"])]
    values, ndv = P.run_expr(lambda leaves: mask_node(leaves[0], value), tiles, ("mask", P._scalar_key(value)))
    return {"values": values, "no_data_value": ndv}


def unary_bool_process(kind, data):
    tiles = [(data["values"], data.get("no_data_value"))]
    builder = {"invert": invert_node, "isdata": isdata_node, "isnodata": isnodata_node}[kind]
    values, _ = P.run_expr(lambda leaves: builder(leaves[0]), tiles, (kind,))
    return {"values": values, "no_data_value": None}


def maskbelow_process(data, value):
    """Drop-in for MaskBelow.process (raster/misc.py:245-251)."""
    if data is None or "values" not in data:
        return data
    tiles = [(data["values"], data["no_data_value"])]
    values, _ = P.run_expr(lambda leaves: maskbelow_node(leaves[0], value), tiles)
    return {"values": values, "no_data_value": data["no_data_value"]}


def step_process(data, left, right, location, at):
    """Drop-in for Step.process (raster/misc.py:308-328)."""
    if data is None or "values" not in data:
        return data
    tiles = [(data["values"], data["no_data_value"])]
    values, _ = P.run_expr(lambda leaves: step_node(leaves[0], left, right, location, at), tiles)
    return {"values": values, "no_data_value": data["no_data_value"]}


def fillnodata_process(process_kwargs, *args):
    """Drop-in for FillNoData.process (raster/elemwise.py:726-757)."""
    tiles = []
    for data in args:
        if data is None:
            continue
        if "time" in data or "meta" in data:
            return data
        if "values" in data and "no_data_value" in data:
            tiles.append((data["values"], data["no_data_value"]))
    dtype = np.dtype(process_kwargs["dtype"])
    fill = utils.get_dtype_max(dtype)
    if not tiles:
        return None
    tiles = _without_empty_indices(tiles)

    def build(leaves):
        node = P.Lit(fill)
        for leaf in leaves:
            node = fillnodata_node(node, leaf, dtype, fill)
        return node

    values, _ = P.run_expr(build, tiles, ("fillnodata", dtype.str, len(tiles)))
    return {"values": values, "no_data_value": fill}
