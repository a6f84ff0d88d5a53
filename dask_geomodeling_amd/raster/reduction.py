"""Reduction over a stack of rasters (counterpart of raster/reduction.py).

``reduce_rasters(stack, statistic, no_data_value=None, dtype=None)`` applies one of the reference's
statistics -- first, last, count, sum, mean, min, max, argmin, argmax, std, var, median, product,
p<number> -- to a list of raster responses, skipping no data (reduction.py:38-119); ``Max`` is the Block
built on it (reduction.py:215-230).  On this path the rasters stay on the device:

* ``first`` / ``last`` / ``count`` work on the rasters themselves: a chain of "take the tile where it has
  data" steps (the FillNoData program op, utils.get_index's isclose rule for float tiles), resp. a sum of
  data masks, as ONE element-wise program over the stack;
* ``max`` is a program too (NaN-coded no data, np.fmax left to right: devops.reduce_max_node) -- it fuses
  with whatever feeds and follows a ``Max`` Block;
* every other statistic follows the reference literally: each raster is cast to the work dtype
  ``result_type(dtype, float16)`` with NaN for no data into one (n, bands, H, W) stack (program op
  NANIFNODATA writing straight into its slice), ``gm_reduce_stack`` (csrc/gm_reduce.hip) is NumPy's
  nan-function along axis 0 of that stack -- same accumulation order, same roundings, float16 work dtypes
  rounded after every operation like NumPy's float16 loops --, cells that are no data everywhere get the
  fill value and the result is cast into the output dtype.
"""
import ctypes

import numpy as np

from ..utils import Extent
from . import devops
from .base import RasterBlock
from .elemwise import BaseElementwise

__all__ = ["Max", "reduce_rasters", "check_statistic"]

# statistic -> code of csrc/gm_reduce.hip (None: handled by an element-wise program)
STATISTICS = {"first": None, "last": None, "count": None, "sum": 0, "mean": 4, "min": 2, "max": 3, "argmin": 9,
              "argmax": 10, "std": 6, "var": 5, "median": 7, "product": 8}
_PERCENTILE_CODE = 11
# median / percentile sort a cell's values in LDS up to this many rasters; beyond, the temporal kernels select the wanted ranks
# (round 5: 64, like their own limit -- the LDS insertion sort of 300 rasters took 0.5 s per 2048^2 cells, the selection 30 ms)
MAX_SORTED_RASTERS = 64
MAX_SORTED_RASTERS_HALF = 599


def parse_percentile_statistic(statistic):
    """'p<number>' -> ('percentile', number); anything else -> (statistic, None)  (utils.py:805-823)."""
    import re
    match = re.match(r"^p(\d+(\.\d+)?)$", statistic)
    if not match:
        return statistic, None
    percentile = float(match.group(1))
    if not 0 <= percentile <= 100:
        raise ValueError("Percentiles must be in the range [0, 100]")
    return "percentile", percentile


def check_statistic(statistic):
    if statistic not in STATISTICS:
        statistic, percentile = parse_percentile_statistic(statistic)
        if percentile is None:
            raise ValueError('Unknown statistic "{}"'.format(statistic))


_INDEX_EXTRA = {}   # (dtype, type of the no-data value, its bytes) -> the parameters below: ~20 us of NumPy scalar work per
                    # raster otherwise, which is what a stack of 60 rasters cost in host time before any kernel ran


def _index_extra(dtype, ndv):
    """flags / tolerance of utils.get_index for a tile: isclose for float tiles, exact != for the others."""
    dtype = np.dtype(dtype)
    try:
        key = (dtype.str, type(ndv), ndv.tobytes() if isinstance(ndv, np.generic) else ndv)
        hit = _INDEX_EXTRA.get(key)
    except TypeError:
        key = hit = None
    if hit is None:
        hit = _index_extra_uncached(dtype, ndv)
        if key is not None and len(_INDEX_EXTRA) < 4096 and not (isinstance(ndv, float) and ndv != ndv):
            _INDEX_EXTRA[key] = hit
    return dict(hit)


def _index_extra_uncached(dtype, ndv):
    if dtype.kind == "f":
        in_f64, y, tol, finite = devops.isclose_params(dtype, ndv)
        return {"flags": (1 if in_f64 else 0) | (2 if finite else 0), "tol": tol, "y": y}
    return {"flags": 4, "tol": 0.0, "y": 0.0}


def _stack_nodes(stack):
    from ..hip import program as P
    return [P.Leaf(None, data["values"].dtype, data["no_data_value"], values=data["values"]) for data in stack]


def _through_program(root):
    from ..hip import program as P
    values, _ = P.evaluate(root)
    return values


def _pick(stack, order, fill_value, dtype):
    """'last' / 'first': out[index] = values[index] raster after raster (reduction.py:78-87)."""
    from ..hip import program as P

    def build(nodes):
        acc = P.Lit(dtype.type(fill_value))
        for node in (nodes if order == "last" else nodes[::-1]):
            acc = devops.fillnodata_node(acc, node, dtype, fill_value)
        return acc

    tiles = [(data["values"], data["no_data_value"]) for data in stack]
    values, _ = P.run_expr(build, tiles, ("pick", order, dtype.str, P._scalar_key(fill_value), len(stack)))
    return values


def _count(stack, dtype):
    """'count': out += get_index(values, ndv) raster after raster (reduction.py:88-91)."""
    from ..hip import program as P
    counter = np.dtype("f8") if dtype.kind == "f" else (np.dtype("i8") if dtype.itemsize == 8 else np.dtype("i4"))

    def build(nodes):
        acc = None
        for node in nodes:
            extra = dict(_index_extra(node.dtype, node.ndv), value=1, fill=0)
            is_data = P.Op("mask", [node], counter, None, extra=extra)
            acc = is_data if acc is None else P.Op("add", [acc, is_data], counter, None, fill=0)
        if counter != dtype:
            acc = P.Op("copy", [acc], dtype, None)  # (integers wrap modulo the output width, like out += index)
        return acc

    tiles = [(data["values"], data["no_data_value"]) for data in stack]
    values, _ = P.run_expr(build, tiles, ("count", dtype.str, len(stack)))
    return values


_declared = False


def _lib():
    global _declared
    from ..hip import runtime
    L = runtime.lib()
    if not _declared:
        c = ctypes
        L.gm_reduce_stack.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_int64, c.c_int, c.c_void_p,
                                      c.c_void_p, c.c_uint64]
        L.gm_reduce_stack.restype = c.c_int
        _declared = True
    return L


def _nan_function(stack, statistic, percentile, fill_value, dtype):
    """NumPy's nan-function of the statistic over the NaN-coded stack in the work dtype (reduction.py:92-117)."""
    from ..hip import program as P
    from ..hip import runtime
    from ..hip.array import DeviceArray
    from ..hip.dtypes import code_of
    from .temporal import percentile_plan
    work = np.result_type(dtype, np.float16)
    half = work.itemsize < 4
    storage = np.dtype("f4") if work.itemsize <= 4 else np.dtype("f8")
    nodes = _stack_nodes(stack)
    n, shape = len(stack), tuple(stack[0]["values"].shape)
    # (float16 stacks -- a 1-byte output dtype -- have no selection kernel: they keep the LDS sort as far as it reaches)
    long_sort = statistic in ("median", "percentile") and n > (MAX_SORTED_RASTERS_HALF if half else MAX_SORTED_RASTERS)
    if long_sort and half:
        raise NotImplementedError("%s over %d rasters in float16 (a 1-byte output dtype)" % (statistic, n))
    bands = shape[0]
    in_place = _rasters_in_place(stack, nodes, statistic, storage, half, shape, fill_value, dtype, percentile)
    if in_place is not None:
        return in_place
    stacked = DeviceArray.empty((n * bands,) + shape[1:], storage)
    for i, node in enumerate(nodes):
        extra = dict(_index_extra(node.dtype, node.ndv), work=storage)
        if half and node.dtype.itemsize != 1:
            extra.update(devops.half_round_extra(node.dtype, extra["flags"]))
        coded = P.Op("nanifnodata", [node], storage, None, extra=extra)
        compiled, leaves, tile_shape = P.plan(coded)
        if tuple(tile_shape) != shape:
            raise ValueError("could not broadcast raster %d of shape %s into shape %s" % (i, tuple(tile_shape), shape))
        P.launch(compiled, [leaf.values for leaf in leaves], shape, out=stacked.band_slice(i * bands, (i + 1) * bands))
    if long_sort:
        # more rasters than a cell's values fit the LDS sort: the stack is n frames of bands * H * W cells, NaN for no
        # data -- what the temporal kernels select the middle pair / the percentile's neighbours from in global memory
        from .temporal import aggregate_frames
        name = "median" if statistic == "median" else "p%r" % float(percentile)
        frames = stacked.reshape((n, bands * shape[1], shape[2]))
        # (a cell without data in any raster gets the fill value, like with every other statistic; a median that is
        #  infinite or NaN -- inf - inf between the middle pair -- stays what it is, reduction.py:113-115)
        result = aggregate_frames(frames, None, [(0, n)], name, storage, fill=storage.type(fill_value),
                                  keep_nonfinite=True).reshape(shape)
        if storage == dtype:
            return result
        like_x86 = dtype.kind in "iub" and dtype.itemsize <= 4 and dtype != np.dtype("u4")
        return _through_program(P.Op("copy", [P.Leaf(None, storage, None, values=result)], dtype, None,
                                     extra={"indefinite": True} if like_x86 else None))
    result = DeviceArray.empty(shape, storage)
    plan = percentile_plan(percentile, n, work) if statistic == "percentile" else None
    code = _PERCENTILE_CODE if statistic == "percentile" else STATISTICS[statistic]
    runtime.check(_lib().gm_reduce_stack(stacked.ptr, code_of(storage), int(half), n, result.size, code,
                                         plan, result.ptr, runtime.scalar_bits(float(fill_value), storage)),
                  "gm_reduce_stack")
    if storage == dtype:
        return result
    # (`out[...] = func(...)` assigns the float statistic into the output raster without a finite check,
    #  reduction.py:115: an integer raster of up to 32 bits gets NumPy's x86 conversion of whatever it is)
    like_x86 = dtype.kind in "iub" and dtype.itemsize <= 4 and dtype != np.dtype("u4")
    return _through_program(P.Op("copy", [P.Leaf(None, storage, None, values=result)], dtype, None,
                                 extra={"indefinite": True} if like_x86 else None))


class gm_raster_ref(ctypes.Structure):
    _fields_ = [("values", ctypes.c_void_p), ("ndv_compared", ctypes.c_double), ("tolerance", ctypes.c_double),
                ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


_RASTERS_IN_PLACE = True  # (the tests switch it off to compare the two routes)


def _rasters_in_place(stack, nodes, statistic, storage, half, shape, fill_value, dtype, percentile=None):
    """Round 6: rasters that already ARE the work dtype (float32 rasters reduced in float32, float64 in float64) are read where
    they are -- one launch over n tile pointers, the NaN coding of reduction.py:99-107 done in the load (gm_reduce_rasters) --
    instead of being coded into a stack by a launch each and read again.  -> the result tile, or None for everything else."""
    from ..hip import program as P
    from ..hip import runtime
    from ..hip.array import DeviceArray, as_device
    from ..hip.dtypes import code_of
    if not _RASTERS_IN_PLACE or half or len(stack) < 2:
        return None
    if statistic in ("median", "percentile") and len(stack) > (64 if storage.itemsize == 4 else 32):
        return None  # (more rasters than a lane sorts in its registers: the coded stack and the LDS sort / the selection)
    if any(np.dtype(node.dtype) != storage or tuple(np.shape(node.values)) != shape for node in nodes):
        return None
    L = _lib()
    _declare_reduce_rasters(L)
    tiles = [as_device(node.values) for node in nodes]  # (windows of resident rasters are compacted: the kernel reads whole tiles)
    refs = (gm_raster_ref * len(nodes))()
    for ref, node, tile in zip(refs, nodes, tiles):
        extra = _index_extra(node.dtype, node.ndv)
        ref.values, ref.ndv_compared, ref.tolerance, ref.flags = tile.ptr, extra["y"], extra["tol"], extra["flags"]
    result = DeviceArray.empty(shape, storage)
    plan = None
    if statistic == "percentile":
        from .temporal import percentile_plan
        plan = percentile_plan(percentile, len(nodes), np.result_type(dtype, np.float16))
    code = _PERCENTILE_CODE if statistic == "percentile" else STATISTICS[statistic]
    runtime.check(L.gm_reduce_rasters(refs, code_of(storage), len(nodes), result.size, code, plan, result.ptr,
                                      runtime.scalar_bits(float(fill_value), storage)), "gm_reduce_rasters")
    del tiles
    if storage == dtype:
        return result
    like_x86 = dtype.kind in "iub" and dtype.itemsize <= 4 and dtype != np.dtype("u4")
    return _through_program(P.Op("copy", [P.Leaf(None, storage, None, values=result)], dtype, None,
                                 extra={"indefinite": True} if like_x86 else None))


_MAX_PICK_AS_PROGRAM = 8  # rasters: up to here first / last / count are ONE fused program (and fuse with what surrounds them)
_PICK_CODES = {"count": 1, "first": 12, "last": 13}


def _picked_in_place(stack, statistic, fill_value, dtype):
    """'first' / 'last' / 'count' over float rasters that already have the output dtype: one launch over n tile pointers
    (gm_reduce_rasters with GM_RED_FIRST / LAST / COUNT) -- the index of utils.get_index evaluated in the load, a NaN cell being
    data like in the reference.  -> the result tile, or None for every other stack."""
    from ..hip import runtime
    from ..hip.array import DeviceArray, as_device
    from ..hip.dtypes import code_of
    if not _RASTERS_IN_PLACE or dtype.kind != "f" or dtype.itemsize < 4 or len(stack) >= 1 << 24:
        return None
    nodes = _stack_nodes(stack)
    shape = tuple(np.shape(nodes[0].values))
    if len(shape) != 3 or any(np.dtype(node.dtype) != dtype or tuple(np.shape(node.values)) != shape for node in nodes):
        return None
    L = _lib()
    _declare_reduce_rasters(L)
    tiles = [as_device(node.values) for node in nodes]
    refs = (gm_raster_ref * len(nodes))()
    for ref, node, tile in zip(refs, nodes, tiles):
        extra = _index_extra(node.dtype, node.ndv)
        ref.values, ref.ndv_compared, ref.tolerance, ref.flags = tile.ptr, extra["y"], extra["tol"], extra["flags"]
    result = DeviceArray.empty(shape, dtype)
    runtime.check(L.gm_reduce_rasters(refs, code_of(dtype), len(nodes), result.size, _PICK_CODES[statistic], None, result.ptr,
                                      runtime.scalar_bits(float(fill_value), dtype)), "gm_reduce_rasters")
    del tiles
    return result


def _declare_reduce_rasters(L):
    if not getattr(L, "_gm_reduce_rasters_declared", False):
        L.gm_reduce_rasters.argtypes = [ctypes.POINTER(gm_raster_ref), ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.gm_reduce_rasters.restype = ctypes.c_int
        L._gm_reduce_rasters_declared = True


_MAX_AS_PROGRAM = 8  # rasters


def reduce_rasters(stack, statistic, no_data_value=None, dtype=None):
    """Apply a statistic (e.g. "mean") to a stack of rasters, skipping 'no data' values
    (reduction.py:38-119; same arguments, same errors).  ``stack`` is a list of raster responses
    (``values`` ndarray or DeviceArray + ``no_data_value``); the result's ``values`` is a DeviceArray."""
    if statistic not in STATISTICS:
        statistic, percentile = parse_percentile_statistic(statistic)
        if percentile is None:
            raise KeyError('Unknown statistic "{}"'.format(statistic))
    else:
        percentile = None
    if len(stack) == 0:
        raise ValueError("Cannot reduce a zero-length stack")
    dtype = np.dtype(stack[0]["values"].dtype if dtype is None else dtype)
    if no_data_value is None:
        no_data_value = stack[0]["no_data_value"]
    # sum and count never hold no data: cells without a contribution are 0
    fill_value = 0 if statistic in ("sum", "count") else no_data_value
    if statistic in ("first", "last", "count"):
        # (more rasters than a program takes inputs -- a Group of many sources -- ran as a chain of staged programs: 'first' over
        #  60 rasters 712 us against 233 for their maximum; float rasters of the output dtype are read in place instead)
        values = _picked_in_place(stack, statistic, fill_value, dtype) if len(stack) > _MAX_PICK_AS_PROGRAM else None
        if values is None:
            values = _pick(stack, statistic, fill_value, dtype) if statistic != "count" else _count(stack, dtype)
    elif statistic == "max" and len(stack) <= _MAX_AS_PROGRAM:
        # (few rasters -- the Max Block's two or three: ONE fused program, fmax from left to right; a long stack takes the
        #  stack kernel like every other statistic: the program of 300 rasters took 0.8 s to plan and run, the kernel 6 ms)
        from ..hip import program as P
        kwargs = {"dtype": dtype, "fillvalue": fill_value}
        values, _ = P.run_expr(lambda leaves: devops.reduce_max_node(kwargs, leaves),
                               [(d["values"], d["no_data_value"]) for d in stack])
    else:
        values = _nan_function(stack, statistic, percentile, fill_value, dtype)
    return {"values": values, "no_data_value": no_data_value}


class BaseReduction(BaseElementwise):
    """A Block over several rasters whose result exists wherever ANY of them has data: extent and geometry
    are unions, and -- unlike the element-wise Blocks -- there is nothing to compute at all when the
    arguments share no period (reduction.py:122-186)."""

    def __init__(self, *args):
        bad = next((arg for arg in args if not isinstance(arg, RasterBlock)), None)
        if bad is not None:
            raise TypeError("'{}' object is not allowed".format(type(bad)))
        super().__init__(*args)

    def get_sources_and_requests(self, **request):
        process_kwargs = {"dtype": self.dtype.name, "fillvalue": self.fillvalue}
        common = self.period
        if common is None:
            return [(process_kwargs, None)]
        # every argument must answer the same frames: move the requested window into the common period
        first, last = common
        start, stop = request.get("start"), request.get("stop")
        if start is None:
            request["start"] = last
        elif stop is None:
            request["start"] = min(max(start, first), last)
        else:
            request["start"], request["stop"] = max(start, first), min(stop, last)
        return [(process_kwargs, None)] + [(raster, request) for raster in self.args]

    @property
    def extent(self):
        boxes = [box for box in (raster.extent for raster in self.args) if box is not None]
        if not boxes:
            return None
        west, south, east, north = zip(*boxes)
        return min(west), min(south), max(east), max(north)

    @property
    def geometry(self):
        shapes = [shape for shape in (raster.geometry for raster in self.args) if shape is not None]
        if not shapes:
            return None
        if len(shapes) == 1:
            return shapes[0]
        union = Extent.from_geometry(shapes[0])
        for shape in shapes[1:]:
            union = union.union(Extent.from_geometry(shape))
        return union.as_geometry()


class Max(BaseReduction):
    """Take the maximum value of two or more rasters, ignoring no data (reduction.py:215-230)."""

    process = staticmethod(devops.reduce_max_process)

    @property
    def dtype(self):
        return np.result_type(*self.args)  # (not lifted to int32 / float32 like the arithmetic Blocks)
