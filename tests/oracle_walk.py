"""The reference's evaluation of a view, request by request, with the ORACLE's process bodies: every Block plans its
request (``get_sources_and_requests`` -- host logic, itself pinned against the reference's planning by the golden
families 'temporal', 'group', 'round4' ... in tests/test_blocks_cpu.py), the sources are walked the same way, and the
responses meet in the oracle's restatement of that Block's ``process`` (oracle/ref_numpy.py) instead of the device's.
What core/graphs.py:157-190 + compute() do for the reference, minus dask -- so that views which need the planning
(time windows, request enlargement, Place, tilers) have an op-by-op CPU answer to be compared with."""
import numpy as np

from dask_geomodeling_amd import raster
from dask_geomodeling_amd.raster.base import RasterBlock
from oracle import ref_numpy as O

MATH = {"Add": "add", "Subtract": "subtract", "Multiply": "multiply", "Divide": "divide", "Power": "power", "Equal": "equal",
        "NotEqual": "not_equal", "Greater": "greater", "GreaterEqual": "greater_equal", "Less": "less", "LessEqual": "less_equal",
        "And": "logical_and", "Or": "logical_or", "Xor": "logical_xor", "Exp": "exp", "Log": "log", "Log10": "log10"}


def _memory_source(block, kwargs):
    """RasterSourceBase.process (sources.py:38-149) on the host copy of the source, GDAL's warp restated by the oracle."""
    mode = kwargs["mode"]
    if mode == "empty_vals":
        return None
    if mode == "empty_time":
        return {"time": []}
    if mode == "empty_meta":
        return {"meta": []}
    first, last = kwargs["bands"]
    if last < first and mode == "vals":
        raise ValueError("negative dimensions are not allowed")  # np.full((length, height, width)) of sources.py:119
    if mode == "time":
        return {"time": [kwargs["start"] + i * kwargs["delta"] for i in range(last - first)]}
    if mode == "meta":
        return {"meta": [block.metadata[i] for i in range(first, last)]}
    frames = np.asarray(block.data)[first:last]
    fill = block.fillvalue
    fill = fill.item() if hasattr(fill, "item") else fill
    bbox, height, width = kwargs["bbox"], kwargs["height"], kwargs["width"]
    if width == 0 or height == 0:
        return np.empty((last - first, height, width), dtype=frames.dtype)
    if bbox[0] == bbox[2] or bbox[1] == bbox[3]:
        values = O.point_read(frames, block.geo_transform, bbox[0], bbox[1], fill)
    else:
        values = O.nearest_warp(frames, block.geo_transform, bbox, height, width, fill)
    return {"values": values, "no_data_value": fill}


def _process(block, args):
    name = type(block).__name__
    if name == "MemorySource":
        return _memory_source(block, args[0])
    if name in MATH:
        return O.math_process(MATH[name], args[0], *args[1:])
    simple = {"Invert": O.invert_process, "IsData": O.isdata_process, "IsNoData": O.isnodata_process, "Clip": O.clip_process,
              "Mask": O.mask_process, "MaskBelow": O.maskbelow_process, "Step": O.step_process, "Classify": O.classify_process,
              "Reclassify": O.reclassify_process, "FillNoData": O.fillnodata_process, "Max": O.reduce_max_process,
              "Dilate": O.dilate_process, "MovingMax": O.movingmax_process, "Smooth": O.smooth_process,
              "TemporalAggregate": O.temporal_aggregate_process, "Cumulative": O.cumulative_process, "Snap": O.snap_process,
              "Shift": O.shift_process, "TemporalSum": O.temporal_sum_process, "Resample": O.resample_process,
              "Place": O.place_process, "RasterTiler": O.tiler_process, "Group": O.group_process}
    if name in simple:
        return simple[name](*args)
    if name == "HillShade":
        return O.hillshade_process(*args, trig="rounded")  # (host-independent trigonometry: see test_stencils_golden)
    if name in ("ArrayLeaf", "TemporalLeaf"):
        return type(block).process(*args)  # (test-only leaves, tests/blocks_util.py: host code)
    raise NotImplementedError("no oracle process for " + name)


def walk(block, request):
    """The response of ``block`` to ``request`` (mode vals / time / meta), computed Block by Block by the oracle."""
    pairs = block.get_sources_and_requests(**request)
    args = []
    for source, ask in pairs:
        if isinstance(source, RasterBlock) and ask is not None:
            args.append(walk(source, ask))
        else:
            args.append(source)
    with np.errstate(all="ignore"):
        return _process(block, args)


# ---------------------------------------------------------------------------------------------------------------------
# random temporal views
# ---------------------------------------------------------------------------------------------------------------------
from datetime import datetime, timedelta  # noqa: E402

T0 = datetime(2021, 3, 1)
SHAPE_YX = (24, 40)
BOX = dict(bbox=(0.0, 0.0, 40.0, 24.0), width=40, height=24, projection="EPSG:28992")
KINDS = [("f4", np.float32(np.finfo("f4").max)), ("f4", -9999.0), ("u1", np.uint8(255)), ("i2", np.int16(-1)), ("i4", -9999),
         ("f8", float(np.finfo("f8").max)), ("u2", np.uint16(65535)), ("i1", np.int8(-128)), ("u4", np.uint32(77)),
         ("i8", -9999), ("f4", np.float32(-3.5))]


# (uint32 with a small no-data value: TemporalSum adds no-data cells like any other, and seven times 4294967295 under a Max that
#  still declares uint32 is a float64 -> uint32 cast out of range -- undefined in C, 0 in this NumPy build, and not emulated;
#  the dtype's maximum as no data is covered by test_merges_of_wide_unsigned_rasters)
def temporal_leaf(rng, frames=None, step_hours=None, shape_yx=SHAPE_YX):
    dt, ndv = KINDS[int(rng.integers(len(KINDS)))]
    n = int(rng.integers(1, 9)) if frames is None else frames
    shape = (n,) + tuple(shape_yx)
    if np.dtype(dt).kind == "f":
        v = (rng.random(shape) * 40 - 10).astype(dt)
    else:
        v = rng.integers(0, 12, size=shape).astype(dt)
    v[rng.random(shape) < 0.15] = ndv
    step = timedelta(hours=int(step_hours if step_hours is not None else [1, 6, 24][int(rng.integers(3))]))
    first = T0 + timedelta(hours=int(rng.integers(0, 3))) * (1 if rng.random() < 0.15 else 0)  # (now and then off the others' grid)
    metadata = ["frame %d of %s" % (i, dt) for i in range(n)] if rng.random() < 0.5 else None
    if rng.random() < 0.3:  # resident in HBM from the start (zero-copy windows, strided point reads)
        from dask_geomodeling_amd.hip.array import DeviceArray
        if np.dtype(dt).kind == "f":
            v[~np.isfinite(v)] = ndv
        v = DeviceArray.from_numpy(v)
    return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, float(shape_yx[0])), time_first=first,
                               time_delta=step if n > 1 else None, metadata=metadata)


def random_temporal_view(rng, depth=3, shape_yx=SHAPE_YX):
    """A random tree over multi-frame MemorySources: element-wise steps, the temporal Blocks, Place and the stencils."""
    step_hours = [1, 6, 24][int(rng.integers(3))]
    leaves = [temporal_leaf(rng, step_hours=step_hours, shape_yx=shape_yx) for _ in range(int(rng.integers(1, 4)))]
    ny, nx = shape_yx

    def node(d):
        if d == 0 or rng.random() < 0.15:
            return leaves[int(rng.integers(len(leaves)))]
        a = node(d - 1)
        c = rng.random()
        if c < 0.22:
            op = ["Add", "Subtract", "Multiply", "Divide"][int(rng.integers(4))]
            b = node(d - 1) if rng.random() < 0.5 else [2.5, 3, -1.5, 0.5][int(rng.integers(4))]
            return getattr(raster, op)(a, b)
        if c < 0.30:
            b = node(d - 1)
            kind = rng.random()
            mask = (raster.Greater(b, [3, 5.5, 0][int(rng.integers(3))]) if kind < 0.5 else
                    raster.IsData(b) if kind < 0.65 else raster.Invert(raster.IsNoData(b)) if kind < 0.8 else b)
            return raster.Clip(a, mask)
        if c < 0.36:
            return raster.FillNoData(a, node(d - 1))
        if c < 0.42:
            return raster.Max(a, node(d - 1))
        if c < 0.47:
            return raster.Mask(a, [1, 7, 2.5][int(rng.integers(3))])
        if c < 0.49:
            return raster.Classify(a, [0.0, 5.0, 11.5, 20.0], bool(rng.random() < 0.5))
        if c < 0.52:
            kind = int(rng.integers(3))
            if kind == 0:
                return raster.MaskBelow(a, [0.0, 3, 5.5][int(rng.integers(3))])
            if kind == 1:
                return raster.Step(a, 1, 5, [2.0, 4][int(rng.integers(2))], 3)
            if a.dtype.kind in "iu":
                table = [[int(k), int(rng.integers(0, 50))] for k in rng.choice(14, size=int(rng.integers(1, 8)), replace=False)]
                return raster.Reclassify(a, table, bool(rng.random() < 0.4))
            return raster.MaskBelow(a, 2.5)
        if c < 0.56:
            return raster.Group(a, node(d - 1)) if rng.random() < 0.7 else raster.Group(a, node(d - 1), node(d - 1))
        if c < 0.60:
            return raster.Snap(a, node(d - 1))
        if c < 0.67:
            return raster.Shift(a, int(rng.integers(-3, 4)) * 3600 * 1000)
        if c < 0.73:
            return raster.TemporalSum(a)
        if c < 0.82:
            statistic = ["sum", "count", "min", "max", "mean", "median", "std", "p90"][int(rng.integers(8))]
            frequency = ["H", "3H", "D", "2D", None][int(rng.integers(5))]
            return raster.TemporalAggregate(a, frequency, statistic, closed=[None, "left", "right"][int(rng.integers(3))],
                                            label=[None, "left", "right"][int(rng.integers(3))])
        if c < 0.87:
            return raster.Cumulative(a, ["sum", "count"][int(rng.integers(2))], [None, "D", "2D"][int(rng.integers(3))])
        if c < 0.92:
            direction = ["nearest", "backward", "forward"][int(rng.integers(3))]
            return raster.Resample(a, ["H", "2H", "6H", "D"][int(rng.integers(4))], direction)
        if c < 0.96:
            anchor = (float(rng.integers(5, nx - 5)), float(rng.integers(5, ny - 4)))
            spots = [(float(rng.integers(0, nx)), float(rng.integers(0, ny))) for _ in range(int(rng.integers(1, 4)))]
            return raster.Place(a, "EPSG:28992", anchor, spots)
        kind = int(rng.integers(3))
        if kind == 0:
            return raster.MovingMax(a, [3, 5][int(rng.integers(2))])
        if kind == 1:
            return raster.Dilate(a, [int(x) for x in rng.integers(0, 12, size=2)])
        return raster.Smooth(a, 1.5, 0)

    for _ in range(30):
        try:
            view = node(depth)
            if rng.random() < 0.2:  # the tiles of a mosaic, each with the planning of the whole tree behind it
                exact = [t for t in (2, 4, 5, 8, 10, 20) if nx % t == 0], [t for t in (2, 3, 4, 6, 8, 12) if ny % t == 0]
                if rng.random() < 0.4 and exact[0] and exact[1]:  # tiles that cover the box exactly
                    tile = [int(nx // exact[0][int(rng.integers(len(exact[0])))]), int(ny // exact[1][int(rng.integers(len(exact[1])))])]
                else:
                    tile = ([int(rng.integers(4, max(nx - 10, 6))), int(rng.integers(4, max(ny + 6, 6)))]
                            if rng.random() < 0.7 else int(rng.integers(5, nx)))
                view = raster.RasterTiler(view, tile)
            view.dtype, view.fillvalue, view.period, view.timedelta  # attribute algebra must hold
            return view
        except (TypeError, ValueError):
            continue
    return leaves[0] + 1


def random_requests(rng, view, shape_yx=SHAPE_YX):
    """Time windows around the view's period (none: the latest frame; a moment; a window; windows that miss it) for the
    three request modes, over one of six kinds of box."""
    ny, nx = shape_yx
    whole = dict(bbox=(0.0, 0.0, float(nx), float(ny)), width=nx, height=ny, projection="EPSG:28992")
    period = view.period
    lo, hi = period if period is not None else (T0, T0)
    span = max((hi - lo).total_seconds() / 3600.0, 1.0)

    def moment():
        return lo + timedelta(hours=float(np.round(rng.uniform(-0.3, 1.3) * span * 2) / 2))

    asks = [dict(), dict(start=moment()), ]
    for _ in range(3):
        a, b = sorted([moment(), moment()])
        asks.append(dict(start=a, stop=b))
    asks.append(dict(start=lo - timedelta(days=30), stop=hi + timedelta(days=30)))
    out = []
    c = rng.random()
    if c < 0.55:
        box = whole
    elif c < 0.70:   # a window of the sources' grid
        x0, y0 = int(rng.integers(0, nx // 2)), int(rng.integers(0, ny // 2))
        w, h = int(rng.integers(1, nx // 2)), int(rng.integers(1, ny // 2))
        box = dict(whole, bbox=(float(x0), float(y0), float(x0 + w), float(y0 + h)), width=w, height=h)
    elif c < 0.80:   # coarser cells
        box = dict(whole, width=nx // 2, height=ny // 2)
    elif c < 0.88:   # finer cells
        box = dict(whole, bbox=(nx / 4.0, ny / 4.0, 3 * nx / 4.0, 3 * ny / 4.0))
    elif c < 0.93:   # half a cell off the grid and partly outside the sources
        box = dict(whole, bbox=(-7.5, -3.5, nx - 7.5, ny - 3.5))
    elif c < 0.96:   # one row, one column or a handful of cells
        x0, y0 = int(rng.integers(0, nx - 3)), int(rng.integers(0, ny - 3))
        w, h = [(nx - x0, 1), (1, ny - y0), (2, 2), (1, 1)][int(rng.integers(4))]
        box = dict(whole, bbox=(float(x0), float(y0), float(x0 + w), float(y0 + h)), width=w, height=h)
    else:            # a point
        x, y = float(rng.integers(0, nx)) + 0.5, float(rng.integers(0, ny)) + 0.5
        box = dict(whole, bbox=(x, y, x, y), width=1, height=1)
    for when in asks:
        out.append(dict(box, mode="vals", **when))
    out.append(dict(mode="time", **asks[int(rng.integers(len(asks)))]))
    out.append(dict(mode="meta", **asks[int(rng.integers(len(asks)))]))
    return out


def zero_sign_is_the_hosts(block):
    """Does the view hold a step whose result for a tie between -0.0 and +0.0 depends on the NumPy build and the host --
    np.fmax / np.fmin reductions (Max; TemporalAggregate min / max run SIMD minima whose tie rule is the instruction's)?"""
    if not isinstance(block, RasterBlock):
        return False
    name = type(block).__name__
    if name == "Max" or (name == "TemporalAggregate" and block.args[2] in ("min", "max")):
        return True
    return any(zero_sign_is_the_hosts(a) for a in block.args)


def comparable(result, view):
    """The response with -0.0 -> +0.0 where the sign of a zero is the host's business (see above); untouched otherwise."""
    if zero_sign_is_the_hosts(view) and isinstance(result, dict) and "values" in result \
            and np.asarray(result["values"]).dtype.kind == "f":
        return dict(result, values=np.asarray(result["values"]) + 0.0)
    return result


def without_lossy_cells(got, want):
    """Both responses with the cells zeroed in which a 64-bit integer result lies beyond 2^53 in either: the reference reaches
    such cells through float64 work arrays (lossy there) and casts that are out of range (undefined in C; INT64_MIN in this
    NumPy build) -- e.g. a TemporalSum over the int64 fill values of a Reclassify, aggregated again.  Nothing else is touched."""
    if not (isinstance(got, dict) and isinstance(want, dict) and "values" in got and "values" in want):
        return got, want
    g, w = np.asarray(got["values"]), np.asarray(want["values"])
    if g.shape != w.shape or g.dtype != w.dtype or g.dtype.kind not in "iu" or g.dtype.itemsize != 8:
        return got, want
    lossy = (np.abs(g.astype(np.float64)) >= 2.0 ** 53) | (np.abs(w.astype(np.float64)) >= 2.0 ** 53)
    if not lossy.any():
        return got, want
    return dict(got, values=np.where(lossy, 0, g).astype(g.dtype)), dict(want, values=np.where(lossy, 0, w).astype(w.dtype))
