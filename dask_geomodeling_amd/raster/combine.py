"""Group: combine rasters along x, y and time (counterpart of raster/combine.py).

The numerics are the paste-where-data merge of FillNoData (``target[index] = source[index]`` with
``index = utils.get_index``), one element-wise program per result band over the band slices of the
sources that contribute to it; everything else is the reference's request planning (which sources
are relevant for a time window, merge by time stamps or by band ranges)."""
import itertools
from datetime import timedelta as Timedelta

import numpy as np

from ..hip.array import DeviceArray, as_device
from ..utils import Extent, GeoTransform, get_dtype_max
from . import devops
from .base import RasterBlock

__all__ = ["Group"]


def filter_none(lst):
    return [x for x in lst if x is not None]


def _union_box(boxes):
    """Smallest (x1, y1, x2, y2) around all boxes."""
    x1, y1, x2, y2 = zip(*boxes)
    return min(x1), min(y1), max(x2), max(y2)


class BaseCombine(RasterBlock):
    """Greedy attributes (combine.py:16-141): period, extent and geometry are unions over the
    arguments that have any; the timedelta survives only when the arguments are equidistant and
    their frames fall on one common grid."""

    def __init__(self, *args):
        bad = [arg for arg in args if not isinstance(arg, RasterBlock)]
        if bad:
            raise TypeError("'{}' object is not allowed".format(type(bad[0])))
        super(BaseCombine, self).__init__(*args)

    @staticmethod
    def get_aligned_timedelta(sources):
        grid = [(src.timedelta, src.period[0]) for src in sources
                if src.period is not None and src.timedelta is not None]
        if not grid:
            return None
        step, origin = grid[0]
        for other_step, other_origin in grid[1:]:
            if other_step != step:
                return None
            # frames of every source must sit a whole number of steps from the first one's
            if (origin - other_origin).total_seconds() % step.total_seconds() != 0:
                return None
        return step

    def _once(self, name, compute):
        """A property of the (immutable) Block computed on first use: request planning reads period / timedelta / dtype
        several times per request."""
        slot = "_once_" + name
        if slot not in self.__dict__:
            self.__dict__[slot] = compute()
        return self.__dict__[slot]

    timedelta = property(lambda self: self._once("timedelta", lambda: self.get_aligned_timedelta(self.args)))
    temporal = property(lambda self: any(arg.temporal for arg in self.args))
    dtype = property(lambda self: self._once("dtype", lambda: np.result_type(*self.args)))
    fillvalue = property(lambda self: get_dtype_max(self.dtype))

    def _period(self):
        known = filter_none([arg.period for arg in self.args])
        if not known:
            return None
        return min(p[0] for p in known), max(p[1] for p in known)

    period = property(lambda self: self._once("period", self._period))

    @property
    def extent(self):
        known = filter_none([arg.extent for arg in self.args])
        return _union_box(known) if known else None

    @property
    def geometry(self):
        known = filter_none([arg.geometry for arg in self.args])
        if not known:
            return None
        joined = Extent.from_geometry(known[0])
        for geometry in known[1:]:
            joined = joined.union(Extent.from_geometry(geometry))
        return joined.as_geometry() if len(known) > 1 else known[0]

    @property
    def projection(self):
        first = self.args[0].projection
        if first is None or any(arg.projection != first for arg in self.args[1:]):
            return None
        return first

    @property
    def geo_transform(self):
        first = self.args[0].geo_transform
        if first is None:
            return None
        first = GeoTransform(first)
        for arg in self.args[1:]:
            other = arg.geo_transform
            if other is None or not first.aligns_with(other):
                return None
        return first


def _band(values, index):
    """Band ``index`` of a (bands, H, W) tile as a (1, H, W) tile, without a copy."""
    if isinstance(values, DeviceArray):
        if values.shape[0] == 1 and index == 0:
            return values  # (a zero-copy window of a resident raster stays one: the merge program reads it in place)
        if values.band_pitch is not None:
            return values.band_slice(index, index + 1)  # (... and so does one frame of a window over a series)
        return values.contiguous().band_slice(index, index + 1)
    return np.asarray(values)[index:index + 1]


def _stack_bands(bands, dtype, fill, shape_yx):
    """(1, H, W) device tiles (None = no source for that band) -> one (nbands, H, W) device tile."""
    import ctypes

    from ..hip import runtime
    from ..hip.dtypes import code_of
    out = DeviceArray.empty((len(bands),) + tuple(shape_yx), dtype)
    L = runtime.lib()
    plane = int(np.prod(shape_yx, dtype=np.int64))
    for i, band in enumerate(bands):
        dst = out.band_slice(i, i + 1)
        if band is None:
            runtime.check(L.gm_fill(ctypes.c_void_p(dst.ptr), code_of(dtype), runtime.scalar_bits(fill, dtype), plane), "gm_fill")
        else:
            band = as_device(band)
            runtime.check(L.gm_d2d(ctypes.c_void_p(dst.ptr), ctypes.c_void_p(band.ptr), plane * dtype.itemsize), "gm_d2d")
    return out


def _merge_bands(dtype, per_band_sources, shape_yx):
    """One merged (1, H, W) band per entry of ``per_band_sources`` (lists of (band tile, ndv), leftmost
    first: later sources are pasted over earlier ones where they have data) -> (nbands, H, W) tile."""
    dtype = np.dtype(dtype)
    fill = get_dtype_max(dtype)
    kwargs = {"dtype": dtype.str, "fillvalue": fill}
    bands = []
    for sources in per_band_sources:
        if sources:
            merged = devops.fillnodata_process(kwargs, *[{"values": v, "no_data_value": n} for v, n in sources])
            bands.append(merged["values"])
        else:
            bands.append(None)
    if len(bands) == 1 and bands[0] is not None:
        return bands[0]
    return _stack_bands(bands, dtype, fill, shape_yx)


class Group(BaseCombine):
    """Combine multiple rasters into a single one (combine.py:143-466).

    Values at equal timesteps are considered starting with the leftmost raster: rasters more 'to
    the right' are shown in the result, 'no data' is transparent."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    NOTHING = [(dict(combine_mode="simple"), None)]
    gm_direct = True  # plan.py: evaluated as process(sources' responses) without a graph per request

    # -- request planning -----------------------------------------------------------------
    def get_relevant_sources(self, start, stop):
        """The arguments that can contribute to the window [start, stop] (combine.py:164-196)."""
        dated = [(src.period[0], src.period[1], src) for src in self.args if src.period is not None]
        if not dated:
            return []
        if start is None:  # latest frame wanted: whoever reaches furthest in time
            horizon = max(last for _, last, _ in dated)
            return [src for _, last, src in dated if last == horizon]
        if stop is None:  # one moment: those that span it, else those whose edge lies nearest to it
            spanning = [src for first, last, src in dated if first <= start <= last]
            if spanning:
                return spanning
            firsts, lasts = [d[0] for d in dated], [d[1] for d in dated]
            nearest = min(firsts + lasts, key=lambda moment: abs(moment - start))
            order = [d[2] for d in dated]
            return [src for edge, src in zip(lasts + firsts, order + order) if edge == nearest]
        return [src for first, last, src in dated if not (stop < first or start > last)]

    def _plan_by_time(self, request, mode, start, stop):
        """No common time grid: frames are lined up by their time stamps, so besides the data every
        source is also asked for its times (combine.py:214-235)."""
        sources = self.get_relevant_sources(start, stop)
        if not sources:
            return self.NOTHING
        asks = [(src, request) for src in sources]
        if mode != "time":
            asks += [(src, dict(mode="time", start=start, stop=stop)) for src in sources]
        kwargs = dict(combine_mode="by_time", mode=mode, start=start, stop=stop)
        if mode == "vals":
            kwargs["dtype"] = self.dtype
        return [(kwargs, None)] + asks

    def _plan_by_bands(self, request, mode, start, stop, step):
        """Common time grid: the window is snapped to the grid and every source fills a range of
        result bands (combine.py:236-297)."""
        step_s = step.total_seconds()
        first, last = self.period
        if start is None:
            start = last
        elif start < first:
            start = first
        else:  # up to the next grid point
            start += Timedelta(seconds=(first - start).total_seconds() % step_s)
        if stop is None:
            stop = start
        elif stop > last:
            stop = last
        else:  # down to the previous grid point
            stop -= Timedelta(seconds=(stop - first).total_seconds() % step_s)
        if mode == "time":
            return [(dict(combine_mode="by_bands", mode=mode, start=start, stop=stop, timedelta=step), None)]

        def band_of(moment):
            return int((moment - start).total_seconds() // step_s)

        asks, ranges = [], []
        for src in self.get_relevant_sources(start, stop):
            lo, hi = max(start, src.period[0]), min(stop, src.period[1])
            ranges.append((band_of(lo), band_of(hi) + 1))
            asks.append((src, dict(request, start=lo, stop=hi)))
        kwargs = dict(combine_mode="by_bands", mode=mode, bands=ranges)
        n_bands = band_of(stop) + 1
        if mode == "meta":
            kwargs["nbands"] = n_bands
        if mode == "vals":
            kwargs["shape"] = n_bands, request["height"], request["width"]
            kwargs["dtype"] = self.dtype
        return [(kwargs, None)] + asks

    def get_sources_and_requests(self, **request):
        start, stop, mode = request.get("start", None), request.get("stop", None), request["mode"]
        period = self.period
        if period is None:
            return self.NOTHING
        if start is not None and stop is not None and (start > period[1] or stop < period[0]):
            return self.NOTHING
        step = self.timedelta
        if step is None:
            return self._plan_by_time(request, mode, start, stop)
        return self._plan_by_bands(request, mode, start, stop, step)

    # -- merging ----------------------------------------------------------------------------
    @staticmethod
    def _unique_times(responses):
        stamps = filter_none([r.get("time", None) for r in responses])
        return sorted(set(itertools.chain(*stamps)))

    @staticmethod
    def _single_frame(times, start, stop):
        """Index of the one frame a moment-request (stop None) is answered with, else None."""
        if stop is not None or len(times) <= 1:
            return None
        if start is None:
            return len(times) - 1
        return min(range(len(times)), key=lambda i: abs(times[i] - start))

    @staticmethod
    def _by_time(mode, responses, stamps, kwargs):
        times = Group._unique_times(stamps)
        slot = {moment: i for i, moment in enumerate(times)}
        if mode == "vals" and all(stamp["time"] == times and data["values"].shape[0] == len(times) for data, stamp in zip(responses, stamps)) \
                and Group._single_frame(times, kwargs["start"], kwargs["stop"]) is None:
            return Group._whole_stacks(responses, kwargs["dtype"])  # (every source has every frame: see _by_bands)
        if mode == "vals":
            frames = [[] for _ in times]
            for data, stamp in zip(responses, stamps):
                held = data["values"].shape[0]
                for k, moment in enumerate(stamp["time"]):
                    if k >= held:  # data["values"][k] of combine.py:331 (a source whose times and frames disagree)
                        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (k, held))
                    frames[slot[moment]].append((_band(data["values"], k), data["no_data_value"]))
        else:
            frames = [None] * len(times)
            for data, stamp in zip(responses, stamps):
                for k, moment in enumerate(stamp["time"]):
                    frames[slot[moment]] = data["meta"][k]
        only = Group._single_frame(times, kwargs["start"], kwargs["stop"])
        if only is not None:
            frames = frames[only:only + 1]
        if mode == "meta":
            return {"meta": frames}
        values = _merge_bands(kwargs["dtype"], frames, responses[0]["values"].shape[1:])
        return {"values": values, "no_data_value": get_dtype_max(kwargs["dtype"])}

    @staticmethod
    def _whole_stacks(responses, dtype):
        dtype = np.dtype(dtype)
        merged = devops.fillnodata_process({"dtype": dtype.str, "fillvalue": get_dtype_max(dtype)}, *responses)
        return {"values": merged["values"], "no_data_value": get_dtype_max(dtype)}

    @staticmethod
    def _by_bands(mode, responses, ranges, kwargs):
        if mode == "meta":
            merged = [""] * kwargs["nbands"]
            for data, (lo, hi) in zip(responses, ranges):
                for i, meta in zip(range(lo, hi), data["meta"]):
                    if meta:
                        merged[i] = meta
            return {"meta": merged}
        shape = kwargs["shape"]
        if shape[0] < 0:  # (a request whose stop lies before its start: np.full(shape, ...) of combine.py:384 refuses it)
            raise ValueError("negative dimensions are not allowed")
        if responses and all((lo, hi) == (0, shape[0]) and data["values"].shape[0] == shape[0] for data, (lo, hi) in zip(responses, ranges)):
            # every source fills every band (series on one grid over one window: the usual Group): the whole stacks are merged
            # by ONE program -- FillNoData's loop is the same paste, band by band or all at once -- instead of a program and a
            # copy per band (two series of 8 frames of 4096^2: 656 -> ~450 us per request)
            return Group._whole_stacks(responses, kwargs["dtype"])
        frames = [[] for _ in range(shape[0])]
        for data, (lo, hi) in zip(responses, ranges):
            held, span = data["values"].shape[0], max(min(hi, shape[0]) - lo, 0)
            if held == 0:
                continue  # values[lo:hi][index] with an empty index assigns nothing (combine.py:388-391)
            if held != span:  # ... and any other index of another length is NumPy's error
                raise IndexError("boolean index did not match indexed array along axis 0; size of axis is %d but size of "
                                 "corresponding boolean axis is %d" % (span, held))
            for i in range(lo, lo + span):
                frames[i].append((_band(data["values"], i - lo), data["no_data_value"]))
        values = _merge_bands(kwargs["dtype"], frames, shape[1:])
        return {"values": values, "no_data_value": get_dtype_max(kwargs["dtype"])}

    @staticmethod
    def process(process_kwargs, *args):
        plan, mode = process_kwargs["combine_mode"], process_kwargs.get("mode", None)
        if plan == "simple":
            return None
        if plan == "by_time":
            if mode == "time":
                times = Group._unique_times(args)
                only = Group._single_frame(times, process_kwargs["start"], process_kwargs["stop"])
                return {"time": times if only is None else times[only:only + 1]}
            if mode in ("meta", "vals"):
                half = len(args) // 2  # data responses, then the time responses in the same order
                responses, stamps = filter_none(args[:half]), filter_none(args[half:])
                if not responses:
                    return None
                return Group._by_time(mode, responses, stamps, process_kwargs)
        elif plan == "by_bands":
            if mode == "time":
                first, step = process_kwargs["start"], process_kwargs["timedelta"]
                count = int((process_kwargs["stop"] - first).total_seconds() // step.total_seconds()) + 1
                return {"time": [first + i * step for i in range(count)]}
            if mode in ("meta", "vals"):
                present = [(data, rng) for data, rng in zip(args, process_kwargs["bands"]) if data is not None]
                return Group._by_bands(mode, [d for d, _ in present], [r for _, r in present], process_kwargs)
        raise ValueError("Unknown combine_mode / mode combination")
