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


class BaseCombine(RasterBlock):
    """combine.py:16-141: attributes are greedy -- period and extent are unions of the arguments',
    the timedelta survives only when the arguments are equidistant and aligned."""

    def __init__(self, *args):
        for arg in args:
            if not isinstance(arg, RasterBlock):
                raise TypeError("'{}' object is not allowed".format(type(arg)))
        super(BaseCombine, self).__init__(*args)

    @staticmethod
    def get_aligned_timedelta(sources):
        timedeltas, periods = [], []
        for arg in sources:
            timedelta, period = arg.timedelta, arg.period
            if period is not None and timedelta is not None:
                timedeltas.append(timedelta)
                periods.append(period)
        if len(timedeltas) == 0:
            return None
        if len(timedeltas) == 1:
            return timedeltas[0]
        if not timedeltas[1:] == timedeltas[:-1]:
            return None
        # the periods must be spaced an integer number of timedeltas apart
        seconds = timedeltas[0].total_seconds()
        first, _ = periods[0]
        for a, _ in periods[1:]:
            if (first - a).total_seconds() % seconds != 0:
                return None
        return timedeltas[0]

    @property
    def timedelta(self):
        return self.get_aligned_timedelta(self.args)

    @property
    def temporal(self):
        return any(x.temporal for x in self.args)

    @property
    def period(self):
        periods = filter_none([x.period for x in self.args])
        if len(periods) == 0:
            return None
        if len(periods) == 1:
            return periods[0]
        return min(p[0] for p in periods), max(p[1] for p in periods)

    @property
    def extent(self):
        extents = filter_none([x.extent for x in self.args])
        if len(extents) == 0:
            return None
        if len(extents) == 1:
            return extents[0]
        return (min(e[0] for e in extents), min(e[1] for e in extents),
                max(e[2] for e in extents), max(e[3] for e in extents))

    @property
    def dtype(self):
        return np.result_type(*self.args)

    @property
    def fillvalue(self):
        return get_dtype_max(self.dtype)

    @property
    def geometry(self):
        geometries = filter_none([x.geometry for x in self.args])
        if len(geometries) == 0:
            return None
        if len(geometries) == 1:
            return geometries[0]
        extent = Extent.from_geometry(geometries[0])
        for geometry in geometries[1:]:
            extent = extent.union(Extent.from_geometry(geometry))
        return extent.as_geometry()

    @property
    def projection(self):
        projection = self.args[0].projection
        if projection is None:
            return None
        for arg in self.args[1:]:
            if projection != arg.projection:
                return None
        return projection

    @property
    def geo_transform(self):
        geo_transform = self.args[0].geo_transform
        if geo_transform is None:
            return None
        geo_transform = GeoTransform(geo_transform)
        for arg in self.args[1:]:
            other = arg.geo_transform
            if other is None or not geo_transform.aligns_with(other):
                return None
        return geo_transform


def _band(values, index):
    """Band ``index`` of a (bands, H, W) tile as a (1, H, W) tile, without a copy."""
    if isinstance(values, DeviceArray):
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

    def get_relevant_sources(self, start, stop):
        stores = [s for s in self.args if s.period is not None]
        if not stores:
            return []
        starts, stops = zip(*(s.period for s in stores))
        if start is None:  # pick the latest store(s) in time
            last = max(stops)
            return [s for b, s in zip(stops, stores) if b == last]
        if stop is None:  # any stores that contain start, else the closest ones
            result = [s for a, b, s in zip(starts, stops, stores) if a <= start <= b]
            if result:
                return result
            closest = min(starts + stops, key=lambda d: abs(d - start))
            return [s for d, s in zip(stops + starts, stores + stores) if d == closest]
        return [s for a, b, s in zip(starts, stops, stores) if not (stop < a or start > b)]

    def get_sources_and_requests(self, **request):
        start, stop = request.get("start", None), request.get("stop", None)
        mode = request["mode"]
        if self.period is None:
            return [(dict(combine_mode="simple"), None)]
        if start is not None and stop is not None and (start > self.period[1] or stop < self.period[0]):
            return [(dict(combine_mode="simple"), None)]
        timedelta = self.timedelta
        if timedelta is None:  # merge by time
            sources = self.get_relevant_sources(start, stop)
            if not sources:
                return [(dict(combine_mode="simple"), None)]
            requests, time_requests = [], []
            for source in sources:
                requests.append((source, request))
                if mode != "time":  # the time stamps of the frames are needed to line them up
                    time_requests.append((source, dict(mode="time", start=start, stop=stop)))
            process_kwargs = dict(combine_mode="by_time", mode=mode, start=start, stop=stop)
            requests = requests + time_requests
        else:  # merge by bands
            td_sec = timedelta.total_seconds()
            period = self.period
            origin = period[0]
            if start is None:
                start = period[1]
            elif start < period[0]:
                start = period[0]
            else:  # ceil start to the closest integer timedelta
                start += Timedelta(seconds=(origin - start).total_seconds() % td_sec)
            if stop is None:
                stop = start
            elif stop > period[1]:
                stop = period[1]
            else:  # floor stop to the closest integer timedelta
                stop -= Timedelta(seconds=(stop - origin).total_seconds() % td_sec)
            if mode == "time":
                return [(dict(combine_mode="by_bands", mode=mode, start=start, stop=stop, timedelta=timedelta), None)]
            sources = self.get_relevant_sources(start, stop)
            requests, bands = [], []
            for source in sources:
                this_start = max(start, source.period[0])
                this_stop = min(stop, source.period[1])
                first_i = int((this_start - start).total_seconds() // td_sec)
                last_i = int((this_stop - start).total_seconds() // td_sec)
                bands.append((first_i, last_i + 1))
                this_request = request.copy()
                this_request.update(start=this_start, stop=this_stop)
                requests.append((source, this_request))
            process_kwargs = dict(combine_mode="by_bands", mode=mode, bands=bands)
            nbands = int((stop - start).total_seconds() // td_sec) + 1
            if mode == "meta":
                process_kwargs["nbands"] = nbands
            if mode == "vals":
                process_kwargs["shape"] = nbands, request["height"], request["width"]
        if mode == "vals":
            process_kwargs["dtype"] = self.dtype
        return [(process_kwargs, None)] + requests

    @staticmethod
    def _unique_times(multi):
        times = filter_none([data.get("time", None) for data in multi])
        return sorted(set(itertools.chain(*times)))

    @staticmethod
    def _nearest_index(time, start):
        if start is None:
            return len(time) - 1
        return min(enumerate(time), key=lambda d: abs(d[1] - start))[0]

    @staticmethod
    def _merge_vals_by_time(multi, times, kwargs):
        sorted_times = Group._unique_times(times)
        bands = dict((y, x) for x, y in enumerate(sorted_times))
        per_band = [[] for _ in sorted_times]
        for data, time in zip(multi, times):
            for source_index, datetime in enumerate(time["time"]):
                per_band[bands[datetime]].append((_band(data["values"], source_index), data["no_data_value"]))
        start, stop = kwargs["start"], kwargs["stop"]
        if stop is None and len(sorted_times) > 1:  # single band result required
            index = Group._nearest_index(sorted_times, start)
            per_band = per_band[index:index + 1]
        shape_yx = multi[0]["values"].shape[1:]
        values = _merge_bands(kwargs["dtype"], per_band, shape_yx)
        return {"values": values, "no_data_value": get_dtype_max(kwargs["dtype"])}

    @staticmethod
    def _merge_meta_by_time(multi, times, kwargs):
        sorted_times = Group._unique_times(times)
        bands = dict((y, x) for x, y in enumerate(sorted_times))
        meta_result = [None] * len(sorted_times)
        for data, time in zip(multi, times):
            for source_index, datetime in enumerate(time["time"]):
                meta_result[bands[datetime]] = data["meta"][source_index]
        start, stop = kwargs["start"], kwargs["stop"]
        if stop is None and len(sorted_times) > 1:
            index = Group._nearest_index(sorted_times, start)
            meta_result = meta_result[index:index + 1]
        return {"meta": meta_result}

    @staticmethod
    def _merge_vals_by_bands(multi, bands, dtype, shape):
        per_band = [[] for _ in range(shape[0])]
        for data, (a, b) in zip(multi, bands):
            for i in range(a, b):
                per_band[i].append((_band(data["values"], i - a), data["no_data_value"]))
        values = _merge_bands(dtype, per_band, shape[1:])
        return {"values": values, "no_data_value": get_dtype_max(dtype)}

    @staticmethod
    def _merge_meta_by_bands(multi, bands, nbands):
        meta_result = [""] * nbands
        for data, (a, b) in zip(multi, bands):
            for i, meta in zip(range(a, b), data["meta"]):
                if meta:
                    meta_result[i] = meta
        return {"meta": meta_result}

    @staticmethod
    def process(process_kwargs, *args):
        combine_mode = process_kwargs["combine_mode"]
        mode = process_kwargs.get("mode", None)
        if combine_mode == "simple":
            return None
        if combine_mode == "by_time" and mode == "time":
            sorted_times = Group._unique_times(args)
            start, stop = process_kwargs["start"], process_kwargs["stop"]
            if stop is None and len(sorted_times) > 1:
                index = Group._nearest_index(sorted_times, start)
                sorted_times = sorted_times[index:index + 1]
            return {"time": sorted_times}
        if combine_mode == "by_time" and mode in ["meta", "vals"]:
            n = int(len(args) // 2)
            multi, times = filter_none(args[:n]), filter_none(args[n:])  # None sits at the same positions
            if len(multi) == 0:
                return None
            if mode == "vals":
                return Group._merge_vals_by_time(multi, times, process_kwargs)
            return Group._merge_meta_by_time(multi, times, process_kwargs)
        if combine_mode == "by_bands" and mode == "time":
            start, stop, delta = process_kwargs["start"], process_kwargs["stop"], process_kwargs["timedelta"]
            length = int((stop - start).total_seconds() // delta.total_seconds()) + 1  # includes the last frame
            return {"time": [start + i * delta for i in range(length)]}
        if combine_mode == "by_bands" and mode in ["meta", "vals"]:
            multi, bands = [], []
            for data, _bands in zip(args, process_kwargs["bands"]):
                if data is None:
                    continue
                multi.append(data)
                bands.append(_bands)
            if mode == "vals":
                return Group._merge_vals_by_bands(multi=multi, bands=bands, dtype=process_kwargs["dtype"],
                                                  shape=process_kwargs["shape"])
            return Group._merge_meta_by_bands(multi=multi, bands=bands, nbands=process_kwargs["nbands"])
        raise ValueError("Unknown combine_mode / mode combination")
