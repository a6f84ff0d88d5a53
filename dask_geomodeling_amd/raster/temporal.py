"""Blocks over the time axis (counterpart of raster/temporal.py): ``TemporalAggregate`` resamples a raster in
time with a statistic (hourly mean of 5-minute frames, ...), ``Cumulative`` accumulates frames inside every period;
``Snap`` / ``Resample`` re-arrange frames (every result frame is the nearest / next / previous source frame),
``Shift`` moves the time axis, ``TemporalSum`` adds all frames up.

Which frames belong to which label is calendar work and stays on the host, done with pandas exactly
as the reference does (``Series.resample(...).indices``); a source's frames ascend in time, so every
label owns one run of consecutive frames.  The per-pixel reduction over those runs is the device
kernel (csrc/gm_temporal.hip): every frame is read once, every label written once."""
import ctypes
import functools
import warnings
from datetime import timedelta as Timedelta

import numpy as np
import pandas as pd
import pytz
from pandas.tseries.frequencies import to_offset

from ..utils import get_dtype_max
from .base import BaseSingle, RasterBlock

__all__ = ["Snap", "Shift", "TemporalSum", "TemporalAggregate", "Cumulative", "Resample"]

# frequencies whose bins pandas closes and labels on the right by default (TimeGrouper.__init__)
_END_ANCHORED = {"M", "A", "Q", "BM", "BA", "BQ", "W", "ME", "YE", "QE", "BME", "BYE", "BQE"}
_TICK = Timedelta(microseconds=1)
STATISTIC_CODES = {"sum": 0, "count": 1, "min": 2, "max": 3, "mean": 4, "var": 5, "std": 6, "median": 7}
MAX_MEDIAN_FRAMES = 599  # GM_MAX_MEDIAN_FRAMES_LONG: a cell's frames are sorted in LDS (float64: 320)
_EXTENSIVE = {"sum", "count"}  # additive statistics: nothing to add up = 0, not 'no data'


@functools.lru_cache(maxsize=256)
def normalize_offset(freq):
    """A pandas offset string in the spelling of the running pandas (utils.py:979-1013).  (Remembered: the Blocks ask per
    request and per tile, and pandas takes its time to parse a rule.)"""
    if freq is None:
        return None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)
        return to_offset(freq).freqstr


@functools.lru_cache(maxsize=256)
def offset_to_timedelta(freq):
    """The fixed length of a frequency, None for calendar-dependent ones (utils.py:960-976)."""
    try:
        return pd.Timedelta(to_offset(freq).nanos, unit="ns").to_pytimedelta()
    except (ValueError, AttributeError):
        return None


def parse_statistic(statistic):
    """'p<number>' -> ("percentile", number), anything else -> (statistic, None) (utils.py:813-823)."""
    import re
    found = re.findall(r"^p(\d+(?:\.\d+)?)$", statistic)
    if found:
        q = float(found[0])
        if not 0 <= q <= 100:
            raise ValueError("Percentiles must be in the range [0, 100]")
        return "percentile", q
    return statistic, None


def dtype_for_statistic(dtype, statistic):
    """Result dtype of a statistic (utils.py:826-845): min / max keep the dtype, sum follows Add,
    count is int32, the rest follows Divide."""
    dtype = np.dtype(dtype)
    if statistic in ("min", "max"):
        return dtype
    if statistic == "sum":
        if dtype.kind in "iub":
            return np.result_type(dtype, np.int32)
        if dtype.kind == "f":
            return np.result_type(dtype, np.float32)
        return dtype
    if statistic == "count":
        return np.dtype(np.int32)
    return np.result_type(np.float32, dtype)


def _remembered(method):
    """Calendar answers depend on the rule and the moments asked about, nothing else -- and pandas takes 100 to 400
    microseconds for each (a one-sample Series, a resampler, its bins).  Every tile of a temporal view asks the same
    questions, so the answers are kept (bounded); a key that does not hash is answered afresh."""
    answers = {}

    def hashable(a):
        if isinstance(a, (list, pd.DatetimeIndex)):
            return tuple(a)
        if isinstance(a, (int, float)) and not isinstance(a, bool):
            return (type(a).__name__, a)  # (1 and 1.0 hash alike but mean different things to shifted(): periods vs a fraction)
        return a

    def wrapper(self, *args):
        try:
            key = (self.frequency, self.closed, self.label, self.timezone) + tuple(hashable(a) for a in args)
            hit = answers.get(key, answers)
        except TypeError:
            return method(self, *args)
        if hit is answers:
            if len(answers) >= 4096:
                answers.clear()
            hit = answers[key] = method(self, *args)
        return list(hit) if isinstance(hit, list) else hit  # (a list is handed out as a copy)

    wrapper.__name__, wrapper.__doc__ = method.__name__, method.__doc__
    return wrapper


@functools.lru_cache(maxsize=256)
def _calendar_for(frequency, closed, label, timezone):
    """The (stateless) calendar of a rule, made once: every request and every tile of a temporal view asks for it."""
    return _Calendar(frequency, closed, label, timezone)


class _Calendar:
    """The bins of one resampling rule: naive-UTC datetimes in and out, pandas in between."""

    def __init__(self, frequency, closed, label, timezone):
        self.frequency, self.timezone = frequency, timezone
        if frequency is None:
            self.closed = self.label = "right"
        else:
            rule = to_offset(frequency).rule_code
            anchored = rule in _END_ANCHORED or rule.split("-")[0] in _END_ANCHORED
            self.closed = closed or ("right" if anchored else "left")
            self.label = label or ("right" if anchored else "left")

    def kwargs(self):
        return {"frequency": self.frequency, "closed": self.closed, "label": self.label, "timezone": self.timezone}

    # -- conversions ---------------------------------------------------------------------
    def stamp(self, dt):
        """naive UTC datetime -> Timestamp in the rule's time zone (temporal.py:257-259)."""
        return pd.Timestamp(dt, tz="UTC").tz_convert(self.timezone)

    def moment(self, ts):
        """Timestamp -> naive UTC datetime (temporal.py:262-268)."""
        if ts.tzinfo is None:
            ts = ts.tz_localize(self.timezone)
        return ts.tz_convert("UTC").tz_localize(None).to_pydatetime(warn=False)

    # -- single moments ------------------------------------------------------------------
    @_remembered
    def label_of(self, dt, closed=None, label=None):
        """Label of the bin ``dt`` falls in (a one-sample resample; temporal.py:271-284)."""
        probe = pd.Series([0], index=[self.stamp(dt)])
        found = None
        for found, members in probe.resample(self.frequency, closed=closed or self.closed, label=label or self.label):
            if len(members):
                break
        return self.moment(found)

    @_remembered
    def bin_start_of(self, dt):
        """Left edge (Timestamp) of the bin ``dt`` falls in (temporal.py:287-296)."""
        probe = pd.Series([0], index=[self.stamp(dt)])
        return probe.resample(self.frequency, closed=self.closed, label="left").first().index[0]

    @_remembered
    def nearest_label(self, dt, side):
        """The label closest to ``dt``; side 'left' / 'right' only looks that way (temporal.py:321-347)."""
        ts = self.stamp(dt)
        step = to_offset(self.frequency)
        centre = self.stamp(self.label_of(dt, "left", "left"))
        around = pd.date_range(centre - step, centre + step, freq=step)
        gaps = (around - ts).to_series()
        gaps.index = around
        if side == "right":
            gaps = gaps[gaps >= pd.Timedelta(0)]
        elif side == "left":
            gaps = gaps[gaps <= pd.Timedelta(0)]
        return self.moment(gaps.abs().idxmin())

    @_remembered
    def first_moment(self, label):
        """First moment inside the bin labelled ``label`` (temporal.py:373-380)."""
        ts = self.stamp(label)
        if self.label == "right":
            ts -= to_offset(self.frequency)
        if self.closed == "right":
            ts += _TICK
        return self.moment(ts)

    @_remembered
    def last_moment(self, label):
        """Last moment inside the bin labelled ``label`` (temporal.py:383-390)."""
        ts = self.stamp(label)
        if self.label == "left":
            ts += to_offset(self.frequency)
        if self.closed == "left":
            ts -= _TICK
        return self.moment(ts)

    @_remembered
    def shifted(self, dt, n):
        """``dt`` moved by n periods of the rule, n possibly fractional: a fraction of a calendar period (a month)
        is that fraction of the period it falls in (temporal.py:298-319)."""
        if n == 0:
            return dt
        ts = self.stamp(dt)
        step = to_offset(self.frequency)
        if isinstance(n, float):
            whole = int(n // 1.0)
            part = n % 1.0 if n > 0 else -(n % 1.0)
            ts = ts + ((ts + step * (whole + 1)) - (ts + step * whole)) * part
        else:
            ts = ts + n * step
        return self.moment(ts)

    @_remembered
    def label_range(self, first, last):
        """All labels from ``first`` to ``last`` inclusive; [first] when last is None (temporal.py:463-473)."""
        if last is None:
            return [first]
        stamps = pd.date_range(start=self.stamp(first), end=self.stamp(last), freq=to_offset(self.frequency))
        return [self.moment(ts) for ts in stamps]

    # -- periods and windows -------------------------------------------------------------
    @_remembered
    def resampled_period(self, period):
        """(first label, last label) of the resampled raster (temporal.py:393-402)."""
        if period is None:
            return None
        if self.frequency is None:
            return period[-1], period[-1]
        return tuple(self.label_of(moment) for moment in period)

    @_remembered
    def snap_window(self, period, start, stop):
        """Requested (start, stop) -> (first label, last label) inside ``period``; (label, None) for a
        one-moment request, (None, None) when no label is in range (temporal.py:405-455)."""
        if period is None:
            return None, None
        first, last = period
        if start is None:
            start = last
        if stop is None:
            if start <= first:
                return first, None
            if start >= last:
                return last, None
            return self.nearest_label(start, "both"), None
        if start > last or stop < first:
            return None, None
        start = first if start <= first else self.nearest_label(start, "right")
        stop = last if stop >= last else self.nearest_label(stop, "left")
        if start > stop:
            return None, None
        return start, stop

    @_remembered
    def labels_between(self, start, stop):
        """The labels from ``start`` to ``stop`` (start alone when stop is None, or without a frequency): as naive
        UTC stamps and in the rule's time zone (temporal.py:696-704)."""
        if self.frequency is None:
            labels = pd.DatetimeIndex([start])
            return labels, labels.tz_localize("UTC")
        labels = pd.date_range(start, stop or start, freq=self.frequency)
        return labels, labels.tz_localize("UTC").tz_convert(self.timezone)

    @_remembered
    def periods_of(self, times, start, stop):
        """The frames of every accumulation period of ``times`` (index arrays, in order) and which frames lie between
        ``start`` and ``stop`` (boolean per frame): Cumulative's bookkeeping (temporal.py:940-975)."""
        stamps = pd.Series(index=times, dtype=float).tz_localize("UTC").tz_convert(self.timezone)
        if self.frequency is None:
            members = (np.arange(len(times)),)
        else:
            resampled = stamps.resample(self.frequency, closed=self.closed, label=self.label)
            members = tuple(np.asarray(frames, dtype=np.int64) for frames in resampled.indices.values())
        wanted = np.asarray((stamps.index >= self.stamp(start)) & (stamps.index <= self.stamp(stop)))
        return members, wanted

    @_remembered
    def frames_of(self, times, labels):
        """[(first frame, count)] per label: the run of ``times`` (naive UTC, ascending) that falls in the
        bin of every label (pandas' Resampler.indices, temporal.py:706-722)."""
        if self.frequency is None:
            return [(0, len(times))]
        series = pd.Series(index=times, dtype=float).tz_localize("UTC").tz_convert(self.timezone)
        members = series.resample(self.frequency, closed=self.closed, label=self.label).indices
        runs = []
        for stamp in labels:
            frames = np.asarray(members.get(stamp, ()), dtype=np.int64)
            if len(frames) and not np.array_equal(frames, np.arange(frames[0], frames[0] + len(frames))):
                raise NotImplementedError("time stamps of the source are not ascending")
            runs.append((int(frames[0]), len(frames)) if len(frames) else (0, 0))
        return runs


class GmFrameBin(ctypes.Structure):
    """gm_frame_bin of include/gmhip.h."""
    _fields_ = [("first", ctypes.c_int32), ("count", ctypes.c_int32), ("out", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def _frame_bins(runs, outs=None):
    bins = (GmFrameBin * max(len(runs), 1))()
    for i, (first, count) in enumerate(runs):
        bins[i].first, bins[i].count, bins[i].out = first, count, i if outs is None else outs[i]
    return bins


_declared = False


def _lib():
    """The loaded library with the temporal entry points declared (once)."""
    global _declared
    from ..hip import runtime
    L = runtime.lib()
    if not _declared:
        c = ctypes
        common = [c.c_void_p, c.c_int, c.c_int, c.c_uint64, c.c_int64, c.c_int64, c.c_void_p, c.c_int]
        L.gm_temporal_out_dtype.argtypes = [c.c_int, c.c_int]
        L.gm_temporal_out_dtype.restype = c.c_int
        L.gm_temporal_aggregate.argtypes = common + [c.c_int, c.c_int, c.c_void_p, c.c_uint64]
        L.gm_temporal_aggregate.restype = c.c_int
        L.gm_temporal_percentile.argtypes = common + [c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_uint64]
        L.gm_temporal_percentile.restype = c.c_int
        L.gm_temporal_cumulative.argtypes = common + [c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_uint64]
        L.gm_temporal_cumulative.restype = c.c_int
        _declared = True
    return L


def _reduce_args(values, no_data_value):
    from ..hip import runtime
    from ..hip.array import as_device
    from ..hip.dtypes import code_of
    from ..hip.program import effective_ndv
    src = as_device(values)
    ndv = effective_ndv(src.dtype, no_data_value)
    has, bits = (0, 0) if ndv is None else (1, runtime.scalar_bits(ndv, src.dtype))
    frames, height, width = src.shape
    return src, code_of(src.dtype), has, bits, frames, height, width


class GmPercentileStep(ctypes.Structure):
    """gm_percentile_step of include/gmhip.h."""
    _fields_ = [("lower", ctypes.c_int32), ("upper", ctypes.c_int32), ("gamma", ctypes.c_double),
                ("one_minus_gamma", ctypes.c_double)]


def percentile_plan(q, longest, work_dtype):
    """What np.nanpercentile(..., q) (method 'linear') takes from n = 1 .. ``longest`` sorted values:
    the two neighbours of the virtual index (n - 1) * (q / 100) and the interpolation weight, all
    rounded the way NumPy's own scalar arithmetic in ``work_dtype`` rounds them
    (numpy/lib/_function_base_impl.py: _quantile, _get_indexes, _get_gamma, _lerp)."""
    number = np.dtype(work_dtype).type
    quantile = np.true_divide(float(q), number(100))  # nanpercentile: q / a.dtype.type(100)
    plan = (GmPercentileStep * longest)()
    for n in range(1, longest + 1):
        virtual = (n - 1) * quantile
        lower = int(np.floor(virtual))
        upper = lower + 1
        if virtual >= n - 1:  # at or past the last value: both neighbours are the last one (index -1)
            lower = upper = -1
        gamma = number(np.float64(virtual) - lower)  # (the difference is taken in float64, then cast back)
        step = plan[n - 1]
        step.lower, step.upper = lower % n, upper % n
        step.gamma, step.one_minus_gamma = float(gamma), float(1 - gamma)
    return plan


def _sums_as_int32(src, no_data_value, kernel_statistic, dtype):
    """The 64-bit integer frames a TemporalSum of an integer series hands on (under a dtype declared for its store:
    temporal.py:548-602) as int32 -- or float32 -- frames, _reduce_args of them, when all values lie within +-2^24 and the
    kernels of such frames give the declared ``dtype``.  The reference works on values.astype(result_type(float32, dtype))
    whatever the frames' own dtype (temporal.py:738-741), which is what the kernels derive from ``dtype`` too: the same
    numbers in, the same statistic out.  Else None."""
    from ..hip import program as P
    if not P._wide_int(src.dtype):
        return None
    for to in ("i4", "f4", "i2", "u2", "i1", "u1"):
        if _lib().gm_temporal_out_dtype(_code(to), kernel_statistic) == _code(dtype):
            if to == "f4" and abs(int(no_data_value if no_data_value is not None else 0)) > P._NARROW_LIMIT:
                continue  # (a no-data value float32 would round)
            narrow = P._narrowed(src, no_data_value, to)
            if narrow is not None:
                return _reduce_args(*narrow)
    return None


def _fill_bits(statistic, dtype):
    from ..hip import runtime
    return runtime.scalar_bits(0 if statistic in _EXTENSIVE else get_dtype_max(dtype), dtype)


def aggregate_frames(values, no_data_value, runs, statistic, dtype, fill=None, keep_nonfinite=False):
    """(frames, h, w) tile + [(first, count)] per label -> (labels, h, w) device tile of ``dtype``
    (the loop of temporal.py:743-766 as one launch).  ``fill``: what a label without a finite result gets
    (default: the reference's, 0 for the additive statistics, else the dtype's maximum).  ``keep_nonfinite``
    (median / percentiles): only the cells without a data frame get ``fill``, an infinite or NaN result stays
    (what reduce_rasters does, reduction.py:113-115)."""
    from ..hip import runtime
    from ..hip.array import DeviceArray
    name, percentile = parse_statistic(statistic)
    if name not in STATISTIC_CODES and percentile is None:
        raise NotImplementedError("TemporalAggregate statistic %r" % statistic)
    src, code, has, bits, frames, height, width = _reduce_args(values, no_data_value)
    if keep_nonfinite and name != "median" and percentile is None:
        raise NotImplementedError("keep_nonfinite with statistic %r" % statistic)
    longest = max([count for _, count in runs] + [1])
    # (median / percentiles: a cell's data frames are sorted in LDS up to MAX_MEDIAN_FRAMES float32 / 320 float64
    #  frames per label; longer bins -- a year of hourly frames -- are selected in global memory: no limit)
    L = _lib()
    kernel_statistic = STATISTIC_CODES["median" if percentile is not None else name]
    if L.gm_temporal_out_dtype(code, kernel_statistic) != _code(dtype):
        narrow = _sums_as_int32(src, no_data_value, kernel_statistic, dtype)
        if narrow is None:
            raise NotImplementedError("%s of a %s raster as %s" % (statistic, src.dtype, np.dtype(dtype)))
        src, code, has, bits, frames, height, width = narrow
    if keep_nonfinite:
        has |= 2  # GM_TEMPORAL_KEEP_NONFINITE (after the narrowing, which hands back flags of its own)
    out = DeviceArray.empty((len(runs), height, width), np.dtype(dtype))
    fill_bits = _fill_bits(statistic, dtype) if fill is None else runtime.scalar_bits(fill, np.dtype(dtype))
    if percentile is not None:
        plan = percentile_plan(percentile, longest, dtype)
        runtime.check(L.gm_temporal_percentile(src.ptr, code, has, bits, frames, height * width, _frame_bins(runs), len(runs),
                                               len(runs), plan, longest, out.ptr, fill_bits),
                      "gm_temporal_percentile")
        return out
    runtime.check(L.gm_temporal_aggregate(src.ptr, code, has, bits, frames, height * width, _frame_bins(runs), len(runs),
                                          len(runs), STATISTIC_CODES[statistic], out.ptr, fill_bits),
                  "gm_temporal_aggregate")
    return out


def accumulate_frames(values, no_data_value, runs, emit, statistic, dtype):
    """Running sum / count inside every run; frames emit[0] .. emit[1]-1 are the result
    (temporal.py:979-1001)."""
    from ..hip import runtime
    from ..hip.array import DeviceArray
    if statistic not in ("sum", "count"):
        raise NotImplementedError("Cumulative statistic %r" % statistic)
    src, code, has, bits, frames, height, width = _reduce_args(values, no_data_value)
    L = _lib()
    if L.gm_temporal_out_dtype(code, STATISTIC_CODES[statistic]) != _code(dtype):
        narrow = _sums_as_int32(src, no_data_value, STATISTIC_CODES[statistic], dtype)
        if narrow is None:
            raise NotImplementedError("cumulative %s of a %s raster as %s" % (statistic, src.dtype, np.dtype(dtype)))
        src, code, has, bits, frames, height, width = narrow
    out = DeviceArray.empty((emit[1] - emit[0], height, width), np.dtype(dtype))
    runtime.check(L.gm_temporal_cumulative(src.ptr, code, has, bits, frames, height * width, _frame_bins(runs), len(runs),
                                           int(emit[0]), int(emit[1]), STATISTIC_CODES[statistic], out.ptr,
                                           _fill_bits(statistic, dtype)), "gm_temporal_cumulative")
    return out


def _code(dtype):
    from ..hip.dtypes import code_of
    return code_of(np.dtype(dtype))


def _expect(value, kind):
    """``value`` if it is a ``kind``; otherwise the reference's refusal of a constructor argument of the wrong type."""
    if not isinstance(value, kind):
        raise TypeError("'{}' object is not allowed.".format(type(value)))
    return value


def _rule_arguments(frequency, timezone, closed=None, label=None, sides=False):
    """(frequency, timezone[, closed, label]) of a calendar rule as the Blocks keep them: the offset string normalised, the
    time zone by its canonical name; ``closed`` / ``label`` one of None, 'left', 'right'."""
    frequency = normalize_offset(_expect(frequency, str))
    if sides:
        for name, side in (("closed", closed), ("label", label)):
            if side not in {None, "left", "right"}:
                raise ValueError("{} must be None, 'left', or 'right'.".format(name))
    timezone = pytz.timezone(_expect(timezone, str)).zone
    return (frequency, timezone, closed, label) if sides else (frequency, timezone)


def _check_statistic(statistic, known):
    statistic, percentile = parse_statistic(_expect(statistic, str).lower())
    if percentile:
        return "p{0}".format(percentile)
    if statistic not in known:
        raise ValueError("Unknown statistic '{}'".format(statistic))
    return statistic



def find_neighbours(array, value, direction="nearest"):
    """Index into the ascending ``array`` of the neighbour of every ``value``: the nearest one (ties go to the
    earlier element), the first one not before it ('forward') or the last one not after it ('backward'); never out
    of bounds (utils.py:929-958, ``find_neigbours``)."""
    array, value = np.asarray(array), np.asarray(value)
    if array.size == 1:
        return np.zeros(value.shape, dtype=int)
    if direction == "nearest":
        halfway = array[:-1] + (array[1:] - array[:-1]) / 2
        found = np.searchsorted(halfway, value)
    elif direction == "forward":
        found = np.searchsorted(array, value, side="left")
    elif direction == "backward":
        found = np.searchsorted(array, value, side="right") - 1
    else:
        raise ValueError("Unknown direction: {}".format(direction))
    return np.clip(found, 0, array.size - 1)


def take_frames(values, nearest):
    """``values[nearest]`` on the device: result frame j is source frame nearest[j] (csrc/gm_temporal.hip
    gm_frames_take)."""
    from ..hip import runtime
    from ..hip.array import DeviceArray, as_device
    src = as_device(values)
    nearest = np.ascontiguousarray(nearest, dtype=np.int32).ravel()
    out = DeviceArray.empty((len(nearest),) + src.shape[1:], src.dtype)
    frame_bytes = int(np.prod(src.shape[1:], dtype=np.int64)) * src.dtype.itemsize
    L = runtime.lib()
    L.gm_frames_take.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.c_void_p]
    L.gm_frames_take.restype = ctypes.c_int
    step = 65535  # (frames per launch: the grid's second dimension)
    for first in range(0, len(nearest), step):
        part = np.ascontiguousarray(nearest[first:first + step])
        runtime.check(L.gm_frames_take(src.ptr, frame_bytes, src.shape[0], part.ctypes.data_as(ctypes.c_void_p),
                                       len(part), out.ptr + first * frame_bytes), "gm_frames_take")
    return out


def sum_frames(values):
    """``values.sum(axis=0)[np.newaxis]`` on the device, in np.sum's accumulator dtype (gm_frames_sum)."""
    from ..hip import runtime
    from ..hip.array import DeviceArray, as_device
    from ..hip.dtypes import code_of, dtype_of
    src = as_device(values)
    L = runtime.lib()
    L.gm_frames_sum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.gm_frames_sum.restype = ctypes.c_int
    code = code_of(src.dtype)
    out = DeviceArray.empty((1,) + src.shape[1:], dtype_of(L.gm_frames_sum_dtype(code)))
    if src.shape[0] == 0:  # (a sum of nothing)
        runtime.check(L.gm_fill(ctypes.c_void_p(out.ptr), code_of(out.dtype), runtime.scalar_bits(0, out.dtype), out.size),
                      "gm_fill")
        return out
    runtime.check(L.gm_frames_sum(src.ptr, code, src.shape[0], int(np.prod(src.shape[1:], dtype=np.int64)), out.ptr),
                  "gm_frames_sum")
    return out


def _time_set(block, start=None, stop=None):
    answer = block.get_data(mode="time", start=start, stop=stop)
    return set() if answer is None else set(answer["time"])


def _frames_near(block, start, stop):
    """The frames of ``block`` near ``start``, between the two and near ``stop``: a result frame may be nearest to
    a source frame outside the requested interval (temporal.py:144-157, 1166-1176)."""
    return sorted(_time_set(block, start=start) | _time_set(block, start=start, stop=stop) | _time_set(block, start=stop))


def _rearranged(process_kwargs, data):
    """data with its frames (values or meta) picked by process_kwargs["nearest"] (temporal.py:172-182)."""
    nearest = process_kwargs["nearest"]
    if "values" in data:
        return dict(data, values=take_frames(data["values"], nearest))
    if "meta" in data:
        return dict(data, meta=[data["meta"][i] for i in nearest])


class Snap(RasterBlock):
    """Snap the time structure of a raster to that of another raster (temporal.py:42-182): the cell values of
    ``store`` at the time stamps of ``index`` -- for every index frame the store frame closest in time.

    Args: store (cell values), index (time stamps)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, index):
        for x in (store, index):
            if not isinstance(x, RasterBlock):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(Snap, self).__init__(store, index)

    store = property(lambda self: self.args[0])
    index = property(lambda self: self.args[1])

    def __len__(self):
        return len(self.index)

    dtype = property(lambda self: self.store.dtype)
    fillvalue = property(lambda self: self.store.fillvalue)
    period = property(lambda self: self.index.period if self.store else None)
    timedelta = property(lambda self: self.index.timedelta)
    temporal = property(lambda self: self.index.temporal)
    extent = property(lambda self: self.store.extent)
    geometry = property(lambda self: self.store.geometry)
    projection = property(lambda self: self.store.projection)
    geo_transform = property(lambda self: self.store.geo_transform)

    def get_sources_and_requests(self, **request):
        store_period, index_period = self.store.period, self.index.period
        if store_period is None or index_period is None:  # an empty raster on either side: empty
            return [(None, None)]
        if request["mode"] == "time":  # the index answers time requests
            return [(None, None), (self.index, request)]
        start, stop = request.get("start"), request.get("stop")
        index_answer = self.index.get_data(mode="time", start=start, stop=stop)
        if index_answer is None:
            return [(None, None)]
        index_time = index_answer["time"]
        if stop is None:  # one frame: the store is asked at the index's time stamp
            request["start"] = index_time[0]
            return [(None, None), (self.store, request)]
        if store_period[0] == store_period[1]:
            store_time = [store_period[0]]
        else:
            store_time = _frames_near(self.store, start, stop)
        request["start"], request["stop"] = store_time[0], store_time[-1]
        return [({"nearest": find_neighbours(store_time, index_time)}, None), (self.store, request)]

    @staticmethod
    def process(process_kwargs, data=None):
        if process_kwargs is None:
            return data
        return _rearranged(process_kwargs, data)


class Shift(BaseSingle):
    """Shift a temporal raster by a timedelta (temporal.py:185-236): positive into the future.

    Args: store, time (milliseconds, or a datetime.timedelta)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, time):
        if isinstance(time, Timedelta):
            time = int(time.total_seconds() * 1000)
        if not isinstance(time, int):
            raise TypeError("'{}' object is not allowed".format(type(time)))
        super(Shift, self).__init__(store, time)

    @property
    def time(self):
        return Timedelta(milliseconds=self.args[1])

    @property
    def period(self):
        start, stop = self.store.period
        return start + self.time, stop + self.time

    def get_sources_and_requests(self, **request):
        for key in ("start", "stop"):
            if request.get(key) is not None:
                request[key] = request[key] - self.time
        return [(self.store, request), (self.time, None)]

    @staticmethod
    def process(data, time):
        if data is None:
            return None
        if "time" in data:
            return dict(data, time=[t + time for t in data["time"]])
        return data


class TemporalSum(BaseSingle):
    """All frames of the store added up into one (temporal.py:239-255): a plain sum -- 'no data' values are added
    like any other value -- in np.sum's accumulator dtype; time and meta answer with their last entry."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    @staticmethod
    def process(data):
        if data is None:
            return data
        if "time" in data:
            return {"time": data["time"][-1:]}
        if "meta" in data:
            return {"meta": data["meta"][-1:]}
        if "values" in data:
            return {"values": sum_frames(data["values"]), "no_data_value": data["no_data_value"]}


class Resample(BaseSingle):
    """Resample a raster in time (temporal.py:1008-1214): the frames of the result lie at the labels of
    ``frequency`` and each takes the nearest source frame, or the first one after / before it.

    Args: source, frequency (pandas offset string), direction ('nearest', 'backward' or 'forward'), timezone."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, source, frequency, direction="nearest", timezone="UTC"):
        _expect(source, RasterBlock)
        # (the order of the checks is the reference's: the frequency, the time zone, then the direction)
        frequency, timezone = _rule_arguments(frequency, timezone)
        if _expect(direction, str) not in {"nearest", "backward", "forward"}:
            raise ValueError("direction must be one of 'nearest', 'backward', or 'forward'.")
        super(Resample, self).__init__(source, frequency, direction, timezone)

    source = property(lambda self: self.args[0])
    frequency = property(lambda self: normalize_offset(self.args[1]))
    direction = property(lambda self: self.args[2])
    timezone = property(lambda self: self.args[3])

    @property
    def _calendar(self):
        return _calendar_for(self.frequency, None, None, self.timezone)

    @property
    def period(self):
        """First and last label that can snap to a source frame (temporal.py:1080-1125): a label t reaches the
        source frames in [t, t + p) forwards, (t - p, t] backwards, [t - p/2, t + p/2) for the nearest one."""
        source_period = self.source.period
        if source_period is None:
            return None
        calendar = self._calendar
        first, last = source_period
        if self.direction in ("forward", "backward"):
            side = "left" if self.direction == "forward" else "right"
            return calendar.nearest_label(first, side), calendar.nearest_label(last, side)
        start = calendar.nearest_label(first, "left")
        if first >= calendar.shifted(start, 0.5):  # too far to the left of the first frame to be nearest to it
            start = calendar.nearest_label(first, "right")
        end = calendar.nearest_label(last, "right")
        if last < calendar.shifted(end, -0.5):
            end = calendar.nearest_label(last, "left")
        return start, end

    @property
    def timedelta(self):
        return offset_to_timedelta(self.frequency)

    def get_sources_and_requests(self, **request):
        calendar = self._calendar
        mode = request["mode"]
        kwargs = {"mode": mode, "direction": self.direction, "frequency": self.frequency, "timezone": self.timezone}
        kwargs["start"], kwargs["stop"] = calendar.snap_window(self.period, request.get("start"), request.get("stop"))
        if kwargs["start"] is None:
            return [({"empty": True, "mode": mode}, None)]
        if mode == "time":  # the labels alone answer a time request
            return [(kwargs, None)]
        index_time = calendar.label_range(kwargs["start"], kwargs["stop"])
        shift = {"forward": 0, "backward": -1}.get(self.direction, -0.5)
        near_start = calendar.shifted(kwargs["start"], shift)
        near_stop = calendar.shifted(kwargs["stop"] or kwargs["start"], shift + 1)
        store_time = _frames_near(self.store, near_start, near_stop)
        if not store_time:
            return [({"empty": True, "mode": mode}, None)]
        nearest = find_neighbours(store_time, index_time, self.direction)
        request["start"], request["stop"] = store_time[nearest.min()], store_time[nearest.max()]
        kwargs["nearest"] = nearest - nearest.min()
        return [(kwargs, None), (self.store, request)]

    @staticmethod
    def process(process_kwargs, data=None):
        mode = process_kwargs["mode"]
        if process_kwargs.get("empty"):
            return None if mode == "vals" else {mode: []}
        if mode == "time":
            calendar = _calendar_for(process_kwargs["frequency"], None, None, process_kwargs["timezone"])
            return {"time": calendar.label_range(process_kwargs["start"], process_kwargs["stop"])}
        return _rearranged(process_kwargs, data)


class TemporalAggregate(BaseSingle):
    """Aggregate a raster in time (temporal.py:476-768).

    Args: source, frequency (pandas offset string, or None for one statistic over the whole period,
    labelled with the end of the period), statistic (sum, count, min, max, mean, median, std, var,
    p<percentile>), closed / label ('left', 'right' or None for the frequency's default), timezone.
    """
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    gm_direct = True  # plan.py: evaluated as process(store's responses) without a graph per request

    STATISTICS = ("sum", "count", "min", "max", "mean", "median", "std", "var")

    def __init__(self, source, frequency, statistic="sum", closed=None, label=None, timezone="UTC"):
        _expect(source, RasterBlock)
        # no frequency: one bin over everything, and the rule's other arguments mean nothing (kept as None)
        rule = (None, None, None, None) if frequency is None else _rule_arguments(frequency, timezone, closed, label, sides=True)
        frequency, timezone, closed, label = rule
        statistic = _check_statistic(statistic, self.STATISTICS)
        super().__init__(source, frequency, statistic, closed, label, timezone)

    source = property(lambda self: self.args[0])
    frequency = property(lambda self: normalize_offset(self.args[1]))
    statistic = property(lambda self: self.args[2])
    closed = property(lambda self: self.args[3])
    label = property(lambda self: self.args[4])
    timezone = property(lambda self: self.args[5])

    @property
    def _calendar(self):
        return _calendar_for(self.frequency, self.closed, self.label, self.timezone)

    @property
    def period(self):
        return self._calendar.resampled_period(self.source.period)

    @property
    def timedelta(self):
        return None if self.frequency is None else offset_to_timedelta(self.frequency)

    @property
    def temporal(self):
        return self.frequency is not None

    @property
    def dtype(self):
        return dtype_for_statistic(self.source.dtype, self.statistic)

    @property
    def fillvalue(self):
        return get_dtype_max(self.dtype)

    def get_sources_and_requests(self, **request):
        calendar = self._calendar
        mode = request["mode"]
        first, last = calendar.snap_window(self.period, request.get("start"), request.get("stop"))
        if first is None:
            return [({"empty": True, "mode": mode}, None)]
        kwargs = dict(calendar.kwargs(), mode=mode, start=first, stop=last)
        if mode == "time":  # the labels alone answer a time request
            return [(kwargs, None)]
        if self.frequency is None:
            request["start"], request["stop"] = self.source.period
        else:
            request["start"], request["stop"] = calendar.first_moment(first), calendar.last_moment(last or first)
        if mode == "vals":
            kwargs["dtype"] = np.dtype(self.dtype).str
            kwargs["statistic"] = self.statistic
        frames_request = {"mode": "time", "start": request["start"], "stop": request["stop"]}
        if "time_resolution" in request:
            frames_request["time_resolution"] = request["time_resolution"]
        return [(kwargs, None), (self.source, frames_request), (self.source, request)]

    @staticmethod
    def process(process_kwargs, time_data=None, data=None):
        mode = process_kwargs["mode"]
        nothing = None if mode == "vals" else {mode: []}
        if process_kwargs.get("empty"):
            return nothing
        calendar = _calendar_for(process_kwargs["frequency"], process_kwargs["closed"], process_kwargs["label"],
                             process_kwargs["timezone"])
        labels, local_labels = calendar.labels_between(process_kwargs["start"], process_kwargs["stop"])
        if mode == "time":
            return {"time": labels.to_pydatetime().tolist()}
        if time_data is None or not time_data.get("time"):
            return nothing
        times = time_data["time"]
        runs = calendar.frames_of(times, local_labels)
        if mode == "meta":
            if data is None or "meta" not in data:
                return {"meta": []}
            return {"meta": [[data["meta"][i] for i in range(first, first + count)] for first, count in runs]}
        if data is None or "values" not in data:
            return None
        if data["values"].shape[0] != len(times):
            raise RuntimeError("Shape of raster does not match number of timestamps")
        dtype = np.dtype(process_kwargs["dtype"])
        statistic, percentile = parse_statistic(process_kwargs["statistic"])
        if percentile is not None:
            statistic = "p%s" % percentile
        values = aggregate_frames(data["values"], data["no_data_value"], runs, statistic, dtype)
        return {"values": values, "no_data_value": get_dtype_max(dtype)}


class Cumulative(BaseSingle):
    """Accumulate a raster over time (temporal.py:776-1003): the time structure of the source is
    kept, every frame holds the sum (or count of data frames) since the start of its period.

    Args: source, statistic ('sum' or 'count'), frequency (period after which the accumulation starts
    again; None = never), timezone."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    gm_direct = True  # plan.py: evaluated as process(store's responses) without a graph per request

    STATISTICS = ("sum", "count")

    def __init__(self, source, statistic="sum", frequency=None, timezone="UTC"):
        _expect(source, RasterBlock)
        statistic = _check_statistic(statistic, self.STATISTICS)
        frequency, timezone = (None, None) if frequency is None else _rule_arguments(frequency, timezone)
        super().__init__(source, statistic, frequency, timezone)

    source = property(lambda self: self.args[0])
    statistic = property(lambda self: self.args[1])
    frequency = property(lambda self: normalize_offset(self.args[2]))
    timezone = property(lambda self: self.args[3])

    @property
    def dtype(self):
        return dtype_for_statistic(self.source.dtype, self.statistic)

    @property
    def fillvalue(self):
        return get_dtype_max(self.dtype)

    def get_sources_and_requests(self, **request):
        if request["mode"] == "time":  # the frames of the source are the frames of the result
            return [({"mode": "time"}, None), (self.source, request)]
        mode = request["mode"]
        # which frames are asked for decides how far back the accumulation has to start
        wanted = self.source.get_data(mode="time", start=request.get("start"), stop=request.get("stop"))
        if wanted is None or not wanted.get("time"):
            return [({"empty": True, "mode": mode}, None)]
        first, last = wanted["time"][0], wanted["time"][-1]
        calendar = _calendar_for(self.frequency, "right", "right", self.timezone)
        if self.frequency is None:
            request["start"], request["stop"] = self.period[0], last
        else:
            request["start"] = calendar.moment(calendar.bin_start_of(first))
            request["stop"] = last + _TICK
        kwargs = dict(calendar.kwargs(), mode=mode, start=first, stop=last)
        if mode == "vals":
            kwargs["dtype"] = np.dtype(self.dtype).str
            kwargs["statistic"] = self.statistic
        frames_request = {"mode": "time", "start": request["start"], "stop": request["stop"]}
        return [(kwargs, None), (self.source, frames_request), (self.source, request)]

    @staticmethod
    def process(process_kwargs, time_data=None, data=None):
        mode = process_kwargs["mode"]
        nothing = None if mode == "vals" else {mode: []}
        if process_kwargs.get("empty"):
            return nothing
        if mode == "time":
            return time_data
        if time_data is None or not time_data.get("time"):
            return nothing
        calendar = _calendar_for(process_kwargs["frequency"], process_kwargs["closed"], process_kwargs["label"],
                             process_kwargs["timezone"])
        times = time_data["time"]
        members, wanted = calendar.periods_of(times, process_kwargs["start"], process_kwargs["stop"])

        if mode == "meta":
            if data is None or "meta" not in data:
                return {"meta": []}
            meta = data["meta"]
            return {"meta": [[meta[i] for i in frames[:n]] for frames in members for n in range(1, len(frames) + 1)
                             if wanted[frames[n - 1]]]}
        if data is None or "values" not in data:
            return None
        if data["values"].shape[0] != len(times):
            raise RuntimeError("Shape of raster does not match number of timestamps")
        runs = []
        for frames in members:
            if len(frames) and not np.array_equal(frames, np.arange(frames[0], frames[0] + len(frames))):
                raise NotImplementedError("time stamps of the source are not ascending")
            if len(frames):
                runs.append((int(frames[0]), len(frames)))
        emit_first = int(np.flatnonzero(wanted)[0])
        emit = (emit_first, emit_first + int(wanted.sum()))
        dtype = np.dtype(process_kwargs["dtype"])
        values = accumulate_frames(data["values"], data["no_data_value"], runs, emit, process_kwargs["statistic"], dtype)
        return {"values": values, "no_data_value": get_dtype_max(dtype)}
