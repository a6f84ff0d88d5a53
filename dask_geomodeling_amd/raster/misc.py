"""Miscellaneous raster Blocks on the hot path: Clip, Mask, Classify, Reclassify
(API of the reference's raster/misc.py:30-515) plus MaskBelow and Step (:225-328).

Construction-time validation, attribute rules and ``process`` signatures follow the
reference; the ``process`` bodies run on the MI355X (``devops`` / ``tables``).
"""
import numpy as np

from .. import utils
from . import devops, tables
from ._checks import argument, need_iterable, need_number, need_raster, not_allowed
from .base import BaseSingle

__all__ = ["Clip", "Classify", "Reclassify", "Mask", "MaskBelow", "Step"]

BOOL = np.dtype("bool")


class Clip(BaseSingle):
    """Keep ``store`` where ``source`` has data (or is True), no-data elsewhere
    (misc.py:30-166)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, source):
        need_raster(source, blame=store)  # (the reference names the store's type here, misc.py:54)
        mismatch = None
        if store.temporal != source.temporal:
            first, second = ("values raster", "clipping mask") if store.temporal else ("clipping mask", "values raster")
            mismatch = "The {} is temporal while the {} is not.".format(first, second)
        elif store.temporal and store.timedelta != source.timedelta:
            mismatch = "Time resolution of the clipping mask does not match that of the values raster."
        if mismatch is not None:
            raise ValueError(mismatch + " Consider using Snap.")
        super().__init__(store, source)

    source = argument(1)

    def get_sources_and_requests(self, **request):
        """Both tiles must answer the same frames: the window is moved into the common period, and there is
        nothing to ask for when it misses that period altogether (misc.py:75-96)."""
        nothing = [(None, None), (None, None)]
        common = self.period
        if common is None:
            return nothing
        first, last = common
        start, stop = request.get("start"), request.get("stop")
        start = last if start is None else start
        if stop is not None:
            if stop < first or start > last:
                return nothing
            request["stop"] = min(max(stop, first), last)
        request["start"] = min(max(start, first), last)
        return [(raster, request) for raster in self.args]

    @staticmethod
    def process(data, source_data):
        return devops.clip_process(data, source_data)

    process.__func__.gm_op = "clip"

    def _both(self, attribute):
        pair = [getattr(raster, attribute) for raster in self.args]
        return None if None in pair else pair

    @property
    def extent(self):
        boxes = self._both("extent")
        if boxes is None:
            return None
        (ax1, ay1, ax2, ay2), (bx1, by1, bx2, by2) = boxes
        west, south, east, north = max(ax1, bx1), max(ay1, by1), min(ax2, bx2), min(ay2, by2)
        return (west, south, east, north) if east > west and north > south else None

    @property
    def geometry(self):
        shapes = self._both("geometry")
        if shapes is None:
            return None
        common = utils.Extent.from_geometry(shapes[0]).intersection(utils.Extent.from_geometry(shapes[1]))
        return None if common is None else common.as_geometry()

    @property
    def period(self):
        periods = self._both("period")
        if periods is None:
            return None
        begin, end = max(p[0] for p in periods), min(p[1] for p in periods)
        return (begin, end) if begin <= end else None


class Mask(BaseSingle):
    """Replace every data cell by one constant; no-data stays no-data (misc.py:169-222)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, value):
        need_number(value)
        super().__init__(store, value)

    value = argument(1)
    fillvalue = property(lambda self: 1 if self.value == 0 else 0)  # whatever differs from the constant
    dtype = property(lambda self: devops.mask_dtype(self.value))
    _dtype_from_value = staticmethod(devops.mask_dtype)

    @staticmethod
    def process(data, value):
        return devops.mask_process(data, value)

    process.__func__.gm_op = "mask"


class MaskBelow(BaseSingle):
    """Cells below ``value`` become no-data (misc.py:225-251)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, value):
        need_number(value)
        super().__init__(store, value)

    @staticmethod
    def process(data, value):
        return devops.maskbelow_process(data, value)

    process.__func__.gm_op = "maskbelow"


class Step(BaseSingle):
    """Three-valued step function around ``value`` (misc.py:254-328)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, left=0, right=1, value=0, at=None):
        if at is None:
            at = (left + right) / 2
        need_number(left, right, value, at)
        super().__init__(store, left, right, value, at)

    left, right, value, at = argument(1), argument(2), argument(3), argument(4)

    @staticmethod
    def process(data, left, right, location, at):
        return devops.step_process(data, left, right, location, at)

    process.__func__.gm_op = "step"


class Classify(BaseSingle):
    """Bin index of every cell for increasing ``bins`` (misc.py:331-399)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, bins, right=False):
        need_raster(store)
        need_iterable(bins)
        edges = np.asarray(bins)
        complaint = None
        if edges.ndim != 1:
            complaint = "one-dimensional"
        elif not np.issubdtype(edges.dtype, np.number):
            complaint = "numeric"
        else:
            # the reference's own acceptance test (misc.py:364-366), quirks included: increasing edges pass;
            # a single edge does not (all() of its empty difference is True on both sides)
            steps = np.diff(bins)
            if not np.all(steps > 0) or np.all(steps < 0):
                complaint = "monotonic"
        if complaint is not None:
            raise TypeError("'bins' should be " + complaint)
        super().__init__(store, edges.tolist(), right)

    bins, right = argument(1), argument(2)
    # n edges -> n + 1 classes, plus one code for no data
    dtype = property(lambda self: utils.get_uint_dtype(len(self.bins) + 2))
    fillvalue = property(lambda self: utils.get_dtype_max(self.dtype))

    @staticmethod
    def process(data, bins, right):
        return tables.classify_process(data, bins, right)

    process.__func__.gm_op = "classify"


def _whole_numbers(dtype):
    return dtype == BOOL or np.issubdtype(dtype, np.integer)


class Reclassify(BaseSingle):
    """Map integer cell values through a list of [from, to] pairs (misc.py:402-515)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, store, data, select=False):
        if not _whole_numbers(store.dtype):
            raise TypeError("The store must be of boolean or integer datatype")
        need_iterable(data)
        try:
            keys, targets = self._data_as_ndarray(data)
        except ValueError:
            raise ValueError("Please supply a list of [from, to] values")
        if not _whole_numbers(keys.dtype):
            raise TypeError("Cannot reclassify from value with type '{}'".format(keys.dtype))
        if np.unique(keys).size != keys.size:
            raise ValueError("There are duplicates in the reclassify values")
        if not np.issubdtype(targets.dtype, np.number):
            raise TypeError("Cannot reclassify to value with type '{}'".format(targets.dtype))
        if select is not True and select is not False:
            raise not_allowed(select)
        # stored with consistent (python) value types: what dtype / JSON round trips see
        super().__init__(store, [[k, t] for k, t in zip(keys.tolist(), targets.tolist())], select)

    @staticmethod
    def _data_as_ndarray(data):
        keys, targets = zip(*data)
        return np.asarray(keys), np.asarray(targets)

    data, select = argument(1), argument(2)
    dtype = property(lambda self: self._data_as_ndarray(self.data)[1].dtype)  # int64 / float64
    fillvalue = property(lambda self: utils.get_dtype_max(self.dtype))

    def get_sources_and_requests(self, **request):
        table = {"dtype": self.dtype.str, "fillvalue": self.fillvalue, "data": self.data, "select": self.select}
        return [(self.store, request), (table, None)]

    @staticmethod
    def process(store_data, process_kwargs):
        return tables.reclassify_process(store_data, process_kwargs)

    process.__func__.gm_op = "reclassify"
