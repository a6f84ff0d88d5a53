"""Miscellaneous raster Blocks on the hot path: Clip, Mask, Classify, Reclassify
(API of the reference's raster/misc.py:30-515) plus MaskBelow and Step (:225-328).

Construction-time validation, attribute rules and ``process`` signatures follow the
reference; the ``process`` bodies run on the MI355X (``devops`` / ``tables``).
"""
import numpy as np

from .. import utils
from . import devops, tables
from .base import BaseSingle, RasterBlock

__all__ = ["Clip", "Classify", "Reclassify", "Mask", "MaskBelow", "Step"]

BOOL = np.dtype("bool")


class Clip(BaseSingle):
    """Keep ``store`` where ``source`` has data (or is True), no-data elsewhere
    (misc.py:30-166)."""

    def __init__(self, store, source):
        if not isinstance(source, RasterBlock):
            raise TypeError("'{}' object is not allowed".format(type(store)))
        if store.temporal and not source.temporal:
            raise ValueError("The values raster is temporal while the clipping mask is not. "
                             "Consider using Snap.")
        if not store.temporal and source.temporal:
            raise ValueError("The clipping mask is temporal while the values raster is not. "
                             "Consider using Snap.")
        if store.temporal and store.timedelta != source.timedelta:
            raise ValueError("Time resolution of the clipping mask does not match that of "
                             "the values raster. Consider using Snap.")
        super(Clip, self).__init__(store, source)

    @property
    def source(self):
        return self.args[1]

    def get_sources_and_requests(self, **request):
        """Clamp start/stop into the common period so both tiles are aligned (misc.py:75-96)."""
        period = self.period
        if period is None:
            return [(None, None), (None, None)]
        start = request.get("start")
        if start is None:
            start = period[1]
        stop = request.get("stop")
        if stop is not None:
            if stop < period[0] or start > period[1]:
                return [(None, None), (None, None)]
            request["stop"] = min(max(stop, period[0]), period[1])
        request["start"] = min(max(start, period[0]), period[1])
        return [(arg, request) for arg in self.args]

    @staticmethod
    def process(data, source_data):
        return devops.clip_process(data, source_data)

    process.__func__.gm_op = "clip"

    @property
    def extent(self):
        store, mask = [x.extent for x in self.args]
        if store is None or mask is None:
            return None
        x1, y1 = max(store[0], mask[0]), max(store[1], mask[1])
        x2, y2 = min(store[2], mask[2]), min(store[3], mask[3])
        if x2 <= x1 or y2 <= y1:
            return None
        return x1, y1, x2, y2

    @property
    def geometry(self):
        store, mask = [x.geometry for x in self.args]
        if store is None or mask is None:
            return None
        extent = utils.Extent.from_geometry(store).intersection(utils.Extent.from_geometry(mask))
        return None if extent is None else extent.as_geometry()

    @property
    def period(self):
        periods = [x.period for x in self.args]
        if any(p is None for p in periods):
            return None
        start = max(p[0] for p in periods)
        stop = min(p[1] for p in periods)
        return None if stop < start else (start, stop)


class Mask(BaseSingle):
    """Replace every data cell by one constant; no-data stays no-data (misc.py:169-222)."""

    def __init__(self, store, value):
        if not isinstance(value, (float, int)):
            raise TypeError("'{}' object is not allowed".format(type(value)))
        super(Mask, self).__init__(store, value)

    @property
    def value(self):
        return self.args[1]

    @property
    def fillvalue(self):
        return 1 if self.value == 0 else 0

    @property
    def dtype(self):
        return devops.mask_dtype(self.value)

    _dtype_from_value = staticmethod(devops.mask_dtype)

    @staticmethod
    def process(data, value):
        return devops.mask_process(data, value)

    process.__func__.gm_op = "mask"


class MaskBelow(BaseSingle):
    """Cells below ``value`` become no-data (misc.py:225-251)."""

    def __init__(self, store, value):
        if not isinstance(value, (float, int)):
            raise TypeError("'{}' object is not allowed".format(type(value)))
        super(MaskBelow, self).__init__(store, value)

    @staticmethod
    def process(data, value):
        return devops.maskbelow_process(data, value)

    process.__func__.gm_op = "maskbelow"


class Step(BaseSingle):
    """Three-valued step function around ``value`` (misc.py:254-328)."""

    def __init__(self, store, left=0, right=1, value=0, at=None):
        at = (left + right) / 2 if at is None else at
        for x in left, right, value, at:
            if not isinstance(x, (float, int)):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(Step, self).__init__(store, left, right, value, at)

    left = property(lambda self: self.args[1])
    right = property(lambda self: self.args[2])
    value = property(lambda self: self.args[3])
    at = property(lambda self: self.args[4])

    @staticmethod
    def process(data, left, right, location, at):
        return devops.step_process(data, left, right, location, at)

    process.__func__.gm_op = "step"


class Classify(BaseSingle):
    """Bin index of every cell for increasing ``bins`` (misc.py:331-399)."""

    def __init__(self, store, bins, right=False):
        if not isinstance(store, RasterBlock):
            raise TypeError("'{}' object is not allowed".format(type(store)))
        if not hasattr(bins, "__iter__"):
            raise TypeError("'{}' object is not allowed".format(type(bins)))
        edges = np.asarray(bins)
        if edges.ndim != 1:
            raise TypeError("'bins' should be one-dimensional")
        if not np.issubdtype(edges.dtype, np.number):
            raise TypeError("'bins' should be numeric")
        steps = np.diff(bins)
        # same acceptance test as the reference (misc.py:364-366), including its rejection
        # of single-edge lists (all() of an empty difference is True)
        if not np.all(steps > 0) or np.all(steps < 0):
            raise TypeError("'bins' should be monotonic")
        super(Classify, self).__init__(store, edges.tolist(), right)

    @property
    def bins(self):
        return self.args[1]

    @property
    def right(self):
        return self.args[2]

    @property
    def dtype(self):
        # n edges -> n + 1 classes, plus one code for no-data
        return utils.get_uint_dtype(len(self.bins) + 2)

    @property
    def fillvalue(self):
        return utils.get_dtype_max(self.dtype)

    @staticmethod
    def process(data, bins, right):
        return tables.classify_process(data, bins, right)

    process.__func__.gm_op = "classify"


class Reclassify(BaseSingle):
    """Map integer cell values through a list of [from, to] pairs (misc.py:402-515)."""

    def __init__(self, store, data, select=False):
        dtype = store.dtype
        if dtype != BOOL and not np.issubdtype(dtype, np.integer):
            raise TypeError("The store must be of boolean or integer datatype")
        if not hasattr(data, "__iter__"):
            raise TypeError("'{}' object is not allowed".format(type(data)))
        try:
            source, target = self._data_as_ndarray(data)
        except ValueError:
            raise ValueError("Please supply a list of [from, to] values")
        if source.dtype != BOOL and not np.issubdtype(source.dtype, np.integer):
            raise TypeError("Cannot reclassify from value with type '{}'".format(source.dtype))
        if len(np.unique(source)) != len(source):
            raise ValueError("There are duplicates in the reclassify values")
        if not np.issubdtype(target.dtype, np.number):
            raise TypeError("Cannot reclassify to value with type '{}'".format(target.dtype))
        data = [list(pair) for pair in zip(source.tolist(), target.tolist())]
        if select is not True and select is not False:
            raise TypeError("'{}' object is not allowed".format(type(select)))
        super().__init__(store, data, select)

    @staticmethod
    def _data_as_ndarray(data):
        source, target = zip(*data)
        return np.asarray(source), np.asarray(target)

    @property
    def data(self):
        return self.args[1]

    @property
    def select(self):
        return self.args[2]

    @property
    def dtype(self):
        return self._data_as_ndarray(self.data)[1].dtype

    @property
    def fillvalue(self):
        return utils.get_dtype_max(self.dtype)

    def get_sources_and_requests(self, **request):
        process_kwargs = {"dtype": self.dtype.str, "fillvalue": self.fillvalue,
                          "data": self.data, "select": self.select}
        return [(self.store, request), (process_kwargs, None)]

    @staticmethod
    def process(store_data, process_kwargs):
        return tables.reclassify_process(store_data, process_kwargs)

    process.__func__.gm_op = "reclassify"
