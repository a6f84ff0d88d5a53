"""Reduction Blocks (counterpart of raster/reduction.py): ``Max`` takes the maximum of two or
more rasters, ignoring no data.  The numerics run as one element-wise program
(devops.reduce_max_node): every raster with its no-data coded as NaN, np.fmax left to right,
NaN -> fill -- fusable with whatever feeds and follows it."""
import numpy as np

from ..utils import Extent
from . import devops
from .base import RasterBlock
from .elemwise import BaseElementwise

__all__ = ["Max"]


def filter_none(lst):
    return [x for x in lst if x is not None]


class BaseReduction(BaseElementwise):
    """reduction.py:122-186: all arguments are rasters; the extent / geometry is the UNION of the
    arguments' (a cell is data as soon as one raster has data there)."""

    def __init__(self, *args):
        for arg in args:
            if not isinstance(arg, RasterBlock):
                raise TypeError("'{}' object is not allowed".format(type(arg)))
        super().__init__(*args)

    def get_sources_and_requests(self, **request):
        # stricter than BaseElementwise: no data at all when the arguments share no period
        period = self.period
        process_kwargs = {"dtype": self.dtype.name, "fillvalue": self.fillvalue}
        if period is None:
            return [(process_kwargs, None)]
        start, stop = request.get("start", None), request.get("stop", None)
        if start is not None:
            if stop is not None:
                request["start"] = max(start, period[0])
                request["stop"] = min(stop, period[1])
            else:  # stop is None but start isn't
                request["start"] = min(max(start, period[0]), period[1])
        else:  # start and stop are both None
            request["start"] = period[1]
        return [(process_kwargs, None)] + [(source, request) for source in self.args]

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


class Max(BaseReduction):
    """Take the maximum value of two or more rasters, ignoring no data (reduction.py:215-230)."""

    process = staticmethod(devops.reduce_max_process)

    @property
    def dtype(self):
        # skip the default behaviour where we use at least int32 / float32
        return np.result_type(*self.args)
