"""Element-wise raster Blocks (API of the reference's raster/elemwise.py).

Same class names, constructor arguments, attribute algebra (dtype/fillvalue/period/
extent/... rules of elemwise.py:37-201) and ``process`` signatures; the ``process``
bodies run on the MI355X through ``devops`` (one fused launch per op, or per fused
sub-graph when evaluated through ``get_data``).
"""
import numpy as np

from ..utils import Extent, GeoTransform, get_dtype_max
from . import devops
from .base import BaseSingle, RasterBlock

__all__ = [
    "Add", "Subtract", "Multiply", "Divide", "Power", "FillNoData", "Equal", "NotEqual",
    "Greater", "GreaterEqual", "Less", "LessEqual", "Invert", "And", "Or", "Xor", "IsData",
    "IsNoData", "Exp", "Log", "Log10",
]

BOOL = np.dtype("bool")


def _overlap(boxes):
    """Intersection of (x1, y1, x2, y2) boxes or None."""
    x1 = max(b[0] for b in boxes)
    y1 = max(b[1] for b in boxes)
    x2 = min(b[2] for b in boxes)
    y2 = min(b[3] for b in boxes)
    if x2 <= x1 or y2 <= y1:
        return None
    return x1, y1, x2, y2


class BaseElementwise(RasterBlock):
    """Base of element-wise operations; extent and period are the intersection of the
    raster arguments' (elemwise.py:37-191)."""

    def __init__(self, *args):
        super(BaseElementwise, self).__init__(*args)
        rasters = self._sources
        if len(rasters) > 1:  # elemwise.py:46-57
            temporal, delta = rasters[0].temporal, rasters[0].timedelta
            if any(r.temporal != temporal for r in rasters[1:]):
                raise ValueError("Temporal properties of input rasters do not match.")
            if temporal and delta is not None:
                if not all(r.timedelta in (None, delta) for r in rasters[1:]):
                    raise ValueError("Time resolutions of input rasters are not equal.")

    @property
    def _sources(self):
        return [arg for arg in self.args if isinstance(arg, RasterBlock)]

    def get_sources_and_requests(self, **request):
        start, stop = request.get("start"), request.get("stop")
        if start is not None and stop is not None:
            period = self.period  # clamp so that all arguments answer the same frames
            if period is not None:
                request["start"] = max(start, period[0])
                request["stop"] = min(stop, period[1])
        process_kwargs = {"dtype": self.dtype.name, "fillvalue": self.fillvalue}
        return [(process_kwargs, None)] + [(arg, request) for arg in self.args]

    @property
    def timedelta(self):
        rasters = self._sources
        if len(rasters) == 1:
            return rasters[0].timedelta
        deltas = [r.timedelta for r in rasters]
        return None if any(d is None for d in deltas) else deltas[0]

    @property
    def temporal(self):
        return self._sources[0].temporal

    @property
    def period(self):
        rasters = self._sources
        if len(rasters) == 1:
            return rasters[0].period
        periods = [r.period for r in rasters]
        if any(p is None for p in periods):
            return None
        start = max(p[0] for p in periods)
        stop = min(p[1] for p in periods)
        return None if stop < start else (start, stop)

    @property
    def extent(self):
        rasters = self._sources
        if len(rasters) == 1:
            return rasters[0].extent
        extents = [r.extent for r in rasters]
        if any(e is None for e in extents):
            return None
        return _overlap(extents)

    @property
    def dtype(self):
        # result_type over all arguments (python scalars are weak), lifted to >= int32 / float32.
        # Blocks are immutable: computed once (request planning asks for it per tile).
        cached = self.__dict__.get("_dtype_cache")
        if cached is not None:
            return cached
        dtype = np.result_type(*[a.dtype if isinstance(a, RasterBlock) else a for a in self.args])
        if np.issubdtype(dtype, np.integer) or dtype == BOOL:
            dtype = np.result_type(dtype, np.int32)
        elif np.issubdtype(dtype, np.floating):
            dtype = np.result_type(dtype, np.float32)
        self.__dict__["_dtype_cache"] = dtype
        return dtype

    @property
    def fillvalue(self):
        dtype = self.dtype
        return None if dtype == BOOL else get_dtype_max(dtype)

    @property
    def geometry(self):
        geometries = [r.geometry for r in self._sources]
        if any(g is None for g in geometries):
            return None
        if len(geometries) == 1:
            return geometries[0]
        extent = Extent.from_geometry(geometries[0])
        for geometry in geometries[1:]:
            extent = extent.intersection(Extent.from_geometry(geometry))
            if extent is None:
                return None
        return extent.as_geometry()

    @property
    def projection(self):
        rasters = self._sources
        projection = rasters[0].projection
        if projection is None:
            return None
        return projection if all(r.projection == projection for r in rasters[1:]) else None

    @property
    def geo_transform(self):
        rasters = self._sources
        first = rasters[0].geo_transform
        if first is None:
            return None
        first = GeoTransform(first)
        for r in rasters[1:]:
            other = r.geo_transform
            if other is None or not first.aligns_with(other):
                return None
        return first


class BaseMath(BaseElementwise):
    def __init__(self, a, b):
        for x in (a, b):
            if not isinstance(x, (RasterBlock, np.ndarray, float, int)):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(BaseMath, self).__init__(a, b)


class BaseComparison(BaseMath):
    @property
    def dtype(self):
        return BOOL


class BaseLogic(BaseElementwise):
    def __init__(self, a, b):
        for x in (a, b):
            if isinstance(x, (RasterBlock, np.ndarray)):
                if x.dtype != BOOL:
                    raise TypeError("inputs must have boolean dtypes")
            elif not isinstance(x, bool):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(BaseLogic, self).__init__(a, b)

    @property
    def dtype(self):
        return BOOL

    @property
    def fillvalue(self):
        return None


class Add(BaseMath):
    """a + b with no-data propagation (elemwise.py:302-318)."""
    process = staticmethod(devops.make_math_process("add"))


class Subtract(BaseMath):
    """a - b (elemwise.py:321-337)."""
    process = staticmethod(devops.make_math_process("subtract"))


class Multiply(BaseMath):
    """a * b (elemwise.py:340-356)."""
    process = staticmethod(devops.make_math_process("multiply"))


class Divide(BaseMath):
    """a / b in at least float32; division by zero becomes no-data (elemwise.py:359-380)."""
    process = staticmethod(devops.make_math_process("divide"))

    @property
    def dtype(self):
        return np.result_type(np.float32, *[a.dtype if isinstance(a, RasterBlock) else a for a in self.args])


class Power(BaseMath):
    """a ** b; out-of-domain / overflowing results become no-data (elemwise.py:383-406)."""
    process = staticmethod(devops.make_math_process("power"))

    def __init__(self, a, b):
        if isinstance(b, int) and b < 0:  # integer ** negative integer is an error in NumPy
            b = float(b)
        super(Power, self).__init__(a, b)


class Equal(BaseComparison):
    process = staticmethod(devops.make_math_process("equal"))


class NotEqual(BaseComparison):
    process = staticmethod(devops.make_math_process("not_equal"))


class Greater(BaseComparison):
    process = staticmethod(devops.make_math_process("greater"))


class GreaterEqual(BaseComparison):
    process = staticmethod(devops.make_math_process("greater_equal"))


class Less(BaseComparison):
    process = staticmethod(devops.make_math_process("less"))


class LessEqual(BaseComparison):
    process = staticmethod(devops.make_math_process("less_equal"))


class And(BaseLogic):
    process = staticmethod(devops.make_math_process("logical_and"))


class Or(BaseLogic):
    process = staticmethod(devops.make_math_process("logical_or"))


class Xor(BaseLogic):
    process = staticmethod(devops.make_math_process("logical_xor"))


def _invert_process(data):
    if "values" in data:
        return devops.unary_bool_process("invert", data)
    return data


_invert_process.gm_op = "invert"


class Invert(BaseSingle):
    """Logical not of a boolean raster (elemwise.py:551-579)."""

    def __init__(self, x):
        super(Invert, self).__init__(x)
        if x.dtype != BOOL:
            raise TypeError("input block must have boolean dtype")

    process = staticmethod(_invert_process)

    @property
    def dtype(self):
        return BOOL


def _isdata_process(data):
    if data is None or "values" not in data:
        return data
    return devops.unary_bool_process("isdata", data)


def _isnodata_process(data):
    if data is None or "values" not in data:
        return data
    return devops.unary_bool_process("isnodata", data)


_isdata_process.gm_op = "isdata"
_isnodata_process.gm_op = "isnodata"


class IsData(BaseSingle):
    """True where the raster has data (elemwise.py:582-615)."""

    def __init__(self, store):
        if store.dtype == BOOL:
            raise TypeError("input block must not have boolean dtype")
        super(IsData, self).__init__(store)

    process = staticmethod(_isdata_process)

    @property
    def dtype(self):
        return BOOL

    @property
    def fillvalue(self):
        return None


class IsNoData(IsData):
    """True where the raster has no data (elemwise.py:618-638)."""
    process = staticmethod(_isnodata_process)


class FillNoData(BaseElementwise):
    """Rightmost data wins, no-data is transparent (elemwise.py:700-757)."""

    def __init__(self, *args):
        for arg in args:
            if not isinstance(arg, RasterBlock):
                raise TypeError("'{}' object is not allowed".format(type(arg)))
        super(FillNoData, self).__init__(*args)

    @staticmethod
    def process(process_kwargs, *args):
        return devops.fillnodata_process(process_kwargs, *args)


class BaseLogExp(BaseSingle):
    def __init__(self, x):
        if x.dtype == BOOL:
            raise TypeError("input block must not have boolean dtype")
        super(BaseLogExp, self).__init__(x)

    def get_sources_and_requests(self, **request):
        process_kwargs = {"dtype": self.dtype.name, "fillvalue": self.fillvalue}
        return [(process_kwargs, None), (self.args[0], request)]

    @property
    def dtype(self):
        return np.result_type(np.float32, self.args[0].dtype)

    @property
    def fillvalue(self):
        return get_dtype_max(self.dtype)


class Exp(BaseLogExp):
    process = staticmethod(devops.make_math_process("exp"))


class Log(BaseLogExp):
    process = staticmethod(devops.make_math_process("log"))


class Log10(BaseLogExp):
    process = staticmethod(devops.make_math_process("log10"))
