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
    west, south = max(b[0] for b in boxes), max(b[1] for b in boxes)
    east, north = min(b[2] for b in boxes), min(b[3] for b in boxes)
    return (west, south, east, north) if east > west and north > south else None


class BaseElementwise(RasterBlock):
    """Base of element-wise operations.  The result exists where ALL raster arguments exist: period, extent
    and geometry are intersections, projection and geo_transform survive only when the arguments agree
    (elemwise.py:37-191)."""
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, *args):
        super().__init__(*args)
        rasters = self._sources
        lead = rasters[0] if rasters else None
        for other in rasters[1:]:  # elemwise.py:46-57
            if other.temporal != lead.temporal:
                raise ValueError("Temporal properties of input rasters do not match.")
        if lead is not None and lead.temporal and lead.timedelta is not None:
            for other in rasters[1:]:
                if other.timedelta not in (None, lead.timedelta):
                    raise ValueError("Time resolutions of input rasters are not equal.")

    @property
    def _sources(self):
        return [arg for arg in self.args if isinstance(arg, RasterBlock)]

    def _of_all(self, attribute):
        """The attribute of every raster argument, or None as soon as one of them has none."""
        values = [getattr(raster, attribute) for raster in self._sources]
        return None if any(v is None for v in values) else values

    def get_sources_and_requests(self, **request):
        if request.get("start") is not None and request.get("stop") is not None:
            common = self.period  # every argument must answer the same frames
            if common is not None:
                request["start"], request["stop"] = max(request["start"], common[0]), min(request["stop"], common[1])
        process_kwargs = {"dtype": self.dtype.name, "fillvalue": self.fillvalue}
        return [(process_kwargs, None)] + [(arg, request) for arg in self.args]

    @property
    def timedelta(self):
        deltas = self._of_all("timedelta")
        return None if deltas is None else deltas[0]

    @property
    def temporal(self):
        return self._sources[0].temporal

    @property
    def period(self):
        periods = self._of_all("period")
        if periods is None:
            return None
        begin, end = max(p[0] for p in periods), min(p[1] for p in periods)
        return (begin, end) if begin <= end else None

    @property
    def extent(self):
        boxes = self._of_all("extent")
        if boxes is None:
            return None
        return boxes[0] if len(boxes) == 1 else _overlap(boxes)

    @property
    def dtype(self):
        # NumPy's result type over all arguments (python scalars are weak), lifted to >= int32 / float32.
        # Blocks are immutable: computed once (request planning asks for it per tile).
        known = self.__dict__.get("_dtype_cache")
        if known is None:
            known = np.result_type(*[a.dtype if isinstance(a, RasterBlock) else a for a in self.args])
            if known == BOOL or np.issubdtype(known, np.integer):
                known = np.result_type(known, np.int32)
            elif np.issubdtype(known, np.floating):
                known = np.result_type(known, np.float32)
            self.__dict__["_dtype_cache"] = known
        return known

    @property
    def fillvalue(self):
        return None if self.dtype == BOOL else get_dtype_max(self.dtype)

    @property
    def geometry(self):
        shapes = self._of_all("geometry")
        if shapes is None:
            return None
        common = Extent.from_geometry(shapes[0])
        for shape in shapes[1:]:
            common = common.intersection(Extent.from_geometry(shape))
            if common is None:
                return None
        return shapes[0] if len(shapes) == 1 else common.as_geometry()

    @property
    def projection(self):
        found = {raster.projection for raster in self._sources}
        return found.pop() if len(found) == 1 else None  # (a lone None stays None)

    @property
    def geo_transform(self):
        grids = self._of_all("geo_transform")
        if grids is None:
            return None
        lead = GeoTransform(grids[0])
        return lead if all(lead.aligns_with(other) for other in grids[1:]) else None


def _allowed(value, kinds):
    if not isinstance(value, kinds):
        raise TypeError("'{}' object is not allowed".format(type(value)))


class BaseMath(BaseElementwise):
    def __init__(self, a, b):
        for operand in (a, b):
            _allowed(operand, (RasterBlock, np.ndarray, float, int))
        super().__init__(a, b)


class BaseComparison(BaseMath):
    dtype = property(lambda self: BOOL)


class BaseLogic(BaseElementwise):
    def __init__(self, a, b):
        for operand in (a, b):
            if not isinstance(operand, (RasterBlock, np.ndarray)):
                _allowed(operand, bool)
            elif operand.dtype != BOOL:
                raise TypeError("inputs must have boolean dtypes")
        super().__init__(a, b)

    dtype = property(lambda self: BOOL)
    fillvalue = property(lambda self: None)


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
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

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
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

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

    process.__func__.gm_op = "fillnodata"  # recognised by the fusion pass (fuse.py)


class BaseLogExp(BaseSingle):
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)
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
