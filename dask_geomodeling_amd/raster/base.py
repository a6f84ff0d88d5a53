"""RasterBlock contract and operator DSL (counterpart of raster/base.py:9-228).

A raster Block exposes ``period, timedelta, extent, dtype, fillvalue, geometry,
projection, geo_transform, temporal`` (None when empty).  Requests are dicts with
``mode ('vals'|'time'|'meta'), bbox, projection, width, height, start, stop``;
responses are ``None`` or ``{"values": (bands, H, W) tile, "no_data_value": x}`` /
``{"time": [...]}`` / ``{"meta": [...]}``.  On this path ``values`` is a
``DeviceArray`` between Blocks and an ndarray at the end of ``get_data``.
"""
from datetime import datetime as Datetime

from ..core import Block


class RasterBlock(Block):
    DEFAULT_ORIGIN = Datetime(1970, 1, 1, 0, 0)

    def __len__(self):
        """Number of temporal bands (base.py:75-94)."""
        period = self.period
        if period is None:
            return 0
        start, stop = period
        if start == stop:
            return 1
        delta = self.timedelta
        if delta is None:
            return len(self.get_data(mode="time", start=start, stop=stop)["time"])
        return int((stop - start).total_seconds() / delta.total_seconds()) + 1

    # operator DSL (base.py:96-174) ------------------------------------------------
    def __add__(self, other):
        from . import Add
        return Add(self, other)

    def __mul__(self, other):
        from . import Multiply
        return Multiply(self, other)

    def __neg__(self):
        from . import Multiply
        return Multiply(self, -1)

    def __sub__(self, other):
        from . import Subtract
        return Subtract(self, other)

    def __truediv__(self, other):
        from . import Divide
        return Divide(self, other)

    def __pow__(self, other):
        from . import Power
        return Power(self, other)

    def __eq__(self, other):
        from . import Equal
        return Equal(self, other)

    def __ne__(self, other):
        from . import NotEqual
        return NotEqual(self, other)

    def __gt__(self, other):
        from . import Greater
        return Greater(self, other)

    def __ge__(self, other):
        from . import GreaterEqual
        return GreaterEqual(self, other)

    def __lt__(self, other):
        from . import Less
        return Less(self, other)

    def __le__(self, other):
        from . import LessEqual
        return LessEqual(self, other)

    def __invert__(self):
        from . import Invert
        return Invert(self)

    def __and__(self, other):
        from . import And
        return And(self, other)

    def __or__(self, other):
        from . import Or
        return Or(self, other)

    def __xor__(self, other):
        from . import Xor
        return Xor(self, other)

    # __eq__ is overloaded: keep Blocks hashable (dict keys in graph tooling)
    __hash__ = object.__hash__


class BaseSingle(RasterBlock):
    """Base of Blocks that adjust a single raster: attributes pass through ``store``."""

    def __init__(self, store, *args):
        if not isinstance(store, RasterBlock):
            raise TypeError("'{}' object is not allowed".format(type(store)))
        super(BaseSingle, self).__init__(store, *args)

    @property
    def store(self):
        return self.args[0]

    def __len__(self):
        return len(self.store)

    extent = property(lambda self: self.store.extent)
    period = property(lambda self: self.store.period)
    timedelta = property(lambda self: self.store.timedelta)
    temporal = property(lambda self: self.store.temporal)
    dtype = property(lambda self: self.store.dtype)
    fillvalue = property(lambda self: self.store.fillvalue)
    geometry = property(lambda self: self.store.geometry)
    projection = property(lambda self: self.store.projection)
    geo_transform = property(lambda self: self.store.geo_transform)
