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

    # The operator DSL (base.py:96-174) is attached below from a table: ``a + b`` -> Add(a, b) ...

    # __eq__ is overloaded: keep Blocks hashable (dict keys in graph tooling)
    __hash__ = object.__hash__


def _attach_operators():
    """``raster + other`` builds the element-wise Block of the same meaning (base.py:96-174).  The
    Block classes live in sibling modules that import this one, hence the lookup at call time."""
    binary = {"__add__": "Add", "__sub__": "Subtract", "__mul__": "Multiply", "__truediv__": "Divide",
              "__pow__": "Power", "__eq__": "Equal", "__ne__": "NotEqual", "__gt__": "Greater",
              "__ge__": "GreaterEqual", "__lt__": "Less", "__le__": "LessEqual", "__and__": "And",
              "__or__": "Or", "__xor__": "Xor"}

    def make_binary(block_name):
        def operator(self, other):
            from . import elemwise
            return getattr(elemwise, block_name)(self, other)
        return operator

    for dunder, block_name in binary.items():
        setattr(RasterBlock, dunder, make_binary(block_name))

    def negate(self):
        from . import elemwise
        return elemwise.Multiply(self, -1)

    def invert(self):
        from . import elemwise
        return elemwise.Invert(self)

    RasterBlock.__neg__ = negate
    RasterBlock.__invert__ = invert


_attach_operators()


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
