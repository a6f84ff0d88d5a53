"""Argument checks shared by the Block constructors.  The reference reports a wrong argument with
``TypeError("'<type>' object is not allowed")`` throughout (raster/misc.py, raster/elemwise.py ...); the texts are
part of its behaviour (its tests match on them), so they are produced here, once."""
from .base import RasterBlock

NUMBER = (float, int)


def not_allowed(obj):
    return TypeError("'{}' object is not allowed".format(type(obj)))


def need_raster(*candidates, blame=None):
    """Every candidate must be a RasterBlock; ``blame`` names the object the message mentions instead."""
    for candidate in candidates:
        if not isinstance(candidate, RasterBlock):
            raise not_allowed(candidate if blame is None else blame)


def need_number(*candidates):
    for candidate in candidates:
        if not isinstance(candidate, NUMBER):
            raise not_allowed(candidate)


def need_iterable(candidate):
    if not hasattr(candidate, "__iter__"):
        raise not_allowed(candidate)


def argument(position):
    """Read-only attribute standing for ``self.args[position]``."""
    return property(lambda self: self.args[position])
