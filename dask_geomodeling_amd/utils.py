"""Host-side helpers of the raster path (GDAL-free counterparts of the reference's
utils.py: dtype/no-data helpers :61-108, Extent :128-204, GeoTransform :208-393,
snap_start_stop :848-915).  Pure scalar / metadata arithmetic -- no tile numerics.
"""
import math
from datetime import datetime, timedelta, timezone
from math import floor, log10

import numpy as np

__all__ = ["get_dtype_max", "get_dtype_min", "get_int_dtype", "get_uint_dtype", "GeoTransform",
           "Extent", "snap_start_stop", "dt_to_ms", "get_epsg_or_wkt", "parse_epsg"]


def _limits(dtype):
    dtype = np.dtype(dtype)
    return np.finfo(dtype) if dtype.kind == "f" else np.iinfo(dtype)


def get_dtype_max(dtype):
    """Largest value of a dtype as a python scalar (utils.py:67-76): float for float dtypes, int otherwise."""
    top = _limits(dtype).max
    return top.item() if hasattr(top, "item") else top


def get_dtype_min(dtype):
    """Smallest value of a dtype as a python scalar (utils.py:79-88)."""
    bottom = _limits(dtype).min
    return bottom.item() if hasattr(bottom, "item") else bottom


def _narrowest(codes, fits, kind, n):
    for code in codes:
        if fits(np.iinfo(code)):
            return np.dtype(code)
    raise ValueError("Value does not fit in {} dtype ({})".format(kind, n))


def get_int_dtype(n):
    """Smallest signed dtype holding ``n`` with its maximum left over as a no-data code (utils.py:91-97)."""
    return _narrowest(("i1", "i2", "i4", "i8"), lambda info: info.min <= n and n - 1 <= info.max, "int", n)


def get_uint_dtype(n):
    """Smallest unsigned dtype holding ``n`` with its maximum left over as a no-data code (utils.py:100-108)."""
    return _narrowest(("u1", "u2", "u4", "u8") if n >= 0 else (), lambda info: n - 1 <= info.max, "uint", n)


def get_rounded_repr(obj, significant=4, fmt="{} (rounded)"):
    digits = (-int(floor(log10(abs(n)))) + (significant - 1) if n else None for n in obj)
    rounded = obj.__class__(round(n, d) if n else n for n, d in zip(obj, digits))
    if obj == rounded:
        return repr(obj)
    return fmt.format(repr(rounded))


def parse_epsg(projection):
    """'EPSG:28992' / 'epsg:28992' / 28992 -> 28992, else None."""
    if isinstance(projection, int):
        return projection
    text = str(projection).strip()
    if text.upper().startswith("EPSG:"):
        try:
            return int(text.split(":", 1)[1])
        except ValueError:
            return None
    return None


def get_epsg_or_wkt(text):
    """Normalise a projection string (the reference resolves WKT through GDAL; this
    path is same-CRS only, so the normal form is 'EPSG:<code>' or the text itself)."""
    code = parse_epsg(text)
    return "EPSG:{}".format(code) if code is not None else str(text)


def same_projection(a, b):
    return get_epsg_or_wkt(a) == get_epsg_or_wkt(b)


class Extent(object):
    """Bounding box + projection.  Re-projection needs PROJ, which this path does not
    carry: ``transformed`` only accepts the extent's own projection."""

    def __init__(self, bbox, sr):
        self.bbox = tuple(bbox)
        self.srs = get_epsg_or_wkt(sr) if not isinstance(sr, Extent) else sr.srs

    def __repr__(self):
        return "<{}: {}>".format(self.__class__.__name__, self.bbox)

    @property
    def width(self):
        return self.bbox[2] - self.bbox[0]

    @property
    def height(self):
        return self.bbox[3] - self.bbox[1]

    def transformed(self, sr):
        if same_projection(sr, self.srs):
            return Extent(self.bbox, self.srs)
        raise NotImplementedError(
            "re-projection ({} -> {}) is outside the MI355X raster path (no PROJ/GDAL)".format(self.srs, sr))

    def intersection(self, other):
        a, b = self.bbox, other.transformed(self.srs).bbox
        result = Extent((max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])), self.srs)
        if result.width > 0 and result.height > 0:
            return result
        return None

    def union(self, other):
        a, b = self.bbox, other.transformed(self.srs).bbox
        return Extent((min(a[0], b[0]), min(a[1], b[1]), max(a[2], b[2]), max(a[3], b[3])), self.srs)

    def as_geometry(self):
        return BoxGeometry(self.bbox, self.srs)

    @classmethod
    def from_geometry(cls, geometry):
        return cls(geometry.bbox, geometry.srs)


class BoxGeometry(object):
    """Axis-aligned rectangle standing in for the OGR polygon of the reference's
    ``geometry`` attribute (enough for envelope / intersection bookkeeping)."""

    def __init__(self, bbox, srs):
        self.bbox = tuple(float(x) for x in bbox)
        self.srs = srs

    def GetEnvelope(self):
        x1, y1, x2, y2 = self.bbox
        return x1, x2, y1, y2

    def ExportToWkt(self):
        x1, y1, x2, y2 = self.bbox
        return "POLYGON (({0} {1},{2} {1},{2} {3},{0} {3},{0} {1}))".format(x1, y1, x2, y2)

    def __eq__(self, other):
        return isinstance(other, BoxGeometry) and self.bbox == other.bbox and self.srs == other.srs

    def __hash__(self):
        return hash((self.bbox, self.srs))

    def __gm_tokenize__(self):
        return ("BoxGeometry", self.bbox, self.srs)


class GeoTransform(tuple):
    """GDAL-style affine (p, a, b, q, c, d):  x = p + a*j + b*i ;  y = q + c*j + d*i  for the cell in row i,
    column j.  North-up grids only: b = c = 0 (utils.py:208-393)."""

    def __new__(cls, tpl):
        return super().__new__(cls, tuple(tpl))

    def __init__(self, tpl):
        if len(self) != 6:
            raise ValueError("GeoTransform expected an iterable of length 6")
        near_zero = [math.isclose(term, 0.0, abs_tol=1e-7) for term in self]
        if not (near_zero[2] and near_zero[4]):
            raise ValueError("Tilted geo_transforms are not supported")
        if near_zero[1] or near_zero[5]:
            raise ValueError("Pixel size should not be zero")

    @classmethod
    def from_bbox(cls, bbox, height, width):
        west, south, east, north = bbox
        return cls((west, (east - west) / width, 0, north, 0, (south - north) / height))

    def __repr__(self):
        return get_rounded_repr(tuple(self))

    def _terms(self):
        return tuple.__iter__(self)

    @property
    def cell_area(self):
        _, a, b, _, c, d = self._terms()
        return abs(a * d - b * c)

    origin = property(lambda self: (self[0], self[3]))
    origin_normalized = property(lambda self: (self[0] % self[1], self[3] % self[5]))

    def get_inverse(self):
        """The 2 x 2 part inverted: (e, f, g, h) with  j = e dx + f dy,  i = g dx + h dy."""
        _, a, b, _, c, d = self._terms()
        det = 1 / (a * d - b * c)
        return d * det, -b * det, -c * det, a * det

    def scale(self, x, y):
        p, a, b, q, c, d = self._terms()
        return type(self)([p, a * x, b * x, q, c * y, d * y])

    def shift(self, origin):
        row, col = origin
        p, a, b, q, c, d = self._terms()
        return type(self)([p + a * col + b * row, a, b, q + c * col + d * row, c, d])

    def get_indices(self, points):
        """(row indices, column indices) of world points, by floor()."""
        e, f, g, h = self.get_inverse()
        xs, ys = np.asarray(points).transpose()
        dx, dy = xs - self[0], ys - self[3]
        return np.floor(g * dx + h * dy).astype(np.int64), np.floor(e * dx + f * dy).astype(np.int64)

    def get_bbox(self, offset, shape):
        """(west, south, east, north) of the ``shape`` = (rows, columns) cells from row / column ``offset`` on."""
        p, a, b, q, c, d = self._terms()
        (row, col), (rows, cols) = offset, shape
        west, north = p + a * col + b * row, q + c * col + d * row
        return west, north + c * cols + d * rows, west + a * cols + b * rows, north

    def aligns_with(self, other):
        other = other if isinstance(other, GeoTransform) else GeoTransform(other)
        same_cells = abs(self[1]) == abs(other[1]) and abs(self[5]) == abs(other[5])
        return same_cells and self.origin_normalized == other.origin_normalized


def dt_to_ms(dt):
    """Naive-UTC (or aware) datetime -> POSIX milliseconds."""
    if dt.tzinfo is None:
        dt = dt.replace(tzinfo=timezone.utc)
    return int(dt.timestamp() * 1000)


def snap_start_stop(start, stop, time_first, time_delta, length):
    """Interpret 'start'/'stop' of a request against an equidistant time axis
    (utils.py:848-915).  Returns (start, stop, first_index, last_index)."""
    if length == 0:
        return (None,) * 4
    if length == 1:
        time_delta = None
        period = (time_first, time_first)
    elif time_delta is None:
        raise ValueError("Length > 1 requires a timedelta")
    else:
        period = (time_first, time_first + (length - 1) * time_delta)

    if start is None:  # latest frame
        return period[1], period[1], length - 1, length - 1
    if stop is None:  # frame nearest to start
        if start <= period[0] or length == 1:
            return period[0], period[0], 0, 0
        if start >= period[1]:
            return period[1], period[1], length - 1, length - 1
        i = int(round((start - period[0]) / time_delta))
        t = period[0] + time_delta * i
        return t, t, i, i
    if start > period[1] or stop < period[0]:
        return (None,) * 4
    if length == 1:
        return period[0], period[0], 0, 0
    first = max(int(np.ceil((start - period[0]) / time_delta)), 0)
    last = min(int(np.floor((stop - period[0]) / time_delta)), length - 1)
    return period[0] + time_delta * first, period[0] + time_delta * last, first, last
