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


def get_dtype_max(dtype):
    """Largest value of a dtype as a python scalar (utils.py:67-76)."""
    d = np.dtype(dtype)
    if d.kind == "f":
        return np.finfo(d).max.item()
    return np.iinfo(d).max


def get_dtype_min(dtype):
    """Smallest value of a dtype as a python scalar (utils.py:79-88)."""
    d = np.dtype(dtype)
    if d.kind == "f":
        return np.finfo(d).min.item()
    return np.iinfo(d).min


def get_int_dtype(n):
    """Smallest signed dtype holding ``n`` plus a spare no-data code (utils.py:91-97)."""
    for code in ("i1", "i2", "i4", "i8"):
        info = np.iinfo(code)
        if n - 1 <= info.max and n >= info.min:
            return np.dtype(code)
    raise ValueError("Value does not fit in int dtype ({})".format(n))


def get_uint_dtype(n):
    """Smallest unsigned dtype holding ``n`` plus a spare no-data code (utils.py:100-108)."""
    if n < 0:
        raise ValueError("Value does not fit in uint dtype ({})".format(n))
    for code in ("u1", "u2", "u4", "u8"):
        if n - 1 <= np.iinfo(code).max:
            return np.dtype(code)
    raise ValueError("Value does not fit in uint dtype ({})".format(n))


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
    """GDAL-style affine (p, a, b, q, c, d):  x = p + a*j + b*i ;  y = q + c*j + d*i."""

    def __new__(cls, tpl):
        return super().__new__(cls, tuple(tpl))

    def __init__(self, tpl):
        if len(self) != 6:
            raise ValueError("GeoTransform expected an iterable of length 6")
        if not all(math.isclose(self[i], 0.0, abs_tol=1e-7) for i in (2, 4)):
            raise ValueError("Tilted geo_transforms are not supported")
        if any(math.isclose(self[i], 0.0, abs_tol=1e-7) for i in (1, 5)):
            raise ValueError("Pixel size should not be zero")

    @classmethod
    def from_bbox(cls, bbox, height, width):
        x1, y1, x2, y2 = bbox
        return cls((x1, (x2 - x1) / width, 0, y2, 0, (y1 - y2) / height))

    def __repr__(self):
        return get_rounded_repr(tuple(self))

    @property
    def cell_area(self):
        _, a, b, _, c, d = self
        return abs(a * d - b * c)

    @property
    def origin(self):
        return self[0], self[3]

    @property
    def origin_normalized(self):
        return self[0] % self[1], self[3] % self[5]

    def get_inverse(self):
        _, a, b, _, c, d = self
        D = 1 / (a * d - b * c)
        return d * D, -b * D, -c * D, a * D

    def scale(self, x, y):
        p, a, b, q, c, d = self
        return self.__class__([p, a * x, b * x, q, c * y, d * y])

    def shift(self, origin):
        p, a, b, q, c, d = self
        i, j = origin
        return self.__class__([p + a * j + b * i, a, b, q + c * j + d * i, c, d])

    def get_indices(self, points):
        """(row indices, column indices) of world points, by floor()."""
        p, _, _, q, _, _ = self
        e, f, g, h = self.get_inverse()
        x, y = np.asarray(points).transpose()
        return (np.floor(g * (x - p) + h * (y - q)).astype(np.int64),
                np.floor(e * (x - p) + f * (y - q)).astype(np.int64))

    def get_bbox(self, offset, shape):
        p, a, b, q, c, d = self
        i, j = offset
        m, n = shape
        west = p + a * j + b * i
        north = q + c * j + d * i
        east = west + a * n + b * m
        south = north + c * n + d * m
        return west, south, east, north

    def aligns_with(self, other):
        if not isinstance(other, GeoTransform):
            other = GeoTransform(other)
        if abs(self[1]) != abs(other[1]) or abs(self[5]) != abs(other[5]):
            return False
        return self.origin_normalized == other.origin_normalized


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
