"""Raster sources (API of the reference's raster/sources.py:157-393, GDAL-free).

``MemorySource`` keeps the constructor, attributes and request planning of the
reference; its ``process`` -- a GDAL nearest-neighbour warp in the reference
(sources.py:119-149) -- is the HIP gather kernel K6 on a device-resident copy of the
array (uploaded once per device).  Only same-CRS requests are served: re-projection
needs PROJ/GDAL, which this path does not carry (out of scope).

``SyntheticSource`` is a procedural, device-side source for benchmarks and the
65536x65536 virtual mosaic: pixels are a counter-based function of their global
index, so any window of any size is generated in place without host data.
"""
import collections
import ctypes
import threading
import weakref
from datetime import datetime, timedelta, timezone

import numpy as np

from .. import utils
from ..hip import runtime
from ..hip.array import DeviceArray
from ..hip.dtypes import code_of
from . import resample
from .base import RasterBlock

__all__ = ["MemorySource", "SyntheticSource"]


def utc_from_ms_timestamp(timestamp):
    """Naive UTC datetime from a POSIX timestamp in milliseconds."""
    return datetime.fromtimestamp(timestamp / 1000, tz=timezone.utc).replace(tzinfo=None)


class RasterData(object):
    """What a source hands to ``process``: the whole array + its geo-referencing
    (sources.py:18-35).  Holds the host array and uploads it once per device."""

    _device_copies = {}  # (id(array), device) -> DeviceArray, dropped with the array

    def __init__(self, array, projection, geo_transform, no_data_value, metadata):
        self.array, self.projection, self.geo_transform = array, projection, geo_transform
        self.no_data_value, self.metadata = no_data_value, metadata

    def on_device(self):
        if isinstance(self.array, DeviceArray):
            return self.array
        key = (id(self.array), runtime.get_device())
        hit = RasterData._device_copies.get(key)
        if hit is None:
            hit = DeviceArray.from_numpy(self.array)
            RasterData._device_copies[key] = hit
            try:
                weakref.finalize(self.array, RasterData._device_copies.pop, key, None)
            except TypeError:
                pass
        return hit

    _finite = {}  # (id(array), device) -> bool

    def all_finite(self):
        """True when the array holds no NaN/inf (computed once per device, on the device): the
        source's non-finite -> no-data rewrite (sources.py:148) is then a no-op."""
        if np.dtype(self.array.dtype).kind != "f":
            return True
        dev = self.on_device()
        key = (id(self.array), dev.device)
        hit = RasterData._finite.get(key)
        if hit is None:
            out = ctypes.c_int(0)
            runtime.check(runtime.lib().gm_all_finite(ctypes.c_void_p(dev.ptr), code_of(dev.dtype), dev.size,
                                                      ctypes.byref(out)), "gm_all_finite")
            hit = bool(out.value)
            RasterData._finite[key] = hit
            try:
                weakref.finalize(self.array, RasterData._finite.pop, key, None)
            except TypeError:
                pass
        return hit

    def __gm_tokenize__(self):
        return ("RasterData", id(self.array), self.projection, tuple(self.geo_transform))


class RasterSourceBase(RasterBlock):
    @staticmethod
    def process(process_kwargs):
        """Counterpart of RasterSourceBase.process (sources.py:38-149) without GDAL."""
        mode = process_kwargs["mode"]
        if mode == "empty_vals":
            return None
        if mode == "empty_time":
            return {"time": []}
        if mode == "empty_meta":
            return {"meta": []}
        bands = process_kwargs["bands"]
        length = bands[1] - bands[0]
        if mode == "time":
            start, delta = process_kwargs["start"], process_kwargs["delta"]
            return {"time": [start + i * delta for i in range(length)]}
        raster_data = process_kwargs["raster_data"]
        if mode == "meta":
            return {"meta": [raster_data.metadata[i] for i in range(bands[0], bands[1])]}

        dtype = np.dtype(process_kwargs["dtype"])
        bbox = process_kwargs["bbox"]
        width, height = process_kwargs["width"], process_kwargs["height"]
        fill = process_kwargs["fillvalue"]
        fill = fill.item() if hasattr(fill, "item") else fill
        window = process_kwargs.get("window")
        if window is not None and length >= 1 and width and height and raster_data.all_finite():
            # pixel-aligned window inside the source (the planning has checked CRS, cell size and bounds: _aligned_window):
            # a zero-copy view (row pitch = source width) that the element-wise program kernel reads in place; other
            # consumers compact it on demand.  First, because it is what every tile of a tiled view asks for.
            runtime.require_device()
            source = raster_data.on_device()
            if source.shape[0] != length:
                source = source.band_slice(bands[0], bands[1])  # (several frames: a window whose bands lie a plane apart)
            if process_kwargs.get("whole"):
                return {"values": source, "no_data_value": fill}  # the request grid IS the source grid
            values = source.window(window[0], window[1], height, width)
            values.resident = True
            return {"values": values, "no_data_value": fill}
        if not utils.same_projection(process_kwargs["projection"], raster_data.projection):
            raise NotImplementedError(
                "request projection {} differs from the source's {}: re-projection is outside the "
                "MI355X raster path".format(process_kwargs["projection"], raster_data.projection))
        if width == 0 or height == 0:  # sources.py:91-92
            return np.empty((length, height, width), dtype=dtype)
        gt = utils.GeoTransform(raster_data.geo_transform)
        if bbox[0] == bbox[2] or bbox[1] == bbox[3]:
            # point request: one pixel by floor() of the source pixel coordinate (:94-117)
            rows, cols = gt.get_indices(((bbox[0], bbox[1]),))
            i, j = int(rows[0]), int(cols[0])
            source = raster_data.array
            count, ny, nx = source.shape
            result = np.full((length, 1, 1), fill, dtype=dtype)
            if 0 <= i < ny and 0 <= j < nx and length > 0:
                if isinstance(source, DeviceArray):
                    # the cell of every requested band in ONE strided copy (rows of one cell, a plane apart) and one transfer
                    # -- the time series of a location; a source may be planes of 16 GiB resident in HBM, and a copy and a
                    # transfer per band cost 45 us each (a request over 8 frames: 380 us)
                    source = source.contiguous()
                    item = source.dtype.itemsize
                    cells = DeviceArray.empty((length, 1, 1), source.dtype)
                    L = runtime.lib()
                    L.gm_copy_2d.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_size_t, ctypes.c_size_t]
                    runtime.check(L.gm_copy_2d(ctypes.c_void_p(cells.ptr), item,
                                               ctypes.c_void_p(source.ptr + ((bands[0] * ny + i) * nx + j) * item), ny * nx * item,
                                               item, length), "gm_copy_2d")
                    if cells.dtype == dtype:
                        # (the series stays where it is: what reads it next -- an aggregation, an element-wise view -- runs
                        #  on the device, and a transfer to the host and back cost 40 us of such a request's 124)
                        return {"values": cells, "no_data_value": fill}
                    result[:, 0, 0] = cells.to_numpy()[:, 0, 0]
                else:
                    result[:, 0, 0] = source[bands[0]:bands[1], i, j]
            return {"values": result, "no_data_value": fill}
        runtime.require_device()
        source = raster_data.on_device().band_slice(bands[0], bands[1])
        if process_kwargs.get("whole") and raster_data.all_finite():
            # the request grid IS the source grid: the warp would be an identity copy, so the
            # resident tile itself is handed on (responses are immutable by contract)
            source.resident = True
            return {"values": source, "no_data_value": fill}
        window = process_kwargs.get("window")
        if window is not None and length == 1 and raster_data.all_finite():
            # pixel-aligned window inside the source: a zero-copy view (row pitch = source width) that
            # the element-wise program kernel reads in place; other consumers compact it on demand
            return {"values": source.window(window[0], window[1], height, width), "no_data_value": fill}
        values = resample.nearest_warp(source, gt, bbox, height, width, fill)
        return {"values": values, "no_data_value": fill}


def _as_frames(data):
    """The pixel values as a (t, y, x) stack: host arrays through NumPy (int64 narrowed to int32: sources.py:201-203,
    a GDAL limitation kept for parity), device-resident arrays as they are; 2-D data is one frame."""
    if not isinstance(data, DeviceArray):
        data = np.asarray(data)
        if data.dtype == np.dtype("i8"):
            data = data.astype("i4")
        if data.ndim == 2:
            data = data[np.newaxis]
    elif data.ndim == 2:
        data = data.reshape((1,) + data.shape)
    if data.ndim != 3:
        raise ValueError("data should be two- or three-dimensional.")
    return data


def _pair(value, name, scalar_ok=False):
    """[float, float] from a sequence of two (or, for the pixel size, one number for both axes)."""
    if scalar_ok and not hasattr(value, "__iter__"):
        value = (value, value)
    pair = [float(x) for x in value]
    if len(pair) != 2:
        raise ValueError("{} should have length 2".format(name))
    return pair


def _milliseconds(delta):
    """A time step as integer milliseconds (None stays None)."""
    if delta is None:
        return None
    if isinstance(delta, timedelta):
        return int(delta.total_seconds() * 1000)
    return int(delta)


class MemorySource(RasterSourceBase):
    """A raster source that interfaces data from memory (sources.py:157-393).

    The pixel with its top-left corner at [x, y] covers [x, x + dx) and (y - dy, y].

    :param data: the pixel values: (y, x) or (t, y, x) ndarray (or a DeviceArray that
        already lives in HBM)
    :param no_data_value: the pixel value that designates 'no data'
    :param projection: the projection of the pixels ('EPSG:<code>' or WKT text)
    :param pixel_size: the size of one pixel; (x, y) if they differ
    :param pixel_origin: the location (x, y) of the pixel with index (0, 0)
    :param time_first: timestamp of the first frame (ms since 1-1-1970 or datetime)
    :param time_delta: difference between two frames (ms or timedelta)
    :param metadata: a list of metadata corresponding to the frames
    """
    gm_local = True  # every cell of the result depends on that cell of the sources alone (plan.py: tiles of a mosaic)

    def __init__(self, data, no_data_value, projection, pixel_size, pixel_origin, time_first=0,
                 time_delta=None, metadata=None):
        frames = _as_frames(data)
        stamp = utils.dt_to_ms(time_first) if isinstance(time_first, datetime) else int(time_first)
        step = _milliseconds(time_delta)
        if step is None and frames.shape[0] > 1:
            raise ValueError("time_delta is required for temporal data")
        if metadata is not None:
            metadata = list(metadata)
            if len(metadata) != frames.shape[0]:
                raise ValueError("Metadata length should match data length")
        super().__init__(frames, frames.dtype.type(no_data_value), utils.get_epsg_or_wkt(projection),
                         _pair(pixel_size, "pixel_size", scalar_ok=True), _pair(pixel_origin, "pixel_origin"),
                         stamp, step, metadata)

    @property
    def token(self):
        # a device-resident array is identified by its checksum rather than by its bytes
        try:
            return self._cached_token
        except AttributeError:
            pass
        if isinstance(self.args[0], DeviceArray):
            from ..core.graphs import tokenize
            d = self.args[0]
            self._cached_token = tokenize(self.get_import_path(), ("device", d.shape, d.dtype.str, d.checksum()),
                                          *self.args[1:])
            return self._cached_token
        return super().token

    data = property(lambda self: self.args[0])
    no_data_value = property(lambda self: self.args[1])
    projection = property(lambda self: self.args[2])
    pixel_size = property(lambda self: self.args[3])
    pixel_origin = property(lambda self: self.args[4])
    time_first = property(lambda self: self.args[5])
    time_delta = property(lambda self: self.args[6])
    metadata = property(lambda self: self.args[7])

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def fillvalue(self):
        return self.no_data_value

    @property
    def geo_transform(self):
        p, q = self.pixel_origin
        a, d = self.pixel_size
        return utils.GeoTransform((p, a, 0, q, 0, -d))

    def _get_extent(self):
        if not self.data.size:
            return None
        bbox = self.geo_transform.get_bbox((0, 0), self.data.shape[1:])
        return utils.Extent(bbox, self.projection)

    @property
    def extent(self):
        extent = self._get_extent()
        if extent is None:
            return None
        return extent.transformed("EPSG:4326").bbox  # needs PROJ unless already WGS84

    @property
    def geometry(self):
        extent = self._get_extent()
        return None if extent is None else extent.as_geometry()

    def __len__(self):
        return self.data.shape[0]

    @property
    def period(self):
        hit = self.__dict__.get("_period_cache")  # (a Block never changes: request planning asks per request and per parent)
        if hit is None:
            n = len(self)
            if n == 0:
                return None
            first = utc_from_ms_timestamp(self.time_first)
            hit = self.__dict__["_period_cache"] = (first, first) if n == 1 else (first, first + (n - 1) * self.timedelta)
        return hit

    @property
    def timedelta(self):
        if self.time_delta is None:
            return None
        hit = self.__dict__.get("_timedelta_cache")
        if hit is None:
            hit = self.__dict__["_timedelta_cache"] = timedelta(milliseconds=self.time_delta)
        return hit

    @property
    def temporal(self):
        return self.time_delta is not None

    def _grid(self):
        """(geo_transform, (west, south, east, north)) of the source, computed once (a tiled view asks
        per tile)."""
        hit = self.__dict__.get("_grid_cache")
        if hit is None:
            gt = self.geo_transform
            hit = self.__dict__["_grid_cache"] = (gt, gt.get_bbox((0, 0), self.data.shape[1:]))
        return hit

    def _same_crs(self, projection):
        memo = self.__dict__.setdefault("_crs_memo", {})
        hit = memo.get(projection)
        if hit is None:
            hit = memo[projection] = utils.same_projection(projection, self.projection)
        return hit

    def _aligned_window(self, request):
        """(row0, col0) when the request grid coincides with source pixels -- same CRS, one request
        cell per source pixel, bbox on pixel edges, entirely inside the source -- else None.  Such
        a request (every tile a RasterTiler cuts from a pixel-aligned view) needs no resampling:
        the nearest-neighbour warp of sources.py:133-144 would copy the window unchanged."""
        if not self.data.size or not self._same_crs(request["projection"]):
            return None
        _, ny, nx = self.data.shape
        p, a, _, q, _, d = self._grid()[0]
        x1, y1, x2, y2 = request["bbox"]
        w, h = request["width"], request["height"]
        if w <= 0 or h <= 0 or a <= 0 or d >= 0:
            return None
        if (x2 - x1) != w * a or (y2 - y1) != h * -d:
            return None
        col0, row0 = (x1 - p) / a, (y2 - q) / d
        if col0 != int(col0) or row0 != int(row0):
            return None
        col0, row0 = int(col0), int(row0)
        if col0 < 0 or row0 < 0 or col0 + w > nx or row0 + h > ny:
            return None
        return row0, col0

    def gm_plan_response(self, request):
        """Short cut for evaluation plans (plan.py): the response ``process`` gives for a 'vals' request that
        is a pixel-aligned window of a single-frame source -- the zero-copy window itself -- without the
        general request planning; None when the request is of any other kind (the caller then takes
        ``get_sources_and_requests`` + ``process``)."""
        if self.data.shape[0] != 1 or request.get("mode") != "vals":
            return None
        start, stop = request.get("start"), request.get("stop")
        if start is not None and stop is not None:
            # a time window (the element-wise Blocks clamp start / stop to the period, elemwise.py:64-78): the one frame
            # answers it exactly when it lies inside (utils.snap_start_stop); anything else takes the general path
            moment = self.period[0]
            try:
                if not (start <= moment <= stop):
                    return None
            except TypeError:
                return None
        window = self._aligned_window(request)
        if window is None:
            return None
        raster_data = self.__dict__.get("_raster_data")
        if raster_data is None:
            return None  # (the general path creates it on first use)
        if not raster_data.all_finite():
            return None
        fill = self.fillvalue
        fill = fill.item() if hasattr(fill, "item") else fill
        values = raster_data.on_device().window(window[0], window[1], request["height"], request["width"])
        values.resident = True  # (a zero-copy window of the resident raster: the same bytes for the same request every time)
        return {"values": values, "no_data_value": fill}

    def get_sources_and_requests(self, **request):
        mode = request["mode"]
        if mode == "meta" and self.metadata is None:
            return [({"mode": "empty_meta"}, None)]
        # (the time planning does not depend on the bounding box: a tiled view asks once per tile)
        when = (request.get("start"), request.get("stop"))
        memo = self.__dict__.get("_snap_memo")
        if memo is None or memo[0] != when:
            memo = self.__dict__["_snap_memo"] = (when, utils.snap_start_stop(
                when[0], when[1], utc_from_ms_timestamp(self.time_first), self.timedelta, len(self)))
        start, stop, band1, band2 = memo[1]
        if start is None:
            return [({"mode": "empty_" + mode}, None)]
        bands = band1, band2 + 1
        raster_data = self.__dict__.get("_raster_data")
        if raster_data is None:  # one RasterData per source: it also carries the per-device upload
            raster_data = self.__dict__["_raster_data"] = RasterData(
                array=self.data, metadata=self.metadata, geo_transform=self.geo_transform,
                no_data_value=self.no_data_value, projection=self.projection)
        if mode == "vals":
            window = self._aligned_window(request)
            # (the source's own grid = the aligned window at (0, 0) that covers it)
            whole = window == (0, 0) and (request["height"], request["width"]) == tuple(self.data.shape[1:])
            kwargs = {"mode": "vals", "raster_data": raster_data, "bbox": request["bbox"],
                      "width": request["width"], "height": request["height"],
                      "projection": request["projection"], "bands": bands, "dtype": self.dtype,
                      "fillvalue": self.fillvalue, "whole": whole, "window": window}
        elif mode == "meta":
            kwargs = {"mode": "meta", "raster_data": raster_data, "bands": bands}
        elif mode == "time":
            kwargs = {"mode": "time", "start": start, "delta": self.timedelta or timedelta(0), "bands": bands}
        else:
            raise RuntimeError("Unknown mode '{}'".format(mode))
        return [(kwargs, None)]


class SyntheticSource(RasterBlock):
    """Procedural single-band raster living on the device: pixel (row, col) of the
    ``shape`` = (H, W) mosaic is ``lo + u * (hi - lo)`` with u = splitmix64-uniform of
    (seed, row * W + col); every ``lattice`` = (ny, nx)-th pixel is no-data.  Requests must be
    aligned with the pixel grid (bbox on pixel edges, one cell per pixel).

    ``resident=True`` keeps every window it has generated in HBM (a per-device tile cache keyed by the
    window): a worker that owns some tiles of a mosaic then holds exactly the leaf windows of those
    tiles -- 1/N of the mosaic per GPU when the tiles are dealt over N workers -- and later requests
    for the same tile are served without regenerating it (the responses are immutable by contract)."""

    # per-device tile cache of the resident sources: (spec, device, row0, col0, h, w) -> DeviceArray, least recently
    # used first.  Bounded by bytes (geomodeling.resident-window-bytes, default 192 GiB of the MI355X's 288 GB): a
    # source asked for ever new windows (sliding boxes, stencil-grown requests) evicts its oldest instead of pinning
    # HBM until drop_resident_windows() is called by hand; guarded by a lock (get_data runs on scheduler threads).
    _windows = collections.OrderedDict()
    _window_bytes = 0
    _window_lock = threading.Lock()

    def __init__(self, shape, dtype, seed, lo, hi, no_data_value, lattice=(0, 0), projection="EPSG:28992",
                 pixel_size=1.0, pixel_origin=(0.0, 0.0), resident=False):
        shape = tuple(int(s) for s in shape)
        if len(shape) != 2:
            raise ValueError("shape should be (height, width)")
        dtype = np.dtype(dtype).str
        code_of(dtype)
        super().__init__(shape, dtype, int(seed), float(lo), float(hi),
                         np.dtype(dtype).type(no_data_value).item(),
                         tuple(int(x) for x in lattice), utils.get_epsg_or_wkt(projection),
                         float(pixel_size), tuple(float(x) for x in pixel_origin), bool(resident))

    shape = property(lambda self: self.args[0])
    dtype = property(lambda self: np.dtype(self.args[1]))
    fillvalue = property(lambda self: np.dtype(self.args[1]).type(self.args[5]))
    projection = property(lambda self: self.args[7])
    temporal = False
    timedelta = None
    period = (RasterBlock.DEFAULT_ORIGIN, RasterBlock.DEFAULT_ORIGIN)

    @property
    def geo_transform(self):
        size, (p, q) = self.args[8], self.args[9]
        return utils.GeoTransform((p, size, 0, q, 0, -size))

    @property
    def geometry(self):
        return utils.Extent(self.geo_transform.get_bbox((0, 0), self.shape), self.projection).as_geometry()

    @property
    def extent(self):
        return utils.Extent(self.geo_transform.get_bbox((0, 0), self.shape),
                            self.projection).transformed("EPSG:4326").bbox

    def get_sources_and_requests(self, **request):
        mode = request["mode"]
        if mode == "time":
            return [({"mode": "time"}, None)]
        if mode == "meta":
            return [({"mode": "meta"}, None)]
        size, (p, q) = self.args[8], self.args[9]
        x1, y1, x2, y2 = request["bbox"]
        col0, row0 = (x1 - p) / size, (q - y2) / size
        w, h = request["width"], request["height"]
        aligned = (col0 == int(col0) and row0 == int(row0) and (x2 - x1) == w * size and (y2 - y1) == h * size)
        if not aligned or not utils.same_projection(request["projection"], self.projection):
            raise NotImplementedError("SyntheticSource only serves pixel-aligned same-CRS requests")
        kwargs = {"mode": "vals", "spec": self.args, "row0": int(row0), "col0": int(col0), "height": h, "width": w}
        return [(kwargs, None)]

    @staticmethod
    def process(process_kwargs):
        mode = process_kwargs["mode"]
        if mode == "time":
            return {"time": [RasterBlock.DEFAULT_ORIGIN]}
        if mode == "meta":
            return {"meta": [None]}
        spec = process_kwargs["spec"]
        shape, dtype, seed, lo, hi, ndv, lattice = spec[:7]
        dtype = np.dtype(dtype)
        window = (process_kwargs["row0"], process_kwargs["col0"], process_kwargs["height"], process_kwargs["width"])
        resident = len(spec) > 10 and spec[10]
        values = None
        cls = SyntheticSource
        if resident:
            runtime.require_device()
            key = (tuple(spec[:7]), runtime.get_device()) + window
            with cls._window_lock:
                values = cls._windows.get(key)
                if values is not None:
                    cls._windows.move_to_end(key)
        if values is None:
            values = generate_window(shape, dtype, seed, lo, hi, ndv, lattice, *window)
            if resident:
                from .. import config as _config
                limit = int(_config.get("geomodeling.resident-window-bytes", 192 << 30))
                size = int(values.size) * dtype.itemsize
                with cls._window_lock:
                    if key not in cls._windows:  # (another thread may have generated the same window meanwhile)
                        values.resident = True
                        cls._windows[key] = values
                        cls._window_bytes += size
                    while cls._window_bytes > limit and len(cls._windows) > 1:
                        _, old = cls._windows.popitem(last=False)
                        cls._window_bytes -= int(old.size) * old.dtype.itemsize
        return {"values": values, "no_data_value": dtype.type(ndv)}

    @classmethod
    def drop_resident_windows(cls):
        """Give the cached windows of all resident SyntheticSources back to the allocator."""
        with cls._window_lock:
            cls._windows.clear()
            cls._window_bytes = 0
        from ..hip import program as _program
        _program.drop_kept_launches()  # (launches kept for those windows hold them)


def generate_window(shape, dtype, seed, lo, hi, ndv, lattice, row0, col0, height, width):
    """Window [row0:row0+height, col0:col0+width] of the procedural (H, W) mosaic -> (1, h, w)
    DeviceArray; pixels outside the mosaic are no-data."""
    runtime.require_device()
    dtype = np.dtype(dtype)
    out = DeviceArray.empty((1, height, width), dtype)
    runtime.check(runtime.lib().gm_generate(ctypes.c_void_p(out.ptr), code_of(dtype), int(seed), int(row0), int(col0),
                                int(height), int(width), int(shape[0]), int(shape[1]), float(lo), float(hi),
                                int(lattice[0]), int(lattice[1]), runtime.scalar_bits(ndv, dtype)), "gm_generate")
    return out
