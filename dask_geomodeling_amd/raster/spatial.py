"""Spatial Blocks: the 3x3 / small-radius stencils of the reference (Dilate, MovingMax, HillShade,
raster/spatial.py:111-213, 320-440), Place (spatial.py:441-731) and SpatialResample.

The stencil Blocks enlarge the request by their radius (``expand_request_pixels``,
spatial.py:27-47) -- the halo is recomputed per tile, never exchanged, so tiles stay independent --
and hand the enlarged tile to the kernels of csrc/gm_stencil.hip, which write the cropped result.


The reference has no stand-alone spatial resample Block (``raster.Resample`` is temporal);
its spatial resampling happens inside the sources (GDAL nearest warp) and in Smooth's zoom
step.  ``SpatialResample`` is the Block form of kernels K6/K7: it asks its store for a
``zoom`` times coarser grid over the same bbox and resamples the answer to the requested
grid with nearest-neighbour or bilinear interpolation (spec in DESIGN.md).
"""
import ctypes
import math

import numpy as np

from .. import utils
from . import resample, stencil
from ._checks import argument, need_number
from .base import BaseSingle, RasterBlock

__all__ = ["SpatialResample", "Dilate", "MovingMax", "HillShade", "Smooth", "Place"]


def expand_request_pixels(request, radius=1):
    """The request grown by ``radius`` cells on every side; None for requests without cells to grow
    (other modes, point requests).  Faithful to spatial.py:27-47, including its swap: the west / south
    edges move by the x amount, the east / north edges by the y amount."""
    if request["mode"] != "vals":
        return None
    west, south, east, north = request["bbox"]
    span_x, span_y = east - west, north - south
    if span_x == 0 or span_y == 0:
        return None
    by_x, by_y = span_x / request["width"] * radius, span_y / request["height"] * radius
    grown = request.copy()
    grown["bbox"] = (west - by_x, south - by_x, east + by_y, north + by_y)
    grown["width"], grown["height"] = request["width"] + 2 * radius, request["height"] + 2 * radius
    return grown


class _Stencil(BaseSingle):
    """A Block that needs ``radius`` cells around every requested cell: the store is asked for the grown
    request and ``process`` crops; requests that cannot grow pass through untouched (spatial.py:130-141)."""

    gm_direct = True  # plan.py: evaluated as process(store's response) without a graph per request
    # plan._whole_instead_of_tiles: a mosaic of this Block's tiles IS its answer for the whole box (every tile's cells see
    # their neighbours through the margin the tile asks its store for; the kernels compute a cell from its window alone)
    gm_tiling_invariant = True

    def _radius(self):
        return 1

    def _process_argument(self, request):
        raise NotImplementedError

    def get_sources_and_requests(self, **request):
        grown = expand_request_pixels(request, radius=self._radius())
        if grown is None:
            return [(self.store, request)]
        return [(self.store, grown), (self._process_argument(request), None)]


def _tile_by_tile(calls, dest, one):
    """gm_batch of a Block whose kernel takes one tile per launch: ``one(data, argument, out)`` tile after tile, the
    result of tile i written into ``dest(i, shape, dtype)`` where the kernel can (no paste into the mosaic afterwards)."""
    if not calls or any(len(args) != 2 or args[0] is None or args[1] is None or "values" not in args[0] for args in calls):
        return None
    return [one(data, argument, lambda shape, dtype, i=i: dest(i, shape, dtype) if dest is not None else None)
            for i, (data, argument) in enumerate(calls)]


class Dilate(_Stencil):
    """Spatially dilate the cells holding one of ``values`` by one cell (spatial.py:111-155).

    Dilation is performed in the order of the values parameter.
    """

    def __init__(self, store, values):
        super().__init__(store, np.asarray(values, dtype=store.dtype).tolist())  # (the reference's cast)

    values = argument(1)

    def _process_argument(self, request):
        return self.values

    @staticmethod
    def process(data, values=None):
        if data is None or values is None or "values" not in data:
            return data
        return {"values": stencil.dilate(data["values"], values), "no_data_value": data["no_data_value"]}

    @staticmethod
    def gm_batch(calls, dest):
        def one(data, values, out_for):
            tile = data["values"]
            out = out_for((tile.shape[0], tile.shape[1] - 2, tile.shape[2] - 2), tile.dtype) if tile.shape[0] == 1 else None
            return {"values": stencil.dilate(tile, values, out=out), "no_data_value": data["no_data_value"]}
        return _tile_by_tile(calls, dest, one)


class MovingMax(_Stencil):
    """Spatial maximum filter with a circular footprint of diameter ``size`` (spatial.py:158-213)."""

    def __init__(self, store, size):
        odd = int(2 * round((size - 1) / 2) + 1)  # the nearest odd integer
        if odd < 3:
            raise ValueError("The size should be odd and larger than 1")
        super().__init__(store, odd)

    size = argument(1)

    def _radius(self):
        return int(self.size // 2)

    def _process_argument(self, request):
        return self.size

    @staticmethod
    def process(data, size=None):
        if data is None or size is None or "values" not in data:
            return data
        ndv = data["no_data_value"]
        return {"values": stencil.moving_max(data["values"], ndv, size), "no_data_value": ndv}

    @staticmethod
    def gm_batch(calls, dest):
        def one(data, size, out_for):
            tile, ndv, radius = data["values"], data["no_data_value"], int(size // 2)
            out = (out_for((1, tile.shape[1] - 2 * radius, tile.shape[2] - 2 * radius), tile.dtype)
                   if tile.shape[0] == 1 and tile.dtype != np.dtype("bool") else None)
            return {"values": stencil.moving_max(tile, ndv, size, out=out), "no_data_value": ndv}
        return _tile_by_tile(calls, dest, one)


class HillShade(_Stencil):
    """Hillshade of an elevation raster (spatial.py:320-440): uint8, 'no data' value 256 on purpose."""

    def __init__(self, store, altitude=45, azimuth=315, fill=0):
        need_number(altitude, azimuth, fill)
        super().__init__(store, float(altitude), float(azimuth), fill)

    altitude, azimuth, fill = argument(1), argument(2), argument(3)
    dtype = property(lambda self: np.dtype("u1"))
    fillvalue = property(lambda self: 256)  # deliberately outside uint8: a shade is never "no data"

    def _process_argument(self, request):
        west, south, east, north = request["bbox"]
        cell = ((east - west) / request["width"], (north - south) / request["height"])
        return dict(resolution=cell, altitude=self.altitude, azimuth=self.azimuth, fill=self.fill)

    @staticmethod
    def process(data, process_kwargs=None):
        if process_kwargs is None:
            return data
        shade = stencil.hillshade(data["values"], data["no_data_value"], process_kwargs["fill"],
                                  process_kwargs["resolution"], process_kwargs["altitude"], process_kwargs["azimuth"])
        return {"values": shade, "no_data_value": 256}

    @staticmethod
    def gm_batch(calls, dest):
        """``process`` for the tiles of one mosaic at once (plan.py: a RasterTiler over this Block): ``calls`` = the
        argument tuples ``process`` would get, tile by tile; ``dest(i, shape, dtype)`` may give the window of the mosaic
        the result of tile i goes to.  -> the list of responses, or None when the tiles have to go one by one."""
        if not calls or any(len(args) != 2 or args[1] is None or args[0] is None or "values" not in args[0] for args in calls):
            return None
        data0, kw0 = calls[0]
        ndv = data0["no_data_value"]
        for data, kw in calls[1:]:
            same_ndv = data["no_data_value"] == ndv or (data["no_data_value"] != data["no_data_value"] and ndv != ndv)
            if kw != kw0 or not same_ndv or type(data["no_data_value"]) is not type(ndv):
                return None
        shades = stencil.hillshade_batch([data["values"] for data, _ in calls], ndv, kw0["fill"], kw0["resolution"],
                                         kw0["altitude"], kw0["azimuth"], dest)
        if shades is None:
            return None
        return [{"values": shade, "no_data_value": 256} for shade in shades]


class SpatialResample(BaseSingle):
    """Upsample ``store`` by an integer or fractional ``zoom`` (> 0).

    Args:
      store (RasterBlock): raster to resample
      zoom (number): requested cells per store cell along each axis (2 = 2x upsample)
      method (str): "nearest" or "bilinear" (float rasters only)
    """

    gm_direct = True

    def __init__(self, store, zoom, method="bilinear"):
        if not isinstance(zoom, (int, float)) or zoom <= 0:
            raise ValueError("'zoom' should be a positive number")
        if method not in ("nearest", "bilinear"):
            raise ValueError("'method' should be 'nearest' or 'bilinear'")
        if method == "bilinear" and np.dtype(store.dtype).kind != "f":
            raise TypeError("bilinear resampling needs a float raster")
        super().__init__(store, zoom, method)

    zoom = property(lambda self: self.args[1])
    method = property(lambda self: self.args[2])

    def get_sources_and_requests(self, **request):
        if request["mode"] != "vals":
            return [(self.store, request), (None, None)]
        width, height = request["width"], request["height"]
        x1, y1, x2, y2 = request["bbox"]
        if x1 == x2 or y1 == y2:  # point request: nothing to resample
            return [(self.store, request), (None, None)]
        src_w = max(int(round(width / self.zoom)), 1)
        src_h = max(int(round(height / self.zoom)), 1)
        src_request = dict(request, width=src_w, height=src_h)
        return [(self.store, src_request), ({"method": self.method, "height": height, "width": width}, None)]

    @staticmethod
    def process(data, process_kwargs):
        if data is None or process_kwargs is None or "values" not in data:
            return data
        h, w = process_kwargs["height"], process_kwargs["width"]
        values, ndv = data["values"], data["no_data_value"]
        if process_kwargs["method"] == "bilinear":
            out = resample.bilinear(values, h, w, ndv)
        else:
            _, hs, ws = values.shape
            # same extent, index space: target pixel centre -> floor of the source coordinate
            gt = (0.0, 1.0, 0, float(hs), 0, -1.0)
            out = resample.nearest_warp(values, gt, (0.0, 0.0, float(ws), float(hs)), h, w, ndv,
                                        finite_to_fill=False)
        return {"values": out, "no_data_value": ndv}


def _metres_per_unit(projection):
    """Length of one projection unit in metres for the projected CRSs this path knows (the
    reference asks GDAL: spatial.py:61-72).  Geographic CRSs would need a re-projection of the
    bounding box (PROJ), which this path does not carry."""
    from ..utils import parse_epsg
    code = parse_epsg(projection)
    metric = code is not None and (code in (28992, 3857, 3035, 31370, 2154, 25831, 25832, 25833, 27700) or
                                   32601 <= code <= 32660 or 32701 <= code <= 32760)
    if not metric:
        raise NotImplementedError("Smooth needs a projected CRS in metres that this path knows; got {}".format(projection))
    return 1.0


class Smooth(BaseSingle):
    """Gaussian smoothing over ``size`` metres, sigma = size / 3 (spatial.py:216-318).  'no data' is
    replaced by ``fill`` before smoothing.  Requests whose margin would exceed 6 pixels are answered
    at the coarser ('zoom') resolution and zoomed back with nearest neighbour, like the reference."""

    gm_direct = True
    # (NOT gm_tiling_invariant: the reference grows the request by `size` cells while the Gaussian reaches 4 / 3 of that,
    #  spatial.py:262-290 -- the border cells of a tile see the edge of the tile's window, a mosaic of tiles is not the whole)

    MARGIN_THRESHOLD = 6

    def __init__(self, store, size, fill=0):
        need_number(size, fill)
        super().__init__(store, size, fill)

    size, fill = argument(1), argument(2)

    def _margins(self, request):
        """(enlarged request, radius in pixels (y, x)) -- expand_request_meters, spatial.py:50-108."""
        radius = self.size / _metres_per_unit(request["projection"])
        x1, y1, x2, y2 = request["bbox"]
        span_y, span_x = y2 - y1, x2 - x1
        grown = request.copy()
        if span_y > 0 and span_x > 0:
            per_metre = request["height"] / span_y, request["width"] / span_x   # pixels per metre
            radius_px = [radius * r for r in per_metre]
            whole_px = [int(round(r)) for r in radius_px]                          # margin in whole pixels
            margin = [px / r for px, r in zip(whole_px, per_metre)]               # ... and back in metres
        else:  # point request: no resolution to speak of
            radius_px = whole_px = [self.MARGIN_THRESHOLD] * 2
            margin = [radius] * 2
        grown["bbox"] = (x1 - margin[1], y1 - margin[0], x2 + margin[1], y2 + margin[0])
        grown["height"] += 2 * whole_px[0]
        grown["width"] += 2 * whole_px[1]
        return grown, radius_px

    def get_sources_and_requests(self, **request):
        if request["mode"] != "vals":
            return [(self.store, request)]
        grown, size = self._margins(request)
        if any(s > self.MARGIN_THRESHOLD for s in size):
            mode = "zoom"  # smooth on the original pixel count over the enlarged box, zoom back afterwards
            zoom = [grown[k] / request[k] for k in ("height", "width")]
            size = [s / z for s, z in zip(size, zoom)]
            grown["height"], grown["width"] = request["height"], request["width"]
        else:
            mode = "exact"
        return [(self.store, grown), (dict(smooth_mode=mode, fill=self.fill, size=size), None)]

    @staticmethod
    def process(data, process_kwargs=None):
        if data is None or process_kwargs is None:
            return data
        size_px, fill = process_kwargs["size"], process_kwargs["fill"]
        sigma = size_px[0] / 3, size_px[1] / 3
        ndv = data["no_data_value"]
        if process_kwargs["smooth_mode"] == "exact":
            crop = [int(round(s)) for s in size_px]
            values = stencil.gaussian_smooth(data["values"], ndv, fill, sigma, crop, out=process_kwargs.get("gm_out"))
        else:
            values = stencil.gaussian_smooth(data["values"], ndv, fill, sigma)
            if values.shape[0] == 0:
                return {"values": values, "no_data_value": ndv}
            _, ny, nx = values.shape
            zoom = 1 - 2 * size_px[0] / ny, 1 - 2 * size_px[1] / nx
            values = resample.affine_nearest(values, zoom, (size_px[0], size_px[1]))
        return {"values": values, "no_data_value": ndv}


def _smooth_batch(calls, dest):
    """Smooth.gm_batch: tile by tile, the exact mode's result written into its window of the mosaic."""
    def one(data, process_kwargs, out_for):
        tile = data["values"]
        if process_kwargs["smooth_mode"] == "exact" and tile.shape[0] == 1:
            crop = [int(round(s)) for s in process_kwargs["size"]]
            out = out_for((1, tile.shape[1] - 2 * crop[0], tile.shape[2] - 2 * crop[1]), tile.dtype)
            process_kwargs = dict(process_kwargs, gm_out=out)
        return Smooth.process(data, process_kwargs)
    return _tile_by_tile(calls, dest, one)


Smooth.gm_batch = staticmethod(_smooth_batch)


def _same_crs_point(point, source_projection, target_projection):
    """A point of one projection in another: the identity between equal projections; anything else needs PROJ."""
    if utils.same_projection(source_projection, target_projection):
        return float(point[0]), float(point[1])
    raise NotImplementedError("re-projection ({} -> {}) is outside the MI355X raster path (no PROJ/GDAL)".format(
        source_projection, target_projection))


def _filled(shape, value, dtype):
    """np.full(shape, value, dtype) on the device."""
    from ..hip import runtime
    from ..hip.array import DeviceArray
    from ..hip.dtypes import code_of
    out = DeviceArray.empty(shape, dtype)
    if out.size:
        runtime.check(runtime.lib().gm_fill(ctypes.c_void_p(out.ptr), code_of(out.dtype),
                                            runtime.scalar_bits(value, out.dtype), out.size), "gm_fill")
    return out


def _shifted_copy(source, fill, out_shape, di, dj):
    """full(out_shape, fill) with ``source`` pasted so that source cell (j, i) lands on (j + dj, i + di); cells that
    fall outside are dropped (the two 'place' branches of spatial.py:700-718 before their data test)."""
    from ..hip import runtime
    from ..hip.array import DeviceArray
    from ..hip.dtypes import code_of
    from .resample import _lib, _params
    bands, src_h, src_w = source.shape
    _, dst_h, dst_w = out_shape
    if bands == 0 or dst_h == 0 or dst_w == 0 or di <= -src_w or di >= dst_w or dj <= -src_h or dj >= dst_h:
        return _filled(out_shape, fill, source.dtype)
    # ONE pass that writes every cell of the result once -- the source cell (row - dj, column - di) or the fill value: the
    # nearest-neighbour kernel in its affine mode (index = floor(k * 1 - shift + 0.5), exact for whole shifts).  A fill
    # of the result followed by a rectangular copy of the runtime wrote the landing zone twice and copied at 2.9 TB/s:
    # 81 + 366 us for 8 frames of 4096^2 float32, against 180 us
    out = DeviceArray.empty(out_shape, source.dtype)
    runtime.check(_lib().gm_resample_nearest(
        ctypes.c_void_p(source.ptr), code_of(source.dtype), bands, src_h, src_w, ctypes.c_void_p(out.ptr), dst_h, dst_w, 1,
        _params((1.0, float(-dj), 0, 0)), _params((1.0, float(-di), 0, 0)), 0.0,
        runtime.scalar_bits(fill, source.dtype), 0), "gm_resample_nearest")
    return out


def _any_data(values, no_data_value):
    from . import devops
    # (Mask is the Block whose cell test is get_index -- isclose for float tiles; IsData compares exactly)
    marked = devops.mask_process({"values": values, "no_data_value": no_data_value}, 1)["values"]  # 1 = data, 0 = not
    return not devops._all_false(marked)


def _has_data(values, no_data_value, landed=None):
    """get_index(values, no_data_value).any() (utils.py:61-64) without leaving the device.  ``landed``: (row0, row1, column0,
    column1) of the cells that can hold data at all (the landing zone of a placed copy)."""
    bands, h, w = values.shape
    if landed is not None and bands * h * w > (1 << 22):
        # a few rows from the middle of the landing zone of the first band answer "is there any data" for nearly every
        # raster (two small launches); the pass over all of it -- a mask and a reduction, 250 us for 8 frames of 4096^2 --
        # is for the copies whose sample holds none
        r0, r1, c0, c1 = landed
        middle = (r0 + r1) // 2
        rows = min(16, r1 - middle)
        if rows > 0 and c1 > c0 and _any_data(values.band_slice(0, 1).window(middle, c0, rows, c1 - c0), no_data_value):
            return True
    return _any_data(values, no_data_value)


class Place(BaseSingle):
    """Place an input raster at given coordinates (spatial.py:441-731): the ``anchor`` of the store is put on each of
    ``coordinates``; where the placed copies overlap, ``statistic`` merges them (reduce_rasters).

    Args: store, place_projection (projection of anchor and coordinates), anchor (x, y), coordinates ([(x, y) ...]),
    statistic (last, first, count, sum, mean, min, max, argmin, argmax, product, std, var, p<number>).

    This path carries no PROJ: the place projection, the store's projection and the requested one have to be the
    same (anything else raises NotImplementedError)."""

    def __init__(self, store, place_projection, anchor, coordinates, statistic="last"):
        from .reduction import check_statistic
        if not isinstance(store, RasterBlock):
            raise TypeError("'{}' object is not allowed".format(type(store)))
        if not isinstance(place_projection, str) or not place_projection.strip():
            raise ValueError("'{}' is not a valid projection string".format(place_projection))
        anchor = list(anchor)
        if len(anchor) != 2:
            raise ValueError("Expected 2 numbers in the 'anchor' parameter")
        for x in anchor:
            if not isinstance(x, (int, float)):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        if coordinates is None or len(coordinates) == 0:
            coordinates = []
        else:
            coordinates = np.asarray(coordinates, dtype=float)
            if coordinates.ndim != 2 or coordinates.shape[1] != 2:
                raise ValueError("Expected a list of lists of 2 numbers in the 'coordinates' parameter")
            coordinates = coordinates.tolist()
        check_statistic(statistic)
        super().__init__(store, place_projection, anchor, coordinates, statistic)

    place_projection = property(lambda self: self.args[1])
    anchor = property(lambda self: self.args[2])
    coordinates = property(lambda self: self.args[3])
    statistic = property(lambda self: self.args[4])

    @property
    def projection(self):
        """The store's projection when the place projection equals it, else None (spatial.py:509-519)."""
        store_projection = self.store.projection
        if store_projection is not None and utils.same_projection(self.place_projection, store_projection):
            return store_projection

    @property
    def geo_transform(self):
        if self.projection is not None:
            return self.store.geo_transform

    @property
    def extent(self):
        geometry = self.geometry
        if geometry is None:
            return
        return utils.Extent.from_geometry(geometry).transformed("EPSG:4326").bbox

    @property
    def geometry(self):
        """The store's box moved onto every coordinate, combined, in the place projection (spatial.py:536-549)."""
        store_geometry = self.store.geometry
        if store_geometry is None:
            return
        extent = utils.Extent.from_geometry(store_geometry).transformed(self.place_projection)
        left, bottom, right, top = extent.bbox
        p, q = self.anchor
        xs, ys = zip(*self.coordinates)
        box = (left + min(xs) - p, bottom + min(ys) - q, right + max(xs) - p, top + max(ys) - q)
        return utils.Extent(box, extent.srs).as_geometry()

    def get_sources_and_requests(self, **request):
        if request["mode"] != "vals":
            return ({"mode": request["mode"]}, None), (self.store, request)
        anchor = _same_crs_point(self.anchor, self.place_projection, request["projection"])
        coordinates = [_same_crs_point(c, self.place_projection, request["projection"]) for c in self.coordinates]
        store_geometry = self.store.geometry
        if store_geometry is None:  # no geometry: no data
            return (({"mode": "null"}, None),)
        xmin, ymin, xmax, ymax = utils.Extent.from_geometry(store_geometry).transformed(request["projection"]).bbox
        x1, y1, x2, y2 = request["bbox"]
        size_x = (x2 - x1) / request["width"]
        size_y = (y2 - y1) / request["height"]
        if size_x > 0 and size_y > 0:  # (point requests never ask for the whole source)
            full_height = math.ceil((ymax - ymin) / size_y)
            full_width = math.ceil((xmax - xmin) / size_x)
            if full_height * full_width <= request["width"] * request["height"]:
                # the whole source at the requested cell size is no larger than the request: fetch it once and paste
                whole = request.copy()
                whole["width"], whole["height"] = full_width, full_height
                whole["bbox"] = (xmin, ymin, xmin + full_width * size_x, ymin + full_height * size_y)
                process_kwargs = {"mode": "warp", "anchor": anchor, "coordinates": coordinates,
                                  "src_bbox": whole["bbox"], "dst_bbox": request["bbox"],
                                  "cellsize": (size_x, size_y), "statistic": self.statistic}
                return [(process_kwargs, None), (self.store, whole)]
        # else one request per coordinate, shifted backwards; cells are [xmin, xmax) x (ymin, ymax]
        pairs = []
        for x, y in coordinates:
            box = [x1 + anchor[0] - x, y1 + anchor[1] - y, x2 + anchor[0] - x, y2 + anchor[1] - y]
            if box[0] >= xmax or box[1] > ymax or box[2] < xmin or box[3] <= ymin:
                continue
            shifted = request.copy()
            shifted["bbox"] = box
            pairs.append((self.store, shifted))
        if not pairs:
            # nothing lands inside: a time request tells how many bands the empty answer has
            probe = request.copy()
            probe["mode"] = "time"
            process_kwargs = {"mode": "empty", "dtype": self.dtype, "fillvalue": self.fillvalue,
                              "width": request["width"], "height": request["height"], "statistic": self.statistic}
            return [(process_kwargs, None), (self.store, probe)]
        return [({"mode": "group", "statistic": self.statistic}, None)] + pairs

    @staticmethod
    def process(process_kwargs, *multi):
        from ..hip.array import as_device
        from .reduction import reduce_rasters
        mode = process_kwargs["mode"]
        if mode in {"meta", "time"}:
            return multi[0]
        if mode == "null":
            return
        if mode == "empty":
            data = multi[0]
            if data is None:
                return
            out_shape = (len(data["time"]), process_kwargs["height"], process_kwargs["width"])
            out_no_data_value, out_dtype = process_kwargs["fillvalue"], np.dtype(process_kwargs["dtype"])
            stack = []
        elif mode == "group":
            stack = [data for data in multi if data is not None]
            if len(stack) == 0:
                return
        elif mode == "warp":
            data = multi[0]
            if data is None:
                return
            out_no_data_value = data["no_data_value"]
            source = as_device(data["values"])
            out_dtype = source.dtype
            anchor, src_bbox = process_kwargs["anchor"], process_kwargs["src_bbox"]
            size_x, size_y = process_kwargs["cellsize"]
            anchor_px = ((anchor[0] - src_bbox[0]) / size_x, (anchor[1] - src_bbox[1]) / size_y)
            x1, y1, x2, y2 = process_kwargs["dst_bbox"]
            dst_h, dst_w = round((y2 - y1) / size_y), round((x2 - x1) / size_x)
            src_d, src_h, src_w = source.shape
            out_shape = (src_d, dst_h, dst_w)
            stack = []
            for x, y in process_kwargs["coordinates"]:
                di = round((x - x1) / size_x - anchor_px[0])
                dj = dst_h - src_h - round((y - y1) / size_y - anchor_px[1])  # (the array's y axis points down)
                if di <= -src_w or di >= dst_w or dj <= -src_h or dj >= dst_h:
                    continue  # shifted completely outside
                placed = _shifted_copy(source, out_no_data_value, out_shape, di, dj)
                landed = (max(0, dj), min(dst_h, src_h + dj), max(0, di), min(dst_w, src_w + di))
                # (a copy none of whose data cells lands inside is left out -- it would count in argmin / argmax)
                if _has_data(placed, out_no_data_value, landed):
                    stack.append({"values": placed, "no_data_value": out_no_data_value})
        if len(stack) == 0:
            return {"values": _filled(out_shape, out_no_data_value, out_dtype), "no_data_value": out_no_data_value}
        return reduce_rasters(stack, process_kwargs["statistic"])
