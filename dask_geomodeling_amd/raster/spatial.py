"""Spatial Blocks: the 3x3 / small-radius stencils of the reference (Dilate, MovingMax, HillShade,
raster/spatial.py:111-213, 320-440) and SpatialResample.

The stencil Blocks enlarge the request by their radius (``expand_request_pixels``,
spatial.py:27-47) -- the halo is recomputed per tile, never exchanged, so tiles stay independent --
and hand the enlarged tile to the kernels of csrc/gm_stencil.hip, which write the cropped result.


The reference has no stand-alone spatial resample Block (``raster.Resample`` is temporal);
its spatial resampling happens inside the sources (GDAL nearest warp) and in Smooth's zoom
step.  ``SpatialResample`` is the Block form of kernels K6/K7: it asks its store for a
``zoom`` times coarser grid over the same bbox and resamples the answer to the requested
grid with nearest-neighbour or bilinear interpolation (spec in DESIGN.md).
"""
import numpy as np

from . import resample, stencil
from .base import BaseSingle

__all__ = ["SpatialResample", "Dilate", "MovingMax", "HillShade", "Smooth"]


def expand_request_pixels(request, radius=1):
    """Expand a request by ``radius`` pixels; None for non-vals and point requests
    (spatial.py:27-47, including its mixed use of the x / y amounts)."""
    if request["mode"] != "vals":
        return None
    width, height = request["width"], request["height"]
    x1, y1, x2, y2 = request["bbox"]
    pwidth, pheight = x2 - x1, y2 - y1
    if pwidth == 0 or pheight == 0:
        return None
    amount_x = pwidth / width * radius
    amount_y = pheight / height * radius
    new_request = request.copy()
    new_request["bbox"] = (x1 - amount_x, y1 - amount_x, x2 + amount_y, y2 + amount_y)
    new_request["width"] += 2 * radius
    new_request["height"] += 2 * radius
    return new_request


class Dilate(BaseSingle):
    """Spatially dilate the cells holding one of ``values`` by one cell (spatial.py:111-155).

    Dilation is performed in the order of the values parameter.
    """

    def __init__(self, store, values):
        values = np.asarray(values, dtype=store.dtype)
        super().__init__(store, values.tolist())

    @property
    def values(self):
        return self.args[1]

    def get_sources_and_requests(self, **request):
        new_request = expand_request_pixels(request, radius=1)
        if new_request is None:  # not an expandable request: do nothing
            return [(self.store, request)]
        return [(self.store, new_request), (self.values, None)]

    @staticmethod
    def process(data, values=None):
        if data is None or values is None or "values" not in data:
            return data
        return {"values": stencil.dilate(data["values"], values), "no_data_value": data["no_data_value"]}


class MovingMax(BaseSingle):
    """Spatial maximum filter with a circular footprint of diameter ``size`` (spatial.py:158-213)."""

    def __init__(self, store, size):
        size = int(2 * round((size - 1) / 2) + 1)  # round to the nearest odd integer
        if size < 3:
            raise ValueError("The size should be odd and larger than 1")
        super(MovingMax, self).__init__(store, size)

    @property
    def size(self):
        return self.args[1]

    def get_sources_and_requests(self, **request):
        size = self.size
        new_request = expand_request_pixels(request, radius=int(size // 2))
        if new_request is None:  # not an expandable request: do nothing
            return [(self.store, request)]
        return [(self.store, new_request), (size, None)]

    @staticmethod
    def process(data, size=None):
        if data is None or size is None or "values" not in data:
            return data
        return {"values": stencil.moving_max(data["values"], data["no_data_value"], size),
                "no_data_value": data["no_data_value"]}


class HillShade(BaseSingle):
    """Hillshade of an elevation raster (spatial.py:320-440): uint8, 'no data' value 256 on purpose."""

    def __init__(self, store, altitude=45, azimuth=315, fill=0):
        for x in (altitude, azimuth, fill):
            if not isinstance(x, (int, float)):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(HillShade, self).__init__(store, float(altitude), float(azimuth), fill)

    @property
    def altitude(self):
        return self.args[1]

    @property
    def azimuth(self):
        return self.args[2]

    @property
    def fill(self):
        return self.args[3]

    @property
    def dtype(self):
        return np.dtype("u1")

    @property
    def fillvalue(self):
        return 256  # on purpose, it does not exist in bytes

    def get_sources_and_requests(self, **request):
        new_request = expand_request_pixels(request, radius=1)
        if new_request is None:  # not an expandable request: do nothing
            return [(self.store, request)]
        bbox = request["bbox"]
        resolution = ((bbox[2] - bbox[0]) / request["width"], (bbox[3] - bbox[1]) / request["height"])
        process_kwargs = dict(resolution=resolution, altitude=self.altitude, azimuth=self.azimuth, fill=self.fill)
        return [(self.store, new_request), (process_kwargs, None)]

    @staticmethod
    def process(data, process_kwargs=None):
        if process_kwargs is None:
            return data
        values = stencil.hillshade(data["values"], data["no_data_value"], process_kwargs["fill"],
                                   process_kwargs["resolution"], process_kwargs["altitude"],
                                   process_kwargs["azimuth"])
        return {"values": values, "no_data_value": 256}


class SpatialResample(BaseSingle):
    """Upsample ``store`` by an integer or fractional ``zoom`` (> 0).

    Args:
      store (RasterBlock): raster to resample
      zoom (number): requested cells per store cell along each axis (2 = 2x upsample)
      method (str): "nearest" or "bilinear" (float rasters only)
    """

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

    MARGIN_THRESHOLD = 6

    def __init__(self, store, size, fill=0):
        for x in (size, fill):
            if not isinstance(x, (int, float)):
                raise TypeError("'{}' object is not allowed".format(type(x)))
        super(Smooth, self).__init__(store, size, fill)

    size = property(lambda self: self.args[1])
    fill = property(lambda self: self.args[2])

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
            values = stencil.gaussian_smooth(data["values"], ndv, fill, sigma, crop)
        else:
            values = stencil.gaussian_smooth(data["values"], ndv, fill, sigma)
            _, ny, nx = values.shape
            zoom = 1 - 2 * size_px[0] / ny, 1 - 2 * size_px[1] / nx
            values = resample.affine_nearest(values, zoom, (size_px[0], size_px[1]))
        return {"values": values, "no_data_value": ndv}
