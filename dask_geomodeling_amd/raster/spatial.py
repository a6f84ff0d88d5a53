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

__all__ = ["SpatialResample", "Dilate", "MovingMax", "HillShade"]


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
