"""Spatial resampling Block.

The reference has no stand-alone spatial resample Block (``raster.Resample`` is temporal);
its spatial resampling happens inside the sources (GDAL nearest warp) and in Smooth's zoom
step.  ``SpatialResample`` is the Block form of kernels K6/K7: it asks its store for a
``zoom`` times coarser grid over the same bbox and resamples the answer to the requested
grid with nearest-neighbour or bilinear interpolation (spec in DESIGN.md).
"""
import numpy as np

from . import resample
from .base import BaseSingle

__all__ = ["SpatialResample"]


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
