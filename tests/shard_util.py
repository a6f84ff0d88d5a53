"""Helpers of the sharded-scheduler tests (tests/test_shard_cpu.py): what the worker processes import.

``install`` is the workers' preload hook (geomodeling.worker-preload = "shard_util:install"): it plugs the oracle's NumPy
interpreter of gm_program in place of the HIP launch -- the cpu_vm test double of conftest.py, here in every worker.
``HostGrid`` is a source Block that carries its pixels in the graph and answers pixel-aligned windows on the host."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from dask_geomodeling_amd.raster.base import RasterBlock  # noqa: E402


def install():
    from dask_driver import install_cpu_vm
    install_cpu_vm()


class HostGrid(RasterBlock):
    """(1, H, W) host array on the grid x in [0, W), y in [0, H), one cell per unit; requests are aligned windows."""

    def __init__(self, data, no_data_value):
        super().__init__(np.asarray(data), no_data_value)

    from datetime import datetime as _dt
    dtype = property(lambda self: self.args[0].dtype)
    fillvalue = property(lambda self: self.args[1])
    period = (_dt(1970, 1, 1), _dt(1970, 1, 1))
    timedelta = None
    temporal = False
    extent = geometry = geo_transform = None
    projection = "EPSG:28992"

    def get_sources_and_requests(self, **request):
        return [({"data": self.args[0], "ndv": self.args[1], "mode": request["mode"], "bbox": request.get("bbox"),
                  "width": request.get("width"), "height": request.get("height")}, None)]

    @staticmethod
    def process(kw):
        from datetime import datetime
        if kw["mode"] == "time":
            return {"time": [datetime(1970, 1, 1)]}
        if kw["mode"] == "meta":
            return {"meta": [None]}
        data = kw["data"]
        H = data.shape[1]
        x1, y1, x2, y2 = kw["bbox"]
        assert x2 - x1 == kw["width"] and y2 - y1 == kw["height"], "HostGrid serves aligned windows only"
        r0, c0 = int(H - y2), int(x1)
        return {"values": data[:, r0:r0 + kw["height"], c0:c0 + kw["width"]], "no_data_value": kw["ndv"]}


def die(code):
    os._exit(code)


def boom(text):
    raise ValueError(text)


def tile_of(value, shape=(1, 3, 5)):
    return {"values": np.full(shape, value, "f4"), "no_data_value": np.float32(-1)}


def whoami():
    return {"pid": os.getpid()}


def setting(key):
    from dask_geomodeling_amd import config
    return {"value": config.get(key), "pid": os.getpid()}
