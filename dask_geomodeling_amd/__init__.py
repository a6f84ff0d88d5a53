"""MI355X-native per-tile raster compute path behind dask-geomodeling's Block API.

    from dask_geomodeling_amd import raster
    view = raster.Clip((a + b) * 2.5 - b + 1.5, a)
    view.get_data(mode="vals", bbox=..., projection=..., width=..., height=...)

Block / get_data / get_compute_graph follow nens/dask-geomodeling; the ``process`` bodies
of the element-wise, Clip/Mask/Classify/Reclassify and resampling Blocks are hand-written
HIP kernels (gfx950) reached through the ctypes C-ABI in ``include/gmhip.h``.
"""
from . import config  # noqa: F401
from .core import Block, construct  # noqa: F401
from . import raster  # noqa: F401

__version__ = "0.1.0"
