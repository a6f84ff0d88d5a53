"""The one consumer of raster tiles outside the raster package: the zonal statistics of
geometry.AggregateRaster (reference geometry/aggregate.py).  Only the numeric core lives here -- the
GeoDataFrame plumbing around it (sources, rasterization through GDAL, sinks) is outside the MI355X path."""
from .aggregate import STATISTICS, aggregate_labelled, check_statistic, label_statistics  # noqa: F401
