"""Zonal statistics of a raster per geometry (counterpart of geometry/aggregate.py:113-203).

The reference's ``aggregate_polygons`` (1) groups the geometries into buckets whose members do not overlap
(``bucketize``, aggregate.py:55-110), (2) paints every bucket into an int32 label raster with GDAL
(``utils.rasterize_geoseries``) and (3) for every frame of the raster runs scipy.ndimage's labelled
statistics over the active cells.  Steps (1) and (2) are geometry work on the host (shapely / GDAL -- not
part of this path); step (3) is the tile arithmetic and runs on the device here:

* ``label_statistics``  -- one bucket: the statistic of every label over its active cells
  (``gm_zonal_statistic``, csrc/gm_zonal.hip), bit for bit what scipy.ndimage.{sum, minimum, maximum, mean,
  median} resp. measurements.percentile return after the cast into the float32 ``agg`` array;
* ``aggregate_labelled`` -- the loop over buckets and frames of ``aggregate_polygons`` with the label rasters
  handed in: same result array (NaN where a geometry has no active cell), same list of geometries without
  cells.
"""
import ctypes
import os

import numpy as np

__all__ = ["STATISTICS", "check_statistic", "label_statistics", "aggregate_labelled"]

# statistic -> code of csrc/gm_zonal.hip (AggregateRaster.STATISTICS, aggregate.py:311-319)
STATISTICS = {"sum": 0, "count": 1, "min": 2, "max": 3, "mean": 4, "median": 5, "percentile": 6}
LABEL_NO_DATA = np.iinfo(np.int32).max  # what utils.rasterize_geoseries leaves on cells no geometry covers


def check_statistic(statistic):
    """'sum' ... 'median' or 'p<number>' -> (statistic, percentile), with AggregateRaster.__init__'s errors
    (aggregate.py:336-342)."""
    from ..raster.reduction import parse_percentile_statistic
    if not isinstance(statistic, str):
        raise TypeError("'{}' object is not allowed".format(type(statistic)))
    statistic, percentile = parse_percentile_statistic(statistic.lower())
    if percentile is None and (statistic not in STATISTICS or statistic == "percentile"):
        raise ValueError("Unknown statistic '{}'".format(statistic))
    return statistic, percentile


_declared = False


def _lib():
    global _declared
    from ..hip import runtime
    L = runtime.lib()
    if not _declared:
        c = ctypes
        L.gm_zonal_statistic.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int64, c.c_int, c.c_uint64, c.c_void_p, c.c_int32,
                                         c.c_void_p, c.c_int, c.c_double, c.c_void_p, c.c_void_p, c.c_void_p]
        L.gm_zonal_statistic.restype = c.c_int
        L.gm_zonal_statistic_host.argtypes = list(L.gm_zonal_statistic.argtypes)
        L.gm_zonal_statistic_host.restype = c.c_int
        _declared = True
    return L


# (GM_ZONAL_HOST=0: results into a device buffer and one copy back by the caller, as before round 6 -- A/B runs, and the tests run both)
_RESULTS_TO_HOST = os.environ.get("GM_ZONAL_HOST", "1") != "0"

_NDV_CACHE = {}  # (dtype, type of the no-data value, its bytes) -> (has_ndv, bits): ~15 us of NumPy scalar work per call otherwise


def _ndv_for(dtype, no_data_value):
    """(has_ndv, the no-data value as the raster's dtype holds it, as gm_scalar bits) for `frame != no_data_value`
    (aggregate.py:177): a no-data value the raster's dtype cannot hold equals no cell."""
    from ..hip import runtime
    if no_data_value is None:
        return False, 0
    try:
        key = (dtype.str, type(no_data_value), no_data_value.tobytes() if isinstance(no_data_value, np.generic) else no_data_value)
        hit = _NDV_CACHE.get(key)
    except TypeError:
        key = hit = None
    if hit is None:
        with np.errstate(all="ignore"):
            held = np.asarray(no_data_value).astype(dtype)[()]
            hit = (bool(held == no_data_value), runtime.scalar_bits(held, dtype))
        if key is not None and len(_NDV_CACHE) < 1024:
            _NDV_CACHE[key] = hit
    return hit


def label_statistics(values, no_data_value, labels, n_labels, statistic, percentile=None, threshold_values=None):
    """The statistic of ``values`` (frames, height, width) per label of ``labels`` (height, width; int32
    geometry indices, anything outside [0, n_labels) = not covered) over the label's active cells.

    Returns ``(agg, counts, present)`` as ndarrays: float32 (frames, n_labels) with NaN for labels without an
    active cell, the number of active cells, and whether the label occurs in ``labels`` at all."""
    from ..hip import runtime
    from ..hip.array import DeviceArray, as_device
    from ..hip.dtypes import code_of
    if statistic not in STATISTICS:
        raise ValueError("Unknown statistic '{}'".format(statistic))
    if statistic == "percentile":
        if percentile is None:
            raise ValueError("the percentile statistic needs a percentile")
        if not 0 <= percentile <= 100:
            raise ValueError("Percentiles must be in the range [0, 100]")
    tile = as_device(values)
    if tile.ndim != 3:
        raise ValueError("expected a (frames, height, width) raster, got shape %s" % (tile.shape,))
    if statistic == "percentile" and tile.dtype == np.dtype("bool"):
        # data[hi] - data[lo] on a boolean array (measurements.py:134)
        raise TypeError("numpy boolean subtract, the `-` operator, is not supported")
    frames, height, width = tile.shape
    label_tile = as_device(np.ascontiguousarray(labels, dtype=np.int32) if isinstance(labels, np.ndarray) else labels)
    if label_tile.dtype != np.dtype("i4") or tuple(label_tile.shape[-2:]) != (height, width) or label_tile.size != height * width:
        raise ValueError("labels must be an int32 raster of shape %s" % ((height, width),))
    n_labels = int(n_labels)
    thresholds = None
    if threshold_values is not None:
        thresholds = np.asarray(threshold_values)
        if thresholds.shape != (n_labels,):
            raise ValueError("expected one threshold per label")
        # `frame >= thresholds` compares in float64 unless both sides are float32 / narrow integers, where the
        # float32 comparison is exact too: the device compares the float32 thresholds in float64
        if thresholds.dtype != np.dtype("f4"):
            raise TypeError("thresholds must be float32 (aggregate.py:532 casts the column with astype('f4'))")
        thresholds = DeviceArray.from_numpy(thresholds)
    has_ndv, ndv_bits = _ndv_for(tile.dtype, no_data_value)
    cells = frames * n_labels
    if _RESULTS_TO_HOST:
        # the three results straight into host arrays: the library queues their copy behind the kernels of the route it takes,
        # ahead of that route's verdict -- one wait per call (gm_zonal_statistic_host)
        counts = np.empty((frames, n_labels), dtype="i8")
        out = np.empty((frames, n_labels), dtype="f4")
        present = np.empty((n_labels,), dtype="u1")
        runtime.check(_lib().gm_zonal_statistic_host(tile.ptr, code_of(tile.dtype), frames, height * width, int(has_ndv), ndv_bits,
                                                     label_tile.ptr, n_labels, thresholds.ptr if thresholds is not None else None,
                                                     STATISTICS[statistic], float(percentile or 0.0), out.ctypes.data, counts.ctypes.data,
                                                     present.ctypes.data),
                      "gm_zonal_statistic_host")
        return out, counts, present.astype(bool)
    # one device buffer for the three results (counts first: 8-byte aligned), one copy back
    packed = DeviceArray.empty((cells * 12 + n_labels,), "u1")
    counts_ptr, out_ptr, present_ptr = packed.ptr, packed.ptr + 8 * cells, packed.ptr + 12 * cells
    runtime.check(_lib().gm_zonal_statistic(tile.ptr, code_of(tile.dtype), frames, height * width, int(has_ndv), ndv_bits,
                                            label_tile.ptr, n_labels, thresholds.ptr if thresholds is not None else None,
                                            STATISTICS[statistic], float(percentile or 0.0), out_ptr, counts_ptr, present_ptr),
                  "gm_zonal_statistic")
    host = packed.to_numpy()
    counts = host[:8 * cells].view("i8").reshape(frames, n_labels)
    out = host[8 * cells:12 * cells].view("f4").reshape(frames, n_labels)
    return out, counts, host[12 * cells:].astype(bool)


def aggregate_labelled(label_rasters, n_geometries, values, no_data_value, threshold_values, statistic, percentile=None):
    """``aggregate_polygons`` (aggregate.py:113-203) from the label rasters on: ``label_rasters`` yields
    ``(select, labels)`` per bucket -- the geometry indices of the bucket and the raster
    ``utils.rasterize_geoseries(geometries.iloc[select], ..., values=select)`` painted for it.

    Returns the float32 array of shape (frames, n_geometries) and the list of geometries that cover no cell."""
    if statistic not in STATISTICS:
        raise KeyError(statistic)
    from ..hip.array import as_device
    tile = as_device(values)
    agg = np.full((tile.shape[0], n_geometries), np.nan, dtype="f4")
    no_cells = set()
    for select, labels in label_rasters:
        select = [int(i) for i in select]
        got, counts, present = label_statistics(tile, no_data_value, labels, n_geometries, statistic, percentile, threshold_values)
        chosen = np.zeros(n_geometries, dtype=bool)
        chosen[select] = True
        no_cells |= set(np.flatnonzero(chosen & ~present).tolist())
        # agg[frame_no][select_and_active] = ...: only geometries of this bucket with an active cell
        take = (counts > 0) & chosen[np.newaxis, :]
        agg[take] = got[take]
    return agg, sorted(no_cells)
