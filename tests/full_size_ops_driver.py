"""The N3 / N4 kernels at the sizes tools/kbench.py times them on (8192 x 8192 tiles, 24 frames of 2048 x 2048, stacks of
4096 x 4096 rasters), against the oracle.  The stencils and the resamplings are local -- a window of the result depends on
a window of the input and its margin -- so random windows of the full-size device result (the far corner and the seams of the
workgroup tiles among them) are compared bit for bit with the oracle run on host copies of just those windows; the temporal
and stack statistics are compared whole.  What only shows at size (indices past 2^31 bytes, the last tiles of a launch, tables
of many tiles) shows here; the unit tests reach every route on small tiles.
Prints 'FULL SIZE OPS ok <checks>'.  Arguments: the parts to run (stencils, resample, temporal, stacks; default: all), then
"quick" for frames / rasters of a quarter of the cells (the oracle's medians over 100 M cells take minutes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dask_geomodeling_amd import raster  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from helpers import assert_values_equal  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
parts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["stencils", "resample", "temporal", "stacks"]
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
rng = np.random.default_rng(808)
checks = 0


def windows(out_h, out_w, size, count):
    """Top-left corners of result windows: the four corners, seams of 256 / 512-column and 16 / 32-row tiles far from the origin,
    random places."""
    yield 0, 0
    yield out_h - size, out_w - size
    yield 0, out_w - size
    yield out_h - size, 0
    for _ in range(count):
        yield int(rng.integers(0, out_h - size)), int(rng.integers(0, out_w - size))
    yield min((out_h // 32 - 3) * 32 - size // 2, out_h - size), min((out_w // 512 - 2) * 512 - size // 2, out_w - size)
    yield min((out_h // 16 - 9) * 16 - 7, out_h - size), min((out_w // 256 - 5) * 256 - 9, out_w - size)


def stencil_windows(name, got, data, margin, oracle_fn, tolerance=0, size=160, reach=None):
    """``margin``: what the result is cropped by; ``reach``: how far a result cell looks (Smooth crops by 3 sigma and looks 4 sigma
    far: the host window is taken that much larger where the raster has the cells)."""
    global checks
    values = data["values"]
    got_values = np.asarray(got["values"])
    _, out_h, out_w = got_values.shape
    _, in_h, in_w = values.shape
    assert (out_h, out_w) == (in_h - 2 * margin, in_w - 2 * margin), (name, got_values.shape)
    extra = max(0, (reach or margin) - margin)
    for r0, c0 in windows(out_h, out_w, size, 6):
        top, left = min(extra, r0), min(extra, c0)
        bottom, right = min(extra, out_h - (r0 + size)), min(extra, out_w - (c0 + size))
        sub = dict(data, values=np.ascontiguousarray(values[:, r0 - top:r0 + size + 2 * margin + bottom, c0 - left:c0 + size + 2 * margin + right]))
        with np.errstate(all="ignore"):
            want = oracle_fn(sub)["values"][:, top:top + size, left:left + size]
        assert_values_equal(got_values[:, r0:r0 + size, c0:c0 + size], want, tolerance, "%s, window at (%d, %d)" % (name, r0, c0))
        checks += 1
    print(name, "ok", flush=True)


if "stencils" in parts:
    side = 8192
    ndv = np.float32(-9999.0)
    v = (rng.standard_normal((1, side, side)) * 50).astype("f4")
    v = np.cumsum(v, axis=2, dtype="f4")  # (terrain-like: slopes of every aspect)
    v[rng.random(v.shape) < 0.02] = ndv
    v[0, 5000:5040, 7000:7080] = ndv      # a hole wider than every footprint
    v[0, 8100, 8101] = -0.0
    data = {"values": v, "no_data_value": ndv}
    for size in (3, 7, 15, 17):
        stencil_windows("MovingMax %d" % size, raster.MovingMax.process(data, size), data, size // 2, lambda sub, size=size: O.movingmax_process(sub, size))
    for alt, az, res in ((45.0, 315.0, (0.5, 0.5)), (30.0, 10.0, (1.0, 2.0))):
        kw = dict(resolution=res, altitude=alt, azimuth=az, fill=1)
        stencil_windows("HillShade %s" % (res,), raster.HillShade.process(data, kw), data, 1, lambda sub, kw=kw: O.hillshade_process(sub, kw, trig="rounded"))
    for sigma_px in (1.0, 2.0):
        kw = dict(smooth_mode="exact", size=(3 * sigma_px, 3 * sigma_px), fill=np.float32(0.0))
        stencil_windows("Smooth sigma %g" % sigma_px, raster.Smooth.process(data, kw), data, int(round(3 * sigma_px)), lambda sub, kw=kw: O.smooth_process(sub, kw),
                        reach=int(4.0 * sigma_px + 0.5))
    classes = rng.integers(0, 6, size=(1, side, side)).astype("u1")
    classes[0, 4000:4100, :] = 0
    for dt in ("u1", "f4"):
        cdata = {"values": classes.astype(dt), "no_data_value": None if dt == "u1" else np.float32(255)}
        stencil_windows("Dilate %s" % dt, raster.Dilate.process(cdata, [2, 5]), cdata, 1, lambda sub: O.dilate_process(sub, [2, 5]))

if "resample" in parts:
    from dask_geomodeling_amd.raster import resample
    ndv = np.float32(np.finfo("f4").max)
    out = 8192
    for zoom_name, src_side in (("1.5 x", 5461), ("3 x", 2731), ("4 x", 2048), ("1.25 x", 6553), ("2 x", 4096), ("0.8 x", 10240)):
        src = (rng.standard_normal((1, src_side, src_side)) * 100).astype("f4")
        src[rng.random(src.shape) < 0.01] = ndv
        src[0, src_side - 1, src_side - 2] = np.nan
        got = np.asarray(resample.bilinear(src, out, out, ndv))
        assert got.shape == (1, out, out)
        size = 200
        for r0, c0 in windows(out, out, size, 6):
            with np.errstate(all="ignore"):
                want = O.bilinear_resample(src, out, out, ndv, window=(r0, r0 + size, c0, c0 + size))
            assert_values_equal(got[:, r0:r0 + size, c0:c0 + size], want, 1, "bilinear %s, window at (%d, %d)" % (zoom_name, r0, c0))
            checks += 1
        print("bilinear", zoom_name, "ok", flush=True)
    # nearest: a 8192^2 window of a source at another resolution and origin (whole: the oracle gathers by index)
    for dt, px, cell in (("f4", 1.0, 0.5), ("u1", 0.5, 0.8), ("i4", 2.0, 0.7)):
        side = 4500
        src = (rng.random((1, side, side)) * 200).astype(dt)
        gt = (1000.0, px, 0, 9000.0, 0, -px)
        x1, y2 = 1000.0 - 37 * cell, 9000.0 + 11 * cell   # (a strip outside the source on two sides)
        bbox = (x1, y2 - out * cell, x1 + out * cell, y2)
        fill = 255 if np.dtype(dt).kind != "f" else float(np.finfo(dt).max)
        got = np.asarray(resample.nearest_warp(src, gt, bbox, out, out, fill))
        assert_values_equal(got, O.nearest_warp(src, gt, bbox, out, out, fill), 0, "nearest %s" % dt)
        checks += 1
        print("nearest", dt, "ok", flush=True)

if "temporal" in parts:
    from dask_geomodeling_amd.raster import temporal
    frames, side = 24, (1024 if quick else 2048)
    ndv = np.float32(-9999.0)
    v = (rng.standard_normal((frames, side, side)) * 100).astype("f4")
    v[rng.random(v.shape) < 0.1] = ndv
    v[:, 100:110, 200:260] = ndv   # cells without data in any frame
    runs = [(0, 6), (6, 6), (12, 5), (17, 7)]
    for stat in ("sum", "mean", "max", "min", "count", "std", "median") + (() if quick else ("p90",)):   # (np.nanpercentile: minutes)
        out_dtype = np.dtype("i4") if stat == "count" else np.dtype("f4")
        with np.errstate(all="ignore"):
            want = O.temporal_aggregate_values(v, ndv, runs, stat, out_dtype)
        got = np.asarray(temporal.aggregate_frames(v, ndv, runs, stat, out_dtype))
        assert_values_equal(got, want, 0, "temporal %s" % stat)
        checks += 1
    for stat in ("sum", "count"):
        out_dtype = np.dtype("i4") if stat == "count" else np.dtype("f4")
        cruns, emit = [(0, 12), (12, 12)], (3, 24)
        want = O.temporal_cumulative_values(v, ndv, cruns, emit, stat, out_dtype)
        got = np.asarray(temporal.accumulate_frames(v, ndv, cruns, emit, stat, out_dtype))
        assert_values_equal(got, want, 0, "cumulative %s" % stat)
        checks += 1
    print("temporal ok", flush=True)

if "stacks" in parts:
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    side, n = (2048 if quick else 4096), 8
    ndv = np.float32(-9999.0)
    stack = []
    for _ in range(n):
        x = (rng.standard_normal((1, side, side)) * 100).astype("f4")
        x[rng.random(x.shape) < 0.15] = ndv
        x[0, 300:310, 500:520] = ndv
        stack.append({"values": x, "no_data_value": ndv})
    for stat in ("sum", "mean", "max", "min", "count", "std", "median", "product") + (() if quick else ("p25",)):
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], stat)
        got = reduce_rasters([dict(d) for d in stack], stat)
        assert_values_equal(np.asarray(got["values"]), want["values"], 0, "reduce_rasters %s" % stat)
        checks += 1
    print("stacks ok", flush=True)

print("FULL SIZE OPS ok", checks)
