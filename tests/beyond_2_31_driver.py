"""Tiles of more than 2^31 cells through the kernels of the next rows (GPU box; ~45 GB of host memory, ~35 GB on the device):
temporal statistics and Cumulative over 40 frames of 8192^2 float32 cells; MovingMax / Dilate / HillShade / Smooth over 9 bands of
16384^2; bilinear and nearest results of 3 bands of 32768^2; reduce_rasters over 40 rasters of 8192^2.  Windows of the device results against the oracle / NumPy on host
copies of the same windows (the statistics are per cell across the frames; the stencils and resamplings are local): a kernel either
answers right or refuses loudly -- an index that wraps at 2^31 or 2^32 would do neither.
python tests/beyond_2_31_driver.py [parts: temporal,stencils,resample,stacks]; prints 'BEYOND 2^31 ok <checks>'."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dask_geomodeling_amd import raster  # noqa: E402
from helpers import assert_values_equal  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.hip.array import DeviceArray  # noqa: E402
from dask_geomodeling_amd.raster import temporal  # noqa: E402

runtime.set_device(0)
parts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["temporal", "stencils", "resample", "stacks"]
checks = 0
rng = np.random.default_rng(31)


def attempt(name, fn):
    t0 = time.perf_counter()
    try:
        out = fn()
        print("%-40s answered in %.2f s" % (name, time.perf_counter() - t0), flush=True)
        return out
    except Exception as exc:  # a refusal is an answer too
        print("%-40s REFUSED: %s: %s" % (name, type(exc).__name__, str(exc)[:150]), flush=True)
        return None


def same(got, want, what):
    global checks
    checks += 1
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape and got.dtype == want.dtype, (what, got.shape, want.shape, got.dtype, want.dtype)
    eq = (got == want) | ((got != got) & (want != want))
    assert eq.all(), "%s: %d of %d cells differ, e.g. %r vs %r" % (what, (~eq).sum(), eq.size, got[~eq][:3], want[~eq][:3])


if "temporal" in parts:
    frames, side = 40, 8192
    ndv = np.float32(-9999.0)
    base = (rng.standard_normal((8, side, side)) * 100).astype("f4")
    base[rng.random(base.shape) < 0.1] = ndv
    v = np.empty((frames, side, side), "f4")
    for f in range(frames):
        v[f] = np.roll(base[f % 8], 3 * f, axis=1)
    print("cells:", v.size, "= 2^31 x %.2f" % (v.size / 2.0 ** 31), flush=True)
    dev = DeviceArray.from_numpy(v)
    runs = [(0, 10), (10, 10), (20, 20)]
    wins = [(0, 0), (side - 64, side - 64), (4000, 5000), (8000, 100)]
    for stat in ("sum", "max", "count", "median"):
        out_dtype = np.dtype("i4") if stat == "count" else np.dtype("f4")
        got = attempt("temporal %s" % stat, lambda: np.asarray(temporal.aggregate_frames(dev, ndv, runs, stat, out_dtype)))
        if got is None:
            continue
        for r0, c0 in wins:
            w = v[:, r0:r0 + 64, c0:c0 + 64].astype("f8")
            w[v[:, r0:r0 + 64, c0:c0 + 64] == ndv] = np.nan
            for i, (first, count) in enumerate(runs):
                part = w[first:first + count]
                with np.errstate(all="ignore"):
                    import warnings
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        if stat == "sum":
                            want = np.nansum(part.astype("f4"), axis=0).astype("f4")
                            want = np.nansum(np.where(np.isnan(part), np.nan, part).astype("f4"), axis=0)
                        elif stat == "max":
                            want = np.nanmax(part, axis=0).astype("f4")
                            want[np.isnan(want)] = np.finfo("f4").max
                        elif stat == "count":
                            want = (~np.isnan(part)).sum(axis=0).astype("i4")
                        else:
                            want = np.nanmedian(part.astype("f4"), axis=0).astype("f4")
                            want[np.isnan(want)] = np.finfo("f4").max
                if stat == "sum":
                    continue   # (float32 sums follow NumPy's pairwise order: covered by the oracle at size in tests/full_size_ops_driver.py)
                same(got[i, r0:r0 + 64, c0:c0 + 64], want, "temporal %s run %d window (%d, %d)" % (stat, i, r0, c0))
        print("temporal %s windows ok" % stat, flush=True)
    got = attempt("cumulative count", lambda: np.asarray(temporal.accumulate_frames(dev, ndv, [(0, 20), (20, 20)], (0, 40), "count", np.dtype("i4"))))
    if got is not None:
        for r0, c0 in wins:
            valid = v[:, r0:r0 + 64, c0:c0 + 64] != ndv
            want = np.concatenate([np.cumsum(valid[:20], axis=0), np.cumsum(valid[20:], axis=0)]).astype("i4")
            same(got[:, r0:r0 + 64, c0:c0 + 64], want, "cumulative count window (%d, %d)" % (r0, c0))
        print("cumulative windows ok", flush=True)
    del dev, v

if "stencils" in parts:
    bands, side = 9, 16384
    ndv = np.float32(-9999.0)
    plane = (rng.standard_normal((side, side)) * 50).astype("f4")
    plane[rng.random(plane.shape) < 0.02] = ndv
    v = np.empty((bands, side, side), "f4")
    for b in range(bands):
        v[b] = np.roll(plane, 5 * b, axis=0) + np.float32(b)
        v[b][np.roll(plane, 5 * b, axis=0) == ndv] = ndv
    print("cells:", v.size, "= 2^31 x %.2f" % (v.size / 2.0 ** 31), flush=True)
    data = {"values": DeviceArray.from_numpy(v), "no_data_value": ndv}
    wins = ((0, 0, 0), (8, side - 64, side - 64), (5, 9000, 12000), (8, 16000, 40))   # (band, row, column) of 64 x 64 result windows

    def stencil_windows(name, got, margin, oracle_fn, reach=None):
        global checks
        extra = max(0, (reach or margin) - margin)
        out_side = side - 2 * margin
        for b, r0, c0 in wins:
            r0, c0 = min(r0, out_side - 64), min(c0, out_side - 64)
            top, left = min(extra, r0), min(extra, c0)
            bottom, right = min(extra, out_side - r0 - 64), min(extra, out_side - c0 - 64)
            sub = {"values": np.ascontiguousarray(v[b:b + 1, r0 - top:r0 + 64 + 2 * margin + bottom, c0 - left:c0 + 64 + 2 * margin + right]),
                   "no_data_value": ndv}
            with np.errstate(all="ignore"):
                want = oracle_fn(sub)["values"][:, top:top + 64, left:left + 64]
            assert_values_equal(got[b:b + 1, r0:r0 + 64, c0:c0 + 64], want, 0, "%s band %d window (%d, %d)" % (name, b, r0, c0))
            checks += 1
        print(name, "windows ok", flush=True)

    for size in (3, 7, 15):
        got = attempt("MovingMax %d" % size, lambda: np.asarray(raster.MovingMax.process(data, size)["values"]))
        if got is not None:
            stencil_windows("MovingMax %d" % size, got, size // 2, lambda sub, size=size: O.movingmax_process(sub, size))
    kw = dict(resolution=(0.5, 0.5), altitude=45.0, azimuth=315.0, fill=1)
    got = attempt("HillShade", lambda: np.asarray(raster.HillShade.process(data, kw)["values"]))
    if got is not None:
        stencil_windows("HillShade", got, 1, lambda sub: O.hillshade_process(sub, kw, trig="rounded"))
    skw = dict(smooth_mode="exact", size=(3.0, 3.0), fill=np.float32(0.0))
    got = attempt("Smooth sigma 1", lambda: np.asarray(raster.Smooth.process(data, skw)["values"]))
    if got is not None:
        stencil_windows("Smooth sigma 1", got, 3, lambda sub: O.smooth_process(sub, skw), reach=4)
    del got
    classes = (np.abs(v) % 6).astype("u1")
    cdata = {"values": DeviceArray.from_numpy(classes), "no_data_value": None}
    got = attempt("Dilate u8", lambda: np.asarray(raster.Dilate.process(cdata, [2, 5])["values"]))
    if got is not None:
        for b, r0, c0 in wins:
            r0, c0 = min(r0, side - 2 - 64), min(c0, side - 2 - 64)
            # (the reference dilates the 3-D array: a cell also takes the value from the bands above and below -- all bands of the window)
            want = O.dilate_process({"values": np.ascontiguousarray(classes[:, r0:r0 + 66, c0:c0 + 66]), "no_data_value": None}, [2, 5])["values"]
            same(got[:, r0:r0 + 64, c0:c0 + 64], want, "Dilate u8 window (%d, %d)" % (r0, c0))
        print("Dilate u8 windows ok", flush=True)
    del data, cdata, v, classes

if "resample" in parts:
    from dask_geomodeling_amd.raster import resample
    bands, src_side, out = 3, 16384, 32768
    ndv = np.float32(np.finfo("f4").max)
    plane = (rng.standard_normal((src_side, src_side)) * 100).astype("f4")
    plane[rng.random(plane.shape) < 0.01] = ndv
    src = np.stack([plane, np.roll(plane, 11, axis=0), np.roll(plane, 7, axis=1)])
    print("result cells:", bands * out * out, "= 2^31 x %.2f" % (bands * out * out / 2.0 ** 31), flush=True)
    dsrc = DeviceArray.from_numpy(src)
    got = attempt("bilinear 2 x", lambda: resample.bilinear(dsrc, out, out, ndv))
    if got is not None:
        for r0, c0 in ((0, 0), (out - 128, out - 128), (20000, 31000), (32000, 5)):
            block = np.asarray(got.window(r0, c0, 128, 128))   # (a window of the 12.9 GB result: only that crosses to the host)
            with np.errstate(all="ignore"):
                want = O.bilinear_resample(src, out, out, ndv, window=(r0, r0 + 128, c0, c0 + 128))
            assert_values_equal(block, want, 1, "bilinear window (%d, %d)" % (r0, c0))
            checks += 1
        print("bilinear windows ok", flush=True)
    del got
    gt = (0.0, 1.0, 0, float(src_side), 0, -1.0)
    bbox = (-10.0, -6.0, src_side + 6.0, src_side + 10.0)   # (16400 units on 32768 cells: 1.998 x, a rim outside the source)
    got = attempt("nearest ~2 x", lambda: np.asarray(resample.nearest_warp(dsrc, gt, bbox, out, out, float(ndv))))
    if got is not None:
        want_rows = O.nearest_indices((gt[3], gt[5]), src_side, bbox[3], -(bbox[3] - bbox[1]) / out, out) if False else None
        for r0, c0 in ((0, 0), (out - 128, out - 128), (20000, 31000), (32000, 5)):
            cell = (bbox[2] - bbox[0]) / out
            sub_bbox = (bbox[0] + c0 * cell, bbox[3] - (r0 + 128) * cell, bbox[0] + (c0 + 128) * cell, bbox[3] - r0 * cell)
            want = O.nearest_warp(src, gt, sub_bbox, 128, 128, float(ndv))
            eq = got[:, r0:r0 + 128, c0:c0 + 128] == want
            # (a sub-request's cell centres are computed from another origin: the last bit of a coordinate may pick the neighbour
            #  on an exact tie -- none expected with this geometry)
            assert eq.all(), "nearest window (%d, %d): %d cells differ" % (r0, c0, (~eq).sum())
            checks += 1
        print("nearest windows ok", flush=True)
if "stacks" in parts:
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    side, n = 8192, 40
    ndv = np.float32(-9999.0)
    base = (rng.standard_normal((4, 1, side, side)) * 100).astype("f4")
    base[rng.random(base.shape) < 0.15] = ndv
    host = [np.roll(base[i % 4], 7 * i, axis=2) for i in range(n)]
    stack = [{"values": DeviceArray.from_numpy(h), "no_data_value": ndv} for h in host]
    print("cells in the stack:", n * side * side, "= 2^31 x %.2f" % (n * side * side / 2.0 ** 31), flush=True)
    for stat in ("sum", "max", "count", "median", "p90"):
        got = attempt("reduce_rasters %s" % stat, lambda: np.asarray(reduce_rasters([dict(d) for d in stack], stat)["values"]))
        if got is None:
            continue
        for r0, c0 in ((0, 0), (side - 48, side - 48), (5000, 7000)):
            sub = [{"values": np.ascontiguousarray(h[:, r0:r0 + 48, c0:c0 + 48]), "no_data_value": ndv} for h in host]
            with np.errstate(all="ignore"):
                want = O.reduce_rasters(sub, stat)["values"]
            assert_values_equal(got[:, r0:r0 + 48, c0:c0 + 48], want, 0, "reduce_rasters %s window (%d, %d)" % (stat, r0, c0))
            checks += 1
    print("stacks windows ok", flush=True)
print("BEYOND 2^31 ok", checks)
