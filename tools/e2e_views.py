"""get_data of mixed views (GPU box): microseconds per 4096^2 tile request on the plan path (PLANS=0: as a fused graph
per request through the built-in scheduler -- what a configured dask scheduler gets), next to the time the
kernels of the view would take at their stand-alone rates (bytes moved / 6 TB/s) -- a view that takes much longer
than that has a slow seam somewhere (a compaction copy, a one-element-per-lane launch, host time)."""
import os, sys, time, numpy as np
from datetime import datetime, timedelta
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import program as P
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
side, big = 4096, 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
dc = generate_window((big, big), "u1", 7003, 0.0, 12.0, 255, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
C = raster.MemorySource(dc, np.uint8(255), "EPSG:28992", 1.0, (0.0, float(big)))
frames = 8
dt = generate_window((frames * 4096, 4096), "f4", 7004, 0.0, 100.0, F32MAX, (97, 89), 0, 0, frames * 4096, 4096).reshape((frames, 4096, 4096))
T = raster.MemorySource(dt, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, 4096.0), time_first=datetime(2020, 1, 1), time_delta=timedelta(hours=6))
px = side * side
views = [
    ("A + B", A + B, 12 * px),
    ("Clip(A * 2, B > 50)", raster.Clip(A * 2, B > 50), 12 * px),
    ("Classify(HillShade(A), 8 bins)", raster.Classify(raster.HillShade(A), [20, 50, 80, 110, 140, 170, 200, 230]), (5 + 2) * px),
    ("Dilate(Reclassify(C), [3])", raster.Dilate(raster.Reclassify(C, [[1, 3], [2, 3], [5, 9]]), [3]), (1 + 8 + 8 + 8) * px),
    ("Mask(Dilate(C, [3, 5]), 1)", raster.Mask(raster.Dilate(C, [3, 5]), 1), (2 + 2) * px),
    ("Smooth(A, 2) (exact)", raster.Smooth(A, 2.0), 8 * px),
    ("Clip(Smooth(A, 2), B > 50)", raster.Clip(raster.Smooth(A, 2.0), B > 50), (8 + 12) * px),
    ("MovingMax(A + B, 7)", raster.MovingMax(A + B, 7), (12 + 8) * px),
    ("Max(A, B, A * 2)", raster.Max(A, B, A * 2), 12 * px),
    ("Group(A, B)", raster.Group(A, B), 12 * px),
    ("FillNoData(A, B)", raster.FillNoData(A, B), 12 * px),
    ("Place(A, 2 coordinates)", raster.Place(A, "EPSG:28992", (8192.0, 8192.0), [(8192.0, 8192.0), (8292.0, 8092.0)]), None),
]
boxes = [(float(i * side + 64), float(j * side + 64), float((i + 1) * side + 64), float((j + 1) * side + 64)) for i in range(3) for j in range(3)]
req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")


def run(name, view, nbytes, requests, n=90):
    with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": os.environ.get("PLANS", "1") == "1"}):
        try:
            for k in range(9):
                view.get_data(**requests[k % len(requests)])
            runtime.sync()
            if P.spec_pending():  # (hot programs get kernels of their own in the background: time those)
                P.spec_wait()
                for k in range(3):
                    view.get_data(**requests[k % len(requests)])
                runtime.sync()
            t0 = time.perf_counter()
            for k in range(n):
                view.get_data(**requests[k % len(requests)])
            runtime.sync()
        except Exception as e:  # noqa: BLE001
            print("%-34s %s: %s" % (name, type(e).__name__, str(e)[:90]), flush=True)
            return
        us = (time.perf_counter() - t0) / n * 1e6
    floor = "" if nbytes is None else "  kernels at 6 TB/s: %6.1f us  (x %.2f)" % (nbytes / 6e6, us / (nbytes / 6e6))
    print("%-34s %8.1f us per request%s" % (name, us, floor), flush=True)


for name, view, nbytes in views:
    run(name, view, nbytes, [dict(req, bbox=b) for b in boxes])
tbox = dict(mode="vals", bbox=(0.0, 0.0, 4096.0, 4096.0), width=4096, height=4096, projection="EPSG:28992",
            start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
run("TemporalAggregate(T, D, mean)", raster.TemporalAggregate(T, "D", "mean"), (frames * 4 + 2 * 4) * px, [tbox], n=30)
run("TemporalAggregate(T * 2, D, mean)", raster.TemporalAggregate(T * 2, "D", "mean"), (frames * 8 + frames * 4 + 2 * 4) * px, [tbox], n=30)
run("Cumulative(T)", raster.Cumulative(T), (frames * 8) * px, [tbox], n=30)
run("TemporalSum(T)", raster.TemporalSum(T), (frames * 4 + 4) * px, [tbox], n=30)
run("Shift(T, 1h)", raster.Shift(T, 3600 * 1000), 0.001, [tbox], n=30)
run("Resample(T, 3h, nearest)", raster.Resample(T, "3h", "nearest"), 2 * 15 * 4 * px, [tbox], n=30)
run("T * 2 (8 frames)", T * 2, frames * 8 * px, [tbox], n=30)
run("Clip(T, T > 50) (8 frames)", raster.Clip(T, T > 50), frames * 8 * px, [tbox], n=30)
run("Snap(A', T) (8 frames of one)", raster.Snap(raster.MemorySource(dt.band_slice(0, 1), np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, 4096.0)), T),
    frames * 8 * px, [tbox], n=30)
# requests off the source's grid: the nearest resample at the source, then the program
zoom_in = [dict(req, bbox=(b[0], b[1], b[0] + side / 2, b[1] + side / 2)) for b in boxes]    # 2x up-sampling
zoom_out = [dict(req, bbox=(b[0] / 2, b[1] / 2, b[0] / 2 + 2 * side, b[1] / 2 + 2 * side)) for b in boxes]  # 2x down
shifted = [dict(req, bbox=(b[0] + 0.5, b[1] + 0.5, b[2] + 0.5, b[3] + 0.5)) for b in boxes]   # half a cell off
run("A + B, 2x zoomed in", A + B, (2 * 5 + 12) * px, zoom_in)
run("A + B, 2x zoomed out", A + B, (2 * 8 + 12) * px, zoom_out)
run("A + B, half a cell off", A + B, (2 * 8 + 12) * px, shifted)
run("HillShade(A), 2x zoomed in", raster.HillShade(A), (5 + 5) * px, zoom_in)
# synthetic (procedural) sources and a Smooth that has to zoom (a margin beyond its threshold)
S1 = raster.SyntheticSource((big, big), "f4", 9001, 0.0, 100.0, F32MAX, (97, 89), "EPSG:28992", 1.0, (0.0, float(big)), resident=False) \
    if hasattr(raster, "SyntheticSource") else None
if S1 is not None:
    try:
        run("Synthetic + B (generated per request)", S1 + B, (4 + 12) * px, [dict(req, bbox=b) for b in boxes])
    except Exception as e:  # noqa: BLE001
        print("SyntheticSource probe skipped:", type(e).__name__, str(e)[:80])
run("Smooth(A, 40) (zoom mode)", raster.Smooth(A, 40.0), 8 * px, [dict(req, bbox=b) for b in boxes], n=30)
run("Reclassify(C) + Mask (u8 table path)", raster.Mask(raster.Reclassify(C, [[1, 3], [2, 3], [5, 9]]), 1), 2 * px, [dict(req, bbox=b) for b in boxes])
run("Classify(A, 200 edges)", raster.Classify(A, list(np.linspace(0, 100, 200))), 5 * px, [dict(req, bbox=b) for b in boxes])
run("A ** 2.5 (Power)", A ** 2.5, 8 * px, [dict(req, bbox=b) for b in boxes])
run("Exp(A / 50)", raster.Exp(A / 50), 8 * px, [dict(req, bbox=b) for b in boxes])
small = [dict(mode="vals", bbox=(float(i * 256), float(j * 256), float(i * 256 + 256), float(j * 256 + 256)), width=256, height=256,
              projection="EPSG:28992") for i in range(1, 4) for j in range(1, 4)]
run("A + B, 256^2 tiles", A + B, 12 * 256 * 256, small, n=900)
run("Clip(A * 2, B > 50), 256^2", raster.Clip(A * 2, B > 50), 12 * 256 * 256, small, n=900)
run("HillShade(A), 256^2", raster.HillShade(A), 5 * 256 * 256, small, n=900)
run("Smooth(A, 2), 256^2", raster.Smooth(A, 2.0), 8 * 256 * 256, small, n=900)
# temporal views under a tiler and in combination (round 4, second half)
run("tiler(TemporalAggregate(T, D, mean), 1024)", raster.RasterTiler(raster.TemporalAggregate(T, "D", "mean"), 1024), (frames * 4 + 2 * 4) * px, [tbox], n=20)
run("tiler(T * 2, 1024) (8 frames)", raster.RasterTiler(T * 2, 1024), frames * 8 * px, [tbox], n=20)
run("TemporalAggregate(Clip(T, T > 50), D, max)", raster.TemporalAggregate(raster.Clip(T, T > 50), "D", "max"), (frames * 8 + frames * 4 + 2 * 4) * px, [tbox], n=20)
run("Cumulative(T * 2)", raster.Cumulative(T * 2), (frames * 8 + frames * 8) * px, [tbox], n=20)
run("TemporalSum(T) / 8", raster.TemporalSum(T) / 8, (frames * 4 + 4 + 8) * px, [tbox], n=20)
run("Group(T, T2) (8 + 8 frames, one grid)", raster.Group(T, T * 0.5), (frames * 8 + frames * 12) * px, [tbox], n=20)
point = dict(tbox, bbox=(2000.5, 2000.5, 2000.5, 2000.5), width=1, height=1)
run("TemporalAggregate(T, D, mean) at a point", raster.TemporalAggregate(T, "D", "mean"), 1.0, [point], n=200)
run("T * 2 at a point (8 frames)", T * 2, 1.0, [point], n=200)
run("TemporalAggregate(Group(T, T2), D, mean)", raster.TemporalAggregate(raster.Group(T, T * 0.5), "D", "mean"), (frames * 8 + frames * 12 + frames * 4 + 8) * px, [tbox], n=20)
run("Resample(T * 2, 3h, nearest)", raster.Resample(T * 2, "3h", "nearest"), (frames * 8 + 2 * 15 * 4) * px, [tbox], n=20)
run("Snap(T * 2, T)", raster.Snap(T * 2, T), (frames * 8 + frames * 8) * px, [tbox], n=20)
run("Place(T, 1 coordinate) (8 frames)", raster.Place(T, "EPSG:28992", (2048.0, 2048.0), [(2100.0, 2000.0)]), frames * 8 * px, [tbox], n=20)
run("Clip(Cumulative(T), T > 50)", raster.Clip(raster.Cumulative(T), T > 50), (frames * 8 + frames * 12) * px, [tbox], n=20)
run("FillNoData(T, T2) (8 frames)", raster.FillNoData(T, T * 0.5), (frames * 8 + frames * 12) * px, [tbox], n=20)
