"""Requests whose width / height are not multiples of 4 (GPU box): microseconds per request of a few views at 4096, 4095, 4094 and
4093 cells a side, windows of two resident 16384^2 sources -- a kernel that falls back to one element per lane for rows that
do not start on 16 bytes shows up as a step between the columns."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import program as P
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
big = 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
dc = generate_window((big, big), "u1", 7003, 0.0, 12.0, 255, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
C = raster.MemorySource(dc, np.uint8(255), "EPSG:28992", 1.0, (0.0, float(big)))
dd = generate_window((big, big), "i4", 7005, 0.0, 256.0, 2147483647, (97, 89), 0, 0, big, big)
D = raster.MemorySource(dd, np.int32(2147483647), "EPSG:28992", 1.0, (0.0, float(big)))
views = [
    ("A + B", A + B),
    ("Clip(A * 2, B > 50)", raster.Clip(A * 2, B > 50)),
    ("Reclassify(C)", raster.Reclassify(C, [[1, 3], [2, 3], [5, 9]])),
    ("Classify(A, 8 bins)", raster.Classify(A, [20, 50, 80, 110, 140, 170, 200, 230])),
    ("HillShade(A)", raster.HillShade(A)),
    ("MovingMax(A, 7)", raster.MovingMax(A, 7)),
    ("Dilate(C, [3, 5])", raster.Dilate(C, [3, 5])),
    ("Smooth(A, 2)", raster.Smooth(A, 2.0)),
    ("Max(A, B, A * 2)", raster.Max(A, B, A * 2)),
    ("Mask(Reclassify(D), 1)", raster.Mask(raster.Reclassify(D, [[k, (7 * k) % 1000] for k in range(0, 256, 2)]), 1)),
    ("Reclassify(D) -> i64", raster.Reclassify(D, [[k, (7 * k) % 1000] for k in range(0, 256, 2)])),
    ("Classify(A, 200 bins)", raster.Classify(A, list(np.linspace(0, 100, 200)))),
    ("chain5", raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)),
    ("Mask(C, 1)", raster.Mask(C, 1)),
]
from datetime import datetime, timedelta
frames = 8
dtm = generate_window((frames * 8192, 8192), "f4", 7004, 0.0, 100.0, F32MAX, (97, 89), 0, 0, frames * 8192, 8192).reshape((frames, 8192, 8192))
T = raster.MemorySource(dtm, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, 8192.0), time_first=datetime(2020, 1, 1), time_delta=timedelta(hours=6))
TREQ = dict(start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
tviews = [
    ("T * 2, 8 frames", T * 2),
    ("TemporalAggregate(T, D, mean)", raster.TemporalAggregate(T, "D", "mean")),
    ("Cumulative(T, sum)", raster.Cumulative(T, "sum")),
    ("Group(A, B)", raster.Group(A, B)),
    ("FillNoData(A, B)", raster.FillNoData(A, B)),
]
CASES = [(4096, 64, 64), (4096, 65, 33), (4095, 64, 64), (4094, 65, 33), (4093, 64, 64)]
print("%-26s" % "view" + "".join("%14s" % ("%d@%d,%d" % c) for c in CASES) + "   (us per request: side @ column, row offset of the window)")
for name, view in views + tviews:
    row = []
    temporal = name in [t[0] for t in tviews[:3]]
    for side, ox, oy in CASES:
        if temporal:
            side, ox, oy = side // 2, ox, oy
        boxes = [(float(i * 4200 + ox), float(j * 4200 + oy), float(i * 4200 + ox + side), float(j * 4200 + oy + side)) for i in range(3) for j in range(3)]
        if temporal:
            boxes = [(float(i * 2100 + ox), float(j * 2100 + oy), float(i * 2100 + ox + side), float(j * 2100 + oy + side)) for i in range(3) for j in range(3)]
        reqs = [dict(mode="vals", bbox=b, width=side, height=side, projection="EPSG:28992", **(TREQ if temporal else {})) for b in boxes]
        with gmconfig.config.set({"geomodeling.materialize": False}):
            try:
                for k in range(9):
                    view.get_data(**reqs[k % 9])
                runtime.sync()
                if P.spec_pending():
                    P.spec_wait()
                    for k in range(3):
                        view.get_data(**reqs[k % 9])
                    runtime.sync()
                t0 = time.perf_counter()
                for k in range(45):
                    view.get_data(**reqs[k % 9])
                runtime.sync()
                row.append("%14.1f" % ((time.perf_counter() - t0) / 45 * 1e6))
            except Exception as e:  # noqa: BLE001
                row.append("%14s" % type(e).__name__)
    print("%-26s" % name + "".join(row), flush=True)
