"""get_data of stencil views with and without plans (GPU box): microseconds per 4096^2 tile request."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
side, big = 4096, 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
views = {"HillShade(A)": raster.HillShade(A), "HillShade(A + B) * 0.5": raster.HillShade(A + B) * 0.5,
         "Smooth(A * 2 - B, 6)": raster.Smooth(A * 2 - B, 6.0), "MovingMax(A, 5) - B": raster.MovingMax(A, 5) - B}
boxes = [(float(i * side + 64), float(j * side + 64), float((i + 1) * side + 64), float((j + 1) * side + 64)) for i in range(3) for j in range(3)]
req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")
for name, view in views.items():
    for plans in (True, False):
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": plans}):
            for k in range(9):
                view.get_data(bbox=boxes[k], **req)
            runtime.sync()
            t0 = time.perf_counter()
            n = 180
            for k in range(n):
                view.get_data(bbox=boxes[k % 9], **req)
            runtime.sync()
            us = (time.perf_counter() - t0) / n * 1e6
        print("%-26s %-18s %8.1f us per 4096^2 request  %9.0f Mpix/s" % (name, "plans" if plans else "graph per request", us, side * side / us), flush=True)
