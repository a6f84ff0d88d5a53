import sys, time, cProfile, pstats, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
side, big = 4096, 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
boxes = [(float(i * side), float(j * side), float((i + 1) * side), float((j + 1) * side)) for i in range(3) for j in range(3)]
req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")
import os
with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": os.environ.get("PLANS", "1") == "1",
                          "geomodeling.templates": os.environ.get("TEMPLATES", "1") == "1"}):  # (PLANS=0 TEMPLATES=0: from scratch)
    for k in range(20):
        view.get_data(bbox=boxes[k % 9], **req)
    runtime.sync()
    pr = cProfile.Profile()
    pr.enable()
    for k in range(500):
        view.get_data(bbox=boxes[k % 9], **req)
    pr.disable()
    runtime.sync()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "45")))
