"""cProfile of get_data (plan path) of Group(A, B) and HillShade(A + B) * 0.5 over two resident 16384^2 sources, 3968^2 requests at
16 boxes in turn: where the host microseconds above the kernels go (round 6)."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
side, big = 4096, 16384
A = raster.MemorySource(generate_window((big, big), "f4", 7001, 0.0, 100.0, F, (97, 89), 0, 0, big, big), np.float32(F), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(generate_window((big, big), "f4", 7002, 0.0, 100.0, F, (97, 89), 0, 0, big, big), np.float32(F), "EPSG:28992", 1.0, (0.0, float(big)))
boxes = [(float(i * side) + 64, float(j * side) + 64, float((i + 1) * side) - 64, float((j + 1) * side) - 64) for i in range(4) for j in range(4)]
req = dict(mode="vals", width=side - 128, height=side - 128, projection="EPSG:28992")
for name, view in (("Group(A, B)", raster.Group(A, B)), ("HillShade(A + B) * 0.5", raster.HillShade(A + B) * 0.5)):
    with gmconfig.config.set({"geomodeling.materialize": False}):
        for k in range(64):
            view.get_data(bbox=boxes[k % 16], **req)
        runtime.sync()
        t0 = time.perf_counter()
        for k in range(1000):
            view.get_data(bbox=boxes[k % 16], **req)
        host = (time.perf_counter() - t0) / 1000 * 1e6
        runtime.sync()
        print("== %s: host %.1f us per request, drained %.1f us" % (name, host, (time.perf_counter() - t0) / 1000 * 1e6))
        pr = cProfile.Profile(); pr.enable()
        for k in range(500):
            view.get_data(bbox=boxes[k % 16], **req)
        pr.disable(); runtime.sync()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
