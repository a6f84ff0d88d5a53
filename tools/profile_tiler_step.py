"""cProfile of the prebuilt tiler graph of bench.py (--workload tiler): where the host microseconds of a step go."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dask_geomodeling_amd import raster
from dask_geomodeling_amd.core.graphs import get_sync
from dask_geomodeling_amd.hip import runtime
runtime.set_device(0)
(graph, names), = bench.build_tiler(raster, 16384, 4096, 0)
for _ in range(20):
    get_sync(graph, names)
runtime.sync()
t0 = time.perf_counter()
for _ in range(50):
    get_sync(graph, names)
host = time.perf_counter() - t0
runtime.sync()
print("host per step %.1f us, with drain %.1f us" % (host / 50 * 1e6, (time.perf_counter() - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    get_sync(graph, names)
pr.disable(); runtime.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
