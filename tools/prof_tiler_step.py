"""One-off: host cost of one step of bench.py --workload tiler (GPU box)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dask_geomodeling_amd import raster
from dask_geomodeling_amd.core.graphs import get_sync
from dask_geomodeling_amd.hip import runtime
runtime.set_device(0)
jobs = bench.build_tiler(raster, 16384, 4096, 0)
graph, names = jobs[0]
for _ in range(5):
    get_sync(graph, names)
runtime.sync()
t0 = time.perf_counter()
for _ in range(50):
    get_sync(graph, names)
t1 = time.perf_counter()
runtime.sync()
t2 = time.perf_counter()
print("host submit %.1f us/step, total %.1f us/step" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    get_sync(graph, names)
runtime.sync(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
