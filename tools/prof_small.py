import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dask_geomodeling_amd import raster
from dask_geomodeling_amd.core.graphs import get_sync
from dask_geomodeling_amd.hip import runtime
runtime.set_device(0)
jobs, seeds, _ = bench.build_batch(raster, 256, 256, 0)
graph, names = jobs[0]
for _ in range(3):
    get_sync(graph, names)
runtime.sync()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    get_sync(graph, names)
runtime.sync()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
