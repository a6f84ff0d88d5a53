"""Host time of one bench step (chain5): the same merged graph on 64^2 tiles, where the kernel is ~5 us and the step
time is what Python + the launch path cost.  If that approaches the 500 us kernel of the real step, a slower host
CPU leaves gaps between launches (what separates avg_launch_us from kernel_only_us in the bench line)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dask_geomodeling_amd import raster  # noqa: E402
from dask_geomodeling_amd.core.graphs import get_sync  # noqa: E402
from dask_geomodeling_amd.hip import program as P  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402

runtime.require_device()
runtime.set_device(0)
for side in (64, 4096):
    jobs, seeds, _ = bench.build_batch(raster, side, 16, 0)

    def step():
        for graph, names in jobs:
            get_sync(graph, names)

    for _ in range(30):
        step()
    runtime.sync()
    if P.spec_pending():
        P.spec_wait()
    for _ in range(10):
        step()
    runtime.sync()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    runtime.sync()
    t2 = time.perf_counter()
    print("side %d: %.1f us per step issued, %.1f us per step completed" % (side, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    if side == 64:
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(300):
            step()
        prof.disable()
        runtime.sync()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
