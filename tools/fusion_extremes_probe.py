"""Element-wise views far larger than the five Blocks of the headline (GPU box): sums of many sources, long chains over one source,
wide trees; us per 4096^2 request with the view's plan kept, and for the first request of a fresh view."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import program as P
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
side = 4096
srcs = [raster.MemorySource(generate_window((side, side), "f4", 300 + i, 0.0, 100.0, F, (97, 89), 0, 0, side, side), np.float32(F), "EPSG:28992", 1.0, (0.0, float(side))) for i in range(24)]
req = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")


def build(kind, n):
    if kind == "sum of n sources":
        v = srcs[0]
        for i in range(1, n):
            v = v + srcs[i]
        return v
    if kind == "chain of n ops on one source":
        v = srcs[0]
        for i in range(n):
            v = (v * 1.0001 + 0.5) if i % 2 else (v - 0.25)
        return v
    if kind == "balanced tree over n sources":
        level = list(srcs[:n])
        while len(level) > 1:
            level = [level[i] + level[i + 1] if i + 1 < len(level) else level[i] for i in range(0, len(level), 2)]
        return level[0]


for kind, sizes in (("sum of n sources", (2, 4, 8, 9, 16, 24)), ("chain of n ops on one source", (5, 20, 60, 200)), ("balanced tree over n sources", (8, 16, 24))):
    for n in sizes:
        with gmconfig.config.set({"geomodeling.materialize": False}):
            try:
                t0 = time.perf_counter()
                view = build(kind, n)
                view.get_data(**req)
                runtime.sync()
                first = (time.perf_counter() - t0) * 1e6
                for _ in range(3):
                    view.get_data(**req)
                runtime.sync()
                if P.spec_pending():
                    P.spec_wait()
                t0 = time.perf_counter()
                for _ in range(10):
                    view.get_data(**req)
                runtime.sync()
                print("%-32s n = %3d   first request %9.0f us   then %8.1f us" % (kind, n, first, (time.perf_counter() - t0) / 10 * 1e6), flush=True)
            except Exception as e:  # noqa: BLE001
                print("%-32s n = %3d   %s: %s" % (kind, n, type(e).__name__, str(e)[:90]), flush=True)
