"""Round 6: host microseconds of one get_data request on the plan path (the canonical chain over two resident sources), with the
GPU taken out of the figure (256^2 requests: the kernel takes 2 us) next to the same requests at 4096^2, and how many of them
were answered from a kept box (plan.ViewPlan.boxes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import raster, config as gmconfig, plan
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
big = 16384
A = raster.MemorySource(generate_window((big, big), "f4", 7001, 0.0, 100.0, F, (97, 89), 0, 0, big, big), np.float32(F), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(generate_window((big, big), "f4", 7002, 0.0, 100.0, F, (97, 89), 0, 0, big, big), np.float32(F), "EPSG:28992", 1.0, (0.0, float(big)))
view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
for side in (256, 4096):
    boxes = [(float(i * side), float(j * side), float((i + 1) * side), float((j + 1) * side)) for i in range(4) for j in range(4)]
    req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")
    for kept in (True, False):
        plan.MAX_KEPT_BOXES = 256 if kept else 0
        plan.clear(view)
        with gmconfig.config.set({"geomodeling.materialize": False}):
            for k in range(64):
                view.get_data(bbox=boxes[k % 16], **req)
            runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            t0 = time.perf_counter(); e0.record()
            for k in range(2000):
                view.get_data(bbox=boxes[k % 16], **req)
            e1.record()
            host = (time.perf_counter() - t0) / 2000 * 1e6
            runtime.sync()
            wall = (time.perf_counter() - t0) / 2000 * 1e6
            print("%4d^2 requests, kept boxes %-5s: host %.1f us per request, with the GPU drained %.1f us, device %.1f us" % (side, kept, host, wall, e0.elapsed_ms(e1) / 2000 * 1e3), flush=True)
