"""reduce_rasters over stacks of other sizes than the golden vectors' handful (GPU box): 2 / 12 / 60 / 300 rasters of 2048^2 float32
cells, every statistic; us per call and the rate over the rasters read."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.reduction import reduce_rasters
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
side = 2048
for n in (2, 12, 60, 300):
    stack = [{"values": generate_window((side, side), "f4", 900 + i, 0.0, 100.0, F, (97, 89), 0, 0, side, side).reshape((1, side, side)),
              "no_data_value": np.float32(F)} for i in range(n)]
    row = "%4d rasters " % n
    for stat in ("sum", "mean", "max", "min", "std", "count", "first", "median", "p90", "product", "argmax"):
        try:
            f = lambda: reduce_rasters(stack, stat)
            f(); runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record()
            for _ in range(2):
                f()
            e1.record(); e1.synchronize()
            us = e0.elapsed_ms(e1) / 2 * 1e3
            row += " %s %.0f (%.1f)" % (stat, us, (n + 1) * side * side * 4.0 / us / 1e6)
        except Exception as e:  # noqa: BLE001
            row += " %s %s" % (stat, type(e).__name__)
    print(row, flush=True)
