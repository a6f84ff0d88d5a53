"""Temporal statistics far from the 24 frames -> 4 labels of tools/kbench.py (GPU box): a year of daily frames into months, bins of
one frame, a single bin of everything, medians of long bins; 1024^2 cells.  us per call and the rate over the frames read."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.hip.array import DeviceArray
from dask_geomodeling_amd.raster import temporal
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
side, frames = 1024, 365
stack = generate_window((frames * side, side), "f4", 60, 0.0, 100.0, F, (97, 89), 0, 0, frames * side, side).reshape((frames, side, side))
months = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
starts = np.concatenate([[0], np.cumsum(months)[:-1]])
layouts = [("12 months of a year", [(int(a), int(n)) for a, n in zip(starts, months)]),
           ("365 bins of one frame", [(i, 1) for i in range(frames)]),
           ("one bin of 365 frames", [(0, frames)]),
           ("52 weeks", [(7 * i, 7) for i in range(52)])]
for name, runs in layouts:
    row = "%-24s" % name
    for stat in ("sum", "mean", "max", "std", "median"):
        out_dt = temporal.dtype_for_statistic("f4", stat)
        f = lambda: temporal.aggregate_frames(stack, np.float32(F), runs, stat, out_dt)
        try:
            f(); runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record()
            for _ in range(3):
                f()
            e1.record(); e1.synchronize()
            us = e0.elapsed_ms(e1) / 3 * 1e3
            read = sum(n for _, n in runs) * side * side * 4.0 * (2 if stat == "std" else 1)
            row += "  %s %8.1f us (%4.2f TB/s)" % (stat, us, (read + len(runs) * side * side * 4.0) / us / 1e6)
        except Exception as e:  # noqa: BLE001
            row += "  %s %s" % (stat, type(e).__name__)
    print(row, flush=True)
