"""Resampling far from the zooms of tools/kbench.py (GPU box): strong down-sampling, large integer and fractional up-sampling, thin
rasters.  us per call and the algorithmic rate (4 bytes per output cell + the source cells an output cell can see)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import resample
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
cases = [(16384, 16384, 1024, 1024), (16384, 16384, 4096, 4096), (8192, 8192, 5461, 5461), (8192, 8192, 8000, 8000), (128, 128, 12800, 12800),
         (1000, 1000, 12345, 12345), (8192, 64, 8192, 4096), (64, 8192, 4096, 8192), (4096, 4096, 4097, 4097)]
for hs, ws, hd, wd in cases:
    src = generate_window((hs, ws), "f4", 70, 0.0, 100.0, F, (97, 89), 0, 0, hs, ws)
    gt = (0.0, ws / float(ws), 0.0, float(hs), 0.0, -1.0)
    row = "%5d x %5d -> %5d x %5d " % (hs, ws, hd, wd)
    for name, f in (("bilinear", lambda: resample.bilinear(src, hd, wd, np.float32(F))),
                    ("nearest", lambda: resample.nearest_warp(src, (0.0, 1.0, 0.0, float(hs), 0.0, -1.0), (0.0, 0.0, float(ws), float(hs)), hd, wd, F))):
        f(); runtime.sync()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record()
        for _ in range(4):
            f()
        e1.record(); e1.synchronize()
        us = e0.elapsed_ms(e1) / 4 * 1e3
        touched = min(hs * ws, hd * wd * (4 if name == "bilinear" else 1))
        row += "  %s %8.1f us (%5.2f TB/s)" % (name, us, (4.0 * hd * wd + 4.0 * touched) / us / 1e6)
    print(row, flush=True)
