"""One-off: cost of staging large LOOKUP windows in the VM (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd.hip import runtime, program as P
from dask_geomodeling_amd.raster import devops, tables
from dask_geomodeling_amd.raster.sources import generate_window
from tools.kbench import timeit
side = 8192; n = side * side
runtime.set_device(0)
I32MAX = 2147483647
i4 = [generate_window((side, side), "i4", 60 + i, 0.0, 256.0, I32MAX, (50, 50), 0, 0, side, side) for i in range(4)]
for W in (256, 1024, 4096, 8000):
    table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[k, (7 * k) % 1000] for k in range(0, W, 2)]}
    leaf = P.Leaf(0, "i4", I32MAX)
    comp = P.compile_expr(devops.mask_node(tables.reclassify_node(leaf, table), 1), 1)
    timeit("window %5d: lookup+mask i32 -> u8 (5 B/px)" % W, 5.0 * n, [lambda t=t: P.launch(comp, [t], (1, side, side)) for t in i4])
    comp = P.compile_expr(tables.reclassify_node(leaf, table), 1)
    timeit("window %5d: lookup i32 -> i64 (12 B/px)" % W, 12.0 * n, [lambda t=t: P.launch(comp, [t], (1, side, side)) for t in i4])
