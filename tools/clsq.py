"""One-off: lone Classify on float32 tiles, dedicated kernel vs program route (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd.hip import runtime, program as P
from dask_geomodeling_amd.raster import tables
from dask_geomodeling_amd.raster.sources import generate_window
from tools.kbench import timeit
side = 8192; n = side * side
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
tiles = [generate_window((side, side), "f4", 40 + i, -100.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(4)]
leaf = P.Leaf(0, "f4", np.float32(F32MAX))
for nb in (2, 5, 12, 30):
    bins = list(np.linspace(-50.0, 150.0, nb))
    timeit("dedicated kernel, %2d edges" % nb, 5.0 * n, [lambda t=t: tables.classify_process({"values": t, "no_data_value": np.float32(F32MAX)}, bins, False) for t in tiles])
    comp = P.compile_expr(tables.classify_node(leaf, bins, False), 1)
    timeit("program route,    %2d edges" % nb, 5.0 * n, [lambda t=t: P.launch(comp, [t], (1, side, side)) for t in tiles])
