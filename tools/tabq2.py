import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd.hip import runtime, program as P
from dask_geomodeling_amd.raster import devops, tables
from dask_geomodeling_amd.raster.sources import generate_window
side = 8192
runtime.set_device(0)
I32MAX = 2147483647
i4 = [generate_window((side, side), "i4", 60 + i, 0.0, 256.0, I32MAX, (50, 50), 0, 0, side, side) for i in range(2)]
table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[k, (7 * k) % 1000] for k in range(0, 256, 2)]}
def prog(build):
    return P.compile_expr(build(P.Leaf(0, "i4", I32MAX)), 1)
which = sys.argv[1]
comp = prog((lambda l: devops.mask_node(tables.reclassify_node(l, table), 1)) if which == "lookup" else (lambda l: devops.mask_node(l, 1)))
P.set_tuning(vec=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for r in range(3):
    for t in i4:
        P.launch(comp, [t], (1, side, side))
runtime.sync()
