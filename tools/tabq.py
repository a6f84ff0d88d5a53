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
u1 = [generate_window((side, side), "u1", 30 + i, 0.0, 6.0, 255, (50, 50), 0, 0, side, side) for i in range(4)]
table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[k, (7 * k) % 1000] for k in range(0, 256, 2)]}
table2 = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[0, 10], [1, 11], [3, 13], [5, 15]]}

def prog(build, leaf_dt, ndv):
    leaf = P.Leaf(0, leaf_dt, ndv)
    return P.compile_expr(build(leaf), 1)

def run(label, comp, tiles, bpp):
    outs = [None] * len(tiles)
    timeit(label, bpp * n, [lambda t=t: P.launch(comp, [t], (1, side, side)) for t in tiles])

for vec in (4, 8):
    P.set_tuning(vec=vec)
    run("vec=%d i32: lookup -> i64 (12 B/px)" % vec, prog(lambda l: tables.reclassify_node(l, table), "i4", I32MAX), i4, 12.0)
    run("vec=%d i32: lookup+mask -> u8 (5 B/px)" % vec, prog(lambda l: devops.mask_node(tables.reclassify_node(l, table), 1), "i4", I32MAX), i4, 5.0)
    run("vec=%d i32: mask -> u8 (5 B/px)" % vec, prog(lambda l: devops.mask_node(l, 1), "i4", I32MAX), i4, 5.0)
    run("vec=%d u8: lookup+mask -> u8 (2 B/px)" % vec, prog(lambda l: devops.mask_node(tables.reclassify_node(l, table2), 1), "u1", 255), u1, 2.0)
    run("vec=%d u8: lookup -> i64 (9 B/px)" % vec, prog(lambda l: tables.reclassify_node(l, table2), "u1", 255), u1, 9.0)
