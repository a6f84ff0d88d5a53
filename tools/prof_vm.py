#!/usr/bin/env python3
"""Run one VM program a few times (for rocprofv3 --pmc runs).  usage: prof_vm.py {add|chain5|mul1} [vec] [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import DeviceArray, program as P, runtime  # noqa: E402
from dask_geomodeling_amd.raster import devops  # noqa: E402
from dask_geomodeling_amd.raster.sources import generate_window  # noqa: E402

F32MAX = float(np.finfo("f4").max)
KW = {"dtype": "float32", "fillvalue": F32MAX}


def build(name, leaves):
    a, b = leaves
    if name == "add":
        return devops.math_node("add", KW, [a, b])
    if name == "mul1":
        return devops.math_node("multiply", KW, [a, P.Lit(2.5)])
    x = devops.math_node("add", KW, [a, b])
    x = devops.math_node("multiply", KW, [x, P.Lit(2.5)])
    x = devops.math_node("subtract", KW, [x, b])
    x = devops.math_node("add", KW, [x, P.Lit(1.5)])
    return devops.clip_node(x, a)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "chain5"
    vec = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    side = 4096
    runtime.set_device(0)
    shape = (1, side, side)
    tiles = [(generate_window((side, side), "f4", 2 * i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side),
              generate_window((side, side), "f4", 2 * i + 1, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side))
             for i in range(4)]
    outs = [DeviceArray.empty(shape, "f4") for _ in range(4)]
    ndv = np.float32(F32MAX)
    comp = P.compile_expr(build(name, [P.Leaf(0, "f4", ndv), P.Leaf(1, "f4", ndv)]), 2)
    P.set_tuning(vec=vec)
    e0, e1 = runtime.Event(), runtime.Event()
    for r in range(reps + 1):
        if r == 1:
            e0.record()
        for i in range(4):
            P.launch(comp, list(tiles[i]), shape, out=outs[i])
    e1.record()
    e1.synchronize()
    us = e0.elapsed_ms(e1) / (4 * reps) * 1e3
    print("%s vec=%d: %.1f us/launch %.0f GB/s" % (name, vec, us, 12 * side * side / us / 1e3))


if __name__ == "__main__":
    main()
