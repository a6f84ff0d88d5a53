#!/usr/bin/env python3
"""Sweep the VM launch knobs on the canonical fused chain (GPU box only).

Prints achieved algorithmic GB/s (12 B/px) per configuration; used to pick the
defaults baked into gm_elemwise.hip.  Not part of the product path."""
import ctypes
import itertools
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import DeviceArray, program as P, runtime  # noqa: E402
from dask_geomodeling_amd.raster import devops  # noqa: E402

F32MAX = float(np.finfo("f4").max)
KW = {"dtype": "float32", "fillvalue": F32MAX}


def chain(leaves):
    a, b = leaves
    x = devops.math_node("add", KW, [a, b])
    x = devops.math_node("multiply", KW, [x, P.Lit(2.5)])
    x = devops.math_node("subtract", KW, [x, b])
    x = devops.math_node("add", KW, [x, P.Lit(1.5)])
    return devops.clip_node(x, a)


def single_add(leaves):
    return devops.math_node("add", KW, list(leaves))


def gen(shape, seed):
    from dask_geomodeling_amd.raster.sources import generate_window
    return generate_window(shape[1:], "f4", seed, 0.0, 100.0, F32MAX, (97, 89), 0, 0, shape[1], shape[2])


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    runtime.set_device(0)
    print(runtime.device_name())
    shape = (1, side, side)
    n = side * side
    ndv = np.float32(F32MAX)
    A = [gen(shape, 2 * i) for i in range(ntiles)]
    B = [gen(shape, 2 * i + 1) for i in range(ntiles)]
    OUT = [DeviceArray.empty(shape, "f4") for _ in range(ntiles)]
    leaves = [P.Leaf(0, "f4", ndv), P.Leaf(1, "f4", ndv)]
    progs = {"chain5": P.compile_expr(chain(leaves), 2), "add": P.compile_expr(single_add(leaves), 2)}
    results = []
    for name, comp in progs.items():
        for vec, nt in itertools.product((4, 8), (3,)):
            P.set_tuning(vec=vec)
            for i in range(ntiles):  # warm-up
                P.launch(comp, [A[i], B[i]], shape, out=OUT[i])
            runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            reps = 3
            e0.record()
            for _ in range(reps):
                for i in range(ntiles):
                    P.launch(comp, [A[i], B[i]], shape, out=OUT[i])
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_ms(e1) / (reps * ntiles)
            gbs = 12.0 * n / (ms * 1e-3) / 1e9
            results.append((name, vec, nt, ms * 1e3, gbs))
            print("%-7s vec=%d nt=%d  %8.1f us/tile  %7.1f GB/s" % results[-1], flush=True)
    P.set_tuning()
    best = {}
    for r in results:
        if r[0] not in best or r[4] > best[r[0]][4]:
            best[r[0]] = r
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
