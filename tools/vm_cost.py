#!/usr/bin/env python3
"""Marginal cost of one VM micro-op: chains of N identical steps over 4096^2 f32 tiles, 16 tiles per
launch (GPU box only).  us/tile at N steps, and (t(N) - t(1)) / (N - 1) per extra step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import DeviceArray, program as P, runtime  # noqa: E402
from dask_geomodeling_amd.raster import devops, tables  # noqa: E402
from tools.tune_vm import gen, F32MAX, KW  # noqa: E402


def main():
    side, ntiles = 4096, 16
    runtime.set_device(0)
    shape = (1, side, side)
    ndv = np.float32(F32MAX)
    A = [gen(shape, 2 * i) for i in range(ntiles)]
    B = [gen(shape, 2 * i + 1) for i in range(ntiles)]
    a, b = P.Leaf(0, "f4", ndv), P.Leaf(1, "f4", ndv)
    BOOLKW = {"dtype": "bool", "fillvalue": None}

    def rep(step, n, start=None):
        x = a if start is None else start
        for k in range(n):
            x = step(x, k)
        return x

    steps = {
        "mul scalar": lambda x, k: devops.math_node("multiply", KW, [x, P.Lit(1.0001)]),
        "add scalar": lambda x, k: devops.math_node("add", KW, [x, P.Lit(0.5)]),
        "add input": lambda x, k: devops.math_node("add", KW, [x, b]),
        "sub input": lambda x, k: devops.math_node("subtract", KW, [x, b]),
        "div scalar": lambda x, k: devops.math_node("divide", KW, [x, P.Lit(1.0001)]),
        "clip input": lambda x, k: devops.clip_node(devops.math_node("add", KW, [x, P.Lit(0.5)]), b),
    }
    def timed(label, root):
        comp = P.compile_expr(root, 2)
        outs = [DeviceArray.empty(shape, comp.out_dtype) for _ in range(ntiles)]
        tiles = [[A[j], B[j]] for j in range(ntiles)]
        P.launch_batch(comp, tiles, shape, outs=outs)
        runtime.sync()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record()
        for _ in range(4):
            P.launch_batch(comp, tiles, shape, outs=outs)
        e1.record()
        e1.synchronize()
        print("%-44s %6.1f us/tile" % (label, e0.elapsed_ms(e1) / (4 * ntiles) * 1e3), flush=True)

    if "tables" in sys.argv[1:]:
        x = devops.math_node("add", KW, [a, P.Lit(0.5)])
        timed("add", x)
        for nb in (2, 5, 12, 30):
            bins = list(np.linspace(-50.0, 150.0, nb))
            timed("add -> classify %d edges" % nb, tables.classify_node(x, bins, False))
        c = tables.classify_node(x, [-50.0, -10.0, 0.0, 10.0, 50.0], False)
        table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[0, 10], [1, 11], [3, 13], [5, 15]]}
        r = tables.reclassify_node(c, table)
        timed("add -> classify 5 -> reclassify -> mask (u8)", devops.mask_node(r, 1))
        table2 = dict(table, data=[[k, k + 10] for k in range(6)])
        timed("add -> classify 5 -> reclassify 6 dense keys -> mask", devops.mask_node(tables.reclassify_node(c, table2), 1))
        return
    if "mosaic" in sys.argv[1:]:
        C = [gen(shape, 100 + i) for i in range(ntiles)]
        D = [gen(shape, 200 + i) for i in range(ntiles)]
        c, d = P.Leaf(2, "f4", ndv), P.Leaf(3, "f4", ndv)
        F64KW = KW
        x = devops.math_node("divide", KW, [devops.math_node("subtract", KW, [devops.math_node("multiply", KW, [devops.math_node("add", KW, [a, b]), P.Lit(0.5)]), c]), P.Lit(2.0)])
        arith = x
        x = devops.clip_node(x, devops.math_node("greater", BOOLKW, [d, P.Lit(20.0)]))
        clipped = x
        x = tables.classify_node(x, [-50.0, -10.0, 0.0, 10.0, 50.0], False)
        table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[0, 10], [1, 11], [3, 13], [5, 15]]}
        full = devops.mask_node(tables.reclassify_node(x, table), 1)
        for vec in (8,):
            P.set_tuning(vec=vec)
            for label, root in (("arith (A+B)*.5-C)/2 [16 B/px]", arith), ("+ clip(greater(D,20)) [20 B/px]", clipped),
                                ("+ classify [17 B/px]", x), ("full mosaic view [17 B/px]", full),
                                ("clipped -> mask(1) u8 [17 B/px]", devops.mask_node(clipped, 1)),
                                ("clipped -> classify 2 edges", tables.classify_node(clipped, [0.0, 10.0], False)),
                                ("arith -> classify 5 edges [13 B/px]", tables.classify_node(arith, [-50.0, -10.0, 0.0, 10.0, 50.0], False)),
                                ("(A+B) -> classify 5 edges [9 B/px]", tables.classify_node(devops.math_node("add", KW, [a, b]), [-50.0, -10.0, 0.0, 10.0, 50.0], False)),
                                ):
                comp = P.compile_expr(root, 4)
                outs = [DeviceArray.empty(shape, comp.out_dtype) for _ in range(ntiles)]
                tiles = [[A[j], B[j], C[j], D[j]] for j in range(ntiles)]
                P.launch_batch(comp, tiles, shape, outs=outs)
                runtime.sync()
                e0, e1 = runtime.Event(), runtime.Event()
                e0.record()
                for _ in range(4):
                    P.launch_batch(comp, tiles, shape, outs=outs)
                e1.record()
                e1.synchronize()
                print("vec=%d %-40s %6.1f us/tile" % (vec, label, e0.elapsed_ms(e1) / (4 * ntiles) * 1e3), flush=True)
        return
    which = sys.argv[1:] or list(steps)
    for name in which:
        times = {}
        for n in (1, 4, 8, 12):
            comp = P.compile_expr(rep(steps[name], n), 2)
            outs = [DeviceArray.empty(shape, comp.out_dtype) for _ in range(ntiles)]
            tiles = [[A[j], B[j]] for j in range(ntiles)]
            P.launch_batch(comp, tiles, shape, outs=outs)
            runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            e0.record()
            for _ in range(4):
                P.launch_batch(comp, tiles, shape, outs=outs)
            e1.record()
            e1.synchronize()
            times[n] = e0.elapsed_ms(e1) / (4 * ntiles) * 1e3
        print("%-12s" % name, "  ".join("N=%2d %6.1f us" % (n, t) for n, t in times.items()),
              "  per extra step: %5.2f us/tile (N=12), %5.2f (N=4)" % ((times[12] - times[1]) / 11, (times[4] - times[1]) / 3), flush=True)


if __name__ == "__main__":
    main()
