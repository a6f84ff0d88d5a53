#!/usr/bin/env python3
"""Time the VM on add / chain5 for a list of tuning configs: vm_quick.py vec,nt,batch [...] (GPU box only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import DeviceArray, program as P, runtime  # noqa: E402
from tools.tune_vm import chain, single_add, gen, F32MAX  # noqa: E402


def main():
    cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(0, 3, 1)]
    side, ntiles = 4096, 16
    runtime.set_device(0)
    shape = (1, side, side)
    ndv = np.float32(F32MAX)
    A = [gen(shape, 2 * i) for i in range(ntiles)]
    B = [gen(shape, 2 * i + 1) for i in range(ntiles)]
    OUT = [DeviceArray.empty(shape, "f4") for _ in range(ntiles)]
    leaves = [P.Leaf(0, "f4", ndv), P.Leaf(1, "f4", ndv)]
    progs = {"chain5": P.compile_expr(chain(leaves), 2), "add": P.compile_expr(single_add(leaves), 2)}
    for cfg in cfgs:
        vec, nt, batch = cfg
        P.set_tuning(vec=vec)
        for name, comp in progs.items():
            def step():
                if batch <= 1:
                    for i in range(ntiles):
                        P.launch(comp, [A[i], B[i]], shape, out=OUT[i])
                else:
                    for i in range(0, ntiles, batch):
                        P.launch_batch(comp, [[A[j], B[j]] for j in range(i, i + batch)], shape, outs=OUT[i:i + batch])
            step()
            runtime.sync()
            e0, e1 = runtime.Event(), runtime.Event()
            reps = 4
            e0.record()
            for _ in range(reps):
                step()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_ms(e1) / (reps * ntiles) * 1e3
            print("%-7s vec=%2d nt=%d batch=%2d  %6.1f us/tile %7.1f GB/s" % (name, vec, nt, batch, us, 12.0 * side * side / us / 1e3), flush=True)


if __name__ == "__main__":
    main()
