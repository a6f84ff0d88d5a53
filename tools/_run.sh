#!/bin/bash
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import stencil
from dask_geomodeling_amd.raster.sources import generate_window
import kbench
runtime.set_device(0)
side = 8192
for dt, ndv in (("i2", 32767), ("u1", 255), ("f4", float(np.finfo("f4").max)), ("i4", 2**31 - 1)):
    src = [generate_window((side, side), dt, 90 + i, 0.0, 100.0, ndv, (97, 89), 0, 0, side, side) for i in range(2)]
    item = np.dtype(dt).itemsize
    kbench.timeit("hillshade %s -> u8 (%d B/px)" % (dt, item + 1), (item + 1.0) * (side - 2) ** 2,
                  [lambda t=t: stencil.hillshade(t, np.dtype(dt).type(ndv), 0, (0.5, 0.5), 45.0, 315.0) for t in src])
PY
