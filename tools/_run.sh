#!/bin/bash
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import raster, config as gmconfig
rng = np.random.default_rng(1)
a = (rng.random((1, 64, 64)) * 10).astype("f4")
A = raster.MemorySource(a, np.float32(-1), "EPSG:28992", 1.0, (0.0, 64.0))
x = raster.HillShade(A + 1)
for level in range(10):
    x = raster.Group(x, x)
req = dict(mode="vals", bbox=(0.0, 0.0, 32.0, 32.0), width=32, height=32, projection="EPSG:28992")
x.get_data(**req)
t0 = time.perf_counter(); got = x.get_data(**req); t1 = time.perf_counter()
with gmconfig.config.set({"geomodeling.plans": False}):
    want = x.get_data(**req)
    t2 = time.perf_counter(); want = x.get_data(**req); t3 = time.perf_counter()
print("10 nested groups: plans %.1f ms, graph %.1f ms, equal: %s" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, np.array_equal(got["values"], want["values"])))
PY
