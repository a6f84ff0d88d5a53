import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd.hip import DeviceArray, runtime
runtime.set_device(0)
a = np.random.default_rng(0).random((1, 4096, 4096), dtype=np.float32)
for _ in range(2):
    d = DeviceArray.from_numpy(a)
t0 = time.perf_counter()
for _ in range(10):
    d = DeviceArray.from_numpy(a)
runtime.sync()
up = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
for _ in range(10):
    b = d.to_numpy()
down = (time.perf_counter() - t0) / 10
print("H2D 64 MiB: %.2f ms = %.1f GB/s ; D2H: %.2f ms = %.1f GB/s" % (up * 1e3, a.nbytes / up / 1e9, down * 1e3, a.nbytes / down / 1e9))
