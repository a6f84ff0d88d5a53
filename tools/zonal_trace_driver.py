"""One zonal statistic on the 3072 x 4096 float32 raster of tools/kbench.py, for ``rocprofv3 --kernel-trace --stats``:
python3 tools/zonal_trace_driver.py [n_labels] [statistic] [repeats] [positive]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.hip.array import DeviceArray  # noqa: E402

n_labels = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
stat = sys.argv[2] if len(sys.argv) > 2 else "sum"
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 5
positive = len(sys.argv) > 4 and sys.argv[4] == "positive"  # elevation-like cells: the sums only grow
h, w = 3072, 4096
rng = np.random.default_rng(5)
values = (rng.standard_normal((1, h, w)) * 100).astype("f4")
if positive:
    values = np.abs(values) + np.float32(3.0)
values[rng.random(values.shape) < 0.1] = -9999.0
layout = os.environ.get("ZT_LAYOUT", "")  # round 6: the layouts of tools/zonal_layouts_probe.py / zonal_noise_labels_probe.py
yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
if layout:
    labels = {"blocks16": lambda: (yy // 16) * (w // 16) + xx // 16,
              "blocks32x64": lambda: (yy // 32) * (w // 64) + xx // 64,
              "blocks48x40": lambda: ((yy + 7) // 48) * ((w + 40 + 13) // 40) + (xx + 13) // 40,
              "stripes25": lambda: np.broadcast_to(xx // 25, (h, w)),
              "bands3": lambda: np.broadcast_to(yy // 3, (h, w)),
              "thirds": lambda: np.where(rng.random((h, w)) < 0.01, 3, np.broadcast_to(yy * 3 // h, (h, w))),
              }.get(layout, lambda: rng.integers(0, int(layout.replace("random", "") or 1), size=(h, w)))()
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n_labels = int(labels.max()) + 1
    if os.environ.get("ZT_INT"):
        values = np.round(values).astype("i4")
elif n_labels == 1:
    labels = np.zeros((h, w), dtype=np.int32)
else:
    labels = ((np.arange(h)[:, None] // 96) * 32 + np.arange(w)[None, :] // 128).astype(np.int32)
    n_labels = 1024
dev_values, dev_labels = DeviceArray.from_numpy(values), DeviceArray.from_numpy(labels)
import time  # noqa: E402
before = runtime.route_stats("zonal")
for _ in range(repeats):
    t0 = time.perf_counter()
    geometry.label_statistics(dev_values, values.dtype.type(-9999.0), dev_labels, n_labels, stat, None)
    print("call %.1f us" % ((time.perf_counter() - t0) * 1e6), flush=True)
runtime.sync()
print("routes", runtime.routes_taken(before, runtime.route_stats("zonal")))
