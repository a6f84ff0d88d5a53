"""Zonal statistics over label layouts of other textures than the 96 x 128 blocks of tools/kbench.py (GPU box): small polygons (tens
of thousands of labels), vertical stripes (every step of a row straddles labels), thin horizontal bands, a few large labels with
strays.  12.6 M float32 cells (zero-mean noise, a tenth no data) and the same rounded to int32; us per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd import geometry
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.hip.array import DeviceArray
runtime.set_device(0)
h, w = 3072, 4096
rng = np.random.default_rng(5)
values = (rng.standard_normal((1, h, w)) * 100).astype("f4")
values[rng.random(values.shape) < 0.1] = -9999.0
ints = np.round(values).astype("i4")
positive = np.abs(values) + np.float32(3.0)   # (elevation-like: the sums only grow)
positive[values == -9999.0] = -9999.0
dv, di, dp = DeviceArray.from_numpy(values), DeviceArray.from_numpy(ints), DeviceArray.from_numpy(positive)
yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
layouts = [
    ("blocks 16 x 16 (49152 labels)", (yy // 16) * (w // 16) + xx // 16),
    ("blocks 32 x 64 (6144 labels)", (yy // 32) * (w // 64) + xx // 64),
    ("blocks 48 x 40, not aligned (6592 labels)", ((yy + 7) // 48) * ((w + 40 + 13) // 40) + (xx + 13) // 40),
    ("vertical stripes 25 wide (164 labels)", np.broadcast_to(xx // 25, (h, w))),
    ("horizontal bands 3 rows (1024 labels)", np.broadcast_to(yy // 3, (h, w))),
    ("3 labels in thirds + 1 % strays", np.where(rng.random((h, w)) < 0.01, 3, np.broadcast_to(yy * 3 // h, (h, w)))),
]
for name, labels in layouts:
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n_labels = int(labels.max()) + 1
    dl = DeviceArray.from_numpy(labels)
    for tname, tile, ndv in (("f32 noise", dv, np.float32(-9999.0)), ("i32", di, -9999), ("f32 > 0", dp, np.float32(-9999.0))):
        line = "%-44s %-10s" % (name, tname)
        for stat in (("sum", "count", "max", "median") if tile is not dp else ("sum", "mean")):
            for _ in range(3):  # (round 6: three calls before the clock -- a route's scratch buffers are first allocated by its
                #  second call where the first call's verdict sizes them, and a cold hipMalloc costs milliseconds: round 5's
                #  figures for the many-label layouts were mostly that)
                geometry.label_statistics(tile, ndv, dl, n_labels, stat, None)
            runtime.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                geometry.label_statistics(tile, ndv, dl, n_labels, stat, None)
            line += "  %s %8.1f" % (stat, (time.perf_counter() - t0) / 5 * 1e6)
        print(line, flush=True)
