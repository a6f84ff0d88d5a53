"""Zonal sums over label rasters that are not made of runs (GPU box): every cell one of n labels at random -- what the chunk records
must give up on quickly.  usage: zonal_noise_labels_probe.py"""
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
dv, di = DeviceArray.from_numpy(values), DeviceArray.from_numpy(ints)
for n_labels in (2, 6, 20, 40, 1000):
    labels = rng.integers(0, n_labels, size=(h, w)).astype(np.int32)
    dl = DeviceArray.from_numpy(labels)
    for name, tile, ndv in (("f32 noise", dv, np.float32(-9999.0)), ("i32", di, -9999)):
        line = "%4d labels at random, %-10s" % (n_labels, name)
        for stat in ("sum", "count", "max", "median"):
            for _ in range(3):  # (round 6: three calls before the clock -- a route's scratch buffers are first allocated by its
                #  second call where the first call's verdict sizes them, and a cold hipMalloc costs milliseconds: round 5's
                #  figures for the many-label layouts were mostly that)
                geometry.label_statistics(tile, ndv, dl, n_labels, stat, None)
            runtime.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                geometry.label_statistics(tile, ndv, dl, n_labels, stat, None)
            line += "  %s %9.1f us" % (stat, (time.perf_counter() - t0) / 5 * 1e6)
        print(line, flush=True)
