"""Zonal sum over label NOISE (every cell its own label out of 1024) -- the input the label partition of
csrc/gm_zonal.hip is worst at -- and over label blocks, with the partition forced on / off (GM_ZONAL_PART=1/0) or chosen
by the library (unset): python tools/zonal_noise_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.hip.array import DeviceArray  # noqa: E402

h, w = 3072, 4096
rng = np.random.default_rng(5)
values = (rng.standard_normal((1, h, w)) * 100).astype("f4")
values[rng.random(values.shape) < 0.1] = -9999.0
dev_values = DeviceArray.from_numpy(values)
blocks = ((np.arange(h)[:, None] // 96) * 32 + np.arange(w)[None, :] // 128).astype(np.int32)
noise = rng.integers(0, 1024, size=(h, w)).astype(np.int32)
speckled = blocks.copy()
speck = rng.random((h, w)) < 0.02
speckled[speck] = noise[speck]
for name, labels in (("blocks", blocks), ("2 % speckle", speckled), ("noise", noise)):
    dev_labels = DeviceArray.from_numpy(labels)
    for stat in ("sum", "median"):
        geometry.label_statistics(dev_values, np.float32(-9999.0), dev_labels, 1024, stat, None)
        runtime.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            geometry.label_statistics(dev_values, np.float32(-9999.0), dev_labels, 1024, stat, None)
        print("zonal %-7s 1024 labels, %-12s GM_ZONAL_PART=%-5s %8.1f us" % (stat, name, os.environ.get("GM_ZONAL_PART", "unset"), (time.perf_counter() - t0) / 5 * 1e6), flush=True)
