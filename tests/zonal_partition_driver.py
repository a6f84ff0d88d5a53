"""The label sort of csrc/gm_zonal.hip as a stable partition (zonal_part_*_kernel) against the oracle: label rasters of
every texture -- long runs, single-cell noise (64 different labels in a wave), one label, labels nobody holds, label
counts around the 2047-label limit of the partition (above it rocPRIM's radix sort takes over) -- over rasters of a few
tiles with ragged ends, for the statistics that sort (sum / mean with GM_ZONAL_SORTED=1, median, percentiles).  Run by
tests/test_gpu_parity.py::test_zonal_label_partition in a process of its own, with GM_ZONAL_PART=1 and =0.  Prints
'PARTITION ok <cases>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
rng = np.random.default_rng(77)
cases = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    texture = it % 6
    h, w = int(rng.integers(60, 400)), int(rng.integers(100, 900))
    n = [1, 7, 300, 2046, 2047, 2500][int(rng.integers(0, 6))] if texture != 5 else int(rng.integers(1, 2047))
    if texture == 0:      # blocks: runs of one label along the rows
        bh, bw = int(rng.integers(3, 40)), int(rng.integers(3, 200))
        labels = ((np.arange(h)[:, None] // bh) * 37 + np.arange(w)[None, :] // bw) % n
    elif texture == 1:    # noise: every cell its own label
        labels = rng.integers(0, n, size=(h, w))
    elif texture == 2:    # runs with noise sprinkled in
        labels = (np.arange(h * w).reshape(h, w) // int(rng.integers(30, 5000))) % n
        noise = rng.random((h, w)) < 0.1
        labels[noise] = rng.integers(0, n, size=int(noise.sum()))
    elif texture == 3:    # only a few of the labels occur
        labels = rng.choice(rng.integers(0, n, size=3), size=(h, w))
    elif texture == 4:    # columns: the label changes with every cell along a row, repeats down the columns
        labels = np.broadcast_to(np.arange(w)[None, :] % n, (h, w)).copy()
    else:
        labels = np.full((h, w), int(rng.integers(0, n)))
    labels = labels.astype(np.int32)
    labels[rng.random((h, w)) < 0.05] = np.iinfo(np.int32).max  # outside every geometry
    dt = ["f4", "f8", "i4", "u1", "i8", "f4"][it % 6]
    if np.dtype(dt).kind == "f":
        values = (rng.standard_normal((1, h, w)) * rng.choice([1e-6, 1.0, 100.0, 1e7], size=(1, h, w))).astype(dt)
        ndv = np.dtype(dt).type(-9999.0)
    else:
        info = np.iinfo(dt)
        values = rng.integers(max(info.min, -10 ** 9), min(info.max, 10 ** 9), size=(1, h, w)).astype(dt)
        ndv = np.dtype(dt).type(info.max)
    values[rng.random(values.shape) < 0.1] = ndv
    buckets = [(list(range(n)), labels)]
    for stat, q in (("sum", None), ("mean", None), ("median", None), ("percentile", float(rng.integers(0, 101)))):
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, None, stat, q)
        got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, None, stat, q)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "case %d texture %d n %d %s %s: %r vs %r at %r" % (
            it, texture, n, dt, stat, got[~same][:3], want[~same][:3], np.argwhere(~same)[:3].tolist())
        assert got_no == want_no
        cases += 1
print("PARTITION ok", cases)
from helpers import print_routes  # noqa: E402
print_routes("zonal")
