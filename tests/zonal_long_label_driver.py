"""Labels of more than 64 groups of 256 blocks (4.2 M cells and more: the composing wave of csrc/gm_zonal.hip holds the
groups' descriptions 64 at a time) against np.bincount's order (the oracle).  Like tests/zonal_certificate_driver.py every
label ends with cells that cancel the sequential float64 sum before them, so that the float32 result shows every bit of
it.  Kinds: cells that only grow the sum (every group one integer addition: the binade analysis), zero-mean noise (groups of
either kind, blocks turned away around the powers of two), noise with a few cells of much finer grain, large integers.
Prints 'LONG LABELS ok <cases>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
rng = np.random.default_rng(4242)
cases = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    kind = it % 4
    h, w = int(rng.integers(2300, 3100)), int(rng.integers(3900, 4100))   # 9 - 12.7 M cells: 137 - 194 groups
    dt = ["f4", "f4", "f4", "i4"][kind]
    labels = np.zeros((h, w), dtype=np.int32)
    if it >= 4:                                      # a second long label below the first, and a few strays
        labels[int(h * 0.55):] = 1
    labels[rng.random((h, w)) < 0.01] = np.iinfo(np.int32).max
    n = int(labels[labels < 2].max()) + 1
    base = rng.standard_normal((h, w))
    if kind == 0:
        values = np.abs(base) * 100 + 3.0
    elif kind == 1:
        values = base * 100
    elif kind == 2:
        values = base * 100
        fine = rng.random((h, w)) < 2e-5
        values[fine] = rng.standard_normal(int(fine.sum())) * 1e-7
    else:
        values = rng.integers(-2 ** 20, 2 ** 20, size=(h, w)).astype(np.float64)
    values = values.astype(dt)
    ndv = np.dtype(dt).type(-9999 if np.dtype(dt).kind == "f" else np.iinfo(dt).max)
    values[rng.random((h, w)) < 0.05] = ndv
    flat_v, flat_l = values.reshape(-1), labels.reshape(-1)
    for l in range(n):
        idx = np.flatnonzero((flat_l == l) & (flat_v != ndv))
        running = np.cumsum(flat_v[idx[:-3]].astype(np.float64))[-1]   # np.cumsum adds in order, like bincount
        for k in range(3):
            piece = np.dtype(dt).type(-running) if np.dtype(dt).kind == "f" else np.dtype(dt).type(int(np.clip(-running, -2 ** 31 + 2, 2 ** 31 - 2)))
            flat_v[idx[-3 + k]] = piece
            running += float(piece)
    buckets = [(list(range(n)), labels)]
    values3 = values[None]
    for stat in ("sum", "mean"):
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values3, ndv, None, stat, None)
        got, got_no = geometry.aggregate_labelled(buckets, n, values3, ndv, None, stat, None)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "case %d kind %d %s %s: %r vs %r" % (it, kind, dt, stat, got, want)
        assert got_no == want_no
        cases += 1
print("LONG LABELS ok", cases)
from helpers import print_routes  # noqa: E402
print_routes("zonal")
