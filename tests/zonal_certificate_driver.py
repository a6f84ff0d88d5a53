"""The exactness certificates of the zonal block pipeline (csrc/gm_zonal.hip: zonal_block_sum_kernel, zonal_group_exact_kernel)
on labels of several whole groups of 256 blocks, against np.bincount's order (the oracle).  The reference's result is a
float32, which hides the low bits of the float64 sum -- so every label ends with cells that CANCEL the sequential float64 sum
of everything before them (computed here with np.cumsum, which adds in order): the expected result is the tiny residue, and a
device sum that is off by one unit in the last place anywhere shows up in full.  Kinds: zero-mean data (the certificates'
home ground), the same with a few cells of much finer grain (groups and blocks that must be turned away), full 53-bit float64
cells, dyadic cells whose prefixes wander around the 2^(53 + grain) limit, large integers, mostly zeros.
Prints 'CERTIFICATES ok <cases>'."""
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
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 28):
    kind = it % 7
    h, w = int(rng.integers(500, 800)), int(rng.integers(900, 1300))   # 450 k - 1 M cells: 7 - 15 whole groups
    dt = ["f4", "f4", "f8", "f8", "i8", "f4", "f8"][kind]
    labels = np.zeros((h, w), dtype=np.int32)
    if it % 3 == 1:                                  # a second long label interleaved row-wise, and a few strays
        labels[1::2] = 1
    labels[rng.random((h, w)) < 0.01] = np.iinfo(np.int32).max
    n = int(labels[labels < 2].max()) + 1
    base = rng.standard_normal((h, w))
    if kind == 0:
        values = base * 100
    elif kind == 1:
        values = base * 100
        fine = rng.random((h, w)) < 2e-5
        values[fine] = rng.standard_normal(int(fine.sum())) * 1e-7
    elif kind == 2:
        values = base * rng.choice([1e-3, 1.0, 1e3], size=(h, w))
    elif kind == 3:   # multiples of 2^-10 up to 2^33: prefixes around 2^43 = 2^(53 - 10)
        values = np.round(base * 2.0 ** 31) + np.round(rng.random((h, w)) * 1024) / 1024
        values[: h // 3] += 2.0 ** 26                # drifts up through the limit and back
        values[h // 3: 2 * h // 3] -= 2.0 ** 26
    elif kind == 4:
        values = rng.integers(-2 ** 40, 2 ** 40, size=(h, w)).astype(np.float64)
    elif kind == 5:   # mostly zeros
        values = np.where(rng.random((h, w)) < 0.01, base * 1e4, 0.0)
    else:             # tiny and huge together
        values = base * np.exp2(rng.integers(-40, 40, size=(h, w)).astype(np.float64))
    values = values.astype(dt)
    ndv = np.dtype(dt).type(-9999 if np.dtype(dt).kind == "f" else np.iinfo(dt).max)
    values[rng.random((h, w)) < 0.05] = ndv
    # the cancelling tail of every label: its last three active cells become -(the sequential sum before them), in pieces
    flat_v, flat_l = values.reshape(-1), labels.reshape(-1)
    for l in range(n):
        idx = np.flatnonzero((flat_l == l) & (flat_v != ndv))
        running = np.cumsum(flat_v[idx[:-3]].astype(np.float64))[-1]   # np.cumsum adds in order, like bincount
        for k in range(3):
            piece = np.dtype(dt).type(-running) if np.dtype(dt).kind == "f" else np.dtype(dt).type(-int(running) if abs(running) < 2 ** 62 else 0)
            if piece == ndv:
                piece = np.dtype(dt).type(0)
            flat_v[idx[-3 + k]] = piece
            running = running + np.float64(piece)
    values = values.reshape(1, h, w)
    buckets = [(list(range(n)), labels)]
    for stat in ("sum", "mean"):
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
        got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "case %d kind %d %s %s: %r vs %r" % (it, kind, dt, stat, got[~same][:3], want[~same][:3])
        assert got_no == want_no
        cases += 1
print("CERTIFICATES ok", cases)
from helpers import print_routes  # noqa: E402
print_routes("zonal")
