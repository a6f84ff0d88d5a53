"""Long labels through the block pipeline of csrc/gm_zonal.hip (zonal_compose_kernel) against np.bincount's order (the
oracle): run by tests/test_gpu_parity.py::test_zonal_block_pipeline in a process of its own with GM_ZONAL_LONG=512, so
that labels of a few hundred cells already take the pipeline and every kind of block shows up often -- valid blocks,
blocks whose prefixes leave the binade, sums that change sign, exact half-unit cells (ties), inf / NaN cells, sums next
to powers of two -- on float32, float64, int32 and int64 rasters.  Prints 'BLOCKS ok <cases>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
rng = np.random.default_rng(2024)
cases = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    kind = it % 8
    h, w = int(rng.integers(40, 300)), int(rng.integers(200, 1500))
    n = int(rng.integers(1, 5))
    dt = ["f4", "f4", "f8", "f4", "i4", "i8", "f4", "f8"][kind]
    labels = np.zeros((h, w), dtype=np.int32)
    if n > 1:
        labels[:, w // 2:] = 1
        labels[rng.random((h, w)) < 0.02] = int(rng.integers(0, n))
    labels[rng.random((h, w)) < 0.05] = np.iinfo(np.int32).max
    if np.dtype(dt).kind == "f":
        base = rng.standard_normal((1, h, w))
        if kind == 0:      # zero mean: the sum wanders through the binades and changes sign
            values = base * 100
        elif kind == 1:    # a large positive sum with small cells: most blocks are one integer addition
            values = np.abs(base) * 3 + 1000
        elif kind == 2:    # wide range of magnitudes (float64 cells: full 53-bit fractions)
            values = base * rng.choice([1e-9, 1e-3, 1.0, 1e6], size=(1, h, w))
        elif kind == 3:    # exact half units: cells that are odd multiples of 2^-k against a sum of about 2^20
            values = np.round(np.abs(base) * 64) / 64 + 4000.0
            values[0, ::3, ::5] += 2.0 ** -33
        elif kind == 6:    # a few non-finite cells: everything behind them is inf / NaN like bincount's
            values = np.abs(base) * 50
            values[0, h // 2, w // 3] = np.inf
            if it % 16 == 6:
                values[0, h // 2 + 1, w // 5] = np.nan
        else:              # sums next to a power of two: +-1 around 2^k
            values = base * 0.5
            values[0, 0, 0] = 2.0 ** int(rng.integers(8, 30))
        values = values.astype(dt)
        ndv = np.dtype(dt).type(-9999.0)
    else:
        span = 5 if kind == 4 else 2 ** 40
        values = rng.integers(-span, span, size=(1, h, w)).astype(dt)
        ndv = np.dtype(dt).type(np.iinfo(dt).max)
    values[rng.random(values.shape) < 0.1] = ndv
    buckets = [(list(range(n)), labels)]
    for stat in ("sum", "mean"):
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
        got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "case %d kind %d %s %s: %r vs %r" % (it, kind, dt, stat, got[~same][:3], want[~same][:3])
        assert got_no == want_no
        cases += 1
print("BLOCKS ok", cases)
from helpers import print_routes  # noqa: E402
print_routes("zonal")
