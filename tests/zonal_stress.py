"""Random rasters / label layouts / statistics through geometry.aggregate_labelled against the oracle (GPU box):
python tests/zonal_stress.py [n].  Lives under tests/ because it checks against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd import geometry
from dask_geomodeling_amd.hip import runtime
from oracle import ref_numpy as O
runtime.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4711)  # [n] [seed]
bad = 0
STATS = [("sum", None), ("mean", None), ("count", None), ("min", None), ("max", None), ("median", None), ("percentile", 0.0),
         ("percentile", 33.3), ("percentile", 100.0)]
for it in range(N):
    dt = ["f4", "f4", "f8", "u1", "i1", "i2", "u2", "i4", "i8", "?"][int(rng.integers(10))]
    frames = int(rng.integers(1, 3))
    h, w = int(rng.integers(1, 400)), int(rng.integers(1, 900))
    if it % 10 == 9:  # every tenth raster up to 4 M cells (labels of millions of cells, many stretches per wave, long partitions)
        h, w = int(rng.integers(600, 2048)), int(rng.integers(900, 2048))
    n = int(rng.integers(1, 60)) if it % 3 else int(rng.integers(1000, 70000))
    layout = int(rng.integers(4))
    if layout == 0:      # blocks (long runs)
        by, bx = int(rng.integers(1, 60)), int(rng.integers(1, 200))
        labels = ((np.arange(h)[:, None] // by) * ((w + bx - 1) // bx) + np.arange(w)[None, :] // bx) % n
    elif layout == 1:    # scattered cells (runs of one)
        labels = rng.integers(0, n, size=(h, w))
    elif layout == 2:    # one label almost everywhere
        labels = np.zeros((h, w), dtype=np.int64)
        labels[rng.random((h, w)) < 0.05] = rng.integers(0, n)
    else:                # vertical stripes (every step of 64 cells straddles labels)
        labels = np.broadcast_to(np.arange(w)[None, :] // int(rng.integers(1, 40)) % n, (h, w)).copy()
    labels = labels.astype(np.int32)
    labels[rng.random((h, w)) < 0.1] = np.iinfo(np.int32).max
    if dt == "?":
        values = rng.random((frames, h, w)) < 0.5
        ndv = None
    elif np.dtype(dt).kind == "f":
        scale = rng.choice([1e-6, 1e-3, 1.0, 1e3, 1e8], size=(frames, h, w)) if it % 2 else 1.0
        values = (rng.standard_normal((frames, h, w)) * 10 * scale).astype(dt)
        if it % 4 == 0:
            values = np.round(values)  # dyadic: the order cannot matter
        if it % 5 == 0:
            values[rng.random(values.shape) < 0.01] = np.nan
        if it % 7 == 0:
            values[rng.random(values.shape) < 0.01] = np.inf
        values[rng.random(values.shape) < 0.05] = 0.0
        values[rng.random(values.shape) < 0.05] = -0.0
        ndv = np.dtype(dt).type(-9999.0)
    else:
        info = np.iinfo(dt)
        span = int(rng.choice([5, 100, 30000, 2**40]))
        values = rng.integers(max(info.min, -span), min(info.max, span), size=(frames, h, w)).astype(dt)
        ndv = np.dtype(dt).type(info.max)
    if ndv is not None:
        values[rng.random(values.shape) < 0.15] = ndv
    thr = None
    if it % 3 == 0:
        thr = (rng.standard_normal(n) * 5).astype("f4")
        thr[rng.random(n) < 0.2] = np.nan
    stat, q = STATS[int(rng.integers(len(STATS)))]
    if dt == "?" and stat == "percentile":
        stat, q = "median", None
    buckets = [(list(range(n)), labels)]
    try:
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, thr, stat, q)
        got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, thr, stat, q)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want)) | (got == want)  # (signed zeros compare equal)
        assert same.all(), "%d of %d results differ" % ((~same).sum(), same.size)
        assert got_no == want_no, "geometries without cells"
    except AssertionError as e:
        bad += 1
        print("FAIL", it, dt, (frames, h, w), "n", n, "layout", layout, stat, q, "thr" if thr is not None else "", e)
print("done", N, "failures", bad)
print("guard violations", runtime.guard_violations(), "guard mode", runtime.lib().gm_guard_enabled())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import print_routes  # noqa: E402
print_routes("zonal")
