"""Zonal statistics at the default raster limit (3072 x 4096 cells) over label rasters of many textures, against the oracle:
blocks of several sizes (up to 49 152 labels), vertical stripes, thin bands, a few large labels with strays, labels drawn
at random per cell (2, 40 and 1000 of them: nothing like runs), one label.  Zero-mean float32 noise with a tenth no data,
the same made positive, and rounded to int32.  The unit tests reach every route on small rasters; this driver is for what
only shows at size (windows of 64 groups, tables of many stretches, second attempts with shorter stretches).
Prints 'FULL SIZE ok <cases>'.  Arguments: [layouts to run, comma separated indices or "all"] [HxW: another raster size, by
hand]; GM_FULL_MEDIANS=1: medians of the layout of
49 152 labels too (the oracle takes a while)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
h, w = (int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (3072, 4096)   # (by hand: 16384x16384 and the like)
big = h * w > 3072 * 4096
rng = np.random.default_rng(55)
noise = (rng.standard_normal((1, h, w)) * 100).astype("f4")
holes = rng.random(noise.shape) < 0.1
positive = (np.abs(noise) + np.float32(3.0)).astype("f4")
ints = np.round(noise).astype("i4")
noise[holes] = -9999.0
positive[holes] = -9999.0
ints[holes] = -9999
yy, xx = np.arange(h)[:, None], np.arange(w)[None, :]
layouts = [
    ("blocks 16 x 16", (yy // 16) * (w // 16) + xx // 16),
    ("blocks 96 x 128", (yy // 96) * 32 + xx // 128),
    ("blocks 48 x 40, not aligned", ((yy + 7) // 48) * ((w + 40 + 13) // 40) + (xx + 13) // 40),
    ("vertical stripes 25 wide", np.broadcast_to(xx // 25, (h, w))),
    ("horizontal bands 3 rows", np.broadcast_to(yy // 3, (h, w))),
    ("3 labels in thirds + 1 % strays", np.where(rng.random((h, w)) < 0.01, 3, np.broadcast_to(yy * 3 // h, (h, w)))),
    ("2 labels at random", rng.integers(0, 2, size=(h, w))),
    ("40 labels at random", rng.integers(0, 40, size=(h, w))),
    ("1000 labels at random", rng.integers(0, 1000, size=(h, w))),
    ("one label", np.zeros((h, w), dtype=np.int64)),
    ("one label with 2 % outside", np.where(rng.random((h, w)) < 0.02, np.iinfo(np.int32).max, 0)),
]
chosen = [int(i) for i in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1] != "all" else range(len(layouts))
cases = 0
for i in chosen:
    name, labels = layouts[i]
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n = int(labels[labels < np.iinfo(np.int32).max].max()) + 1
    buckets = [(list(range(n)), labels)]
    for tname, values, ndv in (("noise", noise, np.float32(-9999.0)), ("positive", positive, np.float32(-9999.0)), ("int32", ints, np.int32(-9999))):
        stats = (("sum", "mean", "count", "max") if big else ("sum", "mean", "count", "min", "max")) + (("median",) if tname == "noise" and (n <= 6592 or os.environ.get("GM_FULL_MEDIANS")) and not big else ())
        for stat in stats:
            with np.errstate(all="ignore"):
                want, want_no = O.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
            got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, None, stat, None)
            got, want = np.asarray(got), np.asarray(want)
            assert got.dtype == want.dtype and got.shape == want.shape, (name, tname, stat, got.dtype, want.dtype)
            if got.dtype.kind == "f":
                same = (got.view("u%d" % got.dtype.itemsize) == want.view("u%d" % got.dtype.itemsize)) | (np.isnan(got) & np.isnan(want))
            else:
                same = got == want
            assert same.all(), "%s, %s, %s: %d of %d differ, e.g. %r vs %r" % (name, tname, stat, (~same).sum(), same.size, got[~same][:3], want[~same][:3])
            assert got_no == want_no, (name, tname, stat)
            cases += 1
    print(name, "ok", flush=True)
print("FULL SIZE ok", cases)
from helpers import print_routes  # noqa: E402
print_routes("zonal")
