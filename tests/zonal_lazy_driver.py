"""The float32 certificate of csrc/gm_zonal.hip (zonal_chunk_combine_kernel, round 6: a label whose float64 sum -- in ANY
order -- lies far enough from every float32 rounding boundary is answered without bincount's order) against the oracle's
np.bincount order, on data made for it and data made to break it:

  noise    zero-mean float32 / float64 noise under a few long labels, blocks, bands and one label: the certificate answers
           (the test asserts that no label was handed to the sorting route);
  ties     labels whose exact sum lies ON a float32 tie or a hair beside it (1 + 2^-24 -/+ cells of 2^-60 ... 2^-30, sums
           next to powers of two, means whose quotient is a tie): the direction is bincount's roundings' to decide;
  near     an integer + half a float32 unit in the last place + a nudge of 2^-1 ... 2^-59 of that, among pairs of noise cells
           that cancel exactly: from nudges the certificate must accept to nudges below bincount's own rounding drift;
  scattered  labels at random per cell: the same certificate over the workgroups' LDS tables (zonal_tables_lazy_kernel), on noise,
           non-finite cells and ties;
  polygons   thousands of small polygons: the certificate on the parallel sums of zonal_exact_kernel for the labels its
           exponent test turns away;
  parcels    thousands of small polygons of positive cells, a handful of them with ties, sums of zero and non-finite cells:
           one pass of parallel sums, the handful added in order (zonal_few_*);
  zeros    sums that cancel to +0.0, to a float32 subnormal, to something that underflows to -0.0 / +0.0;
  wide     tiny and huge cells together, cells beyond float32's range (float64 rasters), int64 cells near 2^62, NaN and
           infinities.

Run by tests/test_gpu_parity.py::test_zonal_float32_certificate with GM_ZONAL_LAZY=1 (the product) and =0 (every sum in
order).  Prints 'LAZY ok <cases>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dask_geomodeling_amd import geometry  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from oracle import ref_numpy as O  # noqa: E402

runtime.set_device(0)
rng = np.random.default_rng(606)
cases = 0
ONLY = sys.argv[1].split(",") if len(sys.argv) > 1 else ["noise", "ties", "near", "scattered", "polygons", "parcels", "zeros", "wide"]


def layout(kind, h, w):
    if kind == 0:      # three labels in thirds and a few strays
        labels = np.minimum(np.arange(h)[:, None] * 3 // h, 2) + np.zeros((1, w), dtype=np.int64)
        stray = rng.random((h, w)) < 0.01
        labels[stray] = rng.integers(0, 3, size=int(stray.sum()))
    elif kind == 1:    # one label
        labels = np.zeros((h, w), dtype=np.int64)
    elif kind == 2:    # blocks
        by, bx = int(rng.integers(40, 130)), int(rng.integers(60, 300))
        labels = (np.arange(h)[:, None] // by) * ((w + bx - 1) // bx) + np.arange(w)[None, :] // bx
    elif kind == 3:    # horizontal bands of whole rows
        labels = np.broadcast_to(np.arange(h)[:, None] // int(rng.integers(3, 40)), (h, w)).copy()
    else:              # vertical stripes: every row holds every label
        labels = np.broadcast_to(np.arange(w)[None, :] // int(rng.integers(100, 500)), (h, w)).copy()
    labels = labels.astype(np.int32)
    n = int(labels.max()) + 1
    labels[rng.random((h, w)) < 0.02] = np.iinfo(np.int32).max
    return labels, n


def check(what, labels, n, values, ndv, thr=None, stats=("sum", "mean")):
    global cases
    buckets = [(list(range(n)), labels)]
    for stat in stats:
        with np.errstate(all="ignore"):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, thr, stat, None)
        got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, thr, stat, None)
        same = (got.view("u4") == want.view("u4")) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s %s %s: %d of %d differ, e.g. label %d: %r vs %r" % (
            what, values.dtype, stat, (~same).sum(), same.size, np.argwhere(~same)[0][1], got[~same][0], want[~same][0])
        assert got_no == want_no, what
        cases += 1


if "noise" in ONLY:
    for it in range(10):
        h, w = int(rng.integers(900, 1600)), int(rng.integers(1500, 2600))
        labels, n = layout(it % 5, h, w)
        dt = "f8" if it % 4 == 3 else "f4"
        frames = 1 + (it % 3 == 0)
        values = rng.standard_normal((frames, h, w)) * 10 ** float(rng.integers(-3, 6))
        if it % 2:
            values *= rng.choice([1e-4, 1.0, 1e2], size=values.shape)
        values = values.astype(dt)
        ndv = np.dtype(dt).type(-9999)
        values[rng.random(values.shape) < 0.07] = ndv
        thr = None
        if it == 4:
            thr = (rng.standard_normal(n) * 5).astype("f4")
        check("noise %d" % it, labels, n, values, ndv, thr)

if "ties" in ONLY:
    # every label a band of whole rows: 1.0, half a float32 unit in the last place of 1.0, then cells that push the exact sum
    # a hair up or down -- some of them too fine for the float64 sum to take up (2^-60: bincount rounds them away one by
    # one, the exact sum takes them all), some not
    for it in range(6):
        h, w = 512, int(rng.integers(700, 1400))
        rows = int(rng.integers(1, 6))
        labels = np.broadcast_to(np.arange(h)[:, None] // rows, (h, w)).copy().astype(np.int32)
        n = int(labels.max()) + 1
        dt = "f8" if it % 2 else "f4"
        values = np.zeros((1, h, w), dtype=dt)
        scale = 2.0 ** rng.integers(-20, 20, size=n)           # (the same picture at other binades)
        for l in range(n):
            band = values[0, l * rows:(l + 1) * rows].reshape(-1)
            m = len(band)
            fine = 2.0 ** -int(rng.integers(25, 61)) if dt == "f8" else 2.0 ** -int(rng.integers(25, 100))
            k = int(rng.integers(0, m - 2))
            sign = rng.choice([-1.0, 1.0])
            pos = rng.permutation(m)
            band[pos[0]] = 1.0 * scale[l]
            band[pos[1]] = 2.0 ** -24 * scale[l]
            band[pos[2:2 + k]] = sign * fine * scale[l]
            if l % 7 == 0:                                           # pairs that cancel: the tie itself
                band[pos[2:2 + k]] = 0
                band[pos[2:2 + (k // 2) * 2:2]] = fine * scale[l]
                band[pos[3:3 + (k // 2) * 2:2]] = -fine * scale[l]
            if l % 11 == 0:                                          # a sum next to a power of two instead
                band[pos[1]] = -(2.0 ** -25) * scale[l]
        check("ties %d" % it, labels, n, values, None)
    # means: count * (a float32 tie) spread over the label's cells
    for it in range(3):
        h, w = 384, 1024
        rows = 3
        labels = np.broadcast_to(np.arange(h)[:, None] // rows, (h, w)).copy().astype(np.int32)
        n = int(labels.max()) + 1
        values = np.full((1, h, w), 1.0 + 2.0 ** -24, dtype="f8")
        jitter = rng.integers(-3, 4, size=(1, h, w)) * 2.0 ** -int(rng.integers(40, 52))
        values += jitter
        check("mean ties %d" % it, labels, n, values, None)

if "near" in ONLY:
    # an integer K, half a float32 unit in the last place of K, a nudge of 2^-j of that half unit to either side, and pairs of
    # noise cells that cancel exactly but make bincount's sum round on the way: from nudges the certificate must accept
    # (j small) to nudges below bincount's own rounding drift
    for it in range(4):
        h, w = 768, int(rng.integers(600, 1500))
        rows = int(rng.integers(2, 40))
        labels = np.broadcast_to(np.arange(h)[:, None] // rows, (h, w)).copy().astype(np.int32)
        n = int(labels.max()) + 1
        dt = "f8" if it % 2 else "f4"
        values = np.zeros((1, h, w), dtype=dt)
        for l in range(n):
            band = values[0, l * rows:(l + 1) * rows].reshape(-1)
            m = len(band)
            K = float(rng.integers(1, 2 ** 20))
            half = float(np.spacing(np.float32(K))) / 2
            j = int(rng.integers(1, 29 if dt == "f8" else 60))
            pos = rng.permutation(m)
            band[pos[0]], band[pos[1]], band[pos[2]] = K, half, rng.choice([-1.0, 1.0]) * half * 2.0 ** -j
            pairs = int(rng.integers(0, (m - 3) // 2 + 1))
            noise = (rng.standard_normal(pairs) * K * 10 ** float(rng.integers(-6, 3))).astype(dt)
            band[pos[3:3 + pairs]] = noise
            band[pos[3 + pairs:3 + 2 * pairs]] = -noise
        check("near %d" % it, labels, n, values, None)

if "scattered" in ONLY:
    # labels at random per cell (no runs: the certificate over the workgroups' LDS tables, zonal_tables_lazy_kernel): noise; noise
    # with non-finite cells under some labels; labels that are all zeros but for a float32 tie and cells too fine to count
    for it in range(6):
        h, w = int(rng.integers(700, 1300)), int(rng.integers(1100, 2300))
        n = [12, 40, 300, 2000, 20, 90][it]
        labels = rng.integers(0, n, size=(h, w)).astype(np.int32)
        labels[rng.random((h, w)) < 0.02] = np.iinfo(np.int32).max
        dt = "f8" if it % 3 == 2 else "f4"
        values = (rng.standard_normal((1, h, w)) * 10 ** float(rng.integers(-2, 4))).astype(dt)
        ndv = np.dtype(dt).type(-9999)
        values[rng.random(values.shape) < 0.07] = ndv
        if it == 4:
            values[rng.random(values.shape) < 1e-6] = np.inf
            values[rng.random(values.shape) < 1e-6] = np.nan
        if it >= 3:
            flat_v, flat_l = values.reshape(-1), labels.reshape(-1)
            for l in range(0, n, 7):
                idx = np.flatnonzero(flat_l == l)
                flat_v[idx] = 0
                pick = rng.permutation(len(idx))
                flat_v[idx[pick[0]]], flat_v[idx[pick[1]]] = 1.0, 2.0 ** -24
                flat_v[idx[pick[2:2 + len(idx) // 2]]] = rng.choice([-1.0, 1.0]) * 2.0 ** -int(rng.integers(30, 62))
        check("scattered %d" % it, labels, n, values, ndv)

if "polygons" in ONLY:
    # thousands of small polygons (more than the chunk records take, beyond the partition's 2047 labels: the parallel sums of
    # zonal_exact_kernel): noise with cells next to zero -- the exponents turn those labels away, the certificate on the parallel
    # sums answers them --, and polygons that hold a float32 tie
    for it in range(4):
        h, w = int(rng.integers(900, 1400)), int(rng.integers(1500, 2500))
        by, bx = int(rng.integers(9, 22)), int(rng.integers(9, 22))
        labels = ((np.arange(h)[:, None] // by) * ((w + bx - 1) // bx) + np.arange(w)[None, :] // bx).astype(np.int32)
        n = int(labels.max()) + 1
        labels[rng.random((h, w)) < 0.02] = np.iinfo(np.int32).max
        dt = "f8" if it == 3 else "f4"
        values = (rng.standard_normal((1, h, w)) * 100).astype(dt)
        values[rng.random(values.shape) < 0.01] *= np.dtype(dt).type(1e-9)
        ndv = np.dtype(dt).type(-9999)
        values[rng.random(values.shape) < 0.07] = ndv
        if it >= 2:
            for py in range(0, h // by, 3):
                for px in range(0, w // bx, 5):
                    block = values[0, py * by:(py + 1) * by, px * bx:(px + 1) * bx]
                    block[...] = 0
                    block[0, 0], block[-1, -1] = 1.0, 2.0 ** -24
                    block[1, 1:] = rng.choice([-1.0, 1.0]) * 2.0 ** -int(rng.integers(30, 62))
        check("polygons %d" % it, labels, n, values, ndv)

if "parcels" in ONLY:
    # thousands of small polygons of elevation-like cells (every label passes the certificate on the parallel sums: one pass,
    # zonal_exact_kernel PHASE 3) but for a handful that hold a float32 tie, a sum that cancels to nothing, a non-finite cell:
    # those are added in order by zonal_few_gather_kernel / zonal_few_ordered_kernel
    for it in range(5):
        h, w = int(rng.integers(900, 1400)), int(rng.integers(1500, 2500))
        by, bx = int(rng.integers(8, 20)), int(rng.integers(8, 24))   # (4 000 - 40 000 parcels)
        labels = ((np.arange(h)[:, None] // by) * ((w + bx - 1) // bx) + np.arange(w)[None, :] // bx).astype(np.int32)
        n = int(labels.max()) + 1
        labels[rng.random((h, w)) < 0.02] = np.iinfo(np.int32).max
        dt = ["f4", "f8", "i4", "f4", "f4"][it]
        base = np.abs(rng.standard_normal((1, h, w))) * 100 + 3
        values = (np.round(base) if dt == "i4" else base).astype(dt)
        ndv = np.dtype(dt).type(-9999)
        values[rng.random(values.shape) < 0.07] = ndv
        rows_of, cols_of = h // by, (w + bx - 1) // bx
        for k in range(int(rng.integers(3, 40 if it < 4 else 120))):   # (the last raster: more such labels than there are slots)
            py, px = int(rng.integers(0, rows_of)), int(rng.integers(0, w // bx))
            block = values[0, py * by:(py + 1) * by, px * bx:(px + 1) * bx]
            kind = k % 4 if dt != "i4" else 1
            if kind == 0:      # a float32 tie and cells too fine to count
                block[...] = 0
                block[0, 0], block[-1, -1] = 1.0, 2.0 ** -24
                block[1, 1:] = rng.choice([-1.0, 1.0]) * 2.0 ** -int(rng.integers(30, 62))
            elif kind == 1:    # pairs that cancel: a sum of exactly zero
                half = block.shape[1] // 2
                block[:, half:2 * half] = -block[:, :half]
                block[:, 2 * half:] = 0
            elif kind == 2:    # zero-mean noise around cells next to zero
                block[...] = (rng.standard_normal(block.shape) * 100).astype(dt)
                block[0, :3] = np.array([1e-12, -1e-13, 1e-20], dtype=dt)
            else:              # a non-finite cell
                block[0, 0] = np.nan if k % 8 == 3 else np.inf
        check("parcels %d" % it, labels, n, values, ndv)

if "zeros" in ONLY:
    for it in range(6):
        h, w = 600, int(rng.integers(800, 1200))
        labels, n = layout(2 + it % 3, h, w)
        dt = "f8" if it % 2 else "f4"
        values = (rng.standard_normal((1, h, w)) * 100).astype(dt)
        flat_v, flat_l = values.reshape(-1), labels.reshape(-1)
        for l in range(min(n, 60)):
            idx = np.flatnonzero(flat_l == l)
            if len(idx) < 8:
                continue
            mode = l % 4
            if mode == 0:        # pairs that cancel exactly: +0.0
                half = len(idx) // 2
                flat_v[idx[half:2 * half]] = -flat_v[idx[:half]]
                flat_v[idx[2 * half:]] = 0
            elif mode == 1:      # all zeros of either sign
                flat_v[idx] = np.where(rng.random(len(idx)) < 0.5, 0.0, -0.0)
            elif mode == 2:      # a float32 subnormal / a sum that underflows float32 (float64 rasters), either sign
                flat_v[idx] = 0
                tiny = 2.0 ** -140 if dt == "f4" else 2.0 ** -int(rng.integers(140, 200))
                flat_v[idx[:5]] = np.array([3, -1, 1, -2, -1.5 if dt == "f8" else -2]) * tiny * rng.choice([-1, 1])
            else:                # the cancelling tail of the other drivers
                running = np.cumsum(flat_v[idx[:-3]].astype(np.float64))[-1]
                for k in range(3):
                    piece = np.dtype(dt).type(-running)
                    flat_v[idx[-3 + k]] = piece
                    running = running + float(piece)
        check("zeros %d" % it, labels, n, values, None)

if "wide" in ONLY:
    for it in range(8):
        h, w = int(rng.integers(500, 900)), int(rng.integers(900, 1500))
        labels, n = layout(it % 5, h, w)
        base = rng.standard_normal((1, h, w))
        if it % 4 == 0:        # tiny and huge together
            values = (base * np.exp2(rng.integers(-60, 60, size=base.shape).astype(np.float64))).astype("f4")
            ndv = np.float32(-9999)
        elif it % 4 == 1:      # float64 cells beyond float32's range: the float32 result is inf or a huge finite number
            values = base * np.exp2(rng.integers(100, 140, size=base.shape).astype(np.float64))
            if it == 5:
                values *= 1e260   # ... and beyond what the certificate's bounds can hold
            ndv = np.float64(-9999)
        elif it % 4 == 2:      # int64 cells near 2^62: (double)cell rounds
            values = rng.integers(-2 ** 62, 2 ** 62, size=base.shape, dtype=np.int64)
            ndv = np.int64(np.iinfo(np.int64).max)
        else:                  # non-finite cells in some labels
            values = (base * 10).astype("f4")
            values[rng.random(base.shape) < 2e-5] = np.inf
            values[rng.random(base.shape) < 2e-5] = -np.inf
            values[rng.random(base.shape) < 2e-5] = np.nan
            ndv = np.float32(-9999)
        values[rng.random(base.shape) < 0.05] = ndv
        check("wide %d" % it, labels, n, values, ndv)

print("LAZY ok", cases)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import print_routes  # noqa: E402
print_routes("zonal")
