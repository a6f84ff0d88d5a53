"""The bound behind the float32 certificate of the zonal sums (csrc/gm_zonal.hip: zonal_chunk_combine_kernel), checked on the host
against exact rational arithmetic: for cells x_1 .. x_n with exact partial sums X_k, np.bincount's sequential float64 sum s_n
satisfies |s_n - X_n| <= 1.000001 u sum_k |X_k| (u = 2^-53), and any float64 summation tree S of the same cells satisfies
|S - X_n| <= u x (the sum of the magnitudes of its additions' results).  From the two: whenever (float32)(S - E) == (float32)(S + E)
with E the sum of both bounds, that float32 is (float32)s_n -- the statement the device kernels rest on.  No GPU here: the
kernels' own parity is tests/zonal_lazy_driver.py."""
from fractions import Fraction

import numpy as np
import pytest

U = 2.0 ** -53


def _sequential(x):
    """np.bincount's order: one float64 addition per cell, from 0.0 (np.cumsum adds in order too: its last element)."""
    return float(np.cumsum(np.asarray(x, dtype=np.float64))[-1])


def _exact_prefix_magnitudes(x):
    total, weight = Fraction(0), Fraction(0)
    for v in x:
        total += Fraction(float(v))
        weight += abs(total)
    return total, weight


def _tree(x, rng):
    """A random summation tree over the cells in float64 -> (its result, the sum of the magnitudes of its additions' results)."""
    parts = [float(v) for v in x]
    cost = 0.0
    while len(parts) > 1:
        i = int(rng.integers(0, len(parts) - 1))
        merged = parts[i] + parts[i + 1]          # (adjacent ranges, like the device's trees -- any tree obeys the bound)
        cost += abs(merged)
        parts[i:i + 2] = [merged]
    return parts[0], cost


def _cases(rng):
    for it in range(160):
        n = int(rng.integers(1, 1500))
        kind = it % 8
        if kind == 0:
            x = rng.standard_normal(n) * 100
        elif kind == 1:      # wide exponents
            x = rng.standard_normal(n) * np.exp2(rng.integers(-40, 40, size=n).astype(np.float64))
        elif kind == 2:      # positive, elevation-like
            x = np.abs(rng.standard_normal(n)) * 50 + 3
        elif kind == 3:      # a float32 tie and cells too fine to count
            x = np.zeros(n)
            x[0], x[-1] = 1.0, 2.0 ** -24
            x[1:-1] = rng.choice([-1.0, 1.0]) * 2.0 ** -float(rng.integers(30, 70))
        elif kind == 4:      # pairs that cancel, shuffled
            half = rng.standard_normal(max(n // 2, 1)) * 1e3
            x = rng.permutation(np.concatenate([half, -half]))
        elif kind == 5:      # float32 cells (what a float32 raster holds)
            x = (rng.standard_normal(n) * 10 ** float(rng.integers(-3, 6))).astype(np.float32).astype(np.float64)
        elif kind == 6:      # large integers
            x = rng.integers(-2 ** 50, 2 ** 50, size=n).astype(np.float64)
        else:                # a big cell among tiny ones
            x = rng.standard_normal(n) * 1e-9
            x[int(rng.integers(0, n))] = 1e9
        yield it, np.asarray(x, dtype=np.float64)


def test_the_sequential_sum_lies_within_u_times_the_partial_sums_of_the_exact_sum():
    rng = np.random.default_rng(20260)
    for it, x in _cases(rng):
        exact, weight = _exact_prefix_magnitudes(x)
        s_n = _sequential(x)
        assert abs(Fraction(s_n) - exact) <= Fraction(1.000001) * Fraction(U) * weight, (it, len(x))


def test_any_summation_tree_lies_within_u_times_the_results_of_its_additions():
    rng = np.random.default_rng(20261)
    for it, x in _cases(rng):
        exact, _ = _exact_prefix_magnitudes(x)
        for _ in range(3):
            S, cost = _tree(x, rng)
            assert abs(Fraction(S) - exact) <= Fraction(1.000001) * Fraction(U) * Fraction(cost), (it, len(x))


@pytest.mark.parametrize("mean", [False, True])
def test_the_certificate_never_accepts_a_float32_that_is_not_bincounts(mean):
    """The decision of the kernels, restated: E from the two bounds (with the kernels' slack factors), accept when both ends round to
    one float32.  On data made to sit on float32 ties and next to them the test must refuse or be right -- and on ordinary data it must
    accept nearly always (or the certificate would be useless)."""
    rng = np.random.default_rng(20262 + mean)
    accepted = refused = 0
    for it, x in _cases(rng):
        n = len(x)
        _, weight = _exact_prefix_magnitudes(x)
        S, cost = _tree(x, rng)
        tau = 1.5 * U * cost
        E = 1.01 * U * (float(weight) + 4.0 * float(np.abs(x).sum()) + (2.0 * n + 4.0) * tau) + tau + (1e-300 if np.any(x != 0) else 0.0)
        below, above = S - E, S + E
        truth = _sequential(x)
        if mean:
            below, above, truth = below / n, above / n, truth / n
        with np.errstate(over="ignore"):
            lo, hi, want = np.float32(below), np.float32(above), np.float32(truth)
        if lo.tobytes() == hi.tobytes() and np.isfinite(lo):
            accepted += 1
            assert lo.tobytes() == want.tobytes(), (it, n, S, E, lo, want)
        else:
            refused += 1
    assert accepted >= 100 and refused >= 10, (accepted, refused)   # (the ties and the cancelling pairs are refused; noise is not)
