"""Shared comparison helpers for the parity tests."""
import os

import numpy as np

from golden import codec

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_family(name):
    return codec.load(os.path.join(GOLDEN_DIR, name))


def ulp_distance(a, b):
    """Element-wise distance in units in the last place (same float dtype)."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.dtype == b.dtype and a.dtype.kind == "f"
    it = {4: np.int32, 8: np.int64}[a.dtype.itemsize]
    ia = a.view(it).astype(np.int64)
    ib = b.view(it).astype(np.int64)
    sign = np.int64(np.iinfo(it).min)
    ia = np.where(ia < 0, sign - ia, ia)
    ib = np.where(ib < 0, sign - ib, ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def assert_values_equal(actual, expected, max_ulp=0, what=""):
    actual = np.asarray(actual)
    expected = np.asarray(expected)
    assert actual.dtype == expected.dtype, "%s dtype %s != %s" % (what, actual.dtype, expected.dtype)
    assert actual.shape == expected.shape, "%s shape %s != %s" % (what, actual.shape, expected.shape)
    if actual.size == 0:
        return
    if expected.dtype.kind == "f":
        if max_ulp == 0:
            same = (actual.view("u%d" % actual.dtype.itemsize) == expected.view("u%d" % expected.dtype.itemsize))
            same |= np.isnan(actual) & np.isnan(expected)
            # +0.0 / -0.0 must match bit-for-bit as well
            assert same.all(), "%s: %d/%d elements differ (bit-exact), first at %s: %r vs %r" % (
                what, (~same).sum(), same.size, np.argwhere(~same)[0],
                actual[~same][0], expected[~same][0])
        else:
            d = ulp_distance(actual, expected)
            # non-finite placement must agree exactly
            assert (np.isfinite(actual) == np.isfinite(expected)).all(), what
            assert d.max() <= max_ulp, "%s: max ulp distance %d > %d" % (what, d.max(), max_ulp)
    else:
        same = actual == expected
        assert same.all(), "%s: %d/%d elements differ, first at %s: %r vs %r" % (
            what, (~same).sum(), same.size, np.argwhere(~same)[0], actual[~same][0], expected[~same][0])


def assert_ndv_equal(actual, expected, what=""):
    if expected is None:
        assert actual is None, "%s ndv %r != None" % (what, actual)
        return
    assert actual is not None, "%s ndv None != %r" % (what, expected)
    assert actual == expected or (actual != actual and expected != expected), \
        "%s ndv %r != %r" % (what, actual, expected)


def assert_result_equal(actual, expected, max_ulp=0, what=""):
    """Compare two raster responses (None | passthrough dict | values dict)."""
    if expected is None:
        assert actual is None, "%s: expected None, got %r" % (what, type(actual))
        return
    assert actual is not None, "%s: got None" % what
    if "values" not in expected:
        assert set(actual) == set(expected), what
        for k in expected:
            assert actual[k] == expected[k], what
        return
    assert_values_equal(actual["values"], expected["values"], max_ulp, what)
    assert_ndv_equal(actual.get("no_data_value"), expected.get("no_data_value"), what)


def print_routes(*families):
    """Last line of a by-hand driver: which kernel routes its calls took (runtime.route_stats), for the test that started it."""
    import json
    from dask_geomodeling_amd.hip import runtime
    print("ROUTES-TAKEN " + json.dumps({f: runtime.route_stats(f) for f in families}), flush=True)


def routes_of(stdout):
    """{family: {route: count}} from the ROUTES-TAKEN line of a driver's output."""
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("ROUTES-TAKEN ")]
    assert lines, "the driver printed no ROUTES-TAKEN line: %s" % stdout[-300:]
    return json.loads(lines[-1][len("ROUTES-TAKEN "):])
