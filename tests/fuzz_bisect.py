"""Where a failing seed of tests/fuzz_temporal.py goes wrong (GPU box): python tests/fuzz_bisect.py <seed> [HxW] [first seed]
walks the view's own requests down and compares every Block's device answer with the walking oracle's: the deepest Block
that differs while all of its sources agree is printed with its arguments."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_util import describe
from oracle_walk import comparable, random_requests, random_temporal_view, walk
from dask_geomodeling_amd import config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.base import RasterBlock
runtime.set_device(0)
seed = int(sys.argv[1])
SHAPE = tuple(int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (24, 40)
BASE = int(sys.argv[3]) if len(sys.argv) > 3 else 90_000


def same(a, b):
    if a is None or b is None or not isinstance(a, dict) or not isinstance(b, dict):
        return (a is None) == (b is None) and (a is None or isinstance(a, np.ndarray) or a == b)
    if "values" not in b:
        return a == b
    x, y = np.asarray(a["values"]), np.asarray(b["values"])
    if not (x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True)):
        return False
    if x.dtype.kind == "f" and not np.array_equal(np.signbit(x), np.signbit(y)):
        return False  # (the sign of a zero counts)
    return a.get("no_data_value") == b.get("no_data_value")


def answer(block, request, how):
    try:
        return how(block, request), None
    except Exception as e:  # noqa
        return None, "%s: %s" % (type(e).__name__, str(e)[:100])


def descend(block, request, depth=0):
    got, gerr = answer(block, request, lambda b, r: b.get_data(**r))
    want, werr = answer(block, request, walk)
    agree = (gerr is None) == (werr is None) and (gerr is not None or same(got, want))
    if agree:
        return True
    culprit_below = False
    for source, sub in block.get_sources_and_requests(**request):
        if isinstance(source, RasterBlock) and sub is not None:
            if not descend(source, sub, depth + 1):
                culprit_below = True
    if not culprit_below:
        print("FIRST DIFFERENCE at", describe(block)[:700])
        print("   request:", {k: str(v) for k, v in request.items()})
        def brief(r, e):
            if e:
                return e
            if r is None or not isinstance(r, dict) or "values" not in r:
                return repr(r)[:200]
            v = np.asarray(r["values"])
            return "values %s %s ndv %r, first cells %s" % (v.shape, v.dtype, r.get("no_data_value"), v.ravel()[:6])
        print("   device:", brief(got, gerr))
        print("   oracle:", brief(want, werr))
    return False


rng = np.random.default_rng(BASE + seed)
view = random_temporal_view(rng, depth=int(rng.integers(1, 5)), shape_yx=SHAPE)
with gmconfig.config.set({"geomodeling.plans": seed % 2 == 0}):
    for request in random_requests(rng, view, SHAPE):
        if not descend(view, request):
            break
    else:
        print("every request agrees")
