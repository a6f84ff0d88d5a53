"""Many more temporal fuzz seeds than the test suite runs (GPU box): python tests/fuzz_temporal.py [n] [HxW] [first seed].  Lives under
tests/ because it checks against the oracle (tests/oracle_walk.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_util import describe, unsigned_zero
from helpers import assert_result_equal
from oracle_walk import comparable, random_requests, random_temporal_view, walk, without_lossy_cells
from dask_geomodeling_amd import config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd import plan as _plan
runtime.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
SHAPE = tuple(int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (24, 40)  # e.g. 150x517
BASE = int(sys.argv[3]) if len(sys.argv) > 3 else 90_000  # first seed
bad = raised = asked = refused = 0
refusals = {}
for seed in range(N):
    rng = np.random.default_rng(BASE + seed)
    # (tilers over element-wise stores: odd seeds tile by tile -- the batched launch into the mosaic --, even seeds the
    #  product's own rule, one request for small tiles: plan._whole_instead_of_tiles)
    _plan._BATCHED_TILE_CELLS = 0 if seed % 2 else 1 << 22
    _plan._STENCILS_WHOLE = seed % 4 < 2  # (likewise the one-request rule for tilers over Dilate / MovingMax / HillShade)
    view = random_temporal_view(rng, depth=int(rng.integers(1, 5)), shape_yx=SHAPE)
    with gmconfig.config.set({"geomodeling.plans": seed % 2 == 0}):
        for request in random_requests(rng, view, SHAPE):
            asked += 1
            try:
                want = walk(view, request)
            except Exception as reference_error:
                raised += 1
                try:
                    view.get_data(**request)
                except Exception:
                    continue
                bad += 1
                print("SEED", seed, "the reference raises, the device path answers:", type(reference_error).__name__, str(reference_error)[:120])
                print("   ", describe(view)[:500])
                print("   ", {k: str(v) for k, v in request.items() if k in ("mode", "start", "stop", "bbox", "width", "height")})
                continue
            try:
                try:
                    got = view.get_data(**request)
                except NotImplementedError as refusal:  # (a refusal DESIGN.md section 9 lists: no answer is not a wrong answer)
                    refused += 1
                    why = str(refusal)[:110]
                    refusals[why] = refusals.get(why, 0) + 1
                    continue
                if isinstance(want, np.ndarray):
                    assert isinstance(got, np.ndarray) and got.shape == want.shape
                else:
                    assert_result_equal(*without_lossy_cells(comparable(got, view), comparable(want, view)), 0, "")
            except Exception as e:  # noqa
                bad += 1
                print("SEED", seed, type(e).__name__, str(e)[:200])
                if os.environ.get("GM_FUZZ_TRACE"):
                    import traceback
                    traceback.print_exc()
                    print("    request:", request)
                print("   ", describe(view)[:500])
                print("   ", {k: str(v) for k, v in request.items() if k in ("mode", "start", "stop")})
    if bad > 25:
        break
print("done: %d views, %d requests (%d of them raise in the reference too, %d refused by the device), %d failures"
      % (N, asked, raised, refused, bad))
for why, count in sorted(refusals.items(), key=lambda kv: -kv[1]):
    print("  refused %4d x  %s" % (count, why))
