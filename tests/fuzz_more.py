"""Many more fuzz seeds than the test suite runs (GPU box): python tests/fuzz_more.py [n] [first seed].  Lives under tests/
because it checks against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_util import REQ, describe, oracle_eval, random_view, unsigned_zero
from helpers import assert_result_equal
from dask_geomodeling_amd import raster
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd import plan as _plan
runtime.set_device(0)
bad = skipped = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000  # first seed
for seed in range(N):
    rng = np.random.default_rng(BASE + seed)
    # (tilers over element-wise stores: odd seeds tile by tile -- the batched launch into the mosaic --, even seeds the
    #  product's own rule, one request for small tiles: plan._whole_instead_of_tiles)
    _plan._BATCHED_TILE_CELLS = 0 if seed % 2 else 1 << 22
    _plan._STENCILS_WHOLE = seed % 4 < 2  # (likewise the one-request rule for tilers over Dilate / MovingMax / HillShade)
    memory = seed % 3 == 0
    view = random_view(rng, "x%d" % seed, depth=int(rng.integers(2, 6)), memory=memory)
    try:
        if memory:
            whole = view.get_data(**REQ)
            tiled = raster.RasterTiler(view, [int(rng.integers(5, 30)), int(rng.integers(5, 20))]).get_data(**REQ)
            assert_result_equal(unsigned_zero(tiled, view), unsigned_zero(whole, view), 0, "tiled")
        else:
            with np.errstate(all="ignore"):
                want = unsigned_zero(oracle_eval(view), view)
            assert_result_equal(unsigned_zero(view.get_data(**REQ), view), want, 0, "fused")
    except NotImplementedError:
        skipped += 1
    except Exception as e:  # noqa
        bad += 1
        print("SEED", seed, type(e).__name__, str(e)[:160])
        print("   ", describe(view)[:400])
        if bad > 40:
            break
print("done: %d views, %d skipped, %d failures" % (N, skipped, bad))
