"""Differential fuzzing of the host side (Blocks -> fusion pass -> planner / compiler) on the CPU:
random views evaluated through get_data with the NumPy interpreter of the programs (cpu_vm) against
the op-by-op oracle evaluation."""
import numpy as np
import pytest

from fuzz_util import REQ, describe, oracle_eval, random_view, unsigned_zero
from helpers import assert_result_equal


@pytest.mark.parametrize("seed", range(120))
def test_random_views_fused_vs_oracle(cpu_vm, monkeypatch, seed):
    from dask_geomodeling_amd.raster import devops
    monkeypatch.setattr(devops, "all_equal", lambda values, ndv: bool(np.all(np.asarray(values) == ndv)))
    rng = np.random.default_rng(seed)
    view = random_view(rng, "c%d" % seed)
    with np.errstate(all="ignore"):
        want = oracle_eval(view)
    from dask_geomodeling_amd.hip.runtime import GmHipError
    try:
        got = view.get_data(**REQ)
    except NotImplementedError as e:
        pytest.skip("not on the device path: %s" % e)
    except GmHipError as e:
        if "no HIP device" not in str(e):
            raise
        pytest.skip("this view needs a stand-alone kernel (covered by the GPU fuzz test)")
    assert_result_equal(unsigned_zero(got), unsigned_zero(want), 0, describe(view))
