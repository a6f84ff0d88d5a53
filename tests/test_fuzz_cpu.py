"""Differential fuzzing of the host side (Blocks -> fusion pass -> planner / compiler) on the CPU:
random views evaluated through get_data with the NumPy interpreter of the programs (cpu_vm) against
the op-by-op oracle evaluation."""
import numpy as np
import pytest

from fuzz_util import REQ, describe, oracle_eval, random_view, unsigned_zero
from helpers import assert_result_equal


@pytest.mark.parametrize("seed", range(400))
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
    assert_result_equal(unsigned_zero(got, view), unsigned_zero(want, view), 0, describe(view))


@pytest.mark.parametrize("seed", range(40))
def test_views_over_nine_to_sixteen_leaves(cpu_vm, monkeypatch, seed):
    """The host side of programs with 9-16 inputs (round 6): fusion, the compiler's operand encoding and the NumPy interpreter of
    gm_program against the op-by-op oracle (the device's side of it: tests/test_gpu_fuzz.py)."""
    from dask_geomodeling_amd.raster import devops
    from dask_geomodeling_amd.hip.runtime import GmHipError
    from fuzz_util import wide_view
    monkeypatch.setattr(devops, "all_equal", lambda values, ndv: bool(np.all(np.asarray(values) == ndv)))
    rng = np.random.default_rng(70_000 + seed)
    view = wide_view(rng, "wc%d" % seed, int(rng.integers(9, 17)))
    with np.errstate(all="ignore"):
        want = oracle_eval(view)
    try:
        got = view.get_data(**REQ)
    except GmHipError as e:
        if "no HIP device" not in str(e):
            raise
        pytest.skip("this view needs a stand-alone kernel (covered by the GPU fuzz test)")
    assert_result_equal(unsigned_zero(got, view), unsigned_zero(want, view), 0, describe(view))


@pytest.mark.parametrize("seed", range(40))
def test_plans_equal_the_graph_path(cpu_vm, seed):
    """get_data through a cached ViewPlan (plan.py) gives what the graph path gives, on first use and on reuse."""
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan
    rng = np.random.default_rng(50_000 + seed)
    view = random_view(rng, "p%d" % seed)
    with np.errstate(all="ignore"):
        try:
            with gmconfig.config.set({"geomodeling.plans": False}):
                want = view.get_data(**REQ)
        except NotImplementedError:
            pytest.skip("not on the device path")
        first = view.get_data(**REQ)
        again = view.get_data(**REQ)
    assert_result_equal(first, want, 0, "first use " + describe(view))
    assert_result_equal(again, want, 0, "reuse " + describe(view))
    plans = view.__dict__.get("_gm_plans", {})
    assert len(plans) <= 1
