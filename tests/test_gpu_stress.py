"""The by-hand stress drivers inside the suite (round 6), fixed seeds, a process each: random rasters / label layouts /
statistics through the zonal kernels (every tenth raster up to 4 M cells), random shapes / dtypes through the stencils, and a
slice of the view fuzzer's larger campaign -- against the oracle.  Once more, shorter, under GM_DEBUG_GUARD=1 (poisoned red
zones around every device block, checked at every free and sync: csrc/gm_runtime.hip): a kernel that writes past
a buffer fails the run even where the answer it leaves behind is right."""
import os
import subprocess
import sys

import pytest

from helpers import routes_of

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, args, env=None, timeout=600):
    res = subprocess.run([sys.executable, os.path.join(HERE, script)] + [str(a) for a in args], capture_output=True, text=True,
                         env=dict(os.environ, **(env or {})), timeout=timeout)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-2500:])
    return res.stdout


@pytest.mark.timeout(900)
def test_zonal_stress_driver(gpu):
    out = _run("zonal_stress.py", [120, 4711])
    assert "done 120 failures 0" in out, out[-2000:]
    took = routes_of(out)["zonal"]
    # the campaign reaches every family of routes (a threshold that moves would show here)
    # (the chunk records have drivers of their own in tests/test_gpu_parity.py: this campaign's sums are answered before them)
    for route in ("direct", "lds_tables", "lds_lazy", "parallel_lazy", "sort_partition", "sort_radix", "select", "raw_select", "segment_sort"):
        assert took[route] > 0, (route, took)


@pytest.mark.timeout(900)
def test_zonal_stress_driver_on_the_older_routes(gpu):
    """The same campaign (shorter) with the round-6 routes switched off: count / min / max through the wave-slot kernel instead of
    the workgroup tables, medians under many labels through the two sorts instead of the segment sort, every sum in bincount's
    order or certified exact (no float32 certificate: the scattered labels' sums through zonal_tables_exact_kernel), the results
    into a device buffer and copied back by the caller (gm_zonal_statistic instead of gm_zonal_statistic_host), medians under a
    few labels through the partition instead of straight off the raster."""
    out = _run("zonal_stress.py", [120, 4711], env={"GM_ZONAL_TABLES": "0", "GM_ZONAL_SEGSORT": "0", "GM_ZONAL_LAZY": "0",
                                                    "GM_ZONAL_HOST": "0", "GM_ZONAL_RAW_SELECT": "0"})
    assert "done 120 failures 0" in out, out[-2000:]
    took = routes_of(out)["zonal"]
    assert took["lds_tables"] == 0 and took["segment_sort"] == 0 and took["direct"] > 0 and took["lexsort"] > 0, took
    assert took["lds_lazy"] == 0 and took["exact_lazy_labels"] == 0 and took["lds_exact"] > 0 and took["ordered_sums"] > 0, took
    assert took["raw_select"] == 0 and took["select"] > 0, took


@pytest.mark.timeout(900)
def test_stencil_stress_driver(gpu):
    out = _run("stencil_stress.py", [100, 2026])
    assert "done 100 failures 0" in out, out[-2000:]
    took = routes_of(out)["stencils"]
    for route in ("hillshade_pairs", "hillshade_single", "movingmax_3x3", "movingmax_stack", "movingmax_columns", "gauss_one_launch"):
        assert took[route] > 0, (route, took)


@pytest.mark.timeout(900)
def test_view_fuzzer_slice(gpu):
    out = _run("fuzz_more.py", [300, 50000])
    assert "done: 300 views" in out and " 0 failures" in out, out[-2000:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,args,done", [("zonal_stress.py", [40, 99], "done 40 failures 0"),
                                               ("stencil_stress.py", [30, 7], "done 30 failures 0")], ids=["zonal", "stencils"])
def test_stress_drivers_under_the_guard_zones(gpu, script, args, done):
    out = _run(script, args, env={"GM_DEBUG_GUARD": "1"})
    assert done in out, out[-2000:]
    assert "guard violations 0 guard mode 1" in out, out[-2000:]


@pytest.mark.timeout(300)
def test_the_guard_zones_catch_a_write_past_a_block(gpu):
    """The debug mode itself: a device block of 1000 bytes, 1016 bytes written (gm_memset past the end) -> gm_sync fails with
    GM_ERR_GUARD, the violation is counted and described; the same without the mode goes unnoticed (the pool's rounding)."""
    code = (
        "import sys, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "from dask_geomodeling_amd.hip import runtime\n"
        "runtime.set_device(0)\n"
        "L = runtime.lib()\n"
        "p = runtime.malloc(1000)\n"
        "runtime.check(L.gm_memset(ctypes.c_void_p(p), 1, 1016), 'gm_memset')\n"
        "try:\n"
        "    runtime.sync()\n"
        "    print('SYNC ok')\n"
        "except runtime.GmHipError as e:\n"
        "    print('SYNC failed:', e)\n"
        "n = ctypes.c_ulonglong(0)\n"
        "L.gm_guard_check(ctypes.byref(n))\n"
        "print('violations', n.value)\n"
        "if L.gm_guard_enabled():\n"
        "    q = runtime.malloc(4096)\n"
        "    runtime.check(L.gm_memset(ctypes.c_void_p(q - 8), 2, 8), 'gm_memset')\n"
        "    print('free rc', L.gm_free(ctypes.c_void_p(q)))\n"
    ) % os.path.dirname(HERE)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GM_DEBUG_GUARD="1"), timeout=200)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "SYNC failed" in res.stdout and "9005" in res.stdout and "violations 1" in res.stdout, res.stdout
    assert "PAST ITS END: 16 bytes differ, the first 0" in res.stderr, res.stderr[-1500:]
    assert "BEFORE ITS START" in res.stderr and "free rc 9005" in res.stdout, (res.stdout, res.stderr[-1500:])
    env = {k: v for k, v in os.environ.items() if k != "GM_DEBUG_GUARD"}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=200)
    assert "SYNC ok" in res.stdout and "violations 0" in res.stdout, (res.stdout, res.stderr[-800:])
