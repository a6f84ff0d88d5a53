"""The product's graphs under a REAL dask (the reference evaluates with whatever scheduler dask is configured
with, core/graphs.py:42-49).  The system interpreter of the GPU image has no dask; the build container carries
one in /opt/conda (dask 2021.10, NumPy 1.26): the driver script runs there.  Skipped where no interpreter with
dask exists."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CANDIDATES = ["/opt/conda/bin/python3.9", "/opt/conda/bin/python"]


def _python_with_dask():
    for exe in CANDIDATES:
        if os.path.exists(exe):
            probe = subprocess.run([exe, "-c", "import dask, numpy, cloudpickle"], capture_output=True)
            if probe.returncode == 0:
                return exe
    return None


@pytest.mark.timeout(900)
def test_views_under_dask_schedulers():
    exe = _python_with_dask()
    if exe is None:
        pytest.skip("no interpreter with dask on this box")
    env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONWARNINGS="ignore")
    system_cxx = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"  # conda ships an older libstdc++ than libgmhip.so needs
    if os.path.exists(system_cxx):
        env["LD_PRELOAD"] = system_cxx
    res = subprocess.run([exe, os.path.join(HERE, "dask_driver.py")], capture_output=True, text=True, timeout=850, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("DASK_REPORT ")][-1]
    report = json.loads(line[len("DASK_REPORT "):])
    assert set(report["schedulers"]) == {"synchronous", "threads", "processes"}
    for name, counts in report["schedulers"].items():
        assert counts["views_checked"] >= 60, (name, counts)
    assert report["batched_tasks_seen"] >= 3
    # graph templates: built once (under the first scheduler), hit under threads and processes -- no view pays the
    # full get_compute_graph + fuse_graph again
    first, threads, processes = (report["schedulers"][k] for k in ("synchronous", "threads", "processes"))
    assert first["templates_built"] >= 50
    for later in (threads, processes):
        assert later["templates_built"] == 0 and later["template_hits"] >= first["templates_built"], report["schedulers"]
        assert later["full_graph_builds"] <= first["full_graph_builds"]
    assert report["task_ran_in_another_process"] == {"synchronous": False, "threads": False, "processes": True}
    print(report)
