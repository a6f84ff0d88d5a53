"""Run under an interpreter that HAS dask (tests/test_real_dask_cpu.py starts it): the product's graphs --
fused, cut into stages, batched, tiled -- evaluated by dask's own schedulers (core/graphs.py:42-49 of the
reference hands the graph to whatever scheduler dask is configured with):

    synchronous   dask.get
    threads       dask.threaded.get           (process bodies from a thread pool)
    processes     dask.multiprocessing.get    (graph optimised by dask -- cull + fuse --, tasks and results
                                               shipped through cloudpickle)

against the op-by-op oracle.  The HIP launch is replaced by the oracle's NumPy interpreter of gm_program (no
GPU here; the same test double as the cpu_vm fixture), installed in every worker process as well."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def install_cpu_vm():
    """The cpu_vm fixture of conftest.py without pytest (also the initializer of the worker processes)."""
    from dask_geomodeling_amd.hip import program, runtime
    from dask_geomodeling_amd.raster import devops, tables
    from oracle import vm_numpy
    runtime.require_device = lambda: None
    tables._PREFER_PROGRAM = True
    devops._all_false = lambda tile: not np.any(np.asarray(tile))
    devops.as_device = lambda v: np.asarray(v)
    devops.all_equal = lambda values, ndv: bool(np.all(np.asarray(values) == ndv))
    program.as_device = lambda v: np.asarray(v)
    program.as_device_view = lambda v: np.asarray(v)
    program.launch = vm_numpy.launch
    program.launch_batch = lambda compiled, tiles, shape, outs=None: [vm_numpy.launch(compiled, t, shape) for t in tiles]


def main():
    import dask
    from concurrent.futures import ProcessPoolExecutor
    install_cpu_vm()
    from blocks_util import build_view, leaves_for
    from fuzz_util import REQ, describe, oracle_eval, random_view, unsigned_zero
    from helpers import assert_result_equal, load_family
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.fuse import BatchedTask, FusedTask, fuse_graph
    from dask_geomodeling_amd.hip.runtime import GmHipError

    views = []
    for seed in range(60):
        view = random_view(np.random.default_rng(90_000 + seed), "d%d" % seed)
        with np.errstate(all="ignore"):
            views.append((describe(view), view, REQ, unsigned_zero(oracle_eval(view), view)))
    def uses(expr, name):
        return isinstance(expr, list) and (expr[0] == name or any(uses(x, name) for x in expr[1:]))

    tolerance = {}  # (Power <= 1 ulp, Exp / Log <= 3 ulp against NumPy's float32 loops; chains of them 16)
    for case in load_family("views"):
        view = build_view(case["expr"], leaves_for(case, kind="array"))
        hits = [ulp for name, ulp in (("Power", 1), ("Exp", 3), ("Log", 3), ("Log10", 3)) if uses(case["expr"], name)]
        tolerance[case["id"]] = 16 if len(hits) > 1 else max(hits or [0])
        views.append((case["id"], view, case["request"], case["expect"]))
    report = {"dask": dask.__version__, "numpy": np.__version__, "schedulers": {}}
    pool = ProcessPoolExecutor(max_workers=2, initializer=install_cpu_vm)
    fused_tasks = batched_tasks = 0
    for name, settings in (("synchronous", {"scheduler": "synchronous"}), ("threads", {"scheduler": "threads"}),
                           ("processes", {"scheduler": "processes", "pool": pool})):
        checked = skipped = 0
        from dask_geomodeling_amd import template
        stats_before = dict(template.stats)
        with dask.config.set(**settings):
            for what, view, request, want in views:
                try:
                    with np.errstate(all="ignore"):
                        got = view.get_data(**request)
                except (NotImplementedError, GmHipError):
                    skipped += 1  # stand-alone kernels need the device
                    continue
                ulp = tolerance.get(what, 0)
                assert_result_equal(unsigned_zero(got, view), unsigned_zero(want, view), ulp, "%s under %s" % (what, name))
                checked += 1
            # structurally identical views evaluated together: one BatchedTask (what a RasterTiler's tiles
            # become) travels through dask as well
            from dask_geomodeling_amd.core.graphs import get_data_many
            from blocks_util import ArrayLeaf
            rng = np.random.default_rng(5)
            same, wants = [], []
            for k in range(4):
                fill = np.float32(np.finfo("f4").max)
                for tag in "ab":
                    v = (rng.random((1, 24, 40)) * 100).astype("f4")
                    v[0, k::5, ::7] = fill
                    ArrayLeaf.ARRAYS["dk/%s%d" % (tag, k)] = {"values": v, "no_data_value": fill}
                A, B = ArrayLeaf("dk/a%d" % k, "<f4", fill), ArrayLeaf("dk/b%d" % k, "<f4", fill)
                view = raster.Clip((A + B) * 2.5 - B + 1.5, A)
                same.append(view)
                wants.append(oracle_eval(view))
            for got, want in zip(get_data_many(same, REQ), wants):
                assert_result_equal(got, want, 0, "batched views under %s" % name)
            graph, keys = {}, []
            for view in same:
                graph, key = view.get_compute_graph(cached_compute_graph=graph, **REQ)
                keys.append(key)
            for task in fuse_graph(graph, keys).values():
                if isinstance(task, tuple) and task:
                    fused_tasks += isinstance(task[0], FusedTask)
                    batched_tasks += isinstance(task[0], BatchedTask)
            from dask_geomodeling_amd.core.graphs import compute
            report.setdefault("task_ran_in_another_process", {})[name] = compute({"pid": (os.getpid,)}, "pid") != os.getpid()
        # the graph every get_data above handed to dask came from the views' templates (template.py): built under
        # the first scheduler, reused -- same FusedTask, same compiled programs -- under the following ones
        report["schedulers"][name] = {"views_checked": checked, "views_skipped": skipped,
                                      "templates_built": template.stats["built"] - stats_before["built"],
                                      "template_hits": template.stats["hits"] - stats_before["hits"],
                                      "full_graph_builds": template.stats["full"] - stats_before["full"]}
    pool.shutdown()
    report["fused_tasks_seen"], report["batched_tasks_seen"] = fused_tasks, batched_tasks
    print("DASK_REPORT " + json.dumps(report))


if __name__ == "__main__":
    main()
