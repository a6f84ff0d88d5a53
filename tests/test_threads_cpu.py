"""Thread- and process-safety of the host bookkeeping (no GPU): the configuration shim's look-up cache, the
per-view template / plan caches and their counters, the specialiser's worker state across fork()."""
import ctypes
import os
import threading

import numpy as np
import pytest

from dask_geomodeling_amd import config as gm_config
from dask_geomodeling_amd import raster, template


def _view():
    a = np.arange(64, dtype="f4").reshape(1, 8, 8)
    source = raster.MemorySource(a, 255.0, "EPSG:28992", 1.0, (0.0, 8.0))
    return raster.Add(source, 2.0) * 3.0


def _request(width=8, bbox=(0.0, 0.0, 8.0, 8.0)):
    return dict(mode="vals", bbox=bbox, width=width, height=8, projection="EPSG:28992")


@pytest.mark.parametrize("width", [8, 8.0, True])
def test_template_keys_are_the_graph_keys_for_every_spelling_of_a_request(width):
    """8, 8.0 and True are equal and hash alike, but core.graphs._request_token hashes their printed form: each
    spelling gets a template (and printed keys) of its own, so the fused graph carries get_compute_graph's keys."""
    view = _view()
    template.fused_compute_graph(view, _request(8))  # the template of the int spelling exists first
    for bbox in ((0.0, 0.0, 8.0, 8.0), (1.0, 1.0, 9.0, 9.0)):
        request = _request(width, bbox)
        graph, name = template.fused_compute_graph(view, request)
        full, full_name = view.get_compute_graph(**request)
        assert name == full_name
        assert set(graph) <= set(full)


def test_templates_are_built_once_under_concurrent_requests():
    view = _view()
    before = dict(template.stats)
    barrier = threading.Barrier(8)
    errors = []

    def work(i):
        try:
            barrier.wait()
            for j in range(50):
                template.fused_compute_graph(view, _request(8, (float(j), 0.0, float(j) + 8.0, 8.0)))
        except Exception as e:  # pragma: no cover - reported below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert template.stats["built"] - before["built"] == 1
    assert template.stats["hits"] - before["hits"] == 8 * 50  # no lost update
    assert len(view.__dict__["_gm_templates"]) == 1


def test_config_lookup_cache_does_not_keep_a_value_that_a_concurrent_set_replaced():
    shim = gm_config.config
    if not hasattr(shim, "_flat"):
        pytest.skip("dask.config is in use, not the shim")
    key = "geomodeling.test-race-key"
    with shim.set({key: "old"}):
        shim._flat.clear()
        original = shim._lookup

        def racing_lookup(k, default):
            value = original(k, default)     # resolved the old value ...
            if k == key:
                shim.set._assign(key, "new")  # ... and a set() of another thread lands before it is remembered
            return value

        shim._lookup = racing_lookup
        try:
            assert shim.get(key) == "old"     # the call that raced may still see the old value
        finally:
            del shim._lookup
        assert shim.get(key) == "new"         # but nobody after it does
        assert shim.get(key) == "new"


def test_spec_wait_returns_in_a_forked_child():
    """pthread_atfork child handler of csrc/gm_spec.hip: a child has no worker thread; gm_spec_wait() must return."""
    from dask_geomodeling_amd.hip import runtime
    lib = runtime.lib()
    lib.gm_spec_wait.restype = ctypes.c_int
    pid = os.fork()
    if pid == 0:
        rc = 1
        try:
            rc = 0 if lib.gm_spec_wait() == 0 else 2
        finally:
            os._exit(rc)
    deadline = 20.0
    import time
    t0 = time.time()
    while time.time() - t0 < deadline:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
            return
        time.sleep(0.05)
    os.kill(pid, 9)
    os.waitpid(pid, 0)
    raise AssertionError("gm_spec_wait() hung in the forked child")


def test_tiler_over_a_stencil_on_a_source_is_one_graph_task(monkeypatch):
    """template.fused_compute_graph of a RasterTiler over a stencil Block on a resident source: ONE MosaicOfTiles task over
    the tiles' source windows (the stencil writes every tile into the mosaic); over a computed store: a task per tile.
    (Where a mosaic of tiles is the whole -- Dilate, MovingMax, HillShade with their margins inside the source -- the graph
    holds the store's graph for the whole box and one mosaic_of_whole task instead: plan._whole_instead_of_tiles.)"""
    from dask_geomodeling_amd import plan
    from dask_geomodeling_amd.fuse import MosaicOfTiles
    a = np.arange(64 * 64, dtype="f4").reshape(1, 64, 64)
    source = raster.MemorySource(a, 255.0, "EPSG:28992", 1.0, (0.0, 64.0))
    request = dict(mode="vals", bbox=(8.0, 8.0, 40.0, 40.0), width=32, height=32, projection="EPSG:28992")
    for block in (raster.HillShade(source), raster.MovingMax(source, 3), raster.Dilate(source, [3.0]), raster.Smooth(source, 2.0)):
        graph, name = template.fused_compute_graph(raster.RasterTiler(block, 16), request)
        assert (graph[name][0] is template.mosaic_of_whole) == (not isinstance(block, raster.Smooth)), type(block).__name__
    monkeypatch.setattr(plan, "_STENCILS_WHOLE", False)
    for block in (raster.HillShade(source), raster.MovingMax(source, 3), raster.Dilate(source, [3.0]), raster.Smooth(source, 2.0)):
        tiler = raster.RasterTiler(block, 16)
        graph, name = template.fused_compute_graph(tiler, request)
        task = graph[name]
        assert isinstance(task[0], MosaicOfTiles), type(block).__name__
        kwargs, counts, args = task[1], task[2], task[3:]
        assert kwargs["count_xy"] == (2, 2) and counts == (2, 2, 2, 2) and len(args) == 8
        assert all(isinstance(args[i], str) and args[i] in graph for i in range(0, 8, 2))  # the tiles' source windows
        full, full_name = tiler.get_compute_graph(**request)
        assert name == full_name
    computed = raster.RasterTiler(raster.HillShade(source * 2), 16)
    graph, name = template.fused_compute_graph(computed, request)
    assert not isinstance(graph[name][0], MosaicOfTiles) and len(graph[name]) == 2 + 4
