"""The one-worker-per-GPU scheduler (dask_geomodeling_amd/shard.py) without a GPU: real worker processes, the HIP launch
replaced in every worker by the oracle's NumPy interpreter of gm_program (the cpu_vm test double, installed through the
workers' preload hook).  What is checked here is the scheduler -- dealing, shipping of views and arrays, jobs that are kept,
assembly of the host mosaic (raster/parallelize.py:108-124), failure of a worker -- not the kernels."""
import gc
import os
import sys
import time

import numpy as np
import pytest

from dask_geomodeling_amd import config, raster, shard

import shard_util
from shard_util import HostGrid

PRELOAD = "shard_util:install"
F32MAX = np.float32(np.finfo("f4").max)


@pytest.fixture
def pool2(cpu_vm, monkeypatch):
    """Two workers ("devices" 0 and 1) with the CPU double; arrays of 1 KiB and more take the shared-memory route."""
    monkeypatch.setattr(shard, "SHARE_FROM", 1024)
    with config.config.set({"geomodeling.devices": [0, 1], "geomodeling.worker-preload": PRELOAD}):
        yield shard.pool_for((0, 1))
    shard.shutdown()
    gc.collect()


def _leaves(seed=0, shape=(80, 96)):
    rng = np.random.default_rng(seed)
    a = (rng.random((1,) + shape) * 100).astype("f4")
    b = (rng.random((1,) + shape) * 100).astype("f4")
    a[0, ::7, ::5] = F32MAX
    return HostGrid(a, F32MAX), HostGrid(b, F32MAX)


def _request(shape=(80, 96)):
    return dict(mode="vals", bbox=(0.0, 0.0, float(shape[1]), float(shape[0])), width=shape[1], height=shape[0],
                projection="EPSG:28992")


def _chain(A, B):
    return raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)


def _bits(values):
    values = np.ascontiguousarray(np.asarray(values))
    return values.view("u%d" % values.dtype.itemsize)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("tile", [[32, 24], [96, 80], [40, 80], 7])
def test_sharded_mosaic_is_the_untiled_view(pool2, tile):
    """RasterTiler.get_data over two workers: tiles dealt t % 2, the mosaic assembled on the host (ragged last tiles too)
    == the view asked for the whole box in this process."""
    A, B = _leaves()
    view = _chain(A, B)
    with config.config.set({"geomodeling.devices": None}):
        whole = view.get_data(**_request())
    got = raster.RasterTiler(view, tile).get_data(**_request())
    assert isinstance(got["values"], np.ndarray) and got["values"].dtype == np.dtype("f4")
    assert got["no_data_value"] == whole["no_data_value"]
    assert np.array_equal(_bits(got["values"]), _bits(whole["values"]))


@pytest.mark.timeout(120)
def test_the_second_request_is_a_rerun_of_the_kept_job(pool2):
    A, B = _leaves(1)
    tiler = raster.RasterTiler(_chain(A, B), [32, 24])
    first = tiler.get_data(**_request())
    before = {w: r for w, r in pool2.request({0: ("ping",), 1: ("ping",)}).items()}
    again = tiler.get_data(**_request())
    after = pool2.request({0: ("ping",), 1: ("ping",)})
    assert np.array_equal(_bits(first["values"]), _bits(again["values"]))
    for w in (0, 1):
        assert after[w][2] == before[w][2] == 1      # one job per worker, not a new one per request
        assert after[w][0] == before[w][0] == pool2.workers[w].pid
    # another box is another job (its tiles are other windows)
    half = dict(_request(), bbox=(0.0, 0.0, 96.0, 40.0), height=40)
    tiler.get_data(**half)
    assert pool2.request({0: ("ping",)})[0][2] == 2
    # the arrays of the view travelled once: one shared-memory range per leaf, however many requests
    assert len(shard._shared_arrays) == 2


@pytest.mark.timeout(120)
def test_results_left_on_the_workers(pool2):
    """materialize off: the response's values are a handle; checksums are computed where the tiles are; np.asarray assembles."""
    A, B = _leaves(2)
    view = _chain(A, B)
    with config.config.set({"geomodeling.devices": None}):
        whole = view.get_data(**_request())
    with config.config.set({"geomodeling.materialize": False}):
        lazy = raster.RasterTiler(view, [32, 24]).get_data(**_request())
    handle = lazy["values"]
    assert isinstance(handle, shard.ShardedTiles) and handle.shape == (1, 80, 96) and handle.dtype == np.dtype("f4")
    sums = handle.checksums()
    assert sorted(sums) == list(range(12))   # 3 x 4 tiles (the last row of tiles is 8 cells high)
    assert len({w for w, _, _ in handle.units}) == 2
    assert [w for w, _, _ in handle.units] == [t % 2 for t in range(12)]
    assert np.array_equal(_bits(np.asarray(handle)), _bits(whole["values"]))
    run = handle._run
    del handle, lazy
    gc.collect()
    assert run in pool2.released  # (the workers hear about it with the next message)
    pool2.request({0: ("ping",), 1: ("ping",)})
    assert pool2.request({0: ("ping",)})[0][3] == 0


@pytest.mark.timeout(120)
def test_a_view_that_is_no_tiler_is_one_unit(pool2):
    A, B = _leaves(3)
    view = _chain(A, B)
    with config.config.set({"geomodeling.devices": None}):
        want = view.get_data(**_request())
    got = view.get_data(**_request())
    assert np.array_equal(_bits(got["values"]), _bits(want["values"]))
    assert view.get_data(**dict(_request(), mode="time")) == {"time": [A.period[0]]}


@pytest.mark.timeout(120)
def test_get_with_dask_calling_convention(pool2):
    """shard.get(graph, keys): tile keys of a RasterTiler end-point and plain keys dealt over the workers; nested key lists;
    results that are no rasters come back as they are; compute() uses it when geomodeling.devices is set."""
    from dask_geomodeling_amd.core.graphs import compute
    A, B = _leaves(4)
    view = _chain(A, B)
    tiler = raster.RasterTiler(view, [48, 40])
    graph, name = tiler.get_compute_graph(**_request())
    graph, plain = view.get_compute_graph(cached_compute_graph=graph, **_request())
    graph, when = view.get_compute_graph(cached_compute_graph=graph, **dict(_request(), mode="time"))
    graph["pid0"] = (shard_util.whoami,)
    graph["pid1"] = (shard_util.whoami,)
    mosaic, [whole, stamp], p0, p1 = shard.get(graph, [name, [plain, when], "pid0", "pid1"])
    with config.config.set({"geomodeling.devices": None}):
        want = view.get_data(**_request())
    assert np.array_equal(_bits(mosaic["values"]), _bits(want["values"]))
    assert np.array_equal(_bits(whole["values"]), _bits(want["values"]))
    assert stamp == {"time": [A.period[0]]}
    pids = {w.pid for w in pool2.workers}
    assert {p0["pid"], p1["pid"]} == pids and os.getpid() not in pids   # units are dealt in turn
    again = compute(graph, name)
    assert np.array_equal(_bits(again["values"]), _bits(want["values"]))


@pytest.mark.timeout(120)
def test_a_later_config_set_reaches_running_workers_and_their_kept_jobs(pool2):
    """The workers evaluate under the parent's geomodeling.* settings as they are at the request, not at the pool's start;
    jobs kept under the old settings are forgotten on both sides (the reference reads raster-limit, strict-file-paths, root
    at evaluation time)."""
    graph = {"a": (shard_util.setting, "geomodeling.raster-limit"), "b": (shard_util.setting, "geomodeling.raster-limit")}
    first = shard.get(graph, ["a", "b"])
    assert {r["value"] for r in first} == {config.get("geomodeling.raster-limit")}
    A, B = _leaves(7)
    tiler = raster.RasterTiler(_chain(A, B), [48, 40])
    want = tiler.get_data(**_request())
    assert len(pool2.jobs) == 1
    with config.config.set({"geomodeling.raster-limit": 4321, "geomodeling.fuse": False}):
        second = shard.get(graph, ["a", "b"])
        assert {r["value"] for r in second} == {4321}
        assert {r["pid"] for r in second} == {w.pid for w in pool2.workers}   # the same workers: no restart
        assert len(pool2.jobs) == 0
        unfused = tiler.get_data(**_request())   # (a new job, built without the fusion pass)
        assert np.array_equal(_bits(unfused["values"]), _bits(want["values"]))
        kept = [r[2] for r in pool2.request({0: ("ping",), 1: ("ping",)}).values()]
        assert kept == [1, 1]
    third = shard.get(graph, ["a", "b"])
    assert {r["value"] for r in third} == {config.get("geomodeling.raster-limit")}
    kept = [r[2] for r in pool2.request({0: ("ping",), 1: ("ping",)}).values()]
    assert kept == [0, 0]


@pytest.mark.timeout(120)
def test_workers_report_their_host_time(pool2):
    """("host_stats",): what bench.py's mosaic line reads per worker -- host seconds in run / rerun messages, their count."""
    A, B = _leaves(8)
    tiler = raster.RasterTiler(_chain(A, B), [48, 40])
    pool2.request({0: ("host_stats", True), 1: ("host_stats", True)})
    for _ in range(3):
        tiler.get_data(**_request())
    stats = pool2.request({0: ("host_stats",), 1: ("host_stats",)})
    for w in (0, 1):
        assert stats[w]["runs"] == 3 and 0.0 < stats[w]["wall_s"] < 60.0 and stats[w]["cpu_s"] >= 0.0
        assert stats[w]["pid"] == pool2.workers[w].pid


@pytest.mark.timeout(120)
def test_tiles_that_are_none_leave_the_fill_value(pool2):
    graph = {"t0": (shard_util.tile_of, 1.0), "t1": (lambda: None,), "t2": (shard_util.tile_of, 3.0),
             "t3": (shard_util.tile_of, 4.0)}
    kw = {"dtype": "float32", "fillvalue": -1.0, "shape_yx": (6, 10), "count_xy": (2, 2), "tilesize_xy": [5, 3]}
    graph["m"] = (raster.RasterTiler.process, kw, "t0", "t1", "t2", "t3")
    got = shard.get(graph, "m")
    want = np.full((1, 6, 10), -1.0, "f4")
    want[0, 3:6, 0:5] = 1.0      # tile 0 = (ix 0, iy 0): low y is the bottom rows
    want[0, 3:6, 5:10] = 3.0     # tile 2 = (ix 1, iy 0)
    want[0, 0:3, 5:10] = 4.0     # tile 3 = (ix 1, iy 1); tile 1 = (ix 0, iy 1) is None: fill
    assert np.array_equal(got["values"], want) and got["no_data_value"] == -1.0
    graph["n"] = (raster.RasterTiler.process, kw, "t1", "t1", "t1", "t1")
    assert shard.get(graph, ["n"]) == [None]


@pytest.mark.timeout(120)
def test_a_dying_worker_fails_the_request_within_seconds_and_is_named(pool2):
    graph = {"ok": (shard_util.tile_of, 1.0), "bad": (shard_util.die, 7)}
    t0 = time.time()
    with pytest.raises(shard.WorkerDied) as info:
        shard.get(graph, ["ok", "bad"])
    assert time.time() - t0 < 10.0
    text = str(info.value)
    assert "shard worker 1 (device 1, pid %d)" % pool2.workers[1].pid in text and "exit code 7" in text
    assert pool2.closed and all(w.proc.poll() is not None for w in pool2.workers)   # nobody is left behind
    # the next request starts a fresh pool
    got = shard.get(graph, ["ok"])
    assert got[0]["values"].shape == (1, 3, 5) and shard.pool_for((0, 1)) is not pool2


@pytest.mark.timeout(120)
def test_an_exception_in_a_worker_travels_with_its_type_and_the_workers_name(pool2):
    """A task that raises fails the request with the worker's exception type, name and traceback -- and nothing else: the
    workers stay (their resident sources and kept jobs with them), the next request is answered by the same processes."""
    graph = {"ok": (shard_util.tile_of, 1.0), "bad": (shard_util.boom, "no such thing")}
    pids = [w.pid for w in pool2.workers]
    with pytest.raises(ValueError) as info:
        shard.get(graph, ["ok", "bad"])
    text = str(info.value)
    assert "shard worker 1 (device 1" in text and "no such thing" in text and "worker traceback" in text
    assert not pool2.closed and [w.proc.poll() for w in pool2.workers] == [None, None]
    got = shard.get(graph, ["ok", "ok"])
    assert got[0]["values"].shape == (1, 3, 5) and [w.pid for w in shard.pool_for((0, 1)).workers] == pids
    # the same through get_data: a view whose evaluation raises in the worker (a refusal, say) leaves no job behind
    A, B = _leaves(7)
    view = raster.RasterTiler(_chain(A, B), [32, 24])
    with pytest.raises(AssertionError):   # (cells of 95 / 96 units: HostGrid serves aligned windows only, and says so in the worker)
        view.get_data(**dict(_request(), bbox=(0.0, 0.0, 95.0, 80.0)))
    assert not pool2.closed and pool2.request({0: ("ping",), 1: ("ping",)})[0][2] == 0   # no kept job
    assert np.array_equal(_bits(view.get_data(**_request())["values"]),
                          _bits(_chain(A, B).get_data(**_request())["values"]))


@pytest.mark.timeout(120)
def test_many_views_at_once_and_a_single_view_left_on_its_worker(pool2):
    """get_data_many under geomodeling.devices hands the merged graph to shard.get (its units dealt over the workers); a view
    that is no tiler, not materialised, is a one-unit ShardedTiles."""
    from dask_geomodeling_amd.core.graphs import get_data_many
    A, B = _leaves(8)
    views = [_chain(A, B), raster.Add(A, B), raster.Multiply(B, 3.0)]
    with config.config.set({"geomodeling.devices": None}):
        wants = [v.get_data(**_request()) for v in views]
    for got, want in zip(get_data_many(views, _request()), wants):
        assert np.array_equal(_bits(got["values"]), _bits(want["values"]))
    with config.config.set({"geomodeling.materialize": False}):
        lazy = views[1].get_data(**_request())
    assert isinstance(lazy["values"], shard.ShardedTiles) and len(lazy["values"].units) == 1
    assert np.array_equal(_bits(np.asarray(lazy["values"])), _bits(wants[1]["values"]))


@pytest.mark.timeout(120)
def test_a_worker_without_device_and_without_double_refuses_to_start(cpu_vm):
    from dask_geomodeling_amd.hip import runtime
    if runtime.lib().gm_device_count() > 0:
        pytest.skip("a GPU is visible: the worker would start")
    with pytest.raises(runtime.GmHipError) as info:
        shard.Pool((0,), preload=None)
    assert "no HIP device visible" in str(info.value) and "shard worker 0" in str(info.value)


def test_shared_arrays_are_dropped_with_their_array(monkeypatch):
    monkeypatch.setattr(shard, "SHARE_FROM", 1024)
    data = np.arange(4096, dtype="f4")
    text = shard._dumps(("x", data, data))
    assert len(text) < 1000 and len(shard._shared_arrays) >= 1
    name = shard._shared_arrays[id(data)][0].name
    assert os.path.exists("/dev/shm/" + name)
    del data
    gc.collect()
    assert not os.path.exists("/dev/shm/" + name)


def test_stale_segments_of_dead_processes_are_swept():
    """A parent that was killed leaves its /dev/shm files behind (they hold memory): the next pool removes the files of pids
    that no longer exist, and only those."""
    import subprocess
    child = subprocess.run([sys.executable, "-c", "import os; print(os.getpid())"], capture_output=True, text=True)
    dead = int(child.stdout)
    ns = shard._pid_namespace()
    stale = "/dev/shm/gm-shard-%d-0-deadbeef-%s-12345" % (dead, ns)
    mine = "/dev/shm/gm-shard-%d-999999-cafecafe-%s-%s" % (os.getpid(), ns, shard._start_time(os.getpid()))
    # a live pid whose start time differs: the creator is gone, a later process got its number
    reused = "/dev/shm/gm-shard-%d-1-deadbeef-%s-1" % (os.getppid(), ns)
    # the same dead pid in ANOTHER pid namespace (a /dev/shm shared between containers): not ours to judge
    foreign = "/dev/shm/gm-shard-%d-0-deadbeef-%s-12345" % (dead, "ffffffff")
    for path in (stale, mine, reused, foreign):
        with open(path, "wb") as f:
            f.write(b"x")
    try:
        removed = shard.sweep_stale_segments()
        assert os.path.basename(stale) in removed and not os.path.exists(stale)
        assert os.path.basename(reused) in removed and not os.path.exists(reused)
        assert os.path.exists(mine) and os.path.exists(foreign)
    finally:
        for path in (stale, mine, reused, foreign):
            if os.path.exists(path):
                os.unlink(path)


def test_a_forked_child_fails_loudly():
    """fork() after the library was entered: every runtime call of the child answers GM_ERR_FORKED -- what dask's `processes`
    scheduler would otherwise hang in on a GPU box (core/graphs.py:42-49 hands the graph to any configured scheduler)."""
    from dask_geomodeling_amd.hip import runtime
    L = runtime.lib()
    L.gm_device_count()  # the parent enters the library
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(r)
            rc = L.gm_init()
            text = L.gm_last_error().decode()
            raised = ""
            try:
                runtime.require_device()
            except runtime.GmHipError as e:
                raised = str(e)
            ptr = __import__("ctypes").c_void_p()
            rc2 = L.gm_malloc(__import__("ctypes").byref(ptr), 64)
            os.write(w, ("%d|%d|%s|%s" % (rc, rc2, text, raised)).encode())
        finally:
            os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.waitpid(pid, 0)
    rc, rc2, text, raised = data.decode().split("|")
    assert int(rc) == 9004 and int(rc2) == 9004
    assert text.startswith("forked after gm_init") and "spawn" in text
    assert raised.startswith("forked after gm_init")
    assert L.gm_init() in (0, 9003)  # the parent is as it was
