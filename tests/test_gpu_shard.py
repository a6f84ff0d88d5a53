"""The one-worker-per-GPU scheduler on the MI355X box (dask_geomodeling_amd/shard.py): worker processes that share the
box's single GPU -- "devices" (0, 0) -- must return what the single-process path returns, bit for bit; host mosaics arrive
through a shared-memory range every worker has pinned (gm_host_register + gm_d2h_2d); a fork()ed child fails loudly."""
import ctypes
import gc
import os

import numpy as np
import pytest

from dask_geomodeling_amd import config, raster, shard

pytestmark = pytest.mark.gpu
F32MAX = np.float32(np.finfo("f4").max)


@pytest.fixture
def two_workers(gpu):
    with config.config.set({"geomodeling.devices": [0, 0]}):
        yield shard.pool_for((0, 0))
    shard.shutdown()
    gc.collect()


def _sources(side, seed=0):
    rng = np.random.default_rng(seed)
    a = (rng.random((side, side)) * 100).astype("f4")
    b = (rng.random((side, side)) * 100).astype("f4")
    a[::97, ::89] = F32MAX
    A = raster.MemorySource(a, F32MAX, "EPSG:28992", 1.0, (0.0, float(side)))
    B = raster.MemorySource(b, F32MAX, "EPSG:28992", 1.0, (0.0, float(side)))
    return A, B


def _request(side):
    return dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")


def _bits(values):
    values = np.ascontiguousarray(np.asarray(values))
    return values.view("u%d" % values.dtype.itemsize)


@pytest.mark.timeout(600)
def test_two_workers_return_the_single_process_mosaic(two_workers):
    """The 16-tile mosaic of the canonical chain (two 2048^2 float32 MemorySources, tiles of 512^2) and of the 10-Block view's
    table Blocks: two worker processes, tiles dealt t % 2, == RasterTiler.get_data in this process, bit for bit."""
    side = 2048
    A, B = _sources(side)
    chain = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    x = raster.Classify(raster.Clip(raster.Divide(raster.Subtract(raster.Multiply(raster.Add(A, B), 0.5), B), 2.0),
                                    raster.Greater(B, 20.0)), [-50.0, -10.0, 0.0, 10.0, 50.0])
    mixed = raster.Mask(raster.Reclassify(x, [[0, 10], [1, 11], [3, 13], [5, 15]]), 1)
    for view, tile in ((chain, 512), (mixed, 512), (chain, [768, 640]), (raster.HillShade(A), 512)):
        tiler = raster.RasterTiler(view, tile)
        with config.config.set({"geomodeling.devices": None}):
            want = tiler.get_data(**_request(side))
        got = tiler.get_data(**_request(side))
        assert isinstance(got["values"], np.ndarray) and got["values"].dtype == want["values"].dtype
        assert got["no_data_value"] == want["no_data_value"]
        assert np.array_equal(_bits(got["values"]), _bits(want["values"])), (type(view).__name__, tile)
        again = tiler.get_data(**_request(side))  # the kept job, the recycled (already pinned) host range
        assert np.array_equal(_bits(again["values"]), _bits(want["values"]))
        # ... which from its second run on keeps its sources' zero-copy windows and the arguments of its batched launches
        # (core/graphs.py PreparedGraph): the third run is the first one that is all replay
        third = tiler.get_data(**_request(side))
        assert np.array_equal(_bits(third["values"]), _bits(want["values"]))
        job = next(reversed(two_workers.jobs.values())).id
        stats = two_workers.request({0: ("job_stats", job), 1: ("job_stats", job)})
        for w in (0, 1):
            assert stats[w]["runs"] == 3, stats
            if isinstance(view, raster.HillShade):
                # (a tile at the mosaic's edge asks its source for a margin outside the raster: warped, not a window; the two
                #  interior tiles of a worker read zero-copy windows)
                assert stats[w]["kept"] >= 2, stats
            else:
                assert stats[w]["kept"] >= 8 and stats[w]["prepared"] >= 1, stats
    assert {w.pid for w in two_workers.workers}.isdisjoint({os.getpid()}) and len({w.pid for w in two_workers.workers}) == 2
    assert all("gfx950" in name for name in two_workers.device_names)
    two_workers.request({0: ("mark",), 1: ("mark",)})
    tiler.get_data(**_request(side))
    assert all(ms is not None and ms > 0 for ms in two_workers.request({0: ("elapsed",), 1: ("elapsed",)}).values())


@pytest.mark.timeout(600)
def test_tiles_left_on_their_devices_and_their_checksums(two_workers):
    side = 1024
    A, B = _sources(side, 3)
    view = raster.Add(raster.Multiply(A, 2.0), B)
    tiler = raster.RasterTiler(view, 256)
    with config.config.set({"geomodeling.devices": None, "geomodeling.materialize": False}):
        from dask_geomodeling_amd import plan
        pairs = list(tiler.get_sources_and_requests(**_request(side)))
        local = [view.get_data(**tile_request)["values"].checksum() for _, tile_request in pairs[1:]]
        want = tiler.get_data(**_request(side))["values"].to_numpy()
    with config.config.set({"geomodeling.materialize": False}):
        lazy = tiler.get_data(**_request(side))
    handle = lazy["values"]
    assert isinstance(handle, shard.ShardedTiles) and handle.shape == (1, side, side)
    sums = handle.checksums()
    assert [sums[t] for t in range(16)] == local        # gm_checksum of every tile where it lives == in this process
    assert np.array_equal(_bits(handle.to_numpy()), _bits(want))


@pytest.mark.timeout(600)
def test_get_of_a_graph_with_memorysources(two_workers):
    """shard.get with dask's calling convention: the data of the MemorySources travels through shared memory (once), every
    worker uploads what its units read; int32 -> int64 Reclassify results come back in their own dtype."""
    side = 1024
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, size=(side, side)).astype("i4")
    d[::50, ::50] = 2147483647
    src = raster.MemorySource(d, 2147483647, "EPSG:28992", 1.0, (0.0, float(side)))
    table = [[k, (7 * k) % 1000] for k in range(0, 256, 2)]
    view = raster.Reclassify(src, table)
    tiler = raster.RasterTiler(view, 256)
    graph, name = tiler.get_compute_graph(**_request(side))
    graph, plain = raster.Mask(view, 1).get_compute_graph(cached_compute_graph=graph, **_request(side))
    with config.config.set({"geomodeling.devices": None}):
        want = view.get_data(**_request(side))
        want_mask = raster.Mask(view, 1).get_data(**_request(side))
    mosaic, mask = shard.get(graph, [name, plain])
    assert mosaic["values"].dtype == np.dtype("i8") and np.array_equal(mosaic["values"], want["values"])
    assert mask["values"].dtype == np.dtype("u1") and np.array_equal(mask["values"], want_mask["values"])
    assert len(shard._shared_arrays) == 1


@pytest.mark.timeout(600)
def test_series_tiles_and_tiles_that_are_cast(two_workers):
    """A tiler over a view of a resident SERIES (tiles of several frames: every frame of a tile lands in its plane of the host
    mosaic) and over a store whose tiles arrive in another dtype than the tiler declares (TemporalSum: int64 sums under a
    declared int16 -- `values[...] = tile` casts on assignment, parallelize.py:118-124; here on the device, before the copy)."""
    from datetime import datetime, timedelta
    rng = np.random.default_rng(11)
    s = (rng.random((5, 192, 256)) * 50).astype("f4")
    s[rng.random(s.shape) < 0.1] = -9999.0
    S = raster.MemorySource(s, -9999.0, "EPSG:28992", 1.0, (0.0, 192.0), time_first=datetime(2021, 3, 1),
                            time_delta=timedelta(hours=6))
    c = rng.integers(0, 40, size=(5, 192, 256)).astype("i2")
    C = raster.MemorySource(c, -1, "EPSG:28992", 1.0, (0.0, 192.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    box = dict(mode="vals", bbox=(0.0, 0.0, 256.0, 192.0), width=256, height=192, projection="EPSG:28992",
               start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))
    for view, tile in ((S * 2 + 1, [64, 48]), (raster.TemporalSum(C), [128, 64]), (raster.TemporalAggregate(S, "D", "max"), 96)):
        tiler = raster.RasterTiler(view, tile)
        with config.config.set({"geomodeling.devices": None}):
            want = tiler.get_data(**box)
        got = tiler.get_data(**box)
        assert got["values"].shape == want["values"].shape and got["values"].dtype == want["values"].dtype, type(view).__name__
        assert np.array_equal(_bits(got["values"]), _bits(want["values"])), type(view).__name__
        with config.config.set({"geomodeling.devices": None}):
            when = tiler.get_data(**dict(box, mode="time"))
        assert tiler.get_data(**dict(box, mode="time")) == when  # (no tiles: one unit on one worker)
    two_workers.sync()


@pytest.mark.timeout(300)
def test_a_forked_child_of_an_initialised_process_fails_loudly(gpu):
    """What dask's `processes` scheduler does by default (fork) after this process has used the GPU: the child's runtime calls
    answer GM_ERR_FORKED at once instead of hanging inside the driver."""
    from dask_geomodeling_amd.hip import runtime
    from dask_geomodeling_amd.hip.array import DeviceArray
    DeviceArray.from_numpy(np.ones((4, 4), "f4")).to_numpy()   # the parent is initialised for real
    L = runtime.lib()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(r)
            ptr = ctypes.c_void_p()
            rc = L.gm_malloc(ctypes.byref(ptr), 1 << 20)
            text = L.gm_last_error().decode()
            rc2 = L.gm_sync()
            os.write(w, ("%d|%d|%s" % (rc, rc2, text)).encode())
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
    rc, rc2, text = data.decode().split("|")
    assert int(rc) == 9004 and int(rc2) != 0 and text.startswith("forked after gm_init")
    # the parent goes on
    assert float(DeviceArray.from_numpy(np.full((2, 2), 3.0, "f4")).to_numpy()[0, 0]) == 3.0
