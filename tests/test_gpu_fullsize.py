"""Parity at BASELINE.json's full sizes.  The oracle cannot sweep 67-1074 Mpix in seconds, so the
device result of the whole tile is spot-checked: the procedural generator has a NumPy twin
(oracle/synth.py), every op on the path is local (a window of the result depends on a window of the
inputs), hence random windows of the full-size result are compared bit for bit with the oracle
evaluated on host copies of just those windows; plus size-independent properties (determinism of
the order-independent checksum, tile-size independence of a tiled evaluation)."""
import numpy as np
import pytest

from helpers import assert_values_equal
from oracle import ref_numpy as O
from oracle import synth

pytestmark = pytest.mark.gpu

F32MAX = float(np.finfo("f4").max)
LATTICE = (97, 89)


def _windows(rng, side, size, count):
    for _ in range(count):
        r0 = int(rng.integers(0, (side - size) // 2)) * 2
        c0 = int(rng.integers(0, (side - size) // 4)) * 4
        yield r0, c0
    yield 0, 0
    yield side - size, side - size


def _host(seed, side, dtype, lo, hi, ndv, lattice, r0, c0, size):
    return synth.generate_window((side, side), dtype, seed, lo, hi, ndv, lattice, r0, c0, size, size)


def test_config2_chain5_4096(gpu):
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.raster.sources import generate_window
    side, ndv = 4096, np.float32(F32MAX)
    da = generate_window((side, side), "f4", 100, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
    db = generate_window((side, side), "f4", 101, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
    A = raster.MemorySource(da, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
    B = raster.MemorySource(db, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
    view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    req = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")
    got = view.get_data(**req)["values"]
    kw = {"dtype": "float32", "fillvalue": F32MAX}
    rng = np.random.default_rng(0)
    for r0, c0 in _windows(rng, side, 128, 12):
        a = {"values": _host(100, side, "f4", 0.0, 100.0, ndv, LATTICE, r0, c0, 128), "no_data_value": ndv}
        b = {"values": _host(101, side, "f4", 0.0, 100.0, ndv, LATTICE, r0, c0, 128), "no_data_value": ndv}
        x = O.math_process("add", kw, a, b)
        x = O.math_process("multiply", kw, x, 2.5)
        x = O.math_process("subtract", kw, x, b)
        x = O.math_process("add", kw, x, 1.5)
        want = O.clip_process(x, a)["values"]
        assert_values_equal(got[:, r0:r0 + 128, c0:c0 + 128], want, 0, "window (%d, %d)" % (r0, c0))
    # tile-size independence: the tiled evaluation (zero-copy windows, batched, in-place mosaic) is the same array
    for tile in (1024, 2048):
        tiled = raster.RasterTiler(view, tile).get_data(**req)["values"]
        assert np.array_equal(tiled.view("u4"), got.view("u4")), "tile size %d" % tile


def test_config3_reclassify_mask_8192(gpu):
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip import DeviceArray
    from dask_geomodeling_amd.raster.sources import generate_window
    from dask_geomodeling_amd import config as gmconfig
    side, ndv = 8192, 2147483647
    table = [[k, (7 * k) % 1000] for k in range(0, 256, 2)]
    d = generate_window((side, side), "i4", 2000, 0.0, 256.0, ndv, (50, 50), 0, 0, side, side)
    src = raster.MemorySource(d, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
    req = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")
    view = raster.Mask(raster.Reclassify(src, table), 1)
    with gmconfig.config.set({"geomodeling.materialize": False}):
        fused = view.get_data(**req)["values"]
        assert isinstance(fused, DeviceArray) and fused.dtype == np.dtype("u1")
        again = view.get_data(**req)["values"]
        assert fused.checksum() == again.checksum()  # deterministic
        with gmconfig.config.set({"geomodeling.fuse": False}):
            unfused = view.get_data(**req)["values"]  # dedicated Reclassify kernel + Mask program
        assert fused.checksum() == unfused.checksum()
        rcl = raster.Reclassify(src, table).get_data(**req)["values"]
    got = np.asarray(fused)
    rcl = np.asarray(rcl)
    kw = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "data": table, "select": False}
    rng = np.random.default_rng(1)
    for r0, c0 in _windows(rng, side, 128, 12):
        h = {"values": _host(2000, side, "i4", 0.0, 256.0, ndv, (50, 50), r0, c0, 128), "no_data_value": ndv}
        want_r = O.reclassify_process(h, kw)
        assert_values_equal(rcl[:, r0:r0 + 128, c0:c0 + 128], want_r["values"], 0, "reclassify window (%d, %d)" % (r0, c0))
        want = O.mask_process(want_r, 1)["values"]
        assert_values_equal(got[:, r0:r0 + 128, c0:c0 + 128], want, 0, "mask window (%d, %d)" % (r0, c0))


def test_config4_bilinear_16384_to_32768(gpu):
    from dask_geomodeling_amd.raster import resample
    from dask_geomodeling_amd.raster.sources import generate_window
    side, ndv = 16384, np.float32(F32MAX)
    src = generate_window((side, side), "f4", 3000, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
    out = resample.bilinear(src, 2 * side, 2 * side, ndv)
    assert out.checksum() == resample.bilinear(src, 2 * side, 2 * side, ndv).checksum()
    rng = np.random.default_rng(2)
    size, pad = 64, 2  # source window + apron; output pixels of the un-padded part are compared
    for r0, c0 in list(_windows(rng, side, size + 2 * pad, 10)):
        s = _host(3000, side, "f4", 0.0, 100.0, ndv, LATTICE, r0, c0, size + 2 * pad)
        want = O.bilinear_resample(s, 2 * (size + 2 * pad), 2 * (size + 2 * pad), ndv)[:, 2 * pad:-2 * pad, 2 * pad:-2 * pad]
        got = np.asarray(out.window(2 * (r0 + pad), 2 * (c0 + pad), 2 * size, 2 * size))
        assert_values_equal(got, want, 1, "bilinear window (%d, %d)" % (r0, c0))  # 1 ulp (spec: float64, rounded once)
    # the clamped borders: first / last rows and columns against the oracle on the matching strips
    edge = _host(3000, side, "f4", 0.0, 100.0, ndv, LATTICE, 0, 0, 40)
    want = O.bilinear_resample(edge, 80, 80, ndv)[:, :64, :64]
    assert_values_equal(np.asarray(out.window(0, 0, 64, 64)), want, 1, "top-left corner")


@pytest.mark.parametrize("leaves", ["whole", "owned"])
def test_config5_mosaic_tiles_65536(gpu, leaves):
    """The 10-Block view over the 65536^2 mosaic -- leaves "whole" (4 x 16 GiB MemorySources, tiles are
    zero-copy windows) and "owned" (only the windows of the evaluated tiles are resident): four of the
    256 tiles are evaluated through the tiler's sub-graphs (one fused float32 program each) and spot-checked."""
    import bench
    from dask_geomodeling_amd import raster, shard
    from dask_geomodeling_amd.core.graphs import get_sync
    from dask_geomodeling_amd.fuse import fuse_graph
    side = bench.MOSAIC_SIDE
    tiler = raster.RasterTiler(bench.mosaic_view(raster, leaves), 4096)
    req = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")
    graph, name = tiler.get_compute_graph(**req)
    _, tile_keys = shard.split_tiler_graph(graph, name)
    assert len(tile_keys) == 256
    picks = [0, 17, 130, 255]
    keys = [tile_keys[t] for t in picks]
    results = get_sync(fuse_graph(graph, keys), keys)
    ndv = np.float32(F32MAX)
    kw = {"dtype": "float32", "fillvalue": F32MAX}
    rng = np.random.default_rng(3)
    for t, res in zip(picks, results):
        ix, iy = divmod(t, 16)  # x-major tile order (parallelize.py:82); low y = high rows
        row_t, col_t = side - (iy + 1) * 4096, ix * 4096
        got = np.asarray(res["values"])
        assert got.dtype == np.dtype("u1") and got.shape == (1, 4096, 4096)
        for _ in range(3):
            r0, c0 = int(rng.integers(0, 4096 - 96)), int(rng.integers(0, 4096 - 96))
            leaf = [{"values": _host(seed, side, "f4", 0.0, 100.0, ndv, LATTICE, row_t + r0, col_t + c0, 96),
                     "no_data_value": ndv} for seed in (11, 12, 13, 14)]
            x = O.math_process("add", kw, leaf[0], leaf[1])
            x = O.math_process("multiply", kw, x, 0.5)
            x = O.math_process("subtract", kw, x, leaf[2])
            x = O.math_process("divide", kw, x, 2.0)
            m = O.math_process("greater", {"dtype": "bool", "fillvalue": None}, leaf[3], 20.0)
            x = O.clip_process(x, m)
            x = O.classify_process(x, [-50.0, -10.0, 0.0, 10.0, 50.0], False)
            x = O.reclassify_process(x, {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False,
                                         "data": [[0, 10], [1, 11], [3, 13], [5, 15]]})
            want = O.mask_process(x, 1)["values"]
            assert_values_equal(got[:, r0:r0 + 96, c0:c0 + 96], want, 0, "tile %d window (%d, %d)" % (t, r0, c0))
    del results, tiler, graph
    raster.SyntheticSource.drop_resident_windows()
