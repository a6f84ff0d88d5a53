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


def test_more_than_2_to_the_32_cells_in_one_tile(gpu):
    """The maximum sizes: one tile of 2^32 + 9.9 M cells (65 600 x 65 622) evaluated in ONE request per view -- flat cell
    indices and byte offsets beyond 32 bits in the interpreter (uint8 and float32 tiles: 4.3 GB and 17 GB per operand), the
    1-byte table path, a stencil, Classify / Reclassify and the reductions -- spot-checked at the far end of the tile (cells
    whose flat index no longer fits 32 bits) and at its start against the oracle on host copies of those windows."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd.hip import DeviceArray
    from dask_geomodeling_amd.raster.sources import generate_window
    h, w = 65600, 65622
    assert h * w > 2 ** 32
    size = 96
    spots = [(0, 0), (h - size, w - size), (h - 2 * size, 4), (h // 2 + 1, w // 2 + 2), (h - size, 0)]
    req = dict(mode="vals", bbox=(0.0, 0.0, float(w), float(h)), width=w, height=h, projection="EPSG:28992")

    def host(seed, dtype, lo, hi, ndv, r0, c0):
        return {"values": synth.generate_window((h, w), dtype, seed, lo, hi, ndv, LATTICE, r0, c0, size, size), "no_data_value": ndv}

    def window(values, r0, c0, margin=0):
        assert isinstance(values, DeviceArray)
        return np.asarray(values.window(r0 - margin if margin else r0, c0 - margin if margin else c0, size, size))

    with gmconfig.config.set({"geomodeling.materialize": False}):
        # -- uint8 --
        dc = generate_window((h, w), "u1", 4100, 0.0, 12.0, 255, LATTICE, 0, 0, h, w)
        C = raster.MemorySource(dc, np.uint8(255), "EPSG:28992", 1.0, (0.0, float(h)))
        table = [[k, (5 * k) % 11] for k in range(0, 12, 2)]
        views = {
            "C > 5": (raster.Greater(C, 5), lambda c: O.math_process("greater", {"dtype": "bool", "fillvalue": None}, c, 5)),
            "Mask(Reclassify(C))": (raster.Mask(raster.Reclassify(C, table), 1),
                                    lambda c: O.mask_process(O.reclassify_process(c, {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max),
                                                                                      "data": table, "select": False}), 1)),
            "Classify(C, 40 edges)": (raster.Classify(C, [0.25 * k for k in range(40)]),
                                      lambda c: O.classify_process(c, [0.25 * k for k in range(40)], False)),
        }
        for name, (view, want_of) in views.items():
            got = view.get_data(**req)["values"]
            assert got.shape == (1, h, w), name
            for r0, c0 in spots:
                want = want_of(host(4100, "u1", 0.0, 12.0, np.uint8(255), r0, c0))["values"]
                assert_values_equal(window(got, r0, c0), want, 0, "%s at (%d, %d)" % (name, r0, c0))
            del got
        # a stencil whose last rows lie beyond 2^32 cells (the request is the tile less a one-cell margin)
        inner = dict(req, bbox=(1.0, 1.0, float(w - 1), float(h - 1)), width=w - 2, height=h - 2)
        got = raster.Dilate(C, [3]).get_data(**inner)["values"]
        for r0, c0 in ((0, 0), (h - 2 - size, w - 2 - size), (h - 2 - size, 0)):
            around = {"values": synth.generate_window((h, w), "u1", 4100, 0.0, 12.0, 255, LATTICE, r0, c0, size + 2, size + 2),
                      "no_data_value": np.uint8(255)}
            assert_values_equal(window(got, r0, c0), O.dilate_process(around, [3])["values"], 0, "Dilate at (%d, %d)" % (r0, c0))
        del got, C, dc
        # -- float32: 17 GB per operand --
        ndv = np.float32(F32MAX)
        da = generate_window((h, w), "f4", 4101, 0.0, 100.0, F32MAX, LATTICE, 0, 0, h, w)
        db = generate_window((h, w), "f4", 4102, 0.0, 100.0, F32MAX, LATTICE, 0, 0, h, w)
        A = raster.MemorySource(da, ndv, "EPSG:28992", 1.0, (0.0, float(h)))
        B = raster.MemorySource(db, ndv, "EPSG:28992", 1.0, (0.0, float(h)))
        kw = {"dtype": "float32", "fillvalue": F32MAX}
        got = raster.Clip(A * 2.5 + B, B > 50).get_data(**req)["values"]
        most = raster.Max(A, B).get_data(**req)["values"]
        for r0, c0 in spots:
            a, b = host(4101, "f4", 0.0, 100.0, ndv, r0, c0), host(4102, "f4", 0.0, 100.0, ndv, r0, c0)
            x = O.math_process("add", kw, O.math_process("multiply", kw, a, 2.5), b)
            want = O.clip_process(x, O.math_process("greater", {"dtype": "bool", "fillvalue": None}, b, 50))["values"]
            assert_values_equal(window(got, r0, c0), want, 0, "Clip(A * 2.5 + B, B > 50) at (%d, %d)" % (r0, c0))
            assert_values_equal(window(most, r0, c0), O.reduce_max_process(kw, a, b)["values"], 0, "Max at (%d, %d)" % (r0, c0))
