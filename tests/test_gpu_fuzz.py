"""Differential fuzzing on the device: random views (fused, cut at register-class boundaries,
stand-alone table kernels where a table does not fit a program) against the op-by-op oracle,
evaluated one by one and all together (batched where structures coincide)."""
import numpy as np
import pytest

from fuzz_util import REQ, describe, oracle_eval, random_view, unsigned_zero
from helpers import assert_result_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(300))
def test_random_views_on_device(gpu, seed):
    from dask_geomodeling_amd import config as gmconfig
    rng = np.random.default_rng(10_000 + seed)
    view = random_view(rng, "g%d" % seed)
    with np.errstate(all="ignore"):
        want = unsigned_zero(oracle_eval(view), view)
    try:
        got = view.get_data(**REQ)
    except NotImplementedError as e:
        pytest.skip("not on the device path: %s" % e)
    assert_result_equal(unsigned_zero(got, view), want, 0, describe(view))
    with gmconfig.config.set({"geomodeling.fuse": False}):
        assert_result_equal(unsigned_zero(view.get_data(**REQ), view), want, 0, "unfused " + describe(view))


@pytest.mark.parametrize("seed", range(60))
def test_views_over_nine_to_sixteen_leaves(gpu, seed):
    """Programs of 9-16 inputs (round 6: GM_MAX_INPUTS 8 -> 16): every leaf folded into the view by a random Block, fused into one
    launch, against the op-by-op oracle -- and unfused.  (The first version fetched the operands of generic steps through a switch
    of eight cases: 'first' over nine rasters picked nothing from rasters 8 on.  No tree of the other fuzzers has more than three
    leaves.)"""
    from dask_geomodeling_amd import config as gmconfig
    from fuzz_util import wide_view
    rng = np.random.default_rng(70_000 + seed)
    n = int(rng.integers(9, 17))
    view = wide_view(rng, "w%d" % seed, n)
    with np.errstate(all="ignore"):
        want = unsigned_zero(oracle_eval(view), view)
    got = view.get_data(**REQ)
    assert_result_equal(unsigned_zero(got, view), want, 0, describe(view))
    with gmconfig.config.set({"geomodeling.fuse": False}):
        assert_result_equal(unsigned_zero(view.get_data(**REQ), view), want, 0, "unfused " + describe(view))


def test_random_views_evaluated_together(gpu):
    from dask_geomodeling_amd.core.graphs import get_data_many
    views, wants = [], []
    for seed in range(40):
        rng = np.random.default_rng(20_000 + seed)
        view = random_view(rng, "m%d" % seed, depth=3)
        try:
            view.get_data(**REQ)
        except NotImplementedError:
            continue
        with np.errstate(all="ignore"):
            wants.append(unsigned_zero(oracle_eval(view), view))
        views.append(view)
    for got, want, view in zip(get_data_many(views, REQ), wants, views):
        assert_result_equal(unsigned_zero(got, view), want, 0, describe(view))


@pytest.mark.parametrize("seed", range(60))
def test_random_views_tiled_equals_untiled(gpu, seed, tiler_route):
    """RasterTiler around a random view on MemorySources (zero-copy windows, batched tile programs,
    in-place mosaic, ragged edge tiles) gives exactly the untiled result."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(30_000 + seed)
    view = random_view(rng, "t%d" % seed, depth=3, memory=True)
    try:
        whole = view.get_data(**REQ)
    except NotImplementedError as e:
        pytest.skip("not on the device path: %s" % e)
    for tile in ([8, 8], [16, 12], 17):
        tiled = raster.RasterTiler(view, tile).get_data(**REQ)
        assert_result_equal(unsigned_zero(tiled, view), unsigned_zero(whole, view), 0, "tile %s of %s" % (tile, describe(view)))


def test_sign_of_zero_normalisation_is_measured(gpu):
    """``unsigned_zero`` compares -0.0 and +0.0 as equal, for views that contain a ``Max`` only (which zero wins an
    np.fmax tie depends on the SIMD maximum NumPy was built with).  How much that hides, over the 300 views of
    test_random_views_on_device: the cells where device and oracle differ in nothing but the sign of a zero.  Measured
    on the MI355X box against this oracle host (round 3): 0 of the 50 880 float cells of the 53 views with a Max; everything else the
    normalisation could mask -- a different non-zero value -- fails test_random_views_on_device itself, because
    x + 0.0 only ever changes -0.0."""
    from fuzz_util import contains
    views = cells = signed = 0
    for seed in range(300):
        rng = np.random.default_rng(10_000 + seed)
        view = random_view(rng, "g%d" % seed)
        if not contains(view, "Max"):
            continue
        with np.errstate(all="ignore"):
            want = oracle_eval(view)
        try:
            got = view.get_data(**REQ)
        except NotImplementedError:
            continue
        a, b = np.asarray(got["values"]), np.asarray(want["values"])
        if a.dtype.kind != "f":
            continue
        views += 1
        cells += a.size
        differ = a.view("u%d" % a.dtype.itemsize) != b.view("u%d" % b.dtype.itemsize)
        differ &= ~(np.isnan(a) & np.isnan(b))
        assert ((a[differ] == 0) & (b[differ] == 0)).all(), "a cell differs in more than the sign of zero: " + describe(view)
        signed += int(differ.sum())
    print("views with a Max: %d, float cells: %d, cells differing in the sign of zero only: %d (%.3g %%)"
          % (views, cells, signed, 100.0 * signed / max(cells, 1)))
    assert views >= 20
    assert signed <= cells // 100, "%d of %d cells differ in the sign of zero" % (signed, cells)


def _same_response(got, want, what):
    if isinstance(want, np.ndarray):  # (an empty window: sources.py:91-92 hands back the bare array)
        assert isinstance(got, np.ndarray) and got.shape == want.shape and got.dtype == want.dtype, what
        return
    assert_result_equal(got, want, 0, what)


@pytest.mark.parametrize("seed", list(range(150)) + list(range(1000, 1030)))
def test_random_temporal_views_against_the_walking_oracle(gpu, seed, monkeypatch):
    """Random trees over multi-frame MemorySources -- element-wise steps, Snap / Shift / TemporalSum / TemporalAggregate /
    Cumulative / Resample, Place, the stencils -- asked for values in six time windows (none, a moment, windows inside, across and
    beyond the period), for their times and their metadata: the device answer equals, bit for bit, the view walked Block by
    Block with the reference's planning and the oracle's process bodies (tests/oracle_walk.py); where the reference's own
    evaluation raises (frames of two sources that do not line up, metadata a source does not have), so does the device path.
    Every other seed runs with a graph per request instead of the plans; every other pair of seeds answers tilers over element-wise
    stores by the product's own rule (small tiles: one request for the whole box) instead of the batched launch over the tiles."""
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan
    from oracle_walk import comparable, random_requests, random_temporal_view, walk, without_lossy_cells
    monkeypatch.setattr(plan, "_BATCHED_TILE_CELLS", 1 << 22 if (seed // 2) % 2 == 0 else 0)
    rng = np.random.default_rng(70_000 + seed)
    shape = (24, 40) if seed < 1000 else (150, 517)  # (the larger rasters: windows with a pitch, several workgroups per tile)
    view = random_temporal_view(rng, depth=int(rng.integers(1, 4)), shape_yx=shape)
    with gmconfig.config.set({"geomodeling.plans": seed % 2 == 0}):
        for request in random_requests(rng, view, shape):
            what = "%s <- %s" % (describe(view)[:600], {k: str(v) for k, v in request.items() if k in ("mode", "start", "stop")})
            try:
                want = walk(view, request)
            except NotImplementedError:
                raise
            except Exception:  # noqa: BLE001 -- whatever NumPy / pandas raise inside the reference's bodies
                with pytest.raises(Exception):
                    view.get_data(**request)
                continue
            try:
                got = view.get_data(**request)
            except NotImplementedError as refusal:  # (e.g. uint64 + float32 has no exact register class)
                pytest.skip("not on the device path: %s" % refusal)
            _same_response(*without_lossy_cells(comparable(got, view), comparable(want, view)), what)


@pytest.mark.parametrize("plans", [True, False])
def test_a_tiler_over_series_that_do_not_line_up(gpu, plans, monkeypatch):
    """What the round-6 campaign found (tests/fuzz_temporal.py 2000 23x39 9000000, seed 358): a tiled Shift(Divide(series of 7
    frames, series of 4 frames)) asked for coarser cells -- the fused task is handed leaf answers that are no raster responses, and
    the key of its kept launches indexed them like one (IndexError instead of the reference's own answer).  Against the walking oracle, every time
    window, tile by tile (the batched launch) with plans and with a graph per request."""
    from datetime import datetime
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan, raster
    from oracle_walk import comparable, walk, without_lossy_cells
    monkeypatch.setattr(plan, "_BATCHED_TILE_CELLS", 1 << 22 if plans else 0)
    rng = np.random.default_rng(358)
    a = (rng.random((7, 23, 39)) * 50).astype("f8")
    b = rng.integers(0, 9, size=(4, 23, 39)).astype("u2")
    a[rng.random(a.shape) < 0.1] = np.finfo("f8").max
    b[rng.random(b.shape) < 0.1] = 65535
    A = raster.MemorySource(a, np.finfo("f8").max, "EPSG:28992", 1.0, (0.0, 23.0), 1614556800000, 86400000, ["frame %d" % i for i in range(7)])
    B = raster.MemorySource(b, np.uint16(65535), "EPSG:28992", 1.0, (0.0, 23.0), 1614556800000, 86400000)
    view = raster.RasterTiler(raster.Shift(raster.Divide(A, B), 3600000), [18, 27])
    box = dict(mode="vals", bbox=(0.0, 0.0, 39.0, 23.0), width=19, height=11, projection="EPSG:28992")  # (coarser cells: a warp)
    windows = [{}, {"start": datetime(2021, 2, 28, 21, 30)}, {"start": datetime(2021, 3, 1, 5), "stop": datetime(2021, 3, 2, 22, 30)},
               {"start": datetime(2021, 3, 2, 14, 30), "stop": datetime(2021, 3, 3, 5, 30)},
               {"start": datetime(2021, 1, 30, 1), "stop": datetime(2021, 4, 3, 1)}]
    with gmconfig.config.set({"geomodeling.plans": plans}):
        for window in windows:
            request = dict(box, **window)
            for turn in range(3):  # (the third time round the launches of resident windows are replayed)
                try:
                    want = walk(view, request)
                except Exception:  # noqa: BLE001 -- where the reference's own evaluation raises, so must the device path
                    with pytest.raises(Exception):
                        view.get_data(**request)
                    continue
                got = view.get_data(**request)
                _same_response(*without_lossy_cells(comparable(got, view), comparable(want, view)), "%s, turn %d" % (window, turn))


@pytest.mark.parametrize("seed", range(120))
def test_random_memory_views_against_the_walking_oracle(gpu, seed, tiler_route):
    """The random views of this file over MemorySources -- stencils (request enlargement) and Smooth (exact and zoom mode)
    included -- asked for the whole grid, a window, coarser and finer cells, a box half a cell off the grid and partly outside
    the sources, and a point: against the view walked with the oracle's process bodies (tests/oracle_walk.py).  Every other seed
    with a graph per request instead of the plans."""
    from dask_geomodeling_amd import config as gmconfig
    from oracle_walk import comparable, walk
    rng = np.random.default_rng(40_000 + seed)
    view = random_view(rng, "w%d" % seed, depth=int(rng.integers(1, 4)), memory=True, smooth=True)
    boxes = [REQ,
             dict(REQ, bbox=(7.0, 3.0, 27.0, 15.0), width=20, height=12),
             dict(REQ, width=20, height=12),
             dict(REQ, bbox=(10.0, 6.0, 30.0, 18.0)),
             dict(REQ, bbox=(-7.5, -3.5, 32.5, 20.5)),
             dict(REQ, bbox=(11.5, 7.5, 11.5, 7.5), width=1, height=1)]
    with gmconfig.config.set({"geomodeling.plans": seed % 2 == 0}):
        for request in boxes:
            what = "%s <- %s" % (describe(view)[:600], {k: request[k] for k in ("bbox", "width", "height")})
            try:
                want = walk(view, request)
            except Exception:  # noqa: BLE001 -- the reference's own bodies raise: so must the device path
                with pytest.raises(Exception):
                    view.get_data(**request)
                continue
            try:
                got = view.get_data(**request)
            except NotImplementedError as refusal:
                pytest.skip("not on the device path: %s" % refusal)
            _same_response(comparable(got, view), comparable(want, view), what)
