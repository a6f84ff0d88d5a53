"""Evaluation plans (dask_geomodeling_amd/plan.py) on the device: a view's cached plan answers requests that
differ in their bounding box only -- aligned windows, unaligned boxes (nearest warp), boxes partly outside the
sources -- exactly like the graph path, for single views and under a RasterTiler."""
import numpy as np
import pytest

from fuzz_util import describe, random_view, unsigned_zero
from helpers import assert_result_equal

pytestmark = pytest.mark.gpu


def _graph_path(view, req):
    """The reference's way: the graph of the request built and fused from scratch (core/graphs.py:157-190)."""
    from dask_geomodeling_amd import config as gmconfig
    with gmconfig.config.set({"geomodeling.plans": False, "geomodeling.templates": False}):
        return view.get_data(**req)


def _template_path(view, req):
    """A real graph for the scheduler, made from the view's template (template.py): what get_data does as soon as
    a scheduler is configured in dask."""
    from dask_geomodeling_amd import config as gmconfig
    with gmconfig.config.set({"geomodeling.plans": False, "geomodeling.templates": True}):
        return view.get_data(**req)


def _oracle_window(view, bbox):
    """The oracle's result for an aligned window: element-wise Blocks are local, so it is the crop of the oracle's
    whole-extent result (fuzz_util.oracle_eval walks the tree op by op with oracle/ref_numpy.py)."""
    from fuzz_util import SHAPE, oracle_eval
    with np.errstate(all="ignore"):
        whole = oracle_eval(view)  # NotImplementedError for views with stencils
    x1, y1, x2, y2 = (int(v) for v in bbox)
    rows = slice(SHAPE[1] - y2, SHAPE[1] - y1)
    return {"values": whole["values"][:, rows, x1:x2], "no_data_value": whole["no_data_value"]}


@pytest.mark.parametrize("seed", range(40))
def test_plan_reused_across_bounding_boxes(gpu, seed):
    rng = np.random.default_rng(40_000 + seed)
    view = random_view(rng, "q%d" % seed, depth=3, memory=True)  # sources: 24 x 40 MemorySources at (0, 0)-(40, 24)
    boxes = [(0.0, 0.0, 16.0, 12.0), (8.0, 4.0, 24.0, 16.0), (24.0, 12.0, 40.0, 24.0),   # aligned windows
             (3.5, 2.25, 19.5, 14.25), (-6.0, -5.0, 10.0, 7.0), (30.0, 18.0, 46.0, 30.0)]  # unaligned / partly outside
    for k, bbox in enumerate(boxes):
        req = dict(mode="vals", bbox=bbox, width=16, height=12, projection="EPSG:28992")
        try:
            want = _graph_path(view, req)
        except NotImplementedError as e:
            pytest.skip("not on the device path: %s" % e)
        got = view.get_data(**req)
        assert_result_equal(unsigned_zero(got, view), unsigned_zero(want, view), 0, "bbox %d of %s" % (k, describe(view)))
        tpl = _template_path(view, req)
        assert_result_equal(unsigned_zero(tpl, view), unsigned_zero(want, view), 0, "template, bbox %d of %s" % (k, describe(view)))
        if k < 3:  # aligned windows inside the sources: the oracle itself, not only the other path
            try:
                oracle = _oracle_window(view, bbox)
            except NotImplementedError:
                continue
            assert_result_equal(unsigned_zero(got, view), unsigned_zero(oracle, view), 0, "plan vs oracle, bbox %d of %s" % (k, describe(view)))
            assert_result_equal(unsigned_zero(tpl, view), unsigned_zero(oracle, view), 0, "template vs oracle, bbox %d of %s" % (k, describe(view)))
    assert len(view.__dict__.get("_gm_plans", {})) <= 1  # one request shape -> at most one plan
    assert len(view.__dict__.get("_gm_templates", {})) <= 1  # ... and at most one template


@pytest.mark.parametrize("seed", range(30))
def test_repeated_requests_replay_their_kept_launches(gpu, seed):
    """Round 6: from the third time a bounding box is asked for, the launch over its resident windows is replayed with its
    arguments marshalled (plan.ViewPlan.boxes; fuse.FusedTask._kept_launch for graphs made per request; run_expr for Blocks with
    a kernel of their own over resident stores).  Five rounds over four boxes, plans and templates: every answer is the first
    round's, bit for bit, and boxes were kept where the view is one fused program over resident windows."""
    from dask_geomodeling_amd import plan
    rng = np.random.default_rng(41_000 + seed)
    view = random_view(rng, "rq%d" % seed, depth=3, memory=True)
    boxes = [(0.0, 0.0, 16.0, 12.0), (8.0, 4.0, 24.0, 16.0), (24.0, 12.0, 40.0, 24.0), (3.5, 2.25, 19.5, 14.25)]
    first = {}
    for turn in range(5):
        for k, bbox in enumerate(boxes):
            req = dict(mode="vals", bbox=bbox, width=16, height=12, projection="EPSG:28992")
            try:
                got = view.get_data(**req)
                tpl = _template_path(view, req)
            except NotImplementedError as e:
                pytest.skip("not on the device path: %s" % e)
            if turn == 0:
                first[k] = got
                want = _graph_path(view, req)
                assert_result_equal(unsigned_zero(got, view), unsigned_zero(want, view), 0, "first round, bbox %d of %s" % (k, describe(view)))
            assert_result_equal(got, first[k], 0, "round %d, bbox %d of %s" % (turn, k, describe(view)))
            assert_result_equal(tpl, first[k], 0, "template, round %d, bbox %d of %s" % (turn, k, describe(view)))
    kept = sum(len(p.boxes) for p in view.__dict__.get("_gm_plans", {}).values() if isinstance(p, plan.ViewPlan))
    print("kept boxes:", kept)


def test_kept_launches_from_several_threads(gpu):
    """Eight scheduler threads asking the same view for the same four boxes at the same time (dask's threaded scheduler shares a
    view's plans and a template's fused tasks): every answer is the box's own -- a kept launch takes a table of result pointers
    of its own per call."""
    import threading
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(77)
    a = (rng.random((1, 96, 128)) * 50).astype("f4")
    b = (rng.random((1, 96, 128)) * 50).astype("f4")
    A = raster.MemorySource(a, np.float32(-1.0), "EPSG:28992", 1.0, (0.0, 96.0))
    B = raster.MemorySource(b, np.float32(-1.0), "EPSG:28992", 1.0, (0.0, 96.0))
    view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    boxes = [(0.0, 0.0, 64.0, 48.0), (64.0, 48.0, 128.0, 96.0), (32.0, 16.0, 96.0, 64.0), (0.0, 48.0, 64.0, 96.0)]
    req = dict(mode="vals", width=64, height=48, projection="EPSG:28992")
    for plans in (True, False):
        with gmconfig.config.set({"geomodeling.plans": plans}):
            want = [np.asarray(view.get_data(bbox=box, **req)["values"]).copy() for box in boxes]
            for _ in range(3):  # (the launches of all four boxes are kept from here on)
                for box in boxes:
                    view.get_data(bbox=box, **req)
            wrong, errors = [], []

            def worker(first):
                try:
                    with gmconfig.config.set({"geomodeling.plans": plans}):
                        for k in range(120):
                            i = (first + k) % 4
                            got = np.asarray(view.get_data(bbox=boxes[i], **req)["values"])
                            if not np.array_equal(got.view("u4"), want[i].view("u4")):
                                wrong.append((first, k, i))
                except Exception as e:  # noqa: BLE001
                    errors.append(e)

            threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not errors, errors[:2]
            assert not wrong, wrong[:5]


def test_plan_exists_for_the_canonical_chain_and_is_reused(gpu, tiler_route):
    from dask_geomodeling_amd import plan, raster
    from dask_geomodeling_amd.hip import DeviceArray
    from dask_geomodeling_amd import config as gmconfig
    rng = np.random.default_rng(1)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 256, 256)) * 100).astype("f4")
    b = (rng.random((1, 256, 256)) * 100).astype("f4")
    a[0, ::9, ::4] = fill
    A = raster.MemorySource(a, fill, "EPSG:28992", 1.0, (0.0, 256.0))
    B = raster.MemorySource(b, fill, "EPSG:28992", 1.0, (0.0, 256.0))
    view = raster.Clip((A + B) * 2.5 - B + 1.5, A)
    req = dict(mode="vals", bbox=(0.0, 0.0, 64.0, 64.0), width=64, height=64, projection="EPSG:28992")
    p = plan.view_plan(view, req)
    assert p is not None and len(p.leaves) == 2
    for x0, y0 in ((0.0, 0.0), (64.0, 128.0), (192.0, 192.0)):
        r = dict(req, bbox=(x0, y0, x0 + 64.0, y0 + 64.0))
        assert plan.view_plan(view, r) is p
        assert_result_equal(view.get_data(**r), _graph_path(view, r), 0, "window at %s" % ((x0, y0),))
        with gmconfig.config.set({"geomodeling.materialize": False}):
            raw = view.get_data(**r)["values"]
        assert isinstance(raw, DeviceArray)
    # another request shape: another plan; a stencil view has no program of its own (it evaluates its store's plan)
    assert plan.view_plan(view, dict(req, width=32, height=32, bbox=(0.0, 0.0, 32.0, 32.0))) is not p
    assert plan.view_plan(raster.MovingMax(view, 3), req) is None
    # RasterTiler through the store's plans == untiled
    big = dict(mode="vals", bbox=(0.0, 0.0, 256.0, 256.0), width=256, height=256, projection="EPSG:28992")
    whole = _graph_path(view, big)
    for tile in (64, [128, 64], 100):
        tiler = raster.RasterTiler(view, tile)
        assert_result_equal(tiler.get_data(**big), whole, 0, "tiler %s through plans" % (tile,))
        assert_result_equal(_graph_path(tiler, big), whole, 0, "tiler %s through the graph" % (tile,))


@pytest.mark.parametrize("seed", range(30))
def test_tiler_through_plans_equals_untiled(gpu, seed, tiler_route):
    from dask_geomodeling_amd import raster
    from fuzz_util import REQ
    rng = np.random.default_rng(30_000 + seed)
    view = random_view(rng, "u%d" % seed, depth=3, memory=True)
    try:
        whole = _graph_path(view, REQ)
    except NotImplementedError as e:
        pytest.skip("not on the device path: %s" % e)
    try:
        from fuzz_util import oracle_eval
        with np.errstate(all="ignore"):
            assert_result_equal(unsigned_zero(whole, view), unsigned_zero(oracle_eval(view), view), 0, "whole vs oracle: " + describe(view))
    except NotImplementedError:
        pass  # (stencil Blocks: compared with the oracle in test_gpu_parity.py)
    for tile in ([8, 8], [16, 12], 17):
        tiler = raster.RasterTiler(view, tile)
        tiled = tiler.get_data(**REQ)
        assert_result_equal(unsigned_zero(tiled, view), unsigned_zero(whole, view), 0, "tile %s of %s" % (tile, describe(view)))
        for _ in range(2):  # (the second request finds the tiler's assembly and the store's templates)
            tiled = _template_path(tiler, REQ)
            assert_result_equal(unsigned_zero(tiled, view), unsigned_zero(whole, view), 0, "template, tile %s of %s" % (tile, describe(view)))


@pytest.mark.parametrize("dtype,cell", [("f4", 1.0), ("f4", 0.75), ("i2", 0.5), ("u1", 2.0), ("i4", 0.75), ("f8", 1.0)])
def test_tiler_over_hillshade_in_one_launch(gpu, dtype, cell, tiler_route):
    """A RasterTiler over HillShade on the plan path: the tiles of a mosaic shaded in ONE launch, written straight
    into the mosaic (gm_hillshade_batch) -- even tilings, ragged ones (tiles of two shapes: tile by tile + paste), dtypes
    and resolutions the batch kernel declines -- always the untiled HillShade, which test_gpu_parity pins to the
    reference, and the graph path's mosaic."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(11)
    dt = np.dtype(dtype)
    ndv = dt.type(np.finfo(dt).max) if dt.kind == "f" else dt.type(np.iinfo(dt).max)
    if dt.kind == "f":
        a = np.cumsum(rng.standard_normal((1, 192, 256)), axis=2).astype(dt) * 3
    else:
        a = rng.integers(0, 100, size=(1, 192, 256)).astype(dt)
    a[0, ::17, ::11] = ndv
    A = raster.MemorySource(a, ndv, "EPSG:28992", cell, (0.0, 192.0 * cell))
    shade = raster.HillShade(A, altitude=40, azimuth=300, fill=3)
    for box in ((16.0, 16.0, 144.0, 144.0), (0.0, 0.0, 256.0, 192.0), (3.0, 5.0, 131.0, 101.0)):
        x0, y0, x1, y1 = (v * cell for v in box)
        width, height = int(round(box[2] - box[0])), int(round(box[3] - box[1]))
        request = dict(mode="vals", bbox=(x0, y0, x1, y1), width=width, height=height, projection="EPSG:28992")
        whole = shade.get_data(**request)
        for tile in (64, [32, 64], 48):
            tiler = raster.RasterTiler(shade, tile)
            what = "%s cell %s box %s tile %s" % (dtype, cell, box, tile)
            assert_result_equal(tiler.get_data(**request), whole, 0, what + " (plans)")
            assert_result_equal(_graph_path(tiler, request), whole, 0, what + " (graph from scratch)")
            assert_result_equal(_template_path(tiler, request), whole, 0, what + " (graph from the template)")


@pytest.mark.parametrize("over", ["source", "computed"])
@pytest.mark.parametrize("dtype", ["f4", "u1", "i2", "i8", "f8"])
@pytest.mark.parametrize("kind", ["dilate", "movingmax3", "movingmax7", "movingmax21", "smooth", "hillshade"])
def test_tiler_over_stencils_writes_in_place(gpu, kind, dtype, over, monkeypatch, tiler_route):
    """A RasterTiler over Dilate / MovingMax / Smooth of a resident source on the plan path: every tile's result is
    written straight into its window of the mosaic where the kernel route can (gm_*_into), pasted otherwise (8-byte
    cells, large footprints, ragged tilings) -- always the untiled Block's result and the graph path's mosaic."""
    from dask_geomodeling_amd import plan, raster
    monkeypatch.setattr(plan, "_STENCILS_WHOLE", False)  # (tile by tile; the one-request rule has its own test below)
    rng = np.random.default_rng(17)
    dt = np.dtype(dtype)
    if kind == "smooth" and dt.kind != "f":
        pytest.skip("Smooth of integer rasters: compared in test_gpu_parity")
    ndv = dt.type(np.finfo(dt).max) if dt.kind == "f" else dt.type(np.iinfo(dt).max)
    a = rng.integers(0, 12, size=(1, 192, 256)).astype(dt)
    if dt.kind == "f":
        a = a + rng.random(a.shape).astype(dt)
    a[0, ::17, ::11] = ndv
    A = raster.MemorySource(a, ndv, "EPSG:28992", 1.0, (0.0, 192.0))
    if over == "computed":  # (a stencil over a computed store: tile after tile, still written in place)
        if dt.kind == "f":
            A = A * 2
        elif kind == "dilate":
            A = raster.Clip(A, A < 11)
        else:
            pytest.skip("integer stores: the element-wise result is at least int32 -- covered by the other dtypes")
    view = {"dilate": lambda: raster.Dilate(A, [3, 5, 7]), "movingmax3": lambda: raster.MovingMax(A, 3),
            "movingmax7": lambda: raster.MovingMax(A, 7), "movingmax21": lambda: raster.MovingMax(A, 21),
            "smooth": lambda: raster.Smooth(A, 4.0), "hillshade": lambda: raster.HillShade(A)}[kind]()
    for box in ((32.0, 32.0, 160.0, 160.0), (0.0, 0.0, 256.0, 192.0), (35.0, 37.0, 163.0, 133.0)):
        width, height = int(box[2] - box[0]), int(box[3] - box[1])
        request = dict(mode="vals", bbox=box, width=width, height=height, projection="EPSG:28992")
        whole = view.get_data(**request)
        for tile in (64, [32, 64], 48):
            tiler = raster.RasterTiler(view, tile)
            what = "%s %s box %s tile %s" % (kind, dtype, box, tile)
            if kind == "smooth":
                # (a tile of a Smooth is not a window of the whole: the reference crops by `size` cells while the Gaussian
                #  reaches 4/3 of that, spatial.py:262-290 -- the tiled results agree with each other)
                whole = _graph_path(tiler, request)
            assert_result_equal(tiler.get_data(**request), whole, 0, what + " (plans)")
            assert_result_equal(_graph_path(tiler, request), whole, 0, what + " (graph from scratch)")
            assert_result_equal(_template_path(tiler, request), whole, 0, what + " (graph from the template)")


def test_stencil_blocks_through_plans(gpu, tiler_route):
    """Blocks with a kernel of their own (gm_direct) are answered without a graph per request: alone over a source,
    over a planned element-wise view, as leaves of an element-wise view above them, nested, and under a RasterTiler --
    every time exactly the graph path's result, and through plan.evaluate."""
    from dask_geomodeling_amd import plan, raster
    rng = np.random.default_rng(4)
    fill = np.float32(np.finfo("f4").max)
    a = np.cumsum(rng.standard_normal((1, 200, 240)), axis=2).astype("f4")
    b = (rng.random((1, 200, 240)) * 10).astype("f4")
    a[0, ::13, ::7] = fill
    A = raster.MemorySource(a, fill, "EPSG:28992", 1.0, (0.0, 200.0))
    B = raster.MemorySource(b, fill, "EPSG:28992", 1.0, (0.0, 200.0))
    classes = raster.Classify(A + B, [-10, 0, 10, 20])
    views = {
        "hillshade(source)": raster.HillShade(A),
        "hillshade(view)": raster.HillShade(A * 2 + B),
        "view(hillshade)": raster.HillShade(A) * 0.5 + B,
        "view(movingmax(view), smooth(view))": raster.MovingMax(A + B, 5) - raster.Smooth(A - B, 2.0),
        "dilate(classify)": raster.Dilate(classes, [1, 3]),
        "movingmax(smooth(view))": raster.MovingMax(raster.Smooth(A + 1, 3.0), 3),
        "tiler(view(hillshade))": raster.RasterTiler(raster.HillShade(A) * 0.5 + B, 64),
        "tiler(hillshade(view))": raster.RasterTiler(raster.HillShade(A * 2 + B), [48, 64]),
        "tiler(smooth(source))": raster.RasterTiler(raster.Smooth(A, 2.0), 64),
    }
    boxes = [(0.0, 0.0, 128.0, 128.0), (40.0, 30.0, 168.0, 158.0), (100.0, 60.0, 228.0, 188.0), (-20.0, -10.0, 108.0, 118.0)]
    for name, view in views.items():
        for bbox in boxes:
            req = dict(mode="vals", bbox=bbox, width=128, height=128, projection="EPSG:28992")
            planned, raw = plan.evaluate(view, req)
            assert planned, name
            got = view.get_data(**req)
            want = _graph_path(view, req)
            assert_result_equal(got, want, 0, "%s at %s" % (name, (bbox,)))
            assert_result_equal(_template_path(view, req), want, 0, "template: %s at %s" % (name, (bbox,)))
    # other modes keep the graph path and agree with it
    for mode in ("time", "meta"):
        assert views["hillshade(view)"].get_data(mode=mode) == _graph_path(views["hillshade(view)"], dict(mode=mode))


def test_temporal_blocks_through_plans(gpu):
    """TemporalAggregate / Cumulative (gm_direct) over planned element-wise views, and under one: plan path == graph path
    for vals, time and meta requests."""
    from datetime import datetime
    from dask_geomodeling_amd import plan, raster
    rng = np.random.default_rng(9)
    frames, h, w = 24, 48, 64
    data = (rng.random((frames, h, w)) * 50).astype("f4")
    data[rng.random(data.shape) < 0.1] = -9999.0
    src = raster.MemorySource(data, -9999.0, "EPSG:28992", 1.0, (0.0, float(h)), time_first=0, time_delta=3600 * 1000,
                              metadata=["frame %d" % i for i in range(frames)])
    views = {"aggregate(view) - 3": raster.TemporalAggregate(src * 2 + 1, "6h", "mean") - 3,
             "aggregate(source, median)": raster.TemporalAggregate(src, "12h", "median"),
             "cumulative(view)": raster.Cumulative(src + 1, "sum", "6h")}
    for name, view in views.items():
        for bbox in ((0.0, 0.0, 32.0, 24.0), (16.0, 8.0, 48.0, 32.0)):
            req = dict(mode="vals", bbox=bbox, width=32, height=24, projection="EPSG:28992",
                       start=datetime(1970, 1, 1, 0), stop=datetime(1970, 1, 1, 23))
            planned, _ = plan.evaluate(view, req)
            assert planned, name
            assert_result_equal(view.get_data(**req), _graph_path(view, req), 0, name)
            for mode in ("time", "meta"):
                other = dict(req, mode=mode)
                assert view.get_data(**other) == _graph_path(view, other)


def test_templates_make_real_graphs_and_are_reused(gpu, tiler_route):
    """template.fused_compute_graph: a dask-style graph any scheduler can run -- leaf tasks + ONE FusedTask (the same
    object, with its compiled programs, for every bounding box) -- equal to the from-scratch graph's result, for a
    single view and for a RasterTiler (one TiledAssembly kept per tile layout)."""
    from dask_geomodeling_amd import raster, template
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd.core.graphs import get_sync
    from dask_geomodeling_amd.fuse import FusedTask, TiledAssembly
    rng = np.random.default_rng(2)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 256, 256)) * 100).astype("f4")
    b = (rng.random((1, 256, 256)) * 100).astype("f4")
    a[0, ::9, ::4] = fill
    A = raster.MemorySource(a, fill, "EPSG:28992", 1.0, (0.0, 256.0))
    B = raster.MemorySource(b, fill, "EPSG:28992", 1.0, (0.0, 256.0))
    view = raster.Clip((A + B) * 2.5 - B + 1.5, A)
    req = dict(mode="vals", width=64, height=64, projection="EPSG:28992")
    runs = []
    before = dict(template.stats)
    for x0, y0 in ((0.0, 0.0), (64.0, 128.0), (192.0, 192.0), (10.5, 3.25)):
        r = dict(req, bbox=(x0, y0, x0 + 64.0, y0 + 64.0))
        graph, name = template.fused_compute_graph(view, r)
        assert name == view.get_compute_graph(**r)[1] and set(graph) >= {name}
        assert len(graph) == 3 and isinstance(graph[name][0], FusedTask)  # two source tasks + the fused view
        runs.append(graph[name][0])
        got = get_sync(graph, [name])[0]
        assert_result_equal({"values": np.asarray(got["values"]), "no_data_value": got["no_data_value"]},
                            _graph_path(view, r), 0, "window at %s" % ((x0, y0),))
    assert all(run is runs[0] for run in runs), "the FusedTask (and its compiled programs) is reused"
    assert template.stats["built"] == before["built"] + 1 and template.stats["hits"] == before["hits"] + 4
    tiler = raster.RasterTiler(view, 64)
    big = dict(mode="vals", bbox=(0.0, 0.0, 256.0, 256.0), width=256, height=256, projection="EPSG:28992")
    assemblies = []
    for _ in range(2):
        graph, name = template.fused_compute_graph(tiler, big)
        if tiler_route == "batched-tiles":
            assert isinstance(graph[name][0], TiledAssembly) and len(graph) == 1 + 2 * 16
        else:  # tiles of 64^2 cells: the product answers the mosaic of a pixel-local view as one request for the whole box
            assert graph[name][0].__name__ == "mosaic_of_whole"
        assemblies.append(graph[name][0])
        got = get_sync(graph, [name])[0]
        assert_result_equal({"values": np.asarray(got["values"]), "no_data_value": got["no_data_value"]},
                            _graph_path(view, big), 0, "tiler through templates")
    assert assemblies[0] is assemblies[1] or tiler_route != "batched-tiles"
    with gmconfig.config.set({"geomodeling.templates": False}):
        graph, name = template.fused_compute_graph(view, dict(req, bbox=(0.0, 0.0, 64.0, 64.0)))
        assert graph[name][0] is not runs[0]  # switched off: built and fused from scratch


@pytest.mark.gpu
@pytest.mark.parametrize("plans", [True, False])
def test_tiled_series_windows_of_several_frames_in_place(gpu, plans, tiler_route):
    """A RasterTiler over element-wise views of resident multi-frame sources: every tile's operands are zero-copy windows of
    several bands (rows a source width apart, bands a source plane apart: DeviceArray.band_pitch), the program runs band by band
    in one call (gm_elemwise_run_batch_bands) and writes into the tile's window of a mosaic of as many frames -- ragged edge
    tiles, a time window that selects a sub-range of the frames, a single-frame operand next to the series, and a consumer
    that needs compact tiles (TemporalAggregate), against the walking oracle and the untiled evaluation."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip.array import DeviceArray
    from helpers import assert_result_equal
    from oracle_walk import walk
    rng = np.random.default_rng(12)

    def series(frames, dt="f4", ndv=-9999.0):
        v = (rng.random((frames, 300, 517)) * 100).astype(dt)
        v[rng.random(v.shape) < 0.1] = ndv
        return raster.MemorySource(DeviceArray.from_numpy(v), ndv, "EPSG:28992", 1.0, (0.0, 300.0), time_first=datetime(2021, 3, 1),
                                   time_delta=timedelta(hours=6) if frames > 1 else None)

    a, b = series(5), series(5, "i2", -1)
    box = dict(mode="vals", bbox=(0.0, 0.0, 517.0, 300.0), width=517, height=300, projection="EPSG:28992")
    windows = [dict(start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1)), dict(start=datetime(2021, 3, 1, 5), stop=datetime(2021, 3, 1, 19)),
               dict(start=datetime(2021, 3, 2))]
    views = [a * 2, raster.Clip(a * 2 + b, b > 40), raster.Classify(a - b, [0.0, 20.0, 50.0]), a + raster.TemporalSum(a),
             raster.TemporalAggregate(a * 2, "D", "mean"), raster.Group(a, b)]
    with gmconfig.config.set({"geomodeling.plans": plans}):
        for view in views:
            for when in windows:
                request = dict(box, **when)
                want = walk(view, request)
                for tile in (128, [200, 77]):
                    got = raster.RasterTiler(view, tile).get_data(**request)
                    assert_result_equal(got, want, 0, "%s tiled by %s, %s" % (type(view).__name__, tile, when))
                assert_result_equal(view.get_data(**request), want, 0, "%s untiled, %s" % (type(view).__name__, when))
        # requests whose width is no multiple of 4 (round 5: run over the next multiple where the rasters have the cells, the
        # result a window of a padded tile of several frames), inside the raster and at its right edge
        for x0, y0, w, h in ((3, 10, 201, 77), (65, 33, 127, 50), (0, 0, 515, 300), (310, 5, 207, 99), (1, 1, 6, 7)):
            request = dict(mode="vals", bbox=(float(x0), float(y0), float(x0 + w), float(y0 + h)), width=w, height=h,
                           projection="EPSG:28992", **windows[0])
            for view in views[:3]:
                assert_result_equal(view.get_data(**request), walk(view, request), 0, "%s on %s" % (type(view).__name__, (x0, y0, w, h)))
    # the mosaic of a multi-frame view is written in place (no paste): the tiles' results are windows of it
    with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": True}):
        out = raster.RasterTiler(a * 2, 128).get_data(**dict(box, **windows[0]))["values"]
        assert isinstance(out, DeviceArray) and out.shape == (5, 300, 517)


@pytest.mark.gpu
def test_tiler_over_pixel_local_temporal_views_is_one_request(gpu, tiler_route):
    """A RasterTiler over a temporal Block (no program to batch over the tiles) whose view is pixel-local and whose tiles cover
    the box exactly is answered by ONE request for the whole box (plan._whole_instead_of_tiles) -- the same cells as the tiles
    pasted next to each other: against the walking oracle (which pastes tiles), for every temporal Block, Group, a declared
    dtype that the values do not have (TemporalSum), time windows that miss the series; and NOT where it could differ: tiles
    that leave cells uncovered, tiles of one cell, a stencil in the view."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan, raster
    from helpers import assert_result_equal
    from oracle_walk import walk
    rng = np.random.default_rng(21)

    def series(frames, dt="f4", ndv=-9999.0, hours=6):
        v = (rng.random((frames, 48, 64)) * 100).astype(dt)
        v[rng.random(v.shape) < 0.1] = ndv
        return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, 48.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=hours))

    a, b = series(9), series(9, "i2", -1)
    box = dict(mode="vals", bbox=(0.0, 0.0, 64.0, 48.0), width=64, height=48, projection="EPSG:28992")
    windows = [dict(start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1)), dict(start=datetime(2021, 3, 1, 5), stop=datetime(2021, 3, 2, 19)),
               dict(start=datetime(2021, 3, 2)), dict(), dict(start=datetime(2022, 1, 1), stop=datetime(2022, 1, 2))]
    stores = [raster.TemporalAggregate(a * 2, "D", "mean"), raster.TemporalAggregate(raster.Clip(a, b > 40), None, "max"),
              raster.Cumulative(a - b, "sum", "D"), raster.TemporalSum(b), raster.Resample(a, "3h", "nearest"), raster.Snap(a, b),
              raster.Shift(a * 2, 3600 * 1000), raster.Group(a, a * 0.5), raster.TemporalAggregate(raster.Group(a, a * 0.5), "D", "median")]
    calls = []
    original = plan._whole_instead_of_tiles

    def spy(*args):
        calls.append(original(*args))
        return calls[-1]
    plan._whole_instead_of_tiles = spy
    try:
        with gmconfig.config.set({"geomodeling.plans": True}):
            for store in stores:
                for when in windows:
                    request = dict(box, **when)
                    for tile, whole in ((16, True), ([32, 12], True), ([20, 16], True), ([64, 48], False)):  # (ragged edge tiles still cover; one tile: nothing to elide)
                        del calls[:]
                        view = raster.RasterTiler(store, tile)
                        assert_result_equal(view.get_data(**request), walk(view, request), 0, "%s tiled by %s, %s" % (type(store).__name__, tile, when))
                        taken = bool(calls and calls[-1])
                        assert taken == whole, (type(store).__name__, tile, taken)
            # a stencil in the view, and tiles of one cell: tile by tile
            for view in (raster.RasterTiler(raster.TemporalAggregate(raster.MovingMax(a, 3), "D", "max"), 16), ):
                del calls[:]
                assert_result_equal(view.get_data(**dict(box, **windows[0])), walk(view, dict(box, **windows[0])), 0, "stencil below")
                assert not any(calls)
            # cells of 0.3 units: the sources do not lie on the request's grid (and the reference's int() truncation of the tile
            # sizes leaves cells uncovered there)
            coarse = dict(mode="vals", bbox=(0.0, 0.0, 19.2, 14.4), width=64, height=48, projection="EPSG:28992", **windows[0])
            view = raster.RasterTiler(raster.TemporalAggregate(a, None, "mean"), 4.8)
            del calls[:]
            assert_result_equal(view.get_data(**coarse), walk(view, coarse), 0, "cells of 0.3")
            assert not any(calls)
            tiny = dict(mode="vals", bbox=(0.0, 0.0, 4.0, 1.0), width=4, height=1, projection="EPSG:28992", **windows[0])
            view = raster.RasterTiler(raster.TemporalAggregate(a, None, "mean"), 1)
            del calls[:]
            assert_result_equal(view.get_data(**tiny), walk(view, tiny), 0, "tiles of one cell")
            assert not any(calls)
    finally:
        plan._whole_instead_of_tiles = original


def test_tiler_over_a_stencil_is_one_request_where_the_margins_lie_inside(gpu):
    """Dilate, MovingMax and HillShade ask their store for the box grown by exactly the margin their kernels look at, so a mosaic
    of their tiles IS their answer for the whole box: a RasterTiler over one of them (alone, or under / over pixel-local Blocks
    with a kernel of their own) is ONE request where the grown box lies on the sources' grid (plan._whole_instead_of_tiles) --
    256 tiles of a HillShade are 256 launches otherwise.  Against the walking oracle, which pastes tiles; on the plan path and
    from the template.  NOT: Smooth (the reference grows by `size` cells, the Gaussian reaches 4 / 3 of that), two stencils on a
    path (the inner one's arguments follow from a request the outer one has grown), a margin past the source's edge."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan, raster
    from helpers import assert_result_equal
    from oracle_walk import walk
    rng = np.random.default_rng(33)
    v = (rng.random((1, 96, 128)) * 12).astype("f4")
    v[rng.random(v.shape) < 0.1] = -9999.0
    A = raster.MemorySource(v, -9999.0, "EPSG:28992", 1.0, (0.0, 96.0))
    half = raster.MemorySource(v, -9999.0, "EPSG:28992", 0.5, (0.0, 48.0))
    c = rng.integers(0, 9, size=(1, 96, 128)).astype("u1")
    C = raster.MemorySource(c, 255, "EPSG:28992", 1.0, (0.0, 96.0))
    s = (rng.random((6, 96, 128)) * 50).astype("f4")
    s[rng.random(s.shape) < 0.1] = -9999.0
    S = raster.MemorySource(s, -9999.0, "EPSG:28992", 1.0, (0.0, 96.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    inner = dict(mode="vals", bbox=(16.0, 8.0, 112.0, 88.0), width=96, height=80, projection="EPSG:28992")
    when = dict(start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))
    calls = []
    original = plan._whole_instead_of_tiles

    def spy(*args):
        calls.append(original(*args))
        return calls[-1]
    plan._whole_instead_of_tiles = spy
    try:
        cases = [
            (raster.Dilate(C, [3, 5]), inner, True), (raster.MovingMax(A, 3), inner, True), (raster.MovingMax(A, 7), inner, True),
            (raster.HillShade(A), inner, True),
            (raster.HillShade(half), dict(inner, bbox=(8.0, 4.0, 56.0, 44.0)), True),                # cells of 0.5
            (raster.TemporalAggregate(raster.MovingMax(S, 5), "D", "max"), dict(inner, **when), True),  # a pixel-local store over it
            (raster.MovingMax(raster.Snap(S, S), 3), dict(inner, **when), True),                       # ... and under it
            (raster.Smooth(A, 2.0), inner, False),
            (raster.Dilate(raster.MovingMax(C, 3), [3]), inner, False),                                # two stencils on a path
            (raster.MovingMax(A, 21), inner, False),                                                   # margin 10 > the 8 rows below the box
            (raster.HillShade(A), dict(inner, bbox=(0.0, 0.0, 128.0, 96.0), width=128, height=96), False),
        ]
        for store, request, whole in cases:
            for tile in (16, [32, 12], [20, 16]):
                view = raster.RasterTiler(store, tile)
                try:
                    want = walk(view, request)
                except OverflowError:
                    # (np.full(shape, 256, uint8), parallelize.py:108-112 over a HillShade: NumPy 2 refuses the fill value the
                    #  reference declares; the mosaic has no cell to fill here and is the untiled answer, tested against the oracle elsewhere)
                    assert isinstance(store, raster.HillShade)
                    want = _graph_path(store, request)
                what = "%s(%s) tiled by %s" % (type(store).__name__, type(store.args[0]).__name__, tile)
                del calls[:]
                with gmconfig.config.set({"geomodeling.plans": True}):
                    assert_result_equal(view.get_data(**request), want, 0, what + " (plans)")
                assert bool(calls and calls[-1]) == whole, (what, calls)
                del calls[:]
                assert_result_equal(_template_path(view, request), want, 0, what + " (template)")
                assert bool(calls and calls[-1]) == whole, (what, calls, "template")
                assert_result_equal(_graph_path(view, request), want, 0, what + " (graph from scratch)")
    finally:
        plan._whole_instead_of_tiles = original


def test_tiler_in_small_tiles_over_an_element_wise_store_is_one_request(gpu, monkeypatch):
    """Element-wise stores under a RasterTiler are launched once over all tiles, written into the mosaic (fuse.TiledAssembly) --
    at the kernel's rate for tiles of 2048^2 cells and more; around SMALL tiles it is the planning per tile that costs, and a
    pixel-local view over sources on the request's grid is asked once for the whole box instead (the suites switch that off to
    keep their small tiles on the batched route: conftest.py).  Both ways the walking oracle's mosaic, on the plan path, from
    the template and as a graph from scratch; series of several frames, a declared dtype, tiles that leave cells uncovered."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import plan, raster
    from helpers import assert_result_equal
    from oracle_walk import walk
    rng = np.random.default_rng(44)
    v = (rng.random((5, 48, 64)) * 100).astype("f4")
    v[rng.random(v.shape) < 0.1] = -9999.0
    T = raster.MemorySource(v, -9999.0, "EPSG:28992", 1.0, (0.0, 48.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    w = rng.integers(0, 50, size=(5, 48, 64)).astype("i2")
    W = raster.MemorySource(w, -1, "EPSG:28992", 1.0, (0.0, 48.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    box = dict(mode="vals", bbox=(0.0, 0.0, 64.0, 48.0), width=64, height=48, projection="EPSG:28992",
               start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))
    stores = [T * 2, raster.Clip(T, T > 20), (T + W) / 3, raster.Classify(T, [10, 50, 90]), W * 2 > 30, raster.FillNoData(T, T * 0.5)]
    calls = []
    original = plan._whole_instead_of_tiles

    def spy(*args):
        calls.append(original(*args))
        return calls[-1]
    monkeypatch.setattr(plan, "_whole_instead_of_tiles", spy)
    for cells, elided in ((1 << 22, True), (0, False), (16 * 16, False), (16 * 16 + 1, True)):
        monkeypatch.setattr(plan, "_BATCHED_TILE_CELLS", cells)
        for store in stores:
            for tile, covers in ((16, True), ([32, 12], True), ([64, 48], False)):
                if cells in (16 * 16, 16 * 16 + 1) and tile != 16:
                    continue
                view = raster.RasterTiler(store, tile)
                want = walk(view, box)
                what = "%s tiled by %s, batched from %d cells" % (type(store).__name__, tile, cells)
                del calls[:]
                with gmconfig.config.set({"geomodeling.plans": True}):
                    assert_result_equal(view.get_data(**box), want, 0, what + " (plans)")
                assert bool(calls and calls[-1]) == (elided and covers), (what, calls)
                assert_result_equal(_template_path(view, box), want, 0, what + " (template)")
                assert_result_equal(_graph_path(view, box), want, 0, what + " (graph from scratch)")
    # cells of 0.3 units: the reference's int() truncation leaves cells uncovered -- tile by tile whatever the threshold
    monkeypatch.setattr(plan, "_BATCHED_TILE_CELLS", 1 << 22)
    coarse = dict(box, bbox=(0.0, 0.0, 19.2, 14.4))
    view = raster.RasterTiler(T * 2, 4.8)
    del calls[:]
    assert_result_equal(view.get_data(**coarse), walk(view, coarse), 0, "cells of 0.3")
    assert not any(calls)
