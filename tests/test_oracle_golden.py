"""Pin the CPU oracle (oracle/ref_numpy.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import assert_result_equal, assert_values_equal, load_family
from oracle import ref_numpy as O

# transcendental ufuncs may dispatch to different SIMD kernels on another host CPU
ULP_OPS = {"power": 1, "exp": 1, "log": 1, "log10": 1}


def _run_or_raises(case, fn):
    if "raises" in case:
        with pytest.raises(Exception) as ei:
            with np.errstate(all="ignore"):
                fn()
        assert type(ei.value).__name__ == case["raises"]
        return None
    with np.errstate(all="ignore"):
        return fn()


@pytest.mark.parametrize("case", load_family("math"), ids=lambda c: c["id"] + "-" + c["op"])
def test_math(case):
    out = _run_or_raises(case, lambda: O.math_process(case["op"], case["kwargs"], *case["args"]))
    if "raises" not in case:
        assert_result_equal(out, case["expect"], ULP_OPS.get(case["op"], 0), case["id"])


@pytest.mark.parametrize("case", load_family("n1"), ids=lambda c: c["id"] + "-" + c["op"])
def test_n1(case):
    op, args = case["op"], case["args"]
    if op in O.UFUNCS:
        out = O.math_process(op, case["kwargs"], *args)
    elif op == "fillnodata":
        out = O.fillnodata_process(case["kwargs"], *args)
    else:
        out = getattr(O, op + "_process")(*args)
    assert_result_equal(out, case["expect"], ULP_OPS.get(op, 0), case["id"])


@pytest.mark.parametrize("case", load_family("clip"), ids=lambda c: c["id"])
def test_clip(case):
    out = O.clip_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])
    # all-no-data early-out hands back the very same object (misc.py:108-109)
    data = case["args"][0]
    if data is not None and "values" in data and np.all(data["values"] == data["no_data_value"]):
        assert out is data


@pytest.mark.parametrize("case", load_family("mask"), ids=lambda c: c["id"])
def test_mask(case):
    out = _run_or_raises(case, lambda: O.mask_process(*case["args"]))
    if "raises" not in case:
        assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("classify"), ids=lambda c: c["id"])
def test_classify(case):
    assert_result_equal(O.classify_process(*case["args"]), case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("reclassify") if c["op"] == "reclassify"],
                         ids=lambda c: c["id"])
def test_reclassify(case):
    out = _run_or_raises(case, lambda: O.reclassify_process(*case["args"]))
    if "raises" not in case:
        assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("tiler"), ids=lambda c: c["id"])
def test_tiler(case):
    if case["op"] == "tiler_process":
        kw = dict(case["kwargs"])
        kw["dtype"] = np.dtype(kw["dtype"])
        assert_result_equal(O.tiler_process(kw, *case["tiles"]), case["expect"], 0, case["id"])
        return
    tile = case["tile_size"]
    tile = [int(x) for x in tile] if hasattr(tile, "__iter__") else [int(tile)] * 2
    kw, reqs = O.tiler_requests(tile, case["request"])
    expect = case["expect"]
    if expect[0][0] == "none":
        assert kw is None and len(reqs) == 1
        assert reqs[0] == expect[1][1]
        return
    ekw = expect[0][1]
    for k in ("shape_yx", "count_xy", "tilesize_xy"):
        assert tuple(kw[k]) == tuple(ekw[k])
    assert len(reqs) == len(expect) - 1
    for got, (_, want) in zip(reqs, expect[1:]):
        assert tuple(float(x) for x in got["bbox"]) == tuple(want["bbox"])
        assert got["width"] == want["width"] and got["height"] == want["height"]


def test_dtype_helpers():
    cases = {c["id"]: c["expect"] for c in load_family("dtypes")}
    for n, want_int, want_uint in cases["dtype_intuint"]:
        for fn, want in ((O.get_int_dtype, want_int), (O.get_uint_dtype, want_uint)):
            try:
                got = fn(n).str
            except Exception as e:
                got = "ERR:" + type(e).__name__
            assert got == want, (n, fn.__name__)
    for dt, mx, mn in cases["dtype_maxmin"]:
        assert O.get_dtype_max(dt) == mx and type(O.get_dtype_max(dt)) is type(mx)
        assert O.get_dtype_min(dt) == mn
    for value, dt, fill in cases["dtype_mask"]:
        assert O.mask_dtype(value).str == dt
        assert (1 if value == 0 else 0) == fill
    for cls_name, da, (kind, ob), order, want, fill in cases["dtype_math"]:
        a = np.dtype(da)
        b = np.dtype(ob) if kind == "r" else ob
        args = [a, b] if order == 0 else [b, a]
        if cls_name == "Power" and isinstance(args[1], int) and args[1] < 0:
            args[1] = float(args[1])  # Power.__init__, elemwise.py:401-406
        try:
            got = O.elemwise_dtype(args, divide=(cls_name == "Divide"))
        except Exception as e:
            assert want.startswith("ERR"), (cls_name, da, ob, e)
            continue
        if want.startswith("ERR"):
            continue  # construction-time type errors are host-logic, tested there
        assert got.str == want, (cls_name, da, kind, ob, order)
        assert O.get_dtype_max(got) == fill


@pytest.mark.parametrize("case", load_family("smoothzoom"), ids=lambda c: c["id"])
def test_affine_nearest(case):
    values, zoom, offset = case["args"]
    assert_values_equal(O.affine_nearest(values, zoom, offset), case["expect"], 0, case["id"])


# ---- the reference's own known-answer tests for the GDAL nearest warp ----------
# expected arrays restated from tests/test_raster_sources.py:68-161 (source: one
# 1x1 pixel of value 5 covering [136700,136705) x (455795,455800], ndv 255)
GT = (136700.0, 5.0, 0, 455800.0, 0, -5.0)
PIX = np.full((1, 1, 1), 5, dtype="u1")


@pytest.mark.parametrize("bbox,h,w,expect", [
    ((136700, 455795, 136705, 455800), 1, 1, [[[5]]]),
    ((136700, 455790, 136705, 455795), 1, 1, [[[255]]]),
    ((136695, 455795, 136700, 455800), 1, 1, [[[255]]]),
    ((136700, 455800, 136705, 455805), 1, 1, [[[255]]]),
    ((136705, 455795, 136710, 455800), 1, 1, [[[255]]]),
    ((136700, 455795, 136710, 455800), 1, 2, [[[5, 255]]]),
    ((136700, 455790, 136705, 455800), 2, 1, [[[5], [255]]]),
    ((136700, 455795, 136710, 455800), 2, 4, [[[5, 5, 255, 255], [5, 5, 255, 255]]]),
    ((136700, 455795, 136705, 455800), 5, 5, np.full((1, 5, 5), 5).tolist()),
])
def test_nearest_warp_reference_kats(bbox, h, w, expect):
    out = O.nearest_warp(PIX, GT, bbox, h, w, 255)
    assert out.tolist() == expect


@pytest.mark.parametrize("dx,dy,hit", [
    (0, 0, True), (0, -4.99, True), (4.99, 0, True), (4.99, -4.99, True),
    (0, -5.0, False), (5.0, 0, False), (-5.0, 5.0, False), (-0.01, 0, False), (0, 0.01, False),
])
def test_point_read_reference_kats(dx, dy, hit):
    out = O.point_read(PIX, GT, 136700 + dx, 455800 + dy, 255)
    assert out.item() == (5 if hit else 255)


def test_bilinear_vs_scipy():
    """Own spec (parity unpinned): agree with SciPy's order-1 spline within 1 ulp (f32)."""
    from scipy import ndimage
    from helpers import ulp_distance

    rng = np.random.default_rng(3)
    for (hs, ws, hd, wd) in ((16, 20, 32, 40), (9, 7, 27, 21), (10, 10, 15, 15), (8, 8, 8, 8), (12, 6, 5, 4)):
        src = (rng.random((2, hs, ws)) * 100).astype("f4")
        got = O.bilinear_resample(src, hd, wd, np.float32(3.4028235e38))
        v = (np.arange(hd) + 0.5) * (hs / hd) - 0.5
        u = (np.arange(wd) + 0.5) * (ws / wd) - 0.5
        vv, uu = np.meshgrid(v, u, indexing="ij")
        for b in range(2):
            want = ndimage.map_coordinates(src[b].astype("f8"), [vv, uu], order=1, mode="nearest")
            assert ulp_distance(got[b], want.astype("f4")).max() <= 1


# ---------------------------------------------------------------------------------
# N3 / N4: stencils and the Max reduction
# ---------------------------------------------------------------------------------
def hillshade_numpy_dependence(data, kwargs):
    """(cells, cells where the oracle with NumPy's own float32 arctan2 / sin differs from the oracle with the
    correctly rounded ones, max |difference|): the part of HillShade's output that depends on the NumPy
    build and host."""
    a = O.hillshade_process(data, kwargs)["values"].astype("i2")
    b = O.hillshade_process(data, kwargs, trig="rounded")["values"].astype("i2")
    return a.size, int((a != b).sum()), int(np.abs(a - b).max(initial=0))


def test_hillshade_trig_rounding_only_moves_truncation_boundaries():
    """HillShade ends in float32 arctan2 -> sin -> ... -> truncating uint8 cast.  NumPy's float32 arctan2 /
    sin are not correctly rounded (SVML on AVX512 hosts, glibc elsewhere: 38 % / 24 % of arguments are 1 ulp
    off on the host this was written on), so the reference's own output depends on the host in a few cells
    per million.  The device evaluates the host-independent variant (trig="rounded"); this pins how far the
    two are apart: never more than 1, and on fewer than 20 cells per million."""
    rng = np.random.default_rng(42)
    cells = differ = 0
    for scale, dt in ((1.0, "f4"), (30.0, "f4"), (200.0, "i2"), (3.0, "f8")):
        v = (rng.standard_normal((1, 1026, 1026)) * scale)
        v = np.cumsum(v, axis=2).astype(dt) if dt != "i2" else np.round(v).astype(dt)
        data = {"values": v, "no_data_value": np.dtype(dt).type(-9999)}
        for alt, az in ((45.0, 315.0), (30.0, 10.0)):
            n, d, worst = hillshade_numpy_dependence(data, dict(resolution=(0.5, 2.0), altitude=alt, azimuth=az, fill=1))
            assert worst <= 1
            cells, differ = cells + n, differ + d
    print("hillshade: %d of %d cells depend on the host's float32 arctan2 / sin (%.2e)" % (differ, cells, differ / cells))
    assert differ <= 20e-6 * cells


@pytest.mark.parametrize("size", [3, 4, 5, 7, 11, 15, 18])
@pytest.mark.parametrize("dt", ["f4", "f8", "i2"])
def test_movingmax_is_scipys_filter_including_nan(size, dt):
    """MovingMax.process calls scipy.ndimage.maximum_filter (spatial.py:205), which is not in /root/reference:
    the oracle restates its published algorithm (the first footprint cell, then every cell that compares
    greater) and is pinned here against SciPy itself, NaN cells and signed zeros included (a NaN only
    survives as the first cell of the footprint; bit patterns compared, not values)."""
    from scipy import ndimage
    rng = np.random.default_rng(int(size * 10))
    v = np.round(rng.standard_normal((2, 60, 70)) * 3, 1)
    v[rng.random(v.shape) < 0.1] = 0.0
    v[rng.random(v.shape) < 0.1] = -0.0
    v = v.astype(dt)
    ndv = np.dtype(dt).type(-9999)
    v[rng.random(v.shape) < 0.1] = ndv
    full_square = bool(O.get_footprint(size).all())
    if dt != "i2" and not full_square:
        # (an all-true footprint -- size 3 -- makes SciPy filter axis by axis with a running-maximum queue whose
        #  output after a NaN depends on the cells met before it along the whole line: no rule to restate)
        v[rng.random(v.shape) < 0.03] = np.nan
    data = {"values": v, "no_data_value": ndv}
    radius = int(size // 2)
    array = v.copy()
    minimum = O.get_dtype_min(array.dtype)
    mask = array == ndv
    array[mask] = minimum
    want = ndimage.maximum_filter(array, footprint=O.get_footprint(size)[np.newaxis])
    want[(want == minimum) & mask] = ndv
    want = want[:, radius:-radius, radius:-radius]
    got = O.movingmax_process(data, size)["values"]
    assert got.dtype == want.dtype
    assert got.tobytes() == want.tobytes()  # (full squares too: the queue keeps the LAST of equal cells, restated)


@pytest.mark.parametrize("case", load_family("spatial"), ids=lambda c: c["id"] + "-" + c["op"])
def test_spatial(case):
    op, args = case["op"], case["args"]
    if op == "expand":
        assert O.expand_request_pixels(*args) == case["expect"]
        return
    out = getattr(O, op + "_process")(*args)
    if op == "hillshade" and isinstance(out, dict) and "values" in out:
        # the host-independent variant reproduces every reference vector exactly; NumPy's own trig may move a
        # cell by 1 on another host (see the test above)
        assert_result_equal(O.hillshade_process(*args, trig="rounded"), case["expect"], 0, case["id"])
        n, differ, worst = hillshade_numpy_dependence(*args)
        assert worst <= 1 and differ <= max(1, n // 1000), case["id"]
    else:
        assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("reduce"), ids=lambda c: c["id"])
def test_reduce_max(case):
    out = O.reduce_max_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("reducestats"), ids=lambda c: c["id"] + "-" + c["op"])
def test_reduce_rasters(case):
    with np.errstate(all="ignore"):
        out = O.reduce_rasters([dict(d) for d in case["stack"]], case["op"], **case["kwargs"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("smooth"), ids=lambda c: c["id"])
def test_smooth(case):
    out = O.smooth_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("temporal"), ids=lambda c: c["id"] + "-" + c["kind"])
def test_temporal(case):
    """TemporalAggregate.process / Cumulative.process on the arguments the reference built."""
    func = O.temporal_aggregate_process if case["kind"] == "aggregate" else O.cumulative_process
    out = func(*case["process_args"])
    want = case["expect"]
    if want is None or "values" not in want:
        assert out == want, case["id"]
    else:
        assert_result_equal(out, want, 0, case["id"])


ROUND4_PROCESS = {"snap": "snap_process", "shift": "shift_process", "temporalsum": "temporal_sum_process",
                  "resample": "resample_process", "place": "place_process"}


@pytest.mark.parametrize("case", load_family("round4"), ids=lambda c: c["id"] + "-" + c["kind"])
def test_round4(case):
    """Snap / Shift / TemporalSum / Resample / Place: the oracle's process() on the arguments the reference built."""
    func = getattr(O, ROUND4_PROCESS[case["kind"]])
    with np.errstate(all="ignore"):
        out = func(*case["process_args"])
    want = case["expect"]
    if want is None or "values" not in want:
        assert out == want, case["id"]
    else:
        assert_result_equal(out, want, 0, case["id"])


@pytest.mark.parametrize("case", load_family("group"), ids=lambda c: c["id"] + "-" + c["setup"] + "-" + c["request"]["mode"])
def test_group_walked(case):
    """Group over temporal sources, every request mode: the view walked with the reference's planning and the oracle's
    group_process (tests/oracle_walk.py) against what the reference's own Group answered."""
    from blocks_util import TemporalLeaf
    from dask_geomodeling_amd import raster
    from oracle_walk import walk
    leaves = []
    for spec in case["sources"]:
        TemporalLeaf.ARRAYS[spec["name"]] = {"values": spec["data"], "no_data_value": spec["ndv"]}
        leaves.append(TemporalLeaf(spec["name"], spec["dtype"], spec["ndv"], spec["times"], spec["delta"]))
    out = walk(raster.Group(*leaves), case["request"])
    if case["expect"] is None:
        assert out is None
    else:
        assert_result_equal(out, case["expect"], 0, case["id"])


# ---------------------------------------------------------------------------------
# N4: zonal statistics (AggregateRaster's numeric core)
# ---------------------------------------------------------------------------------
def zonal_runs():
    for case in load_family("zonal"):
        for k, run in enumerate(case["runs"]):
            yield pytest.param(case, run, id="%s-%s%s%s" % (case["id"], run["statistic"], run["percentile"] or "",
                                                           "-thr" if run["thresholds"] else ""))


@pytest.mark.parametrize("case,run", list(zonal_runs()))
def test_zonal(case, run):
    """The oracle's aggregate_labelled against the reference's own aggregate_polygons (real scipy.ndimage and
    measurements.percentile; only GDAL's rasterization replaced by painted masks, tests/golden/make_golden.py):
    the float32 result array bit for bit, the geometries without cells, and the TypeError of a boolean percentile."""
    thresholds = case["threshold_values"] if run["thresholds"] else None
    args = (case["buckets"], case["n"], case["values"], case["no_data_value"], thresholds, run["statistic"], run["percentile"])
    if "raises" in run:
        with pytest.raises(TypeError):
            O.aggregate_labelled(*args)
        return
    agg, no_cells = O.aggregate_labelled(*args)
    assert_values_equal(agg, run["expect"], 0, case["id"])
    assert no_cells == run["no_cells"]


def test_zonal_label_statistic_is_scipys():
    """scipy.ndimage itself (the reference's STATISTICS table) on a larger labelled raster: long labels,
    NaN cells, signed zeros -- sum / mean bit for bit (same bincount order), the picked statistics by value."""
    from scipy import ndimage
    rng = np.random.default_rng(8)
    labels = (rng.integers(0, 40, size=(1, 1))[0, 0] + np.add.outer(np.arange(300) // 23, 13 * (np.arange(400) // 31))).astype(np.int32)
    for dt in ("f4", "f8", "i2", "u1"):
        values = (rng.standard_normal(labels.shape) * 1000).astype(dt)
        if dt in ("f4", "f8"):
            values[rng.random(labels.shape) < 0.01] = np.nan
            values[rng.random(labels.shape) < 0.05] = -0.0
        index = np.unique(labels)
        for name, fn in (("sum", ndimage.sum), ("mean", ndimage.mean), ("min", ndimage.minimum), ("max", ndimage.maximum),
                         ("median", ndimage.median)):
            want = np.asarray(fn(values, labels=labels, index=index))
            got = np.asarray(O.zonal_label_statistic(values.ravel(), labels.ravel(), index, name))
            if name in ("sum", "mean"):
                assert got.dtype == want.dtype and got.tobytes() == want.tobytes(), (dt, name)
            else:
                assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (dt, name)
        want = np.asarray(ndimage.sum(1, labels=labels, index=index))
        assert np.array_equal(O.zonal_label_statistic(values.ravel(), labels.ravel(), index, "count"), want)


# ---------------------------------------------------------------------------------
# round 3: what the device refused until then (Dilate with many values, HillShade / Smooth on bool and uint64
# rasters, reduce_rasters into a 1-byte dtype from wider rasters), produced by the reference itself
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("case", load_family("round3"), ids=lambda c: c["id"] + "-" + c["op"])
def test_round3(case):
    op = case["op"]
    if op.startswith("reduce:"):
        out = O.reduce_rasters([dict(d) for d in case["stack"]], op.split(":")[1], **case["kwargs"])
        want = case["expect"]
        if op.endswith("median") or op.split(":")[1].startswith("p"):
            from fuzz_util import unsigned_zero
            out, want = unsigned_zero(out), unsigned_zero(want)
        assert_result_equal(out, want, 0, case["id"])
    elif op == "hillshade":
        assert_result_equal(O.hillshade_process(*case["args"], trig="rounded"), case["expect"], 0, case["id"])
    else:
        assert_result_equal(getattr(O, op + "_process")(*case["args"]), case["expect"], 0, case["id"])
