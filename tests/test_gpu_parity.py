"""GPU parity: the HIP path (through the C-ABI) against the reference-generated golden
vectors and against the CPU oracle on seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest

from helpers import ulp_distance, assert_result_equal, assert_values_equal, load_family, routes_of
from dask_geomodeling_amd.raster import devops
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu

# Budgets against the reference-generated vectors, MEASURED on the MI355X (tools/ulp_measure.py,
# profiles/r03_ulp.log), not allowances:
#  * Power: float64 pow rounded once vs NumPy's faithful powf / pow: 1 ulp (north_star tolerance);
#  * Exp / Log / Log10 on float32: the device result IS the correctly rounded value (0 ulp from the function
#    evaluated in extended precision and rounded once: test_exp_log_are_correctly_rounded below); NumPy's own
#    float32 SIMD loops -- the reference -- are 2 ulp off it on these vectors (n10039, n10047, n10049), so 2 is the
#    distance between the reference and the truth, not a device error;
#  * on float64 the device (ocml) and NumPy (libm) are each within 1 ulp of the truth and within 1 ulp of each other.
ULP_OPS = {"power": 1, "exp": 2, "log": 2, "log10": 2}
ULP_F64 = {"power": 1, "exp": 1, "log": 1, "log10": 1}


def _ulp(op, expect):
    values = expect.get("values") if isinstance(expect, dict) else None
    table = ULP_F64 if getattr(values, "dtype", None) == np.dtype("f8") else ULP_OPS
    return table.get(op, 0)


@pytest.mark.parametrize("case", load_family("math"), ids=lambda c: c["id"] + "-" + c["op"])
def test_math_golden(gpu, case):
    if "raises" in case:
        pytest.skip("reference raises %s" % case["raises"])
    out = devops.math_process(case["op"], case["kwargs"], *case["args"])
    assert_result_equal(out, case["expect"], _ulp(case["op"], case["expect"]), case["id"])


def test_power_float32_values_and_special_cases(gpu):
    """np.power on float32 rasters (powf_via_f64, csrc/gm_vm.h): every pair of 31 special bases / exponents (zeros of both
    signs, +-1, +-inf, NaN, subnormals, odd / even / huge integers, values next to 1) agrees with the host's np.power through the
    oracle's wrapper -- finite where it is finite, the same sign of zero, <= 1 ulp -- and sweeps over bases near 1 with huge
    exponents, subnormal bases, negative bases with integer exponents and results in the subnormal range stay within 1 ulp of
    the oracle and equal the extended-precision value rounded once (measured: every one of 8 M samples, tools/ulp_measure.py)."""
    fill = float(np.finfo("f4").max)
    kw = {"dtype": "float32", "fillvalue": fill}
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 2.5, -2.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.0e38,
                        -3.0e38, 16777216.0, 16777217.0, -16777216.0, 33554432.0, 1e10, -1e10, 0.99999994, 1.0000001, 7.0, -7.0, 1e-20,
                        1e20], dtype="f4")
    xs, ys = [a.reshape(1, len(special), len(special)).copy() for a in np.meshgrid(special, special)]
    rng = np.random.default_rng(8)
    n = 1 << 16
    sweeps = [(xs, ys),
              (10.0 ** rng.uniform(-10, 10, n), rng.uniform(-8, 8, n)),
              (rng.uniform(0.5, 2, n), rng.uniform(-120, 120, n)),
              (np.float32(1) + rng.integers(-64, 65, n).astype("f4") * np.float32(2 ** -24), rng.uniform(-1e9, 1e9, n)),
              (rng.uniform(1e-45, 1e-38, n), rng.uniform(-1, 1, n)),
              (-(10.0 ** rng.uniform(-3, 3, n)), rng.integers(-20, 21, n).astype("f8")),
              (rng.uniform(1e-3, 1e-1, n), rng.uniform(15, 45, n))]
    for k, (x, y) in enumerate(sweeps):
        x, y = np.asarray(x, dtype="f4").reshape(1, 1, -1), np.asarray(y, dtype="f4").reshape(1, 1, -1)
        a, b = {"values": x, "no_data_value": fill}, {"values": y, "no_data_value": fill}
        with np.errstate(all="ignore"):
            want = O.math_process("power", kw, a, b)
            true = np.power(x.astype(np.longdouble), y.astype(np.longdouble)).astype("f4")
        got = devops.math_process("power", kw, a, b)
        assert_result_equal(got, want, 1, "sweep %d" % k)
        values = np.asarray(got["values"])
        assert np.array_equal(np.signbit(values), np.signbit(want["values"])), "sign of zero, sweep %d" % k
        finite = np.isfinite(true) & (values != np.float32(fill))
        assert np.array_equal(values[finite], true[finite]), "sweep %d: not the correctly rounded value" % k


def test_integer_power_with_raster_exponent(gpu):
    """Integer ** integer raster: computed (wrapping like NumPy) when every exponent is >= 0, ValueError like
    NumPy's as soon as one is negative -- no-data cells included; inside a view the op is kept out of fused
    programs so that the check runs."""
    from dask_geomodeling_amd import raster
    from blocks_util import ArrayLeaf
    rng = np.random.default_rng(3)
    a = {"values": rng.integers(-4, 9, (1, 24, 40)).astype("i4"), "no_data_value": -9999}
    b = {"values": rng.integers(0, 12, (1, 24, 40)).astype("i4"), "no_data_value": 77}
    kw = {"dtype": "int32", "fillvalue": 2147483647}
    assert_result_equal(devops.math_process("power", kw, a, b), O.math_process("power", kw, a, b), 0, "i4 ** i4")
    ArrayLeaf.ARRAYS["pw/a"], ArrayLeaf.ARRAYS["pw/b"] = a, b
    A, B = ArrayLeaf("pw/a", "<i4", -9999), ArrayLeaf("pw/b", "<i4", 77)
    from fuzz_util import REQ
    view = raster.Add(raster.Power(A, B), 1)
    x = O.math_process("power", kw, a, b)
    assert_result_equal(view.get_data(**REQ), O.math_process("add", kw, x, 1), 0, "Power inside a view")
    b["values"][0, 3, 3] = -1
    with pytest.raises(ValueError, match="negative integer powers"):
        devops.math_process("power", kw, a, b)
    with pytest.raises(ValueError, match="negative integer powers"):
        view.get_data(**REQ)


@pytest.mark.parametrize("case", [c for c in load_family("n1") if c["op"] in devops.P.OPCODES],
                         ids=lambda c: c["id"] + "-" + c["op"])
def test_n1_wrapper_golden(gpu, case):
    op = case["op"]
    if op in ("invert", "isdata", "isnodata"):
        out = devops.unary_bool_process(op, case["args"][0])
    elif op == "maskbelow":
        out = devops.maskbelow_process(*case["args"])
    elif op == "step":
        out = devops.step_process(*case["args"])
    elif op == "fillnodata":
        out = devops.fillnodata_process(case["kwargs"], *case["args"])
    else:
        out = devops.math_process(op, case["kwargs"], *case["args"])
    assert_result_equal(out, case["expect"], _ulp(op, case["expect"]), case["id"])


@pytest.mark.parametrize("dt", ["f4", "f8"])
@pytest.mark.parametrize("op", ["exp", "log", "log10"])
def test_exp_log_are_correctly_rounded(gpu, op, dt):
    """The device against the truth instead of against NumPy: the function in x87 extended precision (64-bit
    significand) rounded once.  float32: bit-exact (the kernel evaluates in float64 and rounds once); float64:
    within 1 ulp (measured maximum, tools/ulp_measure.py).  450 000 arguments over 60 decades, both signs."""
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 long double"
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-80, 80, 200000), rng.uniform(0, 1e-3, 50000),
                        10.0 ** rng.uniform(-30, 30, 200000)]).astype(dt).reshape(1, 450, 1000)
    fill = float(np.finfo(dt).max)
    f = {"exp": np.exp, "log": np.log, "log10": np.log10}[op]
    with np.errstate(all="ignore"):
        true = f(x.astype(np.longdouble)).astype(dt)
    got = np.asarray(devops.math_process(op, {"dtype": np.dtype(dt).name, "fillvalue": fill},
                                         {"values": x, "no_data_value": fill})["values"])
    data = np.isfinite(true)
    assert (got[~data] == np.dtype(dt).type(fill)).all()  # log of a negative number / overflow -> no data
    d = ulp_distance(got[data], true[data])
    assert d.max() <= (0 if dt == "f4" else 1), "%s %s: %d ulp from the correctly rounded value" % (op, dt, d.max())


@pytest.mark.parametrize("case", load_family("clip"), ids=lambda c: c["id"])
def test_clip_golden(gpu, case):
    out = devops.clip_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("mask"), ids=lambda c: c["id"])
def test_mask_golden(gpu, case):
    if "raises" in case:
        with pytest.raises(Exception) as ei:
            devops.mask_process(*case["args"])
        assert type(ei.value).__name__ == case["raises"]
        return
    out = devops.mask_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 255, 1023, 1024, 1025, 4099, 262144 + 7])
@pytest.mark.parametrize("vec", [4, 8])
def test_ragged_sizes_and_tuning(gpu, n, vec):
    """Tail handling (n % 4, n smaller than a block) for both lane widths, all launch knobs."""
    from dask_geomodeling_amd.hip import program
    rng = np.random.default_rng(n)
    a = (rng.random((1, 1, n)) * 100).astype("f4")
    b = (rng.random((1, 1, n)) * 100).astype("f4")
    fill = np.float32(np.finfo("f4").max)
    a[..., ::7] = fill
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    want = O.math_process("multiply", kw, {"values": a, "no_data_value": fill},
                          {"values": b, "no_data_value": fill})
    program.set_tuning(vec=vec)
    try:
        got = devops.math_process("multiply", kw, {"values": a, "no_data_value": fill},
                                  {"values": b, "no_data_value": fill})
    finally:
        program.set_tuning()
    assert_result_equal(got, want, 0, "n=%d vec=%d" % (n, vec))


@pytest.mark.gpu
@pytest.mark.parametrize("n_tiles", [1, 2, 16, 17, 33])
@pytest.mark.parametrize("n", [5, 1024, 4099, 65536 + 3])
def test_batched_launch_matches_per_tile(gpu, n_tiles, n):
    """gm_elemwise_run_batch: many tiles per launch (blockIdx.y = tile), ragged sizes, more tiles
    than one launch takes; every tile must equal the oracle's result for that tile."""
    from dask_geomodeling_amd.hip import DeviceArray, program as P
    fill = np.float32(np.finfo("f4").max)
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    leaves = [P.Leaf(0, "f4", fill), P.Leaf(1, "f4", fill)]
    x = devops.math_node("add", kw, leaves)
    x = devops.math_node("multiply", kw, [x, P.Lit(2.5)])
    x = devops.math_node("subtract", kw, [x, leaves[1]])
    x = devops.clip_node(x, leaves[0])
    comp = P.compile_expr(x, 2)
    tiles, want = [], []
    for t in range(n_tiles):
        rng = np.random.default_rng(1000 * n_tiles + t)
        a = (rng.random((1, 1, n)) * 100).astype("f4")
        b = (rng.random((1, 1, n)) * 100).astype("f4")
        a[..., t % 5::11] = fill
        b[..., ::13] = fill
        A = {"values": a, "no_data_value": fill}
        B = {"values": b, "no_data_value": fill}
        w = O.math_process("add", kw, A, B)
        w = O.math_process("multiply", kw, w, 2.5)
        w = O.math_process("subtract", kw, w, B)
        want.append(O.clip_process(w, A))
        tiles.append([DeviceArray.from_numpy(a), DeviceArray.from_numpy(b)])
    outs = P.launch_batch(comp, tiles, (1, 1, n))
    for t in range(n_tiles):
        assert_values_equal(outs[t].to_numpy(), want[t]["values"], 0, "tile %d of %d, n=%d" % (t, n_tiles, n))


def test_device_array_roundtrip_and_checksum(gpu):
    from dask_geomodeling_amd.hip import DeviceArray
    rng = np.random.default_rng(0)
    for dt in ("u1", "i2", "i4", "i8", "f4", "f8", "bool"):
        x = (rng.random((2, 33, 17)) * 100).astype(dt)
        d = DeviceArray.from_numpy(x)
        assert_values_equal(d.to_numpy(), x)
        assert d.checksum() == DeviceArray.from_numpy(x.copy()).checksum()
        if x.size:
            y = x.copy().reshape(-1)
            y[5] = y[5] + 1 if dt != "bool" else ~y[5]
            assert DeviceArray.from_numpy(y.reshape(x.shape)).checksum() != d.checksum()


# Both routes of the table lookups are pinned to the golden vectors: "program" = the in-program
# op (GM_OP_DIGITIZE / GM_OP_LOOKUP, taken whenever the table is small / dense enough), "kernel" =
# the stand-alone LDS kernels (gm_classify2 / gm_reclassify), forced here for every case.
@pytest.mark.parametrize("route", ["program", "kernel"])
@pytest.mark.parametrize("case", load_family("classify"), ids=lambda c: c["id"])
def test_classify_golden(gpu, case, route, monkeypatch):
    from dask_geomodeling_amd.raster import tables
    monkeypatch.setattr(tables, "_PREFER_PROGRAM", route == "program")
    out = tables.classify_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("route", ["program", "kernel"])
@pytest.mark.parametrize("case", [c for c in load_family("reclassify") if c["op"] == "reclassify"],
                         ids=lambda c: c["id"])
def test_reclassify_golden(gpu, case, route, monkeypatch):
    from dask_geomodeling_amd.raster import tables
    if "raises" in case:
        pytest.skip("reference raises %s" % case["raises"])
    monkeypatch.setattr(tables, "_PREFER_PROGRAM", route == "program")
    out = tables.reclassify_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["program", "kernel"])
@pytest.mark.parametrize("n", [7, 4096, 70001])
@pytest.mark.parametrize("in_dtype", ["u1", "i2", "i4", "i8"])
def test_table_ops_big_tiles_vs_oracle(gpu, n, in_dtype, route, monkeypatch):
    """Random tiles through Classify (few and many edges, both sides) and Reclassify (dense window +
    far no-data key, sparse keys, select on/off, float targets) on both routes, against the oracle."""
    from dask_geomodeling_amd.raster import tables
    monkeypatch.setattr(tables, "_PREFER_PROGRAM", route == "program")
    rng = np.random.default_rng(n)
    ndv = int(np.iinfo(in_dtype).max)
    v = rng.integers(0, 120, size=(1, 1, n)).astype(in_dtype)
    v[..., ::11] = ndv
    data = {"values": v, "no_data_value": ndv}
    tabs = [[[k, (7 * k) % 50] for k in range(0, 120, 2)], [[k * 1000 % 127, k] for k in range(1, 40)],
            [[k, k * 0.5] for k in range(0, 100, 3)]]
    for table in tabs:
        dt = np.asarray([t for _, t in table]).dtype
        for select in (False, True):
            kw = {"dtype": dt.str, "fillvalue": O.get_dtype_max(dt) if hasattr(O, "get_dtype_max") else
                  (np.iinfo(dt).max.item() if dt.kind == "i" else np.finfo(dt).max.item()), "data": table, "select": select}
            want = O.reclassify_process(data, kw)
            assert_result_equal(tables.reclassify_process(data, kw), want, 0, "reclassify %s select=%s" % (in_dtype, select))
    f = (rng.random((1, 1, n)) * 200 - 100).astype("f4")
    fndv = np.float32(np.finfo("f4").max)
    f[..., ::13] = fndv
    f[..., 5:6] = np.nan
    for bins in ([-50.0, -10.5, 0.0, 10.1, 50.0], list(np.linspace(-99.3, 99.1, 77)), [0, 10, 20]):
        for right in (False, True):
            for values, nd in ((f, fndv), (v, ndv)):
                d = {"values": values, "no_data_value": nd}
                want = O.classify_process(d, bins, right)
                assert_result_equal(tables.classify_process(d, bins, right), want, 0, "classify %d edges right=%s" % (len(bins), right))


# ---- resampling -------------------------------------------------------------------
GT = (136700.0, 5.0, 0, 455800.0, 0, -5.0)


@pytest.mark.parametrize("bbox,h,w", [
    ((136700, 455795, 136705, 455800), 1, 1), ((136700, 455790, 136705, 455795), 1, 1),
    ((136695, 455795, 136700, 455800), 1, 1), ((136700, 455795, 136710, 455800), 1, 2),
    ((136700, 455790, 136705, 455800), 2, 1), ((136700, 455795, 136710, 455800), 2, 4),
    ((136700, 455795, 136705, 455800), 5, 5),
])
def test_nearest_reference_kats(gpu, bbox, h, w):
    """The reference's own expectations for the GDAL nearest warp (tests/test_raster_sources.py:
    96-161): one 5 m pixel of value 5, no-data 255."""
    from dask_geomodeling_amd.raster import resample
    pix = np.full((1, 1, 1), 5, dtype="u1")
    got = np.asarray(resample.nearest_warp(pix, GT, bbox, h, w, 255))
    assert_values_equal(got, O.nearest_warp(pix, GT, bbox, h, w, 255))


@pytest.mark.parametrize("dt", ["u1", "i2", "i4", "i8", "f4", "f8"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_nearest_vs_oracle(gpu, dt, seed):
    from dask_geomodeling_amd.raster import resample
    rng = np.random.default_rng(seed)
    bands, hs, ws = int(rng.integers(1, 4)), int(rng.integers(5, 90)), int(rng.integers(5, 90))
    src = (rng.random((bands, hs, ws)) * 200).astype(dt)
    if np.dtype(dt).kind == "f":
        src[0, 1, 1] = np.nan
    px = float(rng.choice([0.5, 1.0, 2.5]))
    gt = (1000.0, px, 0, 5000.0, 0, -px)
    for _ in range(6):
        x1 = 1000.0 + float(rng.integers(-20, ws)) * px * float(rng.choice([1.0, 0.5, 0.37]))
        y2 = 5000.0 - float(rng.integers(-20, hs)) * px * float(rng.choice([1.0, 0.5, 0.37]))
        w, h = int(rng.integers(1, 130)), int(rng.integers(1, 70))
        cell = px * float(rng.choice([1.0, 0.5, 2.0, 0.3, 1.7]))
        bbox = (x1, y2 - h * cell, x1 + w * cell, y2)
        fill = 255 if np.dtype(dt).kind != "f" else float(np.finfo(dt).max)
        got = np.asarray(resample.nearest_warp(src, gt, bbox, h, w, fill))
        assert_values_equal(got, O.nearest_warp(src, gt, bbox, h, w, fill), 0, "%s %s" % (dt, bbox))


@pytest.mark.parametrize("case", load_family("smoothzoom"), ids=lambda c: c["id"])
def test_affine_nearest_golden(gpu, case):
    from dask_geomodeling_amd.raster import resample
    values, zoom, offset = case["args"]
    got = np.asarray(resample.affine_nearest(values, zoom, offset))
    assert_values_equal(got, case["expect"], 0, case["id"])


@pytest.mark.parametrize("dt", ["f4", "f8"])
@pytest.mark.parametrize("shape", [(16, 20, 32, 40), (9, 7, 27, 21), (10, 10, 15, 15), (8, 8, 8, 8),
                                   (12, 6, 5, 4), (64, 300, 128, 600), (500, 40, 37, 91), (1, 1, 4, 4)])
def test_bilinear_vs_oracle(gpu, dt, shape):
    """Own spec (reference has no bilinear: parity unpinned).  Tolerance: 1 ulp (north_star)."""
    from dask_geomodeling_amd.raster import resample
    hs, ws, hd, wd = shape
    rng = np.random.default_rng(hs * 1000 + ws)
    src = (rng.random((2, hs, ws)) * 100).astype(dt)
    ndv = np.dtype(dt).type(np.finfo(dt).max)
    if hs > 4 and ws > 4:
        src[0, 2, 3] = ndv
        src[1, hs - 1, ws - 1] = np.nan
    got = np.asarray(resample.bilinear(src, hd, wd, ndv))
    assert_values_equal(got, O.bilinear_resample(src, hd, wd, ndv), 1, "bilinear %s %s" % (dt, shape))


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["product", "staged", "direct"])
def test_bilinear_routes_vs_oracle(gpu, route):
    """The general bilinear kernel's two routes (round 5: the source window staged in LDS, or the taps read through the caches --
    the product takes the second from a zoom of 3.5 on) and its stores onto rows that do not start on 16 bytes, over several
    workgroups per tile and with no-data cells sprinkled in: against the oracle within 1 ulp, in a process of its own per route
    (the route is read from the environment once)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from dask_geomodeling_amd.raster import resample\n"
        "from oracle import ref_numpy as O\n"
        "from helpers import assert_values_equal\n"
        "rng = np.random.default_rng(11)\n"
        "n = 0\n"
        "for dt in ('f4', 'f8'):\n"
        "    for hs, ws, hd, wd in ((100, 333, 300, 999), (64, 100, 256, 400), (333, 517, 999, 1551), (200, 200, 499, 501),\n"
        "                           (300, 300, 800, 700), (90, 700, 541, 1753), (257, 129, 300, 1290), (40, 2000, 37, 1999)):\n"
        "        src = (rng.standard_normal((2, hs, ws)) * 100).astype(dt)\n"
        "        ndv = np.dtype(dt).type(np.finfo(dt).max)\n"
        "        src[rng.random(src.shape) < 0.02] = ndv\n"
        "        src[1, hs // 2, ws // 3] = np.nan\n"
        "        got = np.asarray(resample.bilinear(src, hd, wd, ndv))\n"
        "        assert_values_equal(got, O.bilinear_resample(src, hd, wd, ndv), 1, 'bilinear %%s %%s' %% (dt, (hs, ws, hd, wd)))\n"
        "        n += 1\n"
        "print('ROUTES ok', n)\n"
        "from helpers import print_routes\n"
        "print_routes('bilinear')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if route != "product":
        env["GM_BL_STAGED"] = "1" if route == "staged" else "0"
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "ROUTES ok 16" in res.stdout
    took = routes_of(res.stdout)["bilinear"]
    assert took["calls"] == 16, took
    if route == "staged":
        assert took["staged"] > 0 and took["direct"] == 0, took
    elif route == "direct":
        assert took["direct"] == took["calls"] - took["two_x"] and took["staged"] == 0, took
    else:   # the product's choice: the cases span both sides of its threshold
        assert took["staged"] > 0 and took["direct"] > 0, took


# ---- whole views through Blocks, fusion pass and MemorySource ------------------------
from blocks_util import build_view, leaves_for  # noqa: E402
from dask_geomodeling_amd import config as gmconfig  # noqa: E402
from dask_geomodeling_amd import raster  # noqa: E402

VIEW_ULP = {"Power": 1, "Exp": 2, "Log": 2, "Log10": 2}  # (see ULP_OPS)
NOT_YET = ()


def _uses(expr, names):
    if isinstance(expr, list):
        return expr[0] in names or any(_uses(x, names) for x in expr[1:])
    return False


def _view_ulp(expr):
    hits = [v for k, v in VIEW_ULP.items() if _uses(expr, (k,))]
    return 16 if len(hits) > 1 else max(hits or [0])


@pytest.mark.parametrize("fuse", [True, False], ids=["fused", "unfused"])
@pytest.mark.parametrize("case", [c for c in load_family("views") if not _uses(c["expr"], NOT_YET)],
                         ids=lambda c: c["id"])
def test_views_golden(gpu, case, fuse):
    # leaves hand the golden arrays over unchanged (a MemorySource would first map NaN/inf to
    # no-data like the reference's source does, sources.py:148 -- covered by the tests below)
    view = build_view(case["expr"], leaves_for(case, kind="array"))
    assert view.dtype.str == case["dtype"] and view.fillvalue == case["fillvalue"]
    with gmconfig.config.set({"geomodeling.fuse": fuse}):
        out = view.get_data(**case["request"])
    if out is not None and "values" in out:
        assert isinstance(out["values"], np.ndarray)  # get_data hands back host arrays
    assert_result_equal(out, case["expect"], _view_ulp(case["expr"]), case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("tiler") if c["op"] == "tiler_process"],
                         ids=lambda c: c["id"])
def test_tiler_process_golden(gpu, case):
    kw = dict(case["kwargs"])
    kw["dtype"] = np.dtype(kw["dtype"])
    out = raster.RasterTiler.process(kw, *case["tiles"])
    out = dict(out, values=np.asarray(out["values"]))
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["u1", "i2", "f4", "f8"])
def test_tiler_process_fill_and_batched_paste(gpu, dtype):
    """RasterTiler.process (parallelize.py:93-125) with the tiles of a mosaic pasted in one launch and the fill skipped
    exactly when the tiles cover the mosaic: full coverage, a tile that is None, tiles short of their slot (the fill
    value has to show there), several bands, widths that are / are not whole 16-byte groups -- against the oracle."""
    rng = np.random.default_rng(abs(hash(dtype)) % 2**31)
    dt = np.dtype(dtype)
    fill = O.get_dtype_max(dt)

    def tile(bands, th, tw):
        v = (rng.random((bands, th, tw)) * 100).astype(dt)
        return {"values": v, "no_data_value": fill}

    cases = []
    for bands, (count_x, count_y), (tx, ty), (height, width), holes, short in (
            (1, (2, 2), (32, 16), (32, 64), (), ()),               # full coverage: no fill
            (1, (2, 2), (32, 16), (32, 64), ((1, 0),), ()),        # one tile missing
            (2, (3, 2), (16, 8), (16, 48), (), ()),                # two bands, full
            (1, (2, 2), (33, 17), (34, 66), (), ()),               # odd sizes (no 16-byte groups), full
            (1, (2, 3), (20, 10), (28, 37), (), ()),               # ragged last column / top row of tiles
            (1, (2, 2), (32, 16), (32, 64), (), ((0, 1),)),        # a tile one row and one column short of its slot
            (3, (1, 1), (8, 8), (8, 8), (), ())):
        kw = {"shape_yx": (height, width), "count_xy": (count_x, count_y), "tilesize_xy": (tx, ty), "dtype": dt, "fillvalue": fill}
        tiles = []
        for ix in range(count_x):
            for iy in range(count_y):
                if (ix, iy) in holes:
                    tiles.append(None)
                    continue
                tw = min(tx, width - ix * tx)
                th = min(ty, height - iy * ty)
                if (ix, iy) in short:
                    tw, th = tw - 1, th - 1
                tiles.append(tile(bands, th, tw))
        cases.append((kw, tiles))
    for kw, tiles in cases:
        want = O.tiler_process(dict(kw, dtype=dt.str), *tiles)
        got = raster.RasterTiler.process(kw, *tiles)
        assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "%s %s" % (dtype, kw["count_xy"]))
    assert raster.RasterTiler.process(cases[0][0], None, None, None, None) is None


@pytest.fixture
def kat_source():
    from datetime import datetime, timedelta
    return raster.MemorySource(
        data=[np.full((10, 10), 1, "u1"), np.full((10, 10), 7, "u1"), np.full((10, 10), 255, "u1")],
        no_data_value=255, projection="EPSG:28992", pixel_size=0.5, pixel_origin=(135000, 456000),
        time_first=datetime(2000, 1, 1), time_delta=timedelta(hours=1))


@pytest.mark.parametrize("off", [(0, -5, 5, 0), (0, -5, 6, 0), (0, -6, 5, 0), (-1, -5, 5, 0), (0, -5, 5, 1)])
def test_tiled_equals_untiled(gpu, kat_source, off, tiler_route):
    """tests/test_raster_parallelize.py:87-116 of the reference."""
    src = kat_source
    block = raster.RasterTiler(src, 2)
    ox, oy = src.pixel_origin
    request = dict(mode="vals", bbox=(ox + off[0], oy + off[1], ox + off[2], oy + off[3]),
                   width=(off[2] - off[0]) * 2, height=(off[3] - off[1]) * 2, projection="EPSG:28992",
                   start=src.period[0])
    actual, expected = block.get_data(**request), src.get_data(**request)
    assert_values_equal(actual["values"], expected["values"])
    assert actual["no_data_value"] == expected["no_data_value"]
    want = O.nearest_warp(src.data[:1], src.geo_transform, request["bbox"], request["height"],
                          request["width"], 255)
    assert_values_equal(expected["values"], want)


def test_reference_kats_through_blocks(gpu, kat_source):
    """Known answers of the reference's own tests, through get_data on the device:
    tests/test_raster.py:369-400, 552-564, 1356-1395, 1447-1467; tests/test_raster_misc.py:134-254."""
    from datetime import datetime, timedelta
    src = kat_source
    req = dict(mode="vals", start=datetime(2000, 1, 1), stop=datetime(2000, 1, 1, 3), width=4, height=6,
               bbox=(135000, 456000 - 3, 135000 + 2, 456000), projection="EPSG:28992")

    def first(view):
        return view.get_data(**req)["values"][:, 0, 0].tolist()

    n32 = 2147483647
    assert first(src + 5) == [6, 12, n32]
    assert first(src - 1) == [0, 6, n32]
    assert first(src * 10) == [10, 70, n32]
    assert first(-src) == [-1, -7, n32]
    out = (src / 10).get_data(**req)
    assert out["values"].dtype == np.float32 and out["values"][0, 0, 0] == np.float32(0.1)
    assert first(src ** 2)[:2] == [1, 49]
    root = (src ** 0.5).get_data(**req)["values"]  # u8 ** python float -> float64, like the reference
    assert root.dtype == np.float64 and abs(root[1, 0, 0] - np.sqrt(7)) <= np.spacing(np.sqrt(7))
    inv = (src ** -1).get_data(**req)["values"]
    assert inv.dtype == np.float64 and abs(inv[1, 0, 0] - 1 / 7) <= np.spacing(1 / 7)
    assert first(src ** 0)[:2] == [1, 1]
    assert first(raster.Clip(src, src == 7)) == [255, 7, 255]
    assert first(raster.Clip(src, src)) == [1, 7, 255]
    assert first(raster.Mask(src, 8)) == [8, 8, 0]
    assert first(raster.Mask(src, 0)) == [0, 0, 1]
    assert raster.Mask(src, 1000).get_data(**req)["values"].dtype == np.uint16
    assert first(raster.Classify(src, [3, 8])) == [0, 1, 255]
    assert first(raster.Reclassify(src, [[7, 1000]])) == [1, 1000, np.iinfo("i8").max]
    assert first(raster.Reclassify(src, [[7, 1000]], select=True)) == [np.iinfo("i8").max, 1000, np.iinfo("i8").max]
    assert first(raster.Reclassify(src, [[7, 8.2]])) == [1.0, 8.2, np.finfo("f8").max]
    assert first(raster.Reclassify(src == 7, [[True, 1000]])) == [0, 1000, 0]
    assert first(raster.Reclassify(src * 1, [[7, 1000]])) == [1, 1000, np.iinfo("i8").max]
    assert first(src == 7) == [False, True, False]
    assert first(src != 7) == [True, False, True]
    assert first((src > 1) & (src < 9)) == [False, True, False]
    # no-data on either side of a Divide -> everything no-data (tests/test_raster.py:552-564)
    nodata = raster.MemorySource(np.full((3, 10, 10), 255, "u1"), 255, "EPSG:28992", 0.5, (135000, 456000),
                                 time_first=datetime(2000, 1, 1), time_delta=timedelta(hours=1))
    fmax = float(np.finfo("f4").max)
    assert first(src / nodata) == [fmax] * 3 and first(nodata / src) == [fmax] * 3
    # zoom-in of the source (nearest), tests/test_raster_sources.py:140-161 pattern
    zoom = dict(req, width=8, height=12)
    assert (src.get_data(**zoom)["values"][0] == 1).all()


def test_canonical_chain_on_memorysources(gpu):
    """cfg 1/2 shape: MemorySource leaves (NaN/inf become no-data at the source, sources.py:148),
    fused 5-op chain, bit-exact against the oracle applied to the same source tiles."""
    rng = np.random.default_rng(0)
    fmax = np.float32(np.finfo("f4").max)
    a = (rng.random((256, 256)) * 100).astype("f4")
    b = (rng.random((256, 256)) * 100).astype("f4")
    a[::97, ::89] = fmax
    a[3, 3], b[7, 7], b[9, 9] = np.nan, np.inf, fmax / 2
    a[9, 9] = fmax / 2
    A = raster.MemorySource(a, fmax, "EPSG:28992", 1.0, (0, 256))
    B = raster.MemorySource(b, fmax, "EPSG:28992", 1.0, (0, 256))
    view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    req = dict(mode="vals", bbox=(0, 0, 256, 256), width=256, height=256, projection="EPSG:28992")
    got = view.get_data(**req)
    ta = {"values": O.nearest_warp(a[None], A.geo_transform, req["bbox"], 256, 256, fmax), "no_data_value": fmax}
    tb = {"values": O.nearest_warp(b[None], B.geo_transform, req["bbox"], 256, 256, fmax), "no_data_value": fmax}
    kw = {"dtype": "float32", "fillvalue": float(fmax)}
    x = O.math_process("add", kw, ta, tb)
    x = O.math_process("multiply", kw, x, 2.5)
    x = O.math_process("subtract", kw, x, tb)
    x = O.math_process("add", kw, x, 1.5)
    want = O.clip_process(x, ta)
    assert_result_equal(got, want, 0, "canonical chain")
    # cfg 1: Add(src, src) and Add(src, 2.5) on one 256x256 tile
    for view, args in ((raster.Add(A, A), (ta, ta)), (raster.Add(A, 2.5), (ta, 2.5))):
        assert_result_equal(view.get_data(**req), O.math_process("add", kw, *args), 0, "cfg1")


# ---- zero-copy windows (strided inputs of the element-wise program kernel) ---------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f4", "i4", "u1", "f8"])
@pytest.mark.parametrize("win", [(0, 0, 64, 128), (5, 8, 40, 64), (3, 4, 33, 60), (7, 1, 21, 35), (0, 12, 1, 4)])
def test_program_on_strided_windows(gpu, dtype, win):
    """Windows of a larger resident raster as program inputs: aligned ones take the 16-byte vector
    path, odd offsets / widths the scalar path; both must equal the oracle on the compacted window,
    alone and inside a batch that mixes windows and contiguous tiles."""
    from dask_geomodeling_amd.hip import DeviceArray, program as P
    r0, c0, h, w = win
    rng = np.random.default_rng(h * 1000 + w)
    big_a = (rng.random((1, 80, 160)) * 100).astype(dtype)
    big_b = (rng.random((1, 80, 160)) * 100).astype(dtype)
    ndv = np.dtype(dtype).type(77)
    big_a[0, ::5, ::7] = ndv
    da, db = DeviceArray.from_numpy(big_a), DeviceArray.from_numpy(big_b)
    wa, wb = da.window(r0, c0, h, w), db.window(r0, c0, h, w)
    assert_values_equal(wa.to_numpy(), big_a[:, r0:r0 + h, c0:c0 + w], 0, "window d2h")
    res_dt = np.result_type(np.dtype(dtype), np.int32 if np.dtype(dtype).kind in "iu" else np.float32)
    kw = {"dtype": res_dt.str, "fillvalue": O.get_dtype_max(res_dt) if hasattr(O, "get_dtype_max") else
          (np.iinfo(res_dt).max.item() if res_dt.kind == "i" else np.finfo(res_dt).max.item())}
    A = {"values": big_a[:, r0:r0 + h, c0:c0 + w].copy(), "no_data_value": ndv}
    B = {"values": big_b[:, r0:r0 + h, c0:c0 + w].copy(), "no_data_value": ndv}
    want = O.clip_process(O.math_process("multiply", kw, O.math_process("add", kw, A, B), 3), B)
    got = devops.clip_process(devops.math_process("multiply", kw, devops.math_process(
        "add", kw, {"values": wa, "no_data_value": ndv}, {"values": wb, "no_data_value": ndv}), 3),
        {"values": wb, "no_data_value": ndv})
    assert_result_equal(got, want, 0, "op-by-op on windows %s %s" % (dtype, win))
    la, lb = P.Leaf(0, dtype, ndv), P.Leaf(1, dtype, ndv)
    comp = P.compile_expr(devops.clip_node(devops.math_node("multiply", kw, [devops.math_node("add", kw, [la, lb]), P.Lit(3)]), lb), 2)
    ca, cb = DeviceArray.from_numpy(A["values"]), DeviceArray.from_numpy(B["values"])
    outs = P.launch_batch(comp, [[wa, wb], [ca, cb], [wa, cb], [ca, wb]], (1, h, w))
    for k, out in enumerate(outs):
        assert_values_equal(out.to_numpy(), want["values"], 0, "batch member %d %s %s" % (k, dtype, win))


@pytest.mark.gpu
def test_tiler_over_memory_source_is_zero_copy_and_exact(gpu, tiler_route):
    """RasterTiler over a view on resident MemorySources: every aligned tile reaches the fused
    program as a strided window (no warp kernel), results equal the untiled evaluation."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip import DeviceArray
    rng = np.random.default_rng(9)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 96, 128)) * 100).astype("f4")
    b = (rng.random((1, 96, 128)) * 100).astype("f4")
    a[0, ::9, ::4] = fill
    A = raster.MemorySource(a, fill, "EPSG:28992", 1.0, (0.0, 96.0))
    B = raster.MemorySource(b, fill, "EPSG:28992", 1.0, (0.0, 96.0))
    view = raster.Clip((A + B) * 2.5 - B + 1.5, A)
    req = dict(mode="vals", bbox=(0.0, 0.0, 128.0, 96.0), width=128, height=96, projection="EPSG:28992")
    whole = view.get_data(**req)
    for tile in (32, [64, 48], 40):  # 40: leftover edge tiles of another shape
        tiled = raster.RasterTiler(view, tile).get_data(**req)
        assert_result_equal(tiled, whole, 0, "tile size %s" % (tile,))
    from dask_geomodeling_amd.fuse import fuse_graph, TiledAssembly
    graph, name = raster.RasterTiler(view, 32).get_compute_graph(**req)
    fused = fuse_graph(graph, name)
    assert isinstance(fused[name][0], TiledAssembly)  # tiles are written straight into the mosaic
    # a second consumer of one tile keeps the reference's paste path (and the same numbers)
    tiler = raster.RasterTiler(view, 32)
    both = raster.Add(tiler, tiler)
    assert_result_equal(both.get_data(**req), (view + view).get_data(**req), 0, "tiler consumed twice")
    sub = dict(req, bbox=(32.0, 16.0, 96.0, 80.0), width=64, height=64)
    got = A.get_data(**sub)
    assert_values_equal(got["values"], a[:, 16:80, 32:96], 0, "aligned sub-window of a MemorySource")
    kwargs = A.get_sources_and_requests(**sub)[0][0]
    assert kwargs["window"] == (16, 32)
    raw = raster.MemorySource.process(kwargs)["values"]
    assert isinstance(raw, DeviceArray) and raw.pitch == 128


@pytest.mark.gpu
@pytest.mark.parametrize("select", [False, True])
@pytest.mark.parametrize("dtype", ["i4", "i8", "i2", "u2"])
def test_reclassify_sparse_keys_hash_table(gpu, dtype, select):
    """Keys too scattered for a dense window run through the two-probe hash table (and, beyond 512 keys, the
    binary search): hits, misses, negative keys, keys that collide in their low bits, the no-data key."""
    from dask_geomodeling_amd.raster import tables
    rng = np.random.default_rng(17)
    info = np.iinfo(dtype)
    span = min(int(info.max), 2_000_000_000)
    for n_keys in (3, 100, 400, 700):
        keys = np.unique(rng.integers(max(int(info.min), -span), span, size=n_keys).astype(dtype)).tolist()
        if dtype == "i8":
            keys += [2 ** 40 + 5, 2 ** 40 + 5 + 2 ** 32, -2 ** 45]  # equal low 32 bits: only the full key tells them apart
        table = [[int(k), float(i) + 0.5 if i % 3 == 0 else int(i * 7)] for i, k in enumerate(sorted(set(keys)))]
        ndv = np.dtype(dtype).type(info.max)
        v = rng.choice(np.asarray([k for k, _ in table] + [0, 1, -1 if info.min < 0 else 2, int(info.max) - 1], dtype=dtype),
                       size=(1, 300, 257))
        v[rng.random(v.shape) < 0.3] = rng.integers(max(int(info.min), -span), span)
        v[0, ::11, ::7] = ndv
        data = {"values": v, "no_data_value": ndv}
        target_dtype = np.asarray([t for _, t in table]).dtype
        kw = {"dtype": target_dtype.str, "fillvalue": (float(np.finfo("f8").max) if target_dtype.kind == "f" else int(np.iinfo("i8").max)),
              "data": table, "select": select}
        got = tables.reclassify_process(data, kw)
        want = O.reclassify_process(data, kw)
        assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "%s %d keys select=%s" % (dtype, n_keys, select))


@pytest.mark.gpu
@pytest.mark.parametrize("right", [False, True])
@pytest.mark.parametrize("kind", ["uniform200", "random37", "clustered", "u16_300", "pairs", "triples"])
def test_classify_many_edges_large_tile(gpu, kind, right):
    """Stand-alone Classify with many edges on a tile large enough for the grid-accelerated kernel (uniform and
    irregular edges; clustered edges fall back to the binary search): exact against np.digitize, values on the
    edges, NaN, +-inf and no data included."""
    from dask_geomodeling_amd.raster import tables
    rng = np.random.default_rng(len(kind))
    if kind == "uniform200":
        bins = np.linspace(-100, 100, 200).tolist()
    elif kind == "random37":
        bins = sorted(set(np.round(rng.random(37) * 180 - 90, 3).tolist()))
    elif kind == "u16_300":
        bins = (np.arange(300) * 0.61 - 90).tolist()
    elif kind == "pairs":    # two edges closer than a grid bucket, pair after pair: two comparisons per value
        bins = np.sort(np.concatenate([np.linspace(-100, 100, 60), np.linspace(-100, 100, 60)[:-1] + 0.01])).tolist()
    elif kind == "triples":  # three: the four-wide scan
        base = np.linspace(-100, 100, 50)
        bins = np.sort(np.concatenate([base, base[:-1] + 0.004, base[:-1] + 0.008])).tolist()
    else:
        bins = sorted(set((np.concatenate([np.linspace(-100, 100, 20), 1.0 + np.arange(40) * 1e-4])).tolist()))
    ndv = np.float32(np.finfo("f4").max)
    v = (rng.random((1, 1024, 1031)) * 220 - 110).astype("f4")
    flat = v.reshape(-1)
    flat[:len(bins)] = np.asarray(bins, dtype="f4")             # values on (float32-rounded) edges
    flat[len(bins):2 * len(bins)] = np.nextafter(np.asarray(bins, dtype="f4"), np.float32(1e9))
    flat[2 * len(bins):3 * len(bins)] = np.nextafter(np.asarray(bins, dtype="f4"), np.float32(-1e9))
    v[0, 5, :7] = [np.nan, np.inf, -np.inf, ndv, 0.0, -0.0, 1e30]
    v[0, ::53, ::47] = ndv
    data = {"values": v, "no_data_value": ndv}
    got = tables.classify_process(data, bins, right)
    want = O.classify_process(data, bins, right)
    assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "%s right=%s" % (kind, right))


@pytest.mark.gpu
@pytest.mark.parametrize("right", [False, True])
@pytest.mark.parametrize("kind", ["uniform200", "clustered", "few"])
def test_classify_many_edges_on_windows_of_a_resident_raster(gpu, kind, right):
    """Classify of a zero-copy window (rows a pitch apart): the grid kernel reads it where it lies, from any first column and
    in rows of any pitch (gm_classify2_window; round 5: element-aligned loads -- until then only on 16-byte boundaries); a
    width that is no multiple of 4 runs over the next one where the raster has cells to its right and is cropped, at the
    raster's right edge it takes a compact copy, like clustered edges (binary search); few edges take the program route.
    Also through the Block, as `Classify(A, edges)` over a resident MemorySource meets it."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip.array import DeviceArray
    from dask_geomodeling_amd.raster import tables
    rng = np.random.default_rng(11)
    if kind == "uniform200":
        bins = np.linspace(-100, 100, 200).tolist()
    elif kind == "few":
        bins = [-50.0, 0.0, 12.5, 70.0]
    else:
        bins = sorted(set((np.concatenate([np.linspace(-100, 100, 20), 1.0 + np.arange(40) * 1e-4])).tolist()))
    ndv = np.float32(np.finfo("f4").max)
    wide = (rng.random((1, 700, 1100)) * 220 - 110).astype("f4")
    wide[0, ::37, ::41] = ndv
    wide[0, 300, 400:406] = [np.nan, np.inf, -np.inf, ndv, 0.0, -0.0]
    edges32 = np.asarray(bins, dtype="f4")
    wide[0, 301, 400:400 + len(bins)] = edges32[:700]
    resident = DeviceArray.from_numpy(wide)
    for row0, col0, h, w in ((0, 0, 700, 1100), (16, 64, 512, 640), (3, 4, 300, 1000), (3, 5, 300, 1000), (9, 8, 600, 1001),
                             (0, 0, 128, 1100), (100, 96, 17, 132), (5, 99, 200, 1001), (7, 97, 100, 1002), (1, 1, 699, 1098)):
        data = {"values": resident.window(row0, col0, h, w), "no_data_value": ndv}
        want = O.classify_process({"values": wide[:, row0:row0 + h, col0:col0 + w], "no_data_value": ndv}, bins, right)
        got = tables.classify_process(data, bins, right)
        assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "%s right=%s %s" % (kind, right, (row0, col0, h, w)))
    finite = np.where(np.isfinite(wide), wide, np.float32(1.0))  # (a source hands non-finite cells on as no data, sources.py:141-143)
    source = raster.MemorySource(DeviceArray.from_numpy(finite), ndv, "EPSG:28992", 1.0, (0.0, 700.0))
    view = raster.Classify(source, bins, right)
    request = dict(mode="vals", bbox=(64.0, 100.0, 704.0, 612.0), width=640, height=512, projection="EPSG:28992")
    got = view.get_data(**request)
    want = O.classify_process({"values": finite[:, 88:600, 64:704], "no_data_value": ndv}, bins, right)
    assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "Block %s right=%s" % (kind, right))


@pytest.mark.gpu
def test_tiles_without_cells_at_the_edge_of_a_tiled_request(gpu, tiler_route):
    """What the round-5 temporal fuzzing found on 17 x 31 rasters: the reference's int() of a tile size (parallelize.py:85-86) can
    leave a tile of NO columns at the edge of a request.  Pasting it pastes nothing (gm_paste_tiles refused the batch: "outside
    mosaic"), and a stencil over such a tile -- all margin -- answers with an empty tile like the reference's filter + crop
    (gm_moving_max: "tile smaller than the footprint").  Tiled requests of that kind against the untiled evaluation."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(355)
    v = (rng.random((3, 17, 31)) * 100).astype("f8")
    ndv = np.float64(np.finfo("f8").max)
    v[rng.random(v.shape) < 0.1] = ndv
    from datetime import datetime, timedelta
    src = raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, 17.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    views = [src * 0.5, raster.MovingMax(src + 2.5, 5), raster.Dilate(raster.Classify(src, [20.0, 50.0]), [1])]
    for x0, y0, w, h in ((3, 2, 15, 8), (0, 0, 31, 17), (16, 9, 15, 8), (1, 1, 29, 3)):
        request = dict(mode="vals", bbox=(float(x0), float(y0), float(x0 + w), float(y0 + h)), width=w, height=h, projection="EPSG:28992",
                       start=datetime(2021, 3, 1), stop=datetime(2021, 3, 3))
        for view in views:
            want = view.get_data(**request)
            for tile in ([14, 17], [8, 14], [5, 5], [17, 1]):
                got = raster.RasterTiler(view, tile).get_data(**request)
                assert_result_equal(got, want, 0, "%s tiled by %s on %s" % (type(view).__name__, tile, (x0, y0, w, h)))
    # the stencils' own entry points on tiles that are all margin
    assert tuple(stencil.moving_max(np.zeros((1, 4, 9), "f4"), None, 5).shape) == (1, 0, 5)
    assert tuple(stencil.moving_max(np.zeros((2, 9, 3), "i4"), None, 5).shape) == (2, 5, 0)
    assert tuple(stencil.hillshade(np.zeros((1, 2, 5), "f4"), None, 0, (1.0, 1.0), 45.0, 315.0).shape) == (1, 0, 3)
    assert tuple(stencil.gaussian_smooth(np.zeros((1, 6, 20), "f4"), None, 0, (1.0, 1.0), crop_yx=(3, 4)).shape) == (1, 0, 12)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", [0.1, 0.3, 0.7])
def test_tiler_with_tiles_one_pixel_short_of_their_slot(gpu, cell, tiler_route):
    """The reference truncates tile sizes with int() (parallelize.py:85-86): for cell sizes that are not
    exact in binary a tile can come out one pixel narrower than its slot, and np.full leaves the fill
    value in the uncovered column / row.  The in-place mosaic must not hand out unwritten memory there."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(12)
    fill = np.float32(np.finfo("f4").max)
    n = 1100
    a = (rng.random((1, n, n)) * 100).astype("f4")
    b = (rng.random((1, n, n)) * 100).astype("f4")
    A = raster.MemorySource(a, fill, "EPSG:28992", cell, (0.0, n * cell))
    B = raster.MemorySource(b, fill, "EPSG:28992", cell, (0.0, n * cell))
    view = raster.Clip((A + B) * 2.5 - B + 1.5, A)
    req = dict(mode="vals", bbox=(cell, cell, cell + 1024 * cell, cell + 1024 * cell), width=1024, height=1024,
               projection="EPSG:28992")
    tiler = raster.RasterTiler(view, 256)
    plan = tiler.get_sources_and_requests(**req)
    kwargs, tile_reqs = plan[0][0], [r for _, r in plan[1:]]
    tiles = []
    for r in tile_reqs:
        t = view.get_data(**r)
        tiles.append(None if t is None else {"values": np.asarray(t["values"]), "no_data_value": t["no_data_value"]})
    want = O.tiler_process(kwargs, *tiles)
    for _ in range(2):  # the second pass runs on recycled (dirty) pool memory
        got = tiler.get_data(**req)
        assert_result_equal(got, want, 0, "cell %s" % cell)
    short = sum(1 for r in tile_reqs if r["width"] * r["height"] != 256 * 256)
    print("cell", cell, "tiles short of their slot:", short)


# ---- N3: stencils (Dilate / MovingMax / HillShade) ------------------------------------------------
def _hillshade_exact(got, data, kw, what):
    """HillShade is bit-exact against the host-independent oracle (float32 arctan2 / sin = roundings of the
    float64 functions, oracle/ref_numpy.py); against THIS host's NumPy trig a cell per ~1e5 may differ by 1 on
    a truncation boundary -- counted and bounded, not tolerated wholesale."""
    want = O.hillshade_process(data, kw, trig="rounded")
    assert got["no_data_value"] == 256
    assert_values_equal(got["values"], want["values"], 0, what)
    host = O.hillshade_process(data, kw)["values"].astype("i2")
    diff = np.abs(np.asarray(got["values"]).astype("i2") - host)
    # (integer terrain has few distinct gradient pairs: an affected pair shows up in several cells at once)
    assert diff.max(initial=0) <= 1 and (diff != 0).sum() <= max(8, 20e-6 * diff.size), \
        "%s: %d of %d cells differ from this host's NumPy" % (what, (diff != 0).sum(), diff.size)
    return int((diff != 0).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in load_family("spatial") if c["op"] != "expand"],
                         ids=lambda c: c["id"] + "-" + c["op"])
def test_stencils_golden(gpu, case):
    """Device Dilate / MovingMax / HillShade against vectors produced by the reference itself:
    bit-exact, HillShade included (the exact tier of the kernel reproduces the reference's float32 sequence; what
    depends on the host's NumPy trigonometry is measured in _hillshade_exact, not allowed for here)."""
    from dask_geomodeling_amd import raster
    fn = {"dilate": raster.Dilate.process, "movingmax": raster.MovingMax.process, "hillshade": raster.HillShade.process}[case["op"]]
    out = fn(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])  # HillShade included: bit-exact


@pytest.mark.gpu
def test_stencils_large_tiles_vs_oracle(gpu):
    """Tiles larger than one workgroup window (several LDS tiles, ragged edges) against the oracle."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(4)
    for dt, ndv in (("f4", np.float32(np.finfo("f4").max)), ("i4", -9999), ("u1", np.uint8(255))):
        shape = (2, 157, 203)
        if np.dtype(dt).kind == "f":
            v = (rng.random(shape) * 100 - 50).astype(dt)
        else:
            v = rng.integers(0, 100, size=shape).astype(dt)
        v[rng.random(shape) < 0.2] = ndv
        v[:, 30:60, 40:90] = ndv
        data = {"values": v, "no_data_value": ndv}
        for size in (3, 9, 15):
            assert_result_equal(raster.MovingMax.process(data, size), O.movingmax_process(data, size), 0, "movingmax %s %d" % (dt, size))
        vals = [v.flat[7], v.flat[11], 3]
        assert_result_equal(raster.Dilate.process(data, vals), O.dilate_process(data, vals), 0, "dilate %s" % dt)
        kw = dict(resolution=(0.5, 2.0), altitude=45.0, azimuth=315.0, fill=1)
        _hillshade_exact(raster.HillShade.process(data, kw), data, kw, "hillshade %s" % dt)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,scale", [("f4", 1.0), ("f4", 40.0), ("i2", 200.0), ("f8", 3.0), ("u1", 60.0)])
def test_hillshade_bit_exact_on_terrain(gpu, dt, scale):
    """4 M cells of random terrain per dtype, two sun positions: every cell equals the host-independent oracle
    (both tiers of the kernel: the algebraic fast tier decides 99.8 % of the cells, the step-by-step tier the
    cells within 1e-3 of a truncation boundary); the cells that differ from this host's NumPy are counted."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(77)
    v = rng.standard_normal((1, 2050, 2050)) * scale
    v = np.cumsum(v, axis=2).astype(dt) if dt in ("f4", "f8") else np.round(np.abs(v) if dt == "u1" else v).astype(dt)
    ndv = np.dtype(dt).type(250 if dt == "u1" else -9999)
    v[0, ::37, ::41] = ndv
    data = {"values": v, "no_data_value": ndv}
    total = 0
    for alt, az, res in ((45.0, 315.0, (0.5, 2.0)), (30.0, 10.0, (1.0, 1.0))):
        kw = dict(resolution=res, altitude=alt, azimuth=az, fill=1)
        total += _hillshade_exact(raster.HillShade.process(data, kw), data, kw, "terrain %s %s" % (dt, alt))
    print("hillshade %s: %d of %d cells differ from this host's NumPy float32 trig" % (dt, total, 2 * 2048 * 2048))


@pytest.mark.gpu
def test_hillshade_float32_kernels_agree(gpu, monkeypatch):
    """float32 tiles have two kernels: two pixels per lane in packed arithmetic (the default) and the one-pixel LDS-tile
    kernel every other dtype uses (GM_HS_SINGLE=1 selects it for float32 too): same bits, on a tile with ragged edges."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(6)
    v = np.cumsum(rng.standard_normal((2, 333, 1111)) * 2, axis=2).astype("f4")
    v[rng.random(v.shape) < 0.05] = -9999.0
    pair = stencil.hillshade(v, np.float32(-9999.0), 1, (0.5, 1.0), 45.0, 315.0).to_numpy()
    from dask_geomodeling_amd.hip import runtime
    before = runtime.route_stats("stencils")
    monkeypatch.setenv("GM_HS_SINGLE", "1")
    single = stencil.hillshade(v, np.float32(-9999.0), 1, (0.5, 1.0), 45.0, 315.0).to_numpy()
    assert np.array_equal(pair, single)
    assert runtime.routes_taken(before, runtime.route_stats("stencils")) == {"calls": 1, "hillshade_single": 1}


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "i2"])
def test_hillshade_full_exact_queue_and_true_division(gpu, dt):
    """The corners of the two-tier kernels: a plateau under a sun in the zenith (255 * cang sits exactly on an
    integer in every cell: the exact-tier queue of every workgroup overflows and lanes shade their own
    columns), mixed with terrain, and resolutions that are not powers of two (true division instead of
    the product with the reciprocal)."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(5)
    v = np.zeros((1, 300, 1100), dtype="f8")
    v[0, 100:200, 300:800] = np.cumsum(rng.standard_normal((100, 500)) * 3, axis=1)
    v = np.round(v).astype(dt) if dt != "f4" else v.astype(dt)
    ndv = np.dtype(dt).type(-9999)
    v[0, ::53, ::29] = ndv
    data = {"values": v, "no_data_value": ndv}
    for alt, az, res in ((90.0, 315.0, (1.0, 1.0)), (90.0, 0.0, (0.3, 0.7)), (45.0, 315.0, (0.3, 0.7)), (60.0, 200.0, (3.0, 0.5))):
        kw = dict(resolution=res, altitude=alt, azimuth=az, fill=0)
        got = raster.HillShade.process(data, kw)
        want = O.hillshade_process(data, kw, trig="rounded")  # (a plateau is one gradient pair: no count against NumPy here)
        assert_values_equal(got["values"], want["values"], 0, "plateau %s %s %s" % (dt, alt, res))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [3, 5, 7, 9, 11, 13, 15, 17, 21])
@pytest.mark.parametrize("dt", ["f4", "f8"])
def test_movingmax_nan_cells_like_scipy(gpu, size, dt):
    """NaN cells: the kernels reduce with the hardware maximum (NaN operands ignored) and make the result NaN
    where the first cell of the footprint is NaN -- SciPy's rule (the oracle is pinned against SciPy on
    exactly this in tests/test_oracle_golden.py).  All three kernels (3 x 3, fixed disks, general), ragged
    tiles, no data next to NaN."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(size)
    v = (rng.standard_normal((2, 90, 300)) * 10).astype(dt)
    ndv = np.dtype(dt).type(-9999)
    v[rng.random(v.shape) < 0.2] = ndv
    v[rng.random(v.shape) < 0.03] = np.nan
    data = {"values": v, "no_data_value": ndv}
    got = raster.MovingMax.process(data, size)
    want = O.movingmax_process(data, size)
    g, x = np.asarray(got["values"]), want["values"]
    assert np.isnan(x).any() and not np.isnan(x).all()
    assert g.dtype == x.dtype and np.array_equal(g, x, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [3, 5, 7, 9, 15, 17, 21])
@pytest.mark.parametrize("dt", ["f4", "f8"])
def test_movingmax_sign_of_a_zero_maximum_like_scipy(gpu, size, dt):
    """Windows whose maximum is zero and which hold zeros of BOTH signs: SciPy keeps the zero its scan leaves standing (the
    first of the footprint in raster order for the disks; for the all-true 3 x 3, which it filters separably, the bottom-most
    zero of the right-most column with one).  The hardware maximum says +0.0; the zero-sign pass behind every float launch
    restores SciPy's.  All three kernels, compact tiles and windows of a resident raster, with and without no data (also a
    no-data value of 0: every zero is no data then), and tiles without any -0.0 (the pass returns at once)."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip.array import DeviceArray
    rng = np.random.default_rng(size)
    kind = np.dtype(dt).type
    for trial, ndv in enumerate((None, kind(-9999), kind(0), None)):
        v = rng.choice(np.array([0.0, -0.0, -1.0, -2.5], dtype=dt), size=(2, 70, 300), p=[.15, .15, .4, .3])
        if trial == 3:
            v[v == 0] = 0.0  # zeros of one sign only
            assert not np.signbit(v[v == 0]).any()
        if ndv is not None and ndv != 0:
            v[rng.random(v.shape) < 0.2] = ndv
        data = {"values": v, "no_data_value": ndv}
        want = O.movingmax_process(data, size)["values"]
        got = np.asarray(raster.MovingMax.process(data, size)["values"])
        assert got.dtype == want.dtype and np.array_equal(got, want)
        assert np.array_equal(np.signbit(got), np.signbit(want)), (size, dt, ndv)
        if trial == 0:
            zeros = want == 0
            assert np.signbit(want[zeros]).any() and not np.signbit(want[zeros]).all()
    # a window of a wider resident raster (rows a pitch apart), one band
    wide = rng.choice(np.array([0.0, -0.0, -1.0], dtype=dt), size=(1, 90, 400), p=[.2, .2, .6])
    resident = DeviceArray.from_numpy(wide)
    got = np.asarray(raster.MovingMax.process({"values": resident.window(7, 13, 64, 290), "no_data_value": None}, size)["values"])
    want = O.movingmax_process({"values": wide[:, 7:71, 13:303], "no_data_value": None}, size)["values"]
    assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["u1", "i2", "f4", "i8", "f8", "?"])
def test_all_equal_heads_tails_and_where_the_odd_cell_sits(gpu, dt):
    """gm_all_equal (np.all(tile == value): Clip's early-out misc.py:108, Place's data test) reads 16 bytes per lane: tiles
    that start off a 16-byte boundary, lengths around the group size, the one differing cell in the head, the body, the tail;
    a tile of several sweeps of the grid (the early exit)."""
    from dask_geomodeling_amd.hip.array import DeviceArray
    from dask_geomodeling_amd.raster import devops
    dtype = np.dtype(dt)
    same = True if dt == "?" else 7
    base = np.full(5000, same, dtype=dtype)

    def view(host, first, count):
        dev = DeviceArray.from_numpy(host.reshape(1, 1, -1))
        return DeviceArray(dev.ptr + first * dtype.itemsize, (1, 1, count), dtype, dev.device, owner=dev)

    for first in (0, 1, 3, 5, 15, 16, 17):
        for count in (0, 1, 2, 15, 16, 17, 31, 33, 100, 4097):
            host = base.copy()
            assert devops.all_equal(view(host, first, count), same) is True, (first, count)
            for at in sorted({0, count // 2, count - 1} - {-1}) if count else ():
                host = base.copy()
                host[first + at] = 0
                assert devops.all_equal(view(host, first, count), same) is False, (first, count, at)
                host = base.copy()
                host[first + count] = 0  # (the cell after the tile is none of its business)
                if first:
                    host[first - 1] = 0
                assert devops.all_equal(view(host, first, count), same) is True, (first, count, "neighbours")
    big = np.full((3, 1500, 2001), same, dtype=dtype)
    assert devops.all_equal(big, same) is True
    for where in ((0, 0, 0), (1, 700, 1000), (2, 1499, 2000)):
        odd = big.copy()
        odd[where] = 0
        assert devops.all_equal(odd, same) is False, where


@pytest.mark.gpu
def test_stencil_blocks_end_to_end(gpu):
    """The Blocks through get_data on a MemorySource: request enlargement + kernel + crop equal the
    oracle applied to the enlarged window (edge cells see the source's no-data outside its extent)."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(12)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 40, 56)) * 100).astype("f4")
    a[0, 5:9, 7:15] = fill
    src = raster.MemorySource(a, fill, "EPSG:28992", 1.0, (0.0, 40.0))
    req = dict(mode="vals", bbox=(8.0, 8.0, 40.0, 32.0), width=32, height=24, projection="EPSG:28992")
    for block, radius, fn in ((raster.MovingMax(src, 5), 2, lambda d: O.movingmax_process(d, 5)),
                              (raster.HillShade(src), 1, lambda d: O.hillshade_process(d, dict(resolution=(1.0, 1.0), altitude=45.0, azimuth=315.0, fill=0), trig="rounded"))):
        big = src.get_data(**O.expand_request_pixels(req, radius))
        want = fn(big)
        got = block.get_data(**req)
        assert_result_equal(got, want, 0, type(block).__name__)
    for mode in ("time", "meta"):
        assert raster.MovingMax(src, 3).get_data(mode=mode) == src.get_data(mode=mode)


# ---- N4: Max = reduce_rasters(stack, "max") ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", load_family("reduce"), ids=lambda c: c["id"])
def test_reduce_max_golden(gpu, case):
    args = case["args"]  # (incl. 64-bit integer rasters: through float64 like the reference)
    assert_result_equal(devops.reduce_max_process(*args), case["expect"], 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", load_family("reducestats"), ids=lambda c: c["id"] + "-" + c["op"])
def test_reduce_rasters_golden(gpu, case):
    """reduce_rasters with every statistic of the reference (first, last, count, sum, mean, min, max, argmin,
    argmax, std, var, median, product, p<number>) on stacks of every dtype, incl. float16 work dtypes (1-byte
    rasters) and casts into another output dtype: bit-exact against the reference's own results."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    out = reduce_rasters([dict(d) for d in case["stack"]], case["op"], **case["kwargs"])
    out, want = dict(out, values=np.asarray(out["values"])), case["expect"]
    if case["op"] == "median" or case["op"].startswith("p"):
        # which of -0.0 / +0.0 a sort puts first is a property of NumPy's SIMD sort on the generating host
        from fuzz_util import unsigned_zero
        out, want = unsigned_zero(out), unsigned_zero(want)
    assert_result_equal(out, want, 0, "%s %s" % (case["id"], case["op"]))


@pytest.mark.gpu
def test_max_block_on_device(gpu):
    """(A tie between +0.0 and -0.0 is left out: NumPy's own answer depends on which SIMD maximum its
    fmax loop was built with.)"""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(21)
    fill = np.float32(np.finfo("f4").max)
    srcs, tiles = [], []
    for k in range(4):
        v = (rng.random((1, 64, 96)) * 100 - 50).astype("f4")
        v[0, k::4, ::3] = fill
        v[0, 5, 5 + k] = np.nan
        srcs.append(raster.MemorySource(v, fill, "EPSG:28992", 1.0, (0.0, 64.0)))
        tiles.append({"values": v, "no_data_value": fill})
    tiles[0]["values"][0, :3, :3] = fill  # one corner where every raster is no data
    for t in tiles[1:]:
        t["values"][0, :3, :3] = fill
    srcs = [raster.MemorySource(t["values"], fill, "EPSG:28992", 1.0, (0.0, 64.0)) for t in tiles]
    req = dict(mode="vals", bbox=(0.0, 0.0, 96.0, 64.0), width=96, height=64, projection="EPSG:28992")
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    got = raster.Max(*srcs).get_data(**req)
    # MemorySource turns non-finite source cells into no data (sources.py:148) before Max sees them
    clean = [{"values": np.where(np.isfinite(t["values"]), t["values"], fill), "no_data_value": fill} for t in tiles]
    assert_result_equal(got, O.reduce_max_process(kw, *clean), 0, "Max of 4 sources")


# ---- threaded scheduler: process bodies are re-entrant --------------------------------------------
@pytest.mark.gpu
def test_process_bodies_from_concurrent_threads(gpu):
    """dask's threaded scheduler calls ``process`` from a pool (ctypes releases the GIL during the
    calls): 8 threads hammer element-wise programs, table kernels and the allocator at once; every
    result must equal the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from dask_geomodeling_amd.raster import tables
    fill = np.float32(np.finfo("f4").max)

    def job(seed):
        rng = np.random.default_rng(seed)
        shape = (1, 64 + seed, 96)
        a = (rng.random(shape) * 100).astype("f4")
        b = (rng.random(shape) * 100).astype("f4")
        a[0, ::7, seed % 5::9] = fill
        A, B = {"values": a, "no_data_value": fill}, {"values": b, "no_data_value": fill}
        kw = {"dtype": "float32", "fillvalue": float(fill)}
        ok = True
        for _ in range(20):
            got = devops.clip_process(devops.math_process("multiply", kw, devops.math_process("add", kw, A, B), 1.0 + seed), A)
            want = O.clip_process(O.math_process("multiply", kw, O.math_process("add", kw, A, B), 1.0 + seed), A)
            ok &= np.array_equal(np.asarray(got["values"]).view("u4"), want["values"].view("u4"))
            bins = [10.0 * k + seed for k in range(1, 6 + seed)]
            c_got, c_want = tables.classify_process(A, bins, bool(seed % 2)), O.classify_process(A, bins, bool(seed % 2))
            ok &= np.array_equal(np.asarray(c_got["values"]), c_want["values"])
            table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": bool(seed % 2),
                     "data": [[k, k * (seed + 2)] for k in range(0, 8)]}
            r_got, r_want = tables.reclassify_process(c_got, table), O.reclassify_process(c_want, table)
            ok &= np.array_equal(np.asarray(r_got["values"]), r_want["values"])
        return ok

    with ThreadPoolExecutor(8) as pool:
        assert all(pool.map(job, range(8)))


# ---- N4: Group -------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in load_family("group") if c["request"]["mode"] == "vals"],
                         ids=lambda c: c["id"] + "-" + c["setup"])
def test_group_golden(gpu, case):
    """Group.get_data on temporal sources: merge programs + band assembly on the device against the
    reference's own results."""
    from blocks_util import TemporalLeaf
    from dask_geomodeling_amd import raster
    leaves = []
    for spec in case["sources"]:
        TemporalLeaf.ARRAYS[spec["name"]] = {"values": spec["data"], "no_data_value": spec["ndv"]}
        leaves.append(TemporalLeaf(spec["name"], spec["dtype"], spec["ndv"], spec["times"], spec["delta"]))
    out = raster.Group(*leaves).get_data(**case["request"])
    if case["expect"] is None:
        assert out is None
    else:
        assert_result_equal(out, case["expect"], 0, case["id"])


# ---- Smooth ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", load_family("smooth"), ids=lambda c: c["id"])
def test_smooth_golden(gpu, case):
    """Gaussian smoothing (SciPy's correlate1d order in float64, axis by axis) + crop / zoom against
    the reference's own results: bit-exact."""
    from dask_geomodeling_amd import raster
    out = raster.Smooth.process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.gpu
def test_smooth_block_requests_and_large_tile(gpu):
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(31)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 300, 420)) * 100).astype("f4")
    a[0, 50:60, 70:90] = fill
    src = raster.MemorySource(a, fill, "EPSG:28992", 0.5, (0.0, 150.0))
    req = dict(mode="vals", bbox=(20.0, 20.0, 120.0, 100.0), width=200, height=160, projection="EPSG:28992")
    for size in (1.4, 12.0):  # exact (margin 3 px) and zoom mode (margin 24 px)
        block = raster.Smooth(src, size, fill=1.5)
        (store, grown), (kw, _) = block.get_sources_and_requests(**req)
        assert kw["smooth_mode"] == ("exact" if size < 3.5 else "zoom")
        want = O.smooth_process(src.get_data(**grown), kw)
        assert_result_equal(block.get_data(**req), want, 0, "Smooth size %s" % size)
    with pytest.raises(NotImplementedError):
        raster.Smooth(src, 2).get_sources_and_requests(**dict(req, projection="EPSG:4326"))


@pytest.mark.gpu
def test_tiler_over_boolean_view(gpu, tiler_route):
    """A boolean view has no fill value (np.full(shape, None, dtype=bool) is all False in the
    reference's assembly): found by the fuzzer."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(5)
    v = rng.integers(0, 10, size=(1, 24, 40)).astype("i4")
    v[0, ::3, ::4] = -1
    src = raster.MemorySource(v, -1, "EPSG:28992", 1.0, (0.0, 24.0))
    view = raster.GreaterEqual(raster.Clip(raster.Reclassify(src, [[1, 5], [2, 7]]), src), 5)
    req = dict(mode="vals", bbox=(0.0, 0.0, 40.0, 24.0), width=40, height=24, projection="EPSG:28992")
    whole = view.get_data(**req)
    assert whole["values"].dtype == np.dtype("bool") and whole["no_data_value"] is None
    for tile in ([7, 5], 16):
        assert_result_equal(raster.RasterTiler(view, tile).get_data(**req), whole, 0, "bool view, tile %s" % (tile,))


# ---- N4: TemporalAggregate / Cumulative -----------------------------------------------------------------
def _temporal_cases():
    from test_blocks_cpu import temporal_on_device
    return [c for c in load_family("temporal") if c["request"]["mode"] == "vals" and temporal_on_device(c)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _temporal_cases(), ids=lambda c: c["id"] + "-" + c["kind"])
def test_temporal_golden(gpu, case):
    """Whole get_data() of TemporalAggregate / Cumulative views: pandas bins on the host, the
    frame-axis reduction on the device (csrc/gm_temporal.hip), bit-exact against the reference."""
    from fuzz_util import unsigned_zero
    from test_blocks_cpu import temporal_view
    out = temporal_view(case).get_data(**case["request"])
    if case["expect"] is None:
        assert out is None
    else:
        want = case["expect"]
        stat = case["block_args"][1] if case["kind"] == "aggregate" else ""
        if stat == "median" or stat.startswith("p"):
            # -0.0 and +0.0 compare equal: which of the two a sort puts first is the sorting algorithm's
            # business (np.partition is not stable), so the sign of a zero result is not pinned
            out, want = unsigned_zero(out), unsigned_zero(want)
        assert_result_equal(out, want, 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,frames", [("f4", 70), ("f4", 365), ("f4", 599), ("f8", 200), ("f8", 320), ("i2", 130)])
def test_temporal_median_and_percentiles_over_long_bins(gpu, dtype, frames):
    """Beyond 64 frames per label (a year of daily frames ...) the wanted ranks are selected in global memory (round 5; until
    then an insertion sort in 64-lane workgroups took bins of up to 599 float32 / 320 float64 frames -- still there behind
    GM_MEDIAN_LONG_SORT=1: test_temporal_medians_over_long_bins_by_the_lds_sort), the percentile plan in device memory."""
    from dask_geomodeling_amd.raster import temporal as T
    rng = np.random.default_rng(frames)
    ndv = np.dtype(dtype).type(-999)
    v = (rng.random((frames, 9, 37)) * 200 - 100).astype(dtype)
    v[rng.random(v.shape) < 0.3] = ndv
    v[:, 0, 0] = ndv
    runs = [(0, frames), (3, frames - 5), (0, 10)]
    for stat in ("median", "p10.0", "p50.0", "p97.5"):
        out_dtype = T.dtype_for_statistic(dtype, stat)
        got = T.aggregate_frames(v, ndv, runs, stat, out_dtype).to_numpy() + 0.0
        want = O.temporal_aggregate_values(v, ndv, runs, stat, out_dtype) + 0.0
        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (stat, dtype, frames)
    # (one frame more than the LDS sort takes: selected in global memory since round 3 -- no limit any more;
    #  test_median_and_percentiles_over_bins_longer_than_the_lds_sort compares those with the oracle)
    one_more = 600 if np.dtype(out_dtype).itemsize == 4 else 321
    zeros = T.aggregate_frames(np.zeros((one_more, 2, 4), dtype), None, [(0, one_more)], "median", out_dtype).to_numpy()
    assert zeros.shape == (1, 2, 4) and not zeros.any()


@pytest.mark.gpu
def test_temporal_medians_over_long_bins_by_the_lds_sort(gpu):
    """The test above once more with the 64-lane LDS insertion sort in place of the selection (GM_MEDIAN_LONG_SORT=1, read once
    per process: a process of its own)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_MEDIAN_LONG_SORT="1")
    here = os.path.abspath(__file__)
    res = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                          "test_temporal_median_and_percentiles_over_long_bins"], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0 and "6 passed" in res.stdout, (res.stdout[-800:], res.stderr[-800:])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8"])
@pytest.mark.parametrize("n,shape", [(2, (1, 7, 13)), (9, (2, 33, 64)), (40, (1, 64, 128)), (300, (1, 5, 12))])
def test_reduce_rasters_in_place_vs_the_stack_and_the_oracle(gpu, monkeypatch, dt, n, shape):
    """Round 6: float rasters of the work dtype are reduced where they are (gm_reduce_rasters: one launch over n tile pointers,
    the NaN coding of reduction.py:99-107 in the load) instead of through a coded stack -- every statistic of the kernel, no-data
    values that are finite, infinite and NaN (np.isclose with each), rasters with their own no-data values, tiles that do and do
    not sit on 16 bytes: bit for bit the stack route's and the oracle's."""
    from dask_geomodeling_amd.raster import reduction
    rng = np.random.default_rng(n * 31 + len(shape))
    ndvs = (dt == "f4" and np.float32(-9999.0) or np.float64(-9999.0), np.dtype(dt).type(np.inf), np.dtype(dt).type(np.nan),
            np.dtype(dt).type(3.0e38 if dt == "f4" else 1e300))
    stats = ("sum", "mean", "min", "max", "std", "var", "product", "argmin", "argmax", "first", "last", "count", "median", "p25", "p90.5")
    if n >= 300:   # (300 uploads per route and statistic: one no-data value of each kind, one statistic per kernel family)
        ndvs, stats = ndvs[::2], ("sum", "min", "std", "argmax", "first", "count", "median", "p90.5")
    for ndv0 in ndvs:
        stack = []
        for i in range(n):
            v = (rng.standard_normal(shape) * 50).astype(dt)
            ndv = ndv0 if i % 3 else np.dtype(dt).type(-1.0)   # (every third raster: another no-data value)
            v[rng.random(shape) < 0.3] = ndv
            v[rng.random(shape) < 0.02] = 0.0
            v[rng.random(shape) < 0.02] = -0.0
            if i == 1:
                v[0, 0, :3] = [np.inf, -np.inf, np.nan]   # (data that is not finite: NaN data drops out like no data)
            stack.append({"values": v, "no_data_value": ndv})
        for i in range(min(n, 3)):
            stack[i]["values"][0, -1, -1] = stack[i]["no_data_value"]   # one cell without data anywhere ...
        for d in stack[3:]:
            d["values"][0, -1, -1] = d["no_data_value"]
        for stat in stats:
            with np.errstate(all="ignore"):
                want = O.reduce_rasters([dict(d) for d in stack], stat)
            got = reduction.reduce_rasters([dict(d) for d in stack], stat)
            monkeypatch.setattr(reduction, "_RASTERS_IN_PLACE", False)
            through_stack = reduction.reduce_rasters([dict(d) for d in stack], stat)
            monkeypatch.setattr(reduction, "_RASTERS_IN_PLACE", True)
            what = "%s %s n=%d ndv=%r" % (stat, dt, n, ndv0)
            a, b = dict(got, values=np.asarray(got["values"])), dict(through_stack, values=np.asarray(through_stack["values"]))
            if stat == "median" or stat.startswith("p"):  # (the register network and the LDS sort may leave the two zeros in either order)
                from fuzz_util import unsigned_zero as _uz
                a, b = _uz(a), _uz(b)
            assert_result_equal(a, b, 0, what + " (routes)")
            mine = dict(got, values=np.asarray(got["values"]))
            if stat in ("min", "max", "median") or stat.startswith("p"):
                # (which zero a tie between -0.0 and +0.0 leaves is a property of the host's SIMD loops: DESIGN.md section 2)
                from fuzz_util import unsigned_zero
                mine, want = unsigned_zero(mine), unsigned_zero(want)
            assert_result_equal(mine, want, 0, what)


@pytest.mark.gpu
def test_reduce_rasters_median_over_many_rasters(gpu):
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    rng = np.random.default_rng(8)
    ndv = np.float32(-1)
    stack = []
    for _ in range(90):
        v = (rng.random((1, 12, 20)) * 50).astype("f4")
        v[rng.random(v.shape) < 0.2] = ndv
        stack.append({"values": v, "no_data_value": ndv})
    for stat in ("median", "p25"):
        got = reduce_rasters([dict(d) for d in stack], stat)
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], stat)
        assert_result_equal(dict(got, values=np.asarray(got["values"]) + 0.0), dict(want, values=want["values"] + 0.0), 0, stat)
    # 1-byte rasters (a float16 stack): the LDS sort as far as it reaches -- 599 rasters --, where float32 / float64 stacks are
    # selected from 65 rasters on (round 5)
    for n in (70, 300):
        stack = []
        for _ in range(n):
            v = rng.integers(0, 200, size=(1, 6, 9)).astype("u1")
            v[rng.random(v.shape) < 0.2] = 255
            stack.append({"values": v, "no_data_value": np.uint8(255)})
        for stat in ("median", "p25"):
            got = reduce_rasters([dict(d) for d in stack], stat)
            with np.errstate(all="ignore"):
                want = O.reduce_rasters([dict(d) for d in stack], stat)
            assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, "%s of %d uint8 rasters" % (stat, n))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,ndv", [("f4", np.float32(np.finfo("f4").max)), ("f8", -9999.0), ("u1", np.uint8(255)),
                                       ("i2", np.int16(-1)), ("i4", 2147483647), ("u2", np.uint16(65535)),
                                       ("i8", -1), ("u4", np.uint32(7))])
@pytest.mark.parametrize("shape_yx", [(64, 96), (37, 53)])
def test_temporal_kernels_against_oracle(gpu, dtype, ndv, shape_yx):
    """Every statistic x dtype on stacks larger than a workgroup (vector and scalar pixel paths, bins
    longer than the unroll, empty bins, more bins than one launch holds) against the oracle."""
    from dask_geomodeling_amd.raster import temporal as T
    rng = np.random.default_rng(hash((dtype,) + shape_yx) % 2**32)
    frames = 23
    if np.dtype(dtype).kind == "f":
        v = (rng.random((frames,) + shape_yx) * 200 - 100).astype(dtype)
        v[3, 2, 3] = np.inf
        v[4, 5, 6] = np.nan
    else:
        v = rng.integers(0, 100, size=(frames,) + shape_yx).astype(dtype)
    v[rng.random(v.shape) < 0.2] = ndv
    v[:, 1, 1] = ndv
    runs = [(0, 1), (1, 2), (3, 0), (3, 9), (12, 11), (0, 23), (22, 1)] + [(k % 20, 3) for k in range(100)]
    for stat in ("sum", "count", "min", "max", "mean", "var", "std", "median", "p0.0", "p37.5", "p50.0", "p99.0", "p100.0"):
        out_dtype = T.dtype_for_statistic(dtype, stat)
        # (np.nanpercentile walks the pixels one by one in Python: a few bins are plenty for the oracle)
        some = runs[:8] if stat.startswith("p") else runs
        got = T.aggregate_frames(v, ndv, some, stat, out_dtype).to_numpy()
        want = O.temporal_aggregate_values(v, ndv, some, stat, out_dtype)
        if stat == "median" or stat.startswith("p"):
            got, want = got + 0.0, want + 0.0  # (the order of -0.0 and +0.0 in a sort is not pinned)
        if stat in ("min", "max") and np.dtype(dtype).kind == "f":
            got, want = got + 0.0, want + 0.0
        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (stat, dtype)
    cruns = [(0, 5), (5, 1), (6, 10), (16, 7)]
    for stat in ("sum", "count"):
        out_dtype = T.dtype_for_statistic(dtype, stat)
        for emit in ((0, 23), (4, 18), (22, 23)):
            got = T.accumulate_frames(v, ndv, cruns, emit, stat, out_dtype).to_numpy()
            want = O.temporal_cumulative_values(v, ndv, cruns, emit, stat, out_dtype)
            assert got.dtype == want.dtype and np.array_equal(got, want), (stat, dtype, emit)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["u1", "i1", "?"])
@pytest.mark.parametrize("shape", [(1, 40, 70), (3, 19, 23), (2, 130, 1031), (1, 11, 6), (1, 21, 64), (2, 35, 2064),
                                   (1, 50, 4115), (3, 3, 66), (1, 18, 2050), (1, 19, 2051)])
def test_dilate_one_byte_tiles(gpu, dtype, shape):
    """The word kernels of 1-byte tiles (8 pixels per lane from 64 columns on, 4 below) (ragged widths, several bands, 1..4 values in
    order, rows that end inside a lane's 8-byte window) against the oracle."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(hash((dtype,) + shape) % 2**32)
    if dtype == "?":
        v = rng.random(shape) < 0.2
        value_sets = [[True], [False, True]]
    else:
        v = rng.integers(-3 if dtype == "i1" else 0, 6, size=shape).astype(dtype)
        value_sets = [[1], [3, 1], [1, 3, 5], [5, 0, 2, 4], [2, 2]]
    for values in value_sets:
        want = O.dilate_process({"values": v, "no_data_value": None}, values)["values"]
        got = stencil.dilate(v, values).to_numpy()
        assert got.dtype == want.dtype and np.array_equal(got, want), (dtype, shape, values)


@pytest.mark.gpu
def test_temporal_views_end_to_end(gpu, tiler_route):
    """TemporalAggregate / Cumulative inside larger views: element-wise programs in front (device tiles
    handed over without a host round trip) and behind, a RasterTiler on top, on a temporal MemorySource."""
    from datetime import datetime
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(5)
    frames, h, w = 12, 40, 56
    data = (rng.random((frames, h, w)) * 50).astype("f4")
    data[rng.random(data.shape) < 0.1] = -9999.0
    hour = 3600 * 1000
    src = raster.MemorySource(data, -9999.0, "EPSG:28992", 1.0, (0.0, float(h)), time_first=0, time_delta=hour)
    req = dict(mode="vals", bbox=(0.0, 0.0, float(w), float(h)), width=w, height=h, projection="EPSG:28992",
               start=datetime(1970, 1, 1, 0), stop=datetime(1970, 1, 1, 11))
    kw32 = {"dtype": "float32", "fillvalue": float(np.finfo("f4").max)}
    # oracle, Block by Block: (src * 2 + 1) -> 4-hourly mean -> minus 3
    x = O.math_process("multiply", kw32, {"values": data, "no_data_value": np.float32(-9999.0)}, 2)
    x = O.math_process("add", kw32, x, 1)
    runs = [(0, 4), (4, 4), (8, 4)]
    agg = O.temporal_aggregate_values(x["values"], x["no_data_value"], runs, "mean", np.dtype("f4"))
    want = O.math_process("subtract", kw32, {"values": agg, "no_data_value": kw32["fillvalue"]}, 3)
    view = raster.TemporalAggregate(src * 2 + 1, "4h", "mean") - 3
    got = view.get_data(**req)
    assert_result_equal(got, want, 0, "aggregate view")
    tiled = raster.RasterTiler(view, [16, 24]).get_data(**req)
    assert_result_equal(tiled, want, 0, "tiled aggregate view")
    # running sum per 6-hour period of the same source
    cum = raster.Cumulative(src, "sum", "6h").get_data(**req)
    want_cum = O.temporal_cumulative_values(data, np.float32(-9999.0), [(1, 6), (7, 5)], (1, 12), "sum", np.dtype("f4"))
    first = O.temporal_cumulative_values(data, np.float32(-9999.0), [(0, 1)], (0, 1), "sum", np.dtype("f4"))
    assert np.array_equal(np.asarray(cum["values"]), np.concatenate([first, want_cum]))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8"])
def test_divide_by_power_of_two_literal(gpu, dt):
    """x / c with c a power of two is lowered to x * (1 / c): must be np.divide bit for bit on every kind of
    float (subnormal results, the largest finite values, infinities, NaN, no data, signed zeros)."""
    from blocks_util import ArrayLeaf
    from dask_geomodeling_amd import raster
    info = np.finfo(dt)
    ndv = dt == "f4" and np.float32(info.max) or float(info.max)
    rng = np.random.default_rng(3)
    special = np.array([0.0, -0.0, info.tiny, info.tiny * 3, info.tiny / 4, -info.tiny / 8, info.smallest_subnormal,
                        info.max, -info.max, info.max / 2, info.max / 3, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0,
                        1e-30, 1e30, 7.5, ndv, np.nextafter(info.max, 0)], dtype=dt)
    values = np.concatenate([special, (rng.standard_normal(42) * 10.0 ** rng.integers(-40, 39, 42)).astype(dt)])[None, None, :]
    ArrayLeaf.ARRAYS["div/p2/" + dt] = {"values": values, "no_data_value": ndv}
    leaf = ArrayLeaf("div/p2/" + dt, np.dtype(dt).str, ndv)
    h, w = values.shape[1:]
    req = dict(mode="vals", bbox=(0.0, 0.0, float(w), float(h)), width=w, height=h, projection="EPSG:28992")
    kw = {"dtype": np.dtype(dt).name, "fillvalue": float(info.max)}
    for c in (2.0, 0.5, 4.0, -8.0, 2.0 ** -20, 2.0 ** 100, 2.0 ** 126, 2.0 ** 127, -(2.0 ** -126), 3.0, 1.0, 1024):
        with np.errstate(all="ignore"):
            want = O.math_process("divide", kw, {"values": values, "no_data_value": ndv}, c)
            want2 = O.math_process("add", kw, want, 1.5)
        got = raster.Divide(leaf, c).get_data(**req)
        assert_result_equal(got, want, 0, "x / %r" % c)
        got2 = (raster.Divide(leaf, c) + 1.5).get_data(**req)
        assert_result_equal(got2, want2, 0, "x / %r + 1.5" % c)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,ndv", [("f4", np.float32(np.finfo("f4").max)), ("f8", -9999.0), ("i4", -1), ("u1", np.uint8(255))])
def test_clip_by_folded_comparison(gpu, dt, ndv):
    """Clip(x, D CMP literal): the comparison is folded into the Clip micro-op (UOP_CLIPCMP) -- all six
    comparisons, mask tiles with no data / NaN / inf / cells equal to the literal, against the oracle's
    two separate process() bodies."""
    from blocks_util import ArrayLeaf
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(11)
    shape = (1, 33, 47)
    if np.dtype(dt).kind == "f":
        d = (rng.random(shape) * 40).astype(dt)
        d[0, 0, :6] = [np.nan, np.inf, -np.inf, 20.0, 20.000001, 19.999999]
    else:
        d = rng.integers(0, 40, size=shape).astype(dt)
        d[0, 0, :3] = [20, 21, 19]
    d[rng.random(shape) < 0.15] = ndv
    a = (rng.random(shape) * 100).astype("f4")
    a[rng.random(shape) < 0.1] = -9999.0
    ArrayLeaf.ARRAYS["cc/d/" + dt] = {"values": d, "no_data_value": ndv}
    ArrayLeaf.ARRAYS["cc/a"] = {"values": a, "no_data_value": np.float32(-9999.0)}
    D, A = ArrayLeaf("cc/d/" + dt, np.dtype(dt).str, ndv), ArrayLeaf("cc/a", "<f4", np.float32(-9999.0))
    req = dict(mode="vals", bbox=(0.0, 0.0, 47.0, 33.0), width=47, height=33, projection="EPSG:28992")
    kw = {"dtype": "float32", "fillvalue": float(np.finfo("f4").max)}
    kwd = {"dtype": np.result_type(np.float32, dt).name, "fillvalue": float(np.finfo(np.result_type(np.float32, dt)).max)}
    for name, op in (("Equal", "equal"), ("NotEqual", "not_equal"), ("Greater", "greater"), ("GreaterEqual", "greater_equal"),
                     ("Less", "less"), ("LessEqual", "less_equal")):
        mask = O.math_process(op, {"dtype": "bool", "fillvalue": None}, {"values": d, "no_data_value": ndv}, 20)
        with np.errstate(all="ignore"):
            store = O.math_process("add", kw, {"values": a, "no_data_value": np.float32(-9999.0)}, 1.5)
            want = O.clip_process(store, mask)
            want_after = O.math_process("multiply", kw, want, 2)
        view = raster.Clip(A + 1.5, getattr(raster, name)(D, 20))
        assert_result_equal(view.get_data(**req), want, 0, name)
        assert_result_equal((view * 2).get_data(**req), want_after, 0, name + " * 2")
        # the same mask used twice is not folded (two users of the temp): still right
        both = raster.Clip(raster.Clip(A + 1.5, getattr(raster, name)(D, 20)) * 2, getattr(raster, name)(D, 20))
        assert_result_equal(both.get_data(**req), O.clip_process(want_after, mask), 0, name + " twice")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8"])
@pytest.mark.parametrize("sigma", [(0.5, 0.5), (2.0, 1.0), (3.0, 3.0), (5.5, 0.0), (1.0, 7.0)])
def test_gauss_axis_kernels_against_oracle(gpu, dt, sigma):
    """The one-axis Gaussian kernel (taps through a sliding register window fed from LDS) on ragged tiles larger
    than a workgroup window, several bands, no data and a non-zero fill, against the oracle's restatement of
    SciPy's correlate1d (bit-exact); radii below 8 (every tap its own read), 8..16 and with a remainder."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(int(sigma[0] * 10 + sigma[1] * 100))
    ndv = np.dtype(dt).type(-9999.0)
    v = (rng.random((2, 75, 531)) * 50).astype(dt)
    v[rng.random(v.shape) < 0.05] = ndv
    for fill, crop in ((0, (0, 0)), (2.5, (3, 5)), (0.1, (0, 2))):
        got = stencil.gaussian_smooth(v, ndv, fill, sigma, crop).to_numpy()
        # (Smooth.process ties the crop to the sigma; here they vary independently, so the oracle's steps
        #  are spelled out: no data -> fill, one correlate1d per axis, crop)
        work = v.copy()
        work[work == ndv] = fill
        for axis, s in ((1, sigma[0]), (2, sigma[1])):
            if s > 1e-15:
                work = O.gaussian_axis(work, s, axis, fill)
        want = work[:, crop[0]:work.shape[1] - crop[0], crop[1]:work.shape[2] - crop[1]]
        assert got.dtype == want.dtype and np.array_equal(got, want), (dt, sigma, fill, crop)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8", "i2", "u1", "i8", "?"])
@pytest.mark.parametrize("shape,sigma,crop", [
    ((2, 300, 517), (13.07, 13.07), (40, 40)),   # the zoom mode of a 40-pixel Smooth: radius 52 both ways
    ((1, 300, 517), (30.0, 2.0), (0, 0)),        # radius 120 along y: a window beyond 64 KiB of LDS
    ((1, 97, 1030), (0.7, 50.3), (3, 7)),        # radius 201 along x, lines shorter than the radius along y
    ((1, 700, 130), (80.0, 80.0), (0, 0)),       # radius 320 > the tile's width
    ((1, 64, 64), (100.0, 100.0), (0, 0)),       # the kernel many times the tile
    ((1, 1200, 70), (200.0, 3.0), (5, 5)),       # radius 800 along y: no window fits, taps from memory
    ((1, 40, 5000), (2.0, 600.0), (0, 9)),       # radius 2400 along x: the same for the other axis
])
def test_gauss_any_radius(gpu, dt, shape, sigma, crop):
    """Smooth has no largest radius (spatial.py:216-318: a zoomed request of N pixels can ask for up to 2 N / 3): the
    sliding-window kernel while its window fits 160 KiB of LDS, taps from memory beyond, weights from device memory either
    way -- bit-exact against the oracle's restatement of SciPy for every dtype family, no data, a fill the dtype cannot
    hold, several bands, crops."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(shape[1] + shape[2])
    if dt == "?":
        v, ndv = rng.random(shape) < 0.5, None
    else:
        v = (rng.random(shape) * 200).astype(dt)
        ndv = np.array(199 if dt == "u1" else 200).astype(dt)[()]
        v[:, ::7, ::5] = ndv
    for fill in (3, 0.25):
        got = stencil.gaussian_smooth(v, ndv, fill, sigma, crop, fused=False).to_numpy()
        work = v.copy()
        if ndv is not None:
            work[work == ndv] = fill
        for axis, s in ((1, sigma[0]), (2, sigma[1])):
            work = O.gaussian_axis(work, s, axis, fill)
        want = work[:, crop[0]:work.shape[1] - crop[0], crop[1]:work.shape[2] - crop[1]]
        assert got.dtype == want.dtype and np.array_equal(got, want), (dt, shape, sigma, fill)


@pytest.mark.gpu
@pytest.mark.parametrize("sigma", [(0.3, 0.3), (0.5, 0.5), (1.0, 1.0), (1.3, 1.4), (2.0, 2.0), (2.2, 2.2), (3.0, 3.0),
                                   (4.0, 4.0), (3.9, 4.1)])
@pytest.mark.parametrize("shape", [(2, 75, 531), (1, 200, 700), (1, 9, 70)])
def test_gauss_both_axes_in_one_launch(gpu, sigma, shape):
    """float32 tiles whose axes share a radius (1..16) are filtered by ONE kernel (y axis into LDS rounded to
    float32, x axis from there): bit-exact against the oracle's axis-by-axis restatement and against the
    two one-axis launches, on tiles spanning several workgroup windows both ways, with no data, non-zero
    fills that float32 cannot hold (the constant extension stays float64) and crops."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(int(sigma[0] * 10 + sigma[1] * 100) + shape[2])
    ndv = np.float32(-9999.0)
    v = (rng.random(shape) * 50).astype("f4")
    v[rng.random(v.shape) < 0.05] = ndv
    for fill, crop in ((0, (0, 0)), (2.5, (3, 5)), (0.1, (1, 0))):
        got = stencil.gaussian_smooth(v, ndv, fill, sigma, crop).to_numpy()
        two = stencil.gaussian_smooth(v, ndv, fill, sigma, crop, fused=False).to_numpy()
        work = v.copy()
        work[work == ndv] = fill
        for axis, s in ((1, sigma[0]), (2, sigma[1])):
            work = O.gaussian_axis(work, s, axis, fill)
        want = work[:, crop[0]:work.shape[1] - crop[0], crop[1]:work.shape[2] - crop[1]]
        assert got.dtype == want.dtype and np.array_equal(got, want), (sigma, fill, crop)
        assert np.array_equal(two, want), (sigma, fill, crop, "two launches")


@pytest.mark.gpu
@pytest.mark.parametrize("dt,ndv", [("u1", np.uint8(255)), ("u1", np.uint8(0)), ("i1", np.int8(-1)), ("?", None)])
def test_single_byte_input_programs_by_table(gpu, dt, ndv, tiler_route):
    """Programs whose only input is a 1-byte tile run as a 256-entry table (built by the VM itself) + the
    byte-LUT kernel from 4096 elements on: every output width, ragged sizes (scalar tail), batches, and
    zero-copy windows of a resident raster (strided tiles; widths the LUT kernel cannot address fall
    back to the VM) -- against the oracle, Block by Block."""
    from blocks_util import ArrayLeaf
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(17)
    shape = (1, 70, 97)  # 6790 elements: not a multiple of 16
    if dt == "?":
        v = rng.random(shape) < 0.4
    else:
        v = rng.integers(-5 if dt == "i1" else 0, 12, size=shape).astype(dt)
        v[rng.random(shape) < 0.1] = ndv
    key = "lut/%s/%s" % (dt, ndv)
    ArrayLeaf.ARRAYS[key] = {"values": v, "no_data_value": ndv}
    leaf = ArrayLeaf(key, np.dtype(dt).str, ndv)
    req = dict(mode="vals", bbox=(0.0, 0.0, 97.0, 70.0), width=97, height=70, projection="EPSG:28992")
    data = {"values": v, "no_data_value": ndv}
    arith = leaf * 2.5 + 1
    flt = {"dtype": arith.dtype.name, "fillvalue": arith.fillvalue}  # (the Block's own dtype rule: float64 for u8 * python float)
    with np.errstate(all="ignore"):
        cases = [("mask -> u8", raster.Mask(leaf, 7), O.mask_process(data, 7)),
                 ("x * 2.5 + 1 -> float", arith, O.math_process("add", flt, O.math_process("multiply", flt, data, 2.5), 1)),
                 ("x > 3 -> bool", leaf > 3, O.math_process("greater", {"dtype": "bool", "fillvalue": None}, data, 3)),
                 ("classify -> u8", raster.Classify(leaf, [1.5, 3.0, 8.0]), O.classify_process(data, [1.5, 3.0, 8.0], False))]
        if dt != "?":
            table = [[1, 10], [2, 20], [3, 7], [11, 5]]
            cases.append(("reclassify -> i64", raster.Reclassify(leaf, table),
                          O.reclassify_process(data, {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "data": table, "select": False})))
            tf = [[1, 0.5], [4, 2.25]]
            cases.append(("reclassify -> f64, select", raster.Reclassify(leaf, tf, True),
                          O.reclassify_process(data, {"dtype": "<f8", "fillvalue": float(np.finfo("f8").max), "data": tf, "select": True})))
            plus = leaf + 100
            cases.append(("x + 100 -> int", plus, O.math_process("add", {"dtype": plus.dtype.name, "fillvalue": plus.fillvalue}, data, 100)))
    for name, view, want in cases:
        assert_result_equal(view.get_data(**req), want, 0, name)
    if dt == "u1":
        # strided: tiles of a resident raster are windows (pitch 256); 64-wide tiles take the table path,
        # 40-wide ones the VM -- both must equal the untiled result
        big = rng.integers(0, 12, size=(1, 192, 256)).astype("u1")
        big[rng.random(big.shape) < 0.1] = ndv
        src = raster.MemorySource(big, ndv, "EPSG:28992", 1.0, (0.0, 192.0))
        view = raster.Reclassify(src, [[1, 10], [2, 20], [3, 7]]) 
        whole_req = dict(mode="vals", bbox=(0.0, 0.0, 256.0, 192.0), width=256, height=192, projection="EPSG:28992")
        whole = view.get_data(**whole_req)
        want = O.reclassify_process({"values": big, "no_data_value": ndv},
                                    {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "data": [[1, 10], [2, 20], [3, 7]], "select": False})
        assert_result_equal(whole, want, 0, "whole")
        for tile in ([64, 64], [40, 96], [64, 48]):
            assert_result_equal(raster.RasterTiler(view, tile).get_data(**whole_req), want, 0, "tiles %s" % tile)
        # windows that start on ANY byte and have any width (round 5: the table kernel's row form; they ran on the VM before):
        # every output width
        views = [("reclassify -> i64", view, want["values"]),
                 ("mask -> u8", raster.Mask(src, 7), O.mask_process({"values": big, "no_data_value": ndv}, 7)["values"]),
                 ("x + 100 -> int", src + 100, None)]
        plus = src + 100
        views[2] = ("x + 100 -> int", plus, O.math_process("add", {"dtype": plus.dtype.name, "fillvalue": plus.fillvalue},
                                                            {"values": big, "no_data_value": ndv}, 100)["values"])
        for col0, row0, w, h in ((1, 3, 97, 70), (65, 33, 127, 50), (3, 0, 253, 192), (17, 5, 200, 111), (255, 191, 1, 1), (0, 7, 256, 3)):
            bbox = (float(col0), float(192 - row0 - h), float(col0 + w), float(192 - row0))
            for name, v, full in views:
                got = v.get_data(mode="vals", bbox=bbox, width=w, height=h, projection="EPSG:28992")
                assert np.array_equal(np.asarray(got["values"]), full[:, row0:row0 + h, col0:col0 + w]), (name, (col0, row0, w, h))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 70, 97), (1, 64, 64), (1, 129, 257), (2, 40, 37)])
def test_int32_reclassify_mask_dedicated_kernel(gpu, shape):
    """Mask(Reclassify(int32 tile)) -- configs[2] -- is one LOOKUP program and runs in the dedicated
    kernel (coalesced loads, wave-local LDS transpose, 16-byte stores): windows from 3 to 8000 slots,
    negative keys, exception keys far from the window, the no-data value inside / outside the table,
    select on and off, ragged tile sizes -- against the oracle's two process() bodies."""
    from blocks_util import ArrayLeaf
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(shape[1] * 1000 + shape[2])
    h, w = shape[1:]
    req = dict(mode="vals", bbox=(0.0, 0.0, float(w), float(h)), width=w, height=h, projection="EPSG:28992")
    tables = {
        "small": [[0, 10], [1, 11], [3, 13], [5, 15]],
        "negative": [[-7, 1], [-3, 2], [0, 3], [4, 9]],
        "wide": [[k, (7 * k) % 251] for k in range(0, 8000, 37)],
        "outliers": [[1, 5], [2, 6], [3, 7], [100000, 8], [-2000000, 9]],
        "ndv key": [[2147483647, 4], [1, 2], [2, 3]],
    }
    for tname, table in tables.items():
        keys = np.array([k for k, _ in table])
        for ndv in (2147483647, -9999):
            v = rng.choice(np.concatenate([keys, keys + 1, [0, 1, 2, 3, -1, 7999, 123456]]), size=shape).astype("i4")
            v[rng.random(shape) < 0.1] = ndv
            key = "rcl32/%s/%d/%s" % (tname, ndv, shape)
            ArrayLeaf.ARRAYS[key] = {"values": v, "no_data_value": ndv}
            leaf = ArrayLeaf(key, "<i4", ndv)
            for select in (False, True):
                for value in (1, 0, 200):
                    view = raster.Mask(raster.Reclassify(leaf, table, select), value)
                    r = O.reclassify_process({"values": v, "no_data_value": ndv},
                                             {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "data": table, "select": select})
                    want = O.mask_process(r, value)
                    assert_result_equal(view.get_data(**req), want, 0, "%s ndv %d select %s value %d" % (tname, ndv, select, value))


@pytest.mark.gpu
def test_int32_reclassify_mask_on_windows(gpu, tiler_route):
    """The dedicated LOOKUP kernel on zero-copy windows of a resident int32 raster (strided tiles written
    straight into the mosaic): tile widths it can address (multiples of 16) and widths that fall back to
    the VM must both equal the untiled result and the oracle."""
    from dask_geomodeling_amd import raster
    rng = np.random.default_rng(23)
    ndv = 2147483647
    big = rng.integers(-3, 40, size=(1, 192, 320)).astype("i4")
    big[rng.random(big.shape) < 0.1] = ndv
    table = [[k, (7 * k) % 251] for k in range(0, 40, 3)]
    src = raster.MemorySource(big, ndv, "EPSG:28992", 1.0, (0.0, 192.0))
    req = dict(mode="vals", bbox=(0.0, 0.0, 320.0, 192.0), width=320, height=192, projection="EPSG:28992")
    for select in (False, True):
        view = raster.Mask(raster.Reclassify(src, table, select), 3)
        r = O.reclassify_process({"values": big, "no_data_value": ndv},
                                 {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "data": table, "select": select})
        want = O.mask_process(r, 3)
        assert_result_equal(view.get_data(**req), want, 0, "whole")
        for tile in ([64, 64], [48, 80], [96, 160], [40, 72], [64, 100]):
            assert_result_equal(raster.RasterTiler(view, tile).get_data(**req), want, 0, "tiles %s select %s" % (tile, select))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [3, 5, 7, 9, 11, 13, 15, 17])
@pytest.mark.parametrize("dt,ndv", [("f4", np.float32(-9999.0)), ("u1", np.uint8(255)), ("i2", np.int16(-1)), ("i4", 2147483647), ("f8", 1e300)])
def test_moving_max_kernels_against_oracle(gpu, size, dt, ndv):
    """The separable 3 x 3 kernel, the fixed-footprint kernels (5 .. 15) and the general kernel (17; 8-byte
    tiles at every size) on ragged multi-band tiles larger than a workgroup tile, with no-data regions
    wide enough to leave all-no-data windows, against the oracle (bit-exact)."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(size * 100 + len(dt))
    shape = (2, 45 + size, 533 + size)
    if np.dtype(dt).kind == "f":
        v = (rng.random(shape) * 200 - 100).astype(dt)
    else:
        v = rng.integers(0, 100, size=shape).astype(dt)
    v[rng.random(shape) < 0.15] = ndv
    v[:, 10:10 + 2 * size, 40:40 + 2 * size] = ndv  # a hole larger than the footprint
    v[0, 0, 0] = np.finfo(dt).min if np.dtype(dt).kind == "f" else np.iinfo(dt).min  # the dtype's minimum as data
    want = O.movingmax_process({"values": v, "no_data_value": ndv}, size)["values"]
    got = stencil.moving_max(v, ndv, size).to_numpy()
    assert got.dtype == want.dtype and np.array_equal(got, want), (size, dt)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["i2", "u2", "i4", "f4", "f8"])
@pytest.mark.parametrize("shape", [(1, 45, 533), (1, 18, 258), (3, 20, 70)])
def test_dilate_kernels_against_oracle(gpu, dt, shape):
    """Dilate on 2- / 4-byte single-band tiles (LDS-tile kernel), on multi-band and 8-byte tiles (column
    walker): ragged shapes around the workgroup tile, 1..3 values in order, against the oracle."""
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(len(dt) * 7 + shape[2])
    v = rng.integers(0, 7, size=shape).astype(dt)
    for values in ([1], [3, 1], [5, 0, 2]):
        want = O.dilate_process({"values": v, "no_data_value": None}, values)["values"]
        got = stencil.dilate(v, values).to_numpy()
        assert got.dtype == want.dtype and np.array_equal(got, want), (dt, shape, values)


# ---- N4: zonal statistics (AggregateRaster's numeric core) -----------------------------------------------
def _zonal_runs():
    for case in load_family("zonal"):
        for run in case["runs"]:
            yield pytest.param(case, run, id="%s-%s%s%s" % (case["id"], run["statistic"], run["percentile"] or "",
                                                           "-thr" if run["thresholds"] else ""))


@pytest.mark.gpu
@pytest.mark.parametrize("case,run", list(_zonal_runs()))
def test_zonal_golden(gpu, case, run):
    """gm_zonal_statistic through geometry.aggregate_labelled against the results of the reference's own
    aggregate_polygons (306 runs: 9 dtypes, disjoint and overlapping geometries = several buckets, thresholds,
    a geometry without cells, a geometry without active cells): bit-exact, sums included."""
    from dask_geomodeling_amd import geometry
    thresholds = case["threshold_values"] if run["thresholds"] else None
    args = (case["buckets"], case["n"], case["values"], case["no_data_value"], thresholds, run["statistic"], run["percentile"])
    if "raises" in run:
        with pytest.raises(TypeError):
            geometry.aggregate_labelled(*args)
        return
    agg, no_cells = geometry.aggregate_labelled(*args)
    assert_values_equal(agg, run["expect"], 0, case["id"])
    assert no_cells == run["no_cells"]


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8", "i4", "u1", "i8"])
def test_zonal_large_tiles_vs_oracle(gpu, dt):
    """A 1500 x 2000 raster, 700 labels from one cell to 400 k cells (the sequential float64 sums run over many
    64-cell steps), NaN cells and signed zeros in the float rasters, thresholds: every statistic against the
    oracle -- sum and mean bit for bit."""
    from dask_geomodeling_amd import geometry
    rng = np.random.default_rng(17)
    h, w, n = 1500, 2000, 700
    labels = np.full((h, w), np.iinfo(np.int32).max, dtype=np.int32)
    labels[:, :] = (np.arange(h)[:, None] // 60) * 26 + np.arange(w)[None, :] // 80   # 25 x 25 blocks -> labels 0..649
    labels[900:1400, 100:900] = 660                                                  # one label of 400 k cells
    labels[5, 7] = 699                                                                # one of a single cell
    labels[rng.random((h, w)) < 0.02] = np.iinfo(np.int32).max
    if np.dtype(dt).kind == "f":
        values = (rng.standard_normal((2, h, w)) * rng.choice([1e-3, 1.0, 1e4], size=(2, h, w))).astype(dt)
        values[rng.random(values.shape) < 0.001] = np.nan
        values[rng.random(values.shape) < 0.01] = -0.0
        ndv = np.dtype(dt).type(-9999.0)
    else:
        info = np.iinfo(dt)
        values = rng.integers(max(info.min, -30000), min(info.max, 30000), size=(2, h, w)).astype(dt)
        ndv = np.dtype(dt).type(info.max)
    values[rng.random(values.shape) < 0.1] = ndv
    thresholds = (rng.standard_normal(n) * 0.5).astype("f4")
    thresholds[3] = np.nan
    buckets = [(list(range(n)), labels)]
    for stat, q in (("sum", None), ("mean", None), ("count", None), ("min", None), ("max", None), ("median", None),
                    ("percentile", 12.5), ("percentile", 50.0)):
        for thr in (None, thresholds):
            want, want_no = O.aggregate_labelled(buckets, n, values, ndv, thr, stat, q)
            got, got_no = geometry.aggregate_labelled(buckets, n, values, ndv, thr, stat, q)
            assert_values_equal(got, want, 0, "%s %s %s" % (dt, stat, q))
            assert got_no == want_no


@pytest.mark.gpu
def test_zonal_sums_without_the_sort_when_order_cannot_matter(gpu, monkeypatch):
    """sum / mean skip the sort when every partial sum of every label is exactly representable (integer rasters,
    dyadic data): the parallel sum then has the bits of bincount's sequential one.  Both routes against the oracle
    on the same data, and data that must NOT qualify (many octaves; one NaN; one huge outlier) still right."""
    from dask_geomodeling_amd import geometry
    rng = np.random.default_rng(23)
    h, w, n = 700, 900, 40
    labels = ((np.arange(h)[:, None] // 140) * 8 + np.arange(w)[None, :] // 120).astype(np.int32)
    labels[rng.random((h, w)) < 0.05] = np.iinfo(np.int32).max
    tiles = {
        "i4": rng.integers(-50000, 50000, size=(2, h, w)).astype("i4"),
        "u1": rng.integers(0, 255, size=(2, h, w)).astype("u1"),
        "i8 large": rng.integers(-2**40, 2**40, size=(1, h, w)).astype("i8"),
        "f4 quarters": (rng.integers(-4000, 4000, size=(2, h, w)) / 4).astype("f4"),
        "f4 octaves": (rng.standard_normal((2, h, w)) * rng.choice([1e-6, 1.0, 1e6], size=(2, h, w))).astype("f4"),
        "f8 outlier": rng.integers(0, 100, size=(1, h, w)).astype("f8"),
        "f4 nan": rng.integers(0, 100, size=(1, h, w)).astype("f4"),
    }
    tiles["f8 outlier"][0, 3, 3] = 1e300
    tiles["f8 outlier"][0, 3, 4] = 1e-300
    tiles["f4 nan"][0, 650, 10] = np.nan
    thresholds = (rng.standard_normal(n) * 10).astype("f4")
    for name, values in tiles.items():
        ndv = values.dtype.type(77)
        for stat in ("sum", "mean"):
            for thr in (None, thresholds):
                want, _ = O.aggregate_labelled([(list(range(n)), labels)], n, values, ndv, thr, stat)
                got, _ = geometry.aggregate_labelled([(list(range(n)), labels)], n, values, ndv, thr, stat)
                assert_values_equal(got, want, 0, "%s %s" % (name, stat))
                monkeypatch.setenv("GM_ZONAL_SORTED", "1")
                forced, _ = geometry.aggregate_labelled([(list(range(n)), labels)], n, values, ndv, thr, stat)
                monkeypatch.delenv("GM_ZONAL_SORTED")
                assert_values_equal(forced, want, 0, "%s %s (sorted route)" % (name, stat))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "u1", "i2", "i4", "f8"])
def test_stencils_on_zero_copy_windows(gpu, dt):
    """Dilate / MovingMax / HillShade / Smooth on a strided window of a resident raster (rows `pitch` elements apart):
    the LDS-tile kernels read the window in place (gm_*_window); dtypes / shapes they do not cover are compacted first
    (GM_ERR_UNSUPPORTED -> copy -> plain call).  Either way the result is that of the compact tile and of the oracle;
    windows at the raster's corners make the staging's clamped reads touch the first / last rows of the allocation."""
    from dask_geomodeling_amd.hip.array import DeviceArray
    from dask_geomodeling_amd.raster import stencil
    rng = np.random.default_rng(31)
    H, W = 420, 2300
    if np.dtype(dt).kind == "f":
        full = np.cumsum(rng.standard_normal((1, H, W)), axis=2).astype(dt)
        ndv = np.dtype(dt).type(-9999.0)
    else:
        full = rng.integers(0, 9, size=(1, H, W)).astype(dt)
        ndv = np.dtype(dt).type(7)
    full[rng.random(full.shape) < 0.1] = ndv
    resident = DeviceArray.from_numpy(full)
    for row0, col0, h, w in ((0, 0, 150, 700), (37, 111, 300, 2100), (H - 90, W - 530, 90, 530), (5, 3, 40, 66)):
        win = resident.window(row0, col0, h, w)
        assert win.pitch == W
        host = np.ascontiguousarray(full[:, row0:row0 + h, col0:col0 + w])
        data = {"values": host, "no_data_value": ndv}
        for size in (3, 7, 21):
            got = stencil.moving_max(win, ndv, size).to_numpy()
            assert np.array_equal(got, O.movingmax_process(data, size)["values"]), ("movingmax", size)
        for values in ([1], [3, 1, 5]):
            got = stencil.dilate(win, values).to_numpy()
            assert np.array_equal(got, O.dilate_process(data, values)["values"]), ("dilate", values)
        kw = dict(resolution=(0.5, 2.0), altitude=45.0, azimuth=315.0, fill=1)
        got = stencil.hillshade(win, ndv, 1, kw["resolution"], 45.0, 315.0).to_numpy()
        with np.errstate(all="ignore"):
            assert np.array_equal(got, O.hillshade_process(data, kw, trig="rounded")["values"]), "hillshade"
        for sigma in ((1.0, 1.0), (2.0, 0.7), (6.0, 6.0), (0.0, 9.1)):  # (one launch; the one-axis kernel, small and large radii)
            got = stencil.gaussian_smooth(win, ndv, 0.5, sigma, (1, 2)).to_numpy()
            assert np.array_equal(got, stencil.gaussian_smooth(host, ndv, 0.5, sigma, (1, 2)).to_numpy()), ("smooth", sigma)


@pytest.mark.gpu
def test_zonal_edge_cases(gpu):
    """Corners of geometry.label_statistics: no labels at all, nothing labelled, a raster without cells, zero frames, a
    million labels of which three occur, thresholds that are all NaN, a no-data value the dtype cannot hold, labels outside
    [0, n), int8 percentiles whose difference wraps -- against the oracle's per-bucket restatement."""
    from dask_geomodeling_amd import geometry
    rng = np.random.default_rng(3)
    values = (rng.standard_normal((2, 30, 40)) * 10).astype("f4")
    labels = rng.integers(0, 5, size=(30, 40)).astype(np.int32)

    def check(values, ndv, labels, n, stat, q=None, thr=None):
        got, counts, present = geometry.label_statistics(values, ndv, labels, n, stat, q, thr)
        want, _ = O.aggregate_labelled([(list(range(n)), labels)], n, values, ndv, thr, stat, q)
        assert got.shape == (values.shape[0], n) and counts.shape == got.shape and present.shape == (n,)
        merged = np.where(counts > 0, got, np.float32(np.nan)).astype("f4")
        assert_values_equal(merged, want, 0, "%s n=%d" % (stat, n))
        return got, counts, present

    for stat, q in (("sum", None), ("count", None), ("max", None), ("median", None), ("percentile", 30.0)):
        got, counts, present = check(values, np.float32(-1.0), labels, 0, stat, q)                       # no labels
        check(values, np.float32(-1.0), np.full((30, 40), np.iinfo(np.int32).max, np.int32), 4, stat, q)  # nothing labelled
        check(values[:, :0, :], np.float32(-1.0), labels[:0, :], 3, stat, q)                              # no cells
        check(values[:0], np.float32(-1.0), labels, 5, stat, q)                                           # no frames
        check(values, np.float32(-1.0), labels - 2, 3, stat, q)                                           # labels -2 .. 2: negative ones ignored
        check(values, np.float32(-1.0), labels, 5, stat, q, np.full(5, np.nan, "f4"))                     # no geometry has a threshold
        check(values, 1e300, labels, 5, stat, q)                                                          # a no-data value float32 cannot hold
        sparse = np.where(labels < 3, labels * 499_999, np.iinfo(np.int32).max).astype(np.int32)
        got, counts, present = check(values, np.float32(-1.0), sparse, 1_000_000, stat, q)                # a million labels, three occur
        assert present.sum() == 3 and (counts[0] > 0).sum() == 3
    small = rng.integers(-128, 128, size=(1, 30, 40)).astype("i1")
    for q in (10.0, 50.0, 99.0):
        check(small, np.int8(-128), labels, 5, "percentile", q)  # int8: data[hi] - data[lo] wraps modulo 256
    with pytest.raises(TypeError):
        geometry.label_statistics(values > 0, None, labels, 5, "percentile", 50.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8"])
def test_bilinear_on_zero_copy_windows(gpu, dt):
    """gm_resample_bilinear_window reads a strided window of a resident raster in place (aligned windows take the 16-byte
    staging path, the others the scalar one): same result as the compacted copy, up- and down-sampling, windows at the
    corners of the allocation."""
    from dask_geomodeling_amd.hip.array import DeviceArray
    from dask_geomodeling_amd.raster import resample
    rng = np.random.default_rng(2)
    H, W = 300, 1100
    full = (rng.random((1, H, W)) * 100).astype(dt)
    ndv = np.dtype(dt).type(np.finfo(dt).max)
    full[rng.random(full.shape) < 0.02] = ndv
    resident = DeviceArray.from_numpy(full)
    for row0, col0, h, w in ((0, 0, 128, 512), (40, 64, 200, 1024), (H - 77, W - 333, 77, 333), (3, 5, 50, 66)):
        win = resident.window(row0, col0, h, w)
        host = np.ascontiguousarray(full[:, row0:row0 + h, col0:col0 + w])
        for hd, wd in ((2 * h, 2 * w), (h * 3 // 2, w * 3 // 2), (h // 2, w // 3), (h, w)):
            got = resample.bilinear(win, hd, wd, ndv).to_numpy()
            want = resample.bilinear(host, hd, wd, ndv).to_numpy()
            assert np.array_equal(got, want, equal_nan=True), (dt, (row0, col0, h, w), (hd, wd))
            assert_values_equal(got, O.bilinear_resample(host, hd, wd, ndv), 1, "bilinear window %s" % dt)


# ---------------------------------------------------------------------------------------------------------
# round 3: the last refusals, closed -- each against vectors produced by the reference itself
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", load_family("round3"), ids=lambda c: c["id"] + "-" + c["op"])
def test_round3_golden(gpu, case):
    """Dilate with more than 16 values (value groups chained over the original tile, spatial.py:147-154), HillShade and
    Smooth on bool / uint64 rasters (spatial.py:276-308, 354-417), reduce_rasters into a 1-byte dtype from wider
    rasters (float16 work dtype: the stack assignment rounds, reduction.py:99-107): bit-exact."""
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    op = case["op"]
    if op.startswith("reduce:"):
        out = reduce_rasters([dict(d) for d in case["stack"]], op.split(":")[1], **case["kwargs"])
        out, want = dict(out, values=np.asarray(out["values"])), case["expect"]
        if op.endswith("median") or op.split(":")[1].startswith("p"):
            from fuzz_util import unsigned_zero
            out, want = unsigned_zero(out), unsigned_zero(want)
        assert_result_equal(out, want, 0, "%s %s" % (case["id"], op))
        return
    fn = {"dilate": raster.Dilate.process, "hillshade": raster.HillShade.process, "smooth": raster.Smooth.process}[op]
    assert_result_equal(fn(*case["args"]), case["expect"], 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8", "i2", "u1"])
def test_median_and_percentiles_over_bins_longer_than_the_lds_sort(gpu, dt):
    """Medians / percentiles over bins of 599, 600, 601 and 1500 frames (temporal.py:724-768): up to 599 float32
    frames a cell's values are sorted in LDS, beyond that they are selected in global memory (radix select of the middle
    pair / the percentile's neighbours) -- 600 is also where np.nanmedian itself changes its route.  Against the oracle
    (NumPy's own nan-functions), bit for bit up to the sign of a zero median."""
    from dask_geomodeling_amd.raster import temporal
    from fuzz_util import unsigned_zero
    rng = np.random.default_rng(77)
    frames, h, w = 3800, 6, 40
    ndv = {"f4": np.float32(-9999.0), "f8": -9999.0, "i2": np.int16(-1), "u1": np.uint8(255)}[dt]
    if np.dtype(dt).kind == "f":
        v = (rng.standard_normal((frames, h, w)) * 50).astype(dt)
        v[rng.random(v.shape) < 0.02] = np.nan
        v[5:900, 0, 0] = 7.25   # long runs of equal values: the upper rank is the same value again
        v[:, 0, 1] = ndv        # a cell without data
    else:
        v = rng.integers(0, 100, size=(frames, h, w)).astype(dt)
    v[rng.random(v.shape) < 0.1] = ndv
    runs = [(0, 599), (599, 600), (1199, 601), (1800, 1500), (3300, 1), (3301, 499)]
    for stat in ("median", "p25", "p50", "p99.5", "p0", "p100"):
        out_dt = temporal.dtype_for_statistic(dt, stat)
        got = np.asarray(temporal.aggregate_frames(v, ndv, runs, stat, out_dt))
        with np.errstate(all="ignore"):
            want = O.temporal_aggregate_values(v, ndv, runs, stat, out_dt)
        assert_values_equal(unsigned_zero({"values": got})["values"], unsigned_zero({"values": want})["values"], 0,
                            "%s of %s frames" % (stat, dt))


@pytest.mark.gpu
def test_reduce_rasters_median_over_650_rasters(gpu):
    """reduce_rasters median / percentile over more rasters than the LDS sort takes (reduction.py:92-117)."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    from fuzz_util import unsigned_zero
    rng = np.random.default_rng(78)
    ndv = np.float32(-1.0)
    stack = []
    for _ in range(650):
        v = (rng.random((2, 5, 12)) * 80).astype("f4")
        v[rng.random(v.shape) < 0.2] = ndv
        v[0, 0, 0] = ndv
        stack.append({"values": v, "no_data_value": ndv})
    for stat, kwargs in (("median", {}), ("p90", {}), ("median", {"dtype": "i4", "no_data_value": 2147483647})):
        out = reduce_rasters([dict(d) for d in stack], stat, **kwargs)
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], stat, **kwargs)
        assert_result_equal(unsigned_zero(dict(out, values=np.asarray(out["values"]))), unsigned_zero(want), 0, stat)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [9, 40, 90, 650])
def test_reduce_rasters_median_keeps_non_finite_results(gpu, n):
    """reduction.py:113-115 protects only the cells without a data raster: a median that is +-inf, the mean of a middle
    pair that overflows and the NaN of inf - inf stay what they are -- on every route (register network, LDS sort,
    selection in global memory beyond 64 rasters)."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    rng = np.random.default_rng(100 + n)
    ndv = np.float32(-1.0)
    big = np.float32(3.0e38)
    stack = []
    for i in range(n):
        v = (rng.random((1, 6, 16)) * 80).astype("f4")
        v[rng.random(v.shape) < 0.1] = ndv
        v[0, 0, 0] = ndv                                   # no data anywhere: the fill value
        v[0, 0, 1] = np.inf                                # median +inf
        v[0, 0, 2] = -np.inf                               # median -inf
        v[0, 0, 3] = np.inf if i % 2 else -np.inf          # even n: (-inf + inf) / 2 = NaN
        v[0, 0, 4] = big                                   # (big + big) overflows before the halving
        v[0, 0, 5] = np.inf if i < n // 2 else np.float32(i)  # percentile interpolation with an infinite neighbour
        stack.append({"values": v, "no_data_value": ndv})
    for stat in ("median", "p50", "p30", "p99"):
        out = reduce_rasters([dict(d) for d in stack], stat)
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], stat)
        got = np.asarray(out["values"])
        assert got[0, 0, 0] == ndv and want["values"][0, 0, 0] == ndv
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want["values"]), err_msg=stat)
        assert_values_equal(np.nan_to_num(got, nan=0.0) + 0.0, np.nan_to_num(want["values"], nan=0.0) + 0.0, 0, "%s of %d" % (stat, n))
    assert not np.isfinite(np.asarray(reduce_rasters([dict(d) for d in stack], "median")["values"])[0, 0, 1:3]).any()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f4", "f8", "u1", "i2", "i4", "i8"])
@pytest.mark.parametrize("n", [8, 9, 40])
def test_reduce_rasters_max_of_long_stacks(gpu, n, dtype):
    """reduce_rasters(..., "max") of more rasters than the fused fmax program takes (round 5: beyond 8 the stack kernel, like
    every other statistic -- the program of 300 rasters took 0.8 s) against the oracle: no data in some rasters / in all,
    NaN and infinities, zeros of both signs (the later raster wins a tie), every dtype, also into a narrower dtype."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    rng = np.random.default_rng(100 * n + len(dtype))
    dt = np.dtype(dtype)
    stack = []
    for i in range(n):
        if dt.kind == "f":
            v = (rng.random((1, 9, 13)) * 200 - 100).astype(dt)
            ndv = dt.type(-9999.0)
            v[0, 1, :4] = [np.nan, np.inf, -np.inf, 0.0]
            v[0, 2, 0] = -0.0 if i % 2 else 0.0
        else:
            info = np.iinfo(dt)
            v = rng.integers(max(info.min, -1000), min(info.max, 1000), size=(1, 9, 13)).astype(dt)
            ndv = dt.type(info.max)
        v[rng.random(v.shape) < 0.3] = ndv
        v[0, 0, 0] = ndv  # no data anywhere
        stack.append({"values": v, "no_data_value": ndv})
    # (into a 1-byte raster only from the dtypes whose float16 stack the 8-raster program already answers like the oracle: the
    #  32- and 64-bit integers' no-data values overflow float16 there, which is not what this test is about)
    # (int64 rasters into float32: a refusal of the device -- no exact register class --)
    for out_dtype in ((None,) if dtype == "i8" else (None, "f4" if dt.kind != "f" else "f8")) + (("u1",) if dtype not in ("i4", "i8") else ()):
        kwargs = {} if out_dtype is None else {"dtype": out_dtype}
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], "max", **kwargs)
        got = reduce_rasters([dict(d) for d in stack], "max", **kwargs)
        from fuzz_util import unsigned_zero
        assert_result_equal(unsigned_zero(dict(got, values=np.asarray(got["values"]))), unsigned_zero(want), 0,
                            "max of %d %s rasters as %s" % (n, dtype, out_dtype))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 4097, 8193, 9001, 17000])
@pytest.mark.parametrize("dtype", ["f4", "f8"])
def test_reduce_rasters_sums_over_deep_stacks(gpu, n, dtype):
    """sum / mean / var / std over more rasters than the unrolled pairwise recursion covers (128 * 2^5): NumPy's
    pairwise order walked with an explicit stack (csrc/gm_reduce.hip pw_sum, DEPTH < 0), bit for bit."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    rng = np.random.default_rng(n)
    ndv = np.dtype(dtype).type(-5.0)
    base = (rng.random((n, 1, 3, 7)) * 1000 - 300).astype(dtype)
    base[rng.random(base.shape) < 0.05] = ndv
    base[:, 0, 0, 0] = ndv
    stack = [{"values": base[i], "no_data_value": ndv} for i in range(n)]
    for stat in ("sum", "mean", "var", "std"):
        out = reduce_rasters([dict(d) for d in stack], stat)
        with np.errstate(all="ignore"):
            want = O.reduce_rasters([dict(d) for d in stack], stat)
        assert_values_equal(np.asarray(out["values"]), want["values"], 0, "%s over %d %s rasters" % (stat, n, dtype))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f4", "u1", "i2", "i4", "f8", "i8"])
def test_programs_over_windows_at_any_column(gpu, dtype):
    """Element-wise programs over zero-copy windows of resident rasters that start at ANY column and have ANY width --
    what a stencil Block asks of its store (4096 + 2 columns from an odd column on): 16-byte groups loaded from
    element-aligned addresses, widths that are no multiple of 4 run over the next multiple (the result comes back as
    a window of a padded tile), windows at the right edge of their raster, where there is nothing to pad with, take
    the one-element path.  All against the oracle on the same slices."""
    from dask_geomodeling_amd.hip import program as P
    from dask_geomodeling_amd.hip.array import DeviceArray
    rng = np.random.default_rng(abs(hash(dtype)) % 2**31)
    dt = np.dtype(dtype)
    H, W = 70, 520
    if dt.kind == "f":
        a = (rng.random((1, H, W)) * 200 - 100).astype(dt)
        b = (rng.random((1, H, W)) * 200 - 100).astype(dt)
        ndv = dt.type(np.finfo(dt).max)
    else:
        a = rng.integers(0, 100, size=(1, H, W)).astype(dt)
        b = rng.integers(0, 100, size=(1, H, W)).astype(dt)
        ndv = dt.type(np.iinfo(dt).max)
    a[rng.random(a.shape) < 0.05] = ndv
    da, db = DeviceArray.from_numpy(a), DeviceArray.from_numpy(b)
    out_dt = np.result_type(dt, np.int32) if dt.kind in "iu" else np.result_type(dt, np.float32)
    kw = {"dtype": out_dt.name, "fillvalue": O.get_dtype_max(out_dt)}
    cases = [(0, 0, 16, 64), (3, 1, 16, 64), (5, 2, 33, 66), (7, 3, 20, 67), (1, 5, 64, 129), (2, 7, 9, 509),
             (0, 517, 70, 3), (4, 453, 30, 67), (6, 9, 1, 5), (0, 1, 70, 518), (9, 258, 17, 262)]
    for row0, col0, h, w in cases:
        wa, wb = da.window(row0, col0, h, w), db.window(row0, col0, h, w)
        sa, sb = a[:, row0:row0 + h, col0:col0 + w], b[:, row0:row0 + h, col0:col0 + w]
        A = {"values": wa, "no_data_value": ndv}
        B = {"values": wb, "no_data_value": ndv}
        got = devops.math_process("add", kw, A, B)
        want = O.math_process("add", kw, {"values": sa, "no_data_value": ndv}, {"values": sb, "no_data_value": ndv})
        what = "%s window (%d, %d, %d, %d)" % (dtype, row0, col0, h, w)
        assert tuple(got["values"].shape) == (1, h, w)
        assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, what)
        # the (possibly pitched) result feeds the next program as it is
        again = devops.math_process("multiply", kw, got, 3)
        want2 = O.math_process("multiply", kw, want, 3)
        assert_result_equal(dict(again, values=np.asarray(again["values"])), want2, 0, what + " * 3")
        masked = devops.mask_process(A, 7)
        assert_result_equal(dict(masked, values=np.asarray(masked["values"])),
                            O.mask_process({"values": sa, "no_data_value": ndv}, 7), 0, what + " mask")


# ---------------------------------------------------------------------------------
# round 4: Snap / Shift / TemporalSum / Resample (frame gather + frame sum on the device), Place
# ---------------------------------------------------------------------------------
def _round4_vals_cases():
    return [c for c in load_family("round4") if c["kind"] != "place" and c["request"]["mode"] == "vals"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _round4_vals_cases(), ids=lambda c: c["id"] + "-" + c["kind"])
def test_round4_temporal_golden(gpu, case):
    """Whole get_data() of Snap / Shift / TemporalSum / Resample views (temporal.py:42-255, 1008-1214) with the frames
    re-arranged (gm_frames_take) and added up (gm_frames_sum) on the device, bit for bit what the reference returns."""
    from test_blocks_cpu import round4_view
    out = round4_view(case).get_data(**case["request"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in load_family("round4") if c["kind"] == "place"],
                         ids=lambda c: c["id"] + "-" + c["process_args"][0]["mode"] + "-" + str(c["process_args"][0].get("statistic")))
def test_place_process_golden(gpu, case):
    """Place.process (spatial.py:620-731) on the device -- shifted copies, the data test that decides which copies
    take part, reduce_rasters -- against what the reference's own process() returned for the same arguments."""
    from dask_geomodeling_amd import raster
    out = raster.Place.process(*case["process_args"])
    want = case["expect"]
    if want is None or "values" not in want:
        assert out == want, case["id"]
    else:
        assert_result_equal(dict(out, values=np.asarray(out["values"])), want, 0, case["id"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f4", "f8"])
def test_place_leaves_out_copies_whose_cells_are_only_close_to_no_data(gpu, dtype):
    """The data test of Place.process is utils.get_index -- np.isclose for float rasters -- so a copy whose only cells
    inside the window are merely CLOSE to the no-data value does not take part: the sum over nothing is the no-data
    raster (not zeros), and argmax counts the copies that remain."""
    from dask_geomodeling_amd import raster
    dt = np.dtype(dtype)
    ndv = dt.type(-9999.0)
    near = np.nextafter(ndv, dt.type(0))  # isclose(near, ndv), near != ndv
    assert near != ndv and np.isclose(near, ndv)
    source = np.full((1, 4, 4), ndv, dt)
    source[0, 1, 2] = near
    source[0, 3, 3] = 5.0
    kwargs = {"mode": "warp", "anchor": (2.0, 2.0), "src_bbox": (0.0, 0.0, 4.0, 4.0), "dst_bbox": (10.0, 10.0, 13.0, 13.0),
              "cellsize": (1.0, 1.0)}
    # first copy: only the close-to-no-data cell lands inside (the 5.0 falls outside); second copy: the 5.0 lands inside
    for coordinates, statistic in (([(12.0, 11.0)], "sum"), ([(12.0, 11.0), (11.0, 12.0)], "argmax"), ([(12.0, 11.0), (11.0, 12.0)], "count")):
        kw = dict(kwargs, coordinates=coordinates, statistic=statistic)
        data = {"values": source, "no_data_value": ndv}
        want = O.place_process(kw, dict(data))
        got = raster.Place.process(kw, dict(data))
        assert_result_equal(dict(got, values=np.asarray(got["values"])), want, 0, statistic)
    only_close = O.place_process(dict(kwargs, coordinates=[(12.0, 11.0)], statistic="sum"), {"values": source, "no_data_value": ndv})
    assert (only_close["values"] == ndv).all()
    two = O.place_process(dict(kwargs, coordinates=[(12.0, 11.0), (11.0, 12.0)], statistic="argmax"), {"values": source, "no_data_value": ndv})
    assert two["values"][0, 2, 2] == 0  # (the copy that remains is the first of the stack)


@pytest.mark.gpu
def test_place_known_answers(gpu):
    """Place through get_data on the raster of the reference's own tests (10 x 10 cells of 10 m, value 7, anchored
    at its centre; tests/test_raster_spatial.py:119-257 of the reference): exact / shifted placement at three zoom
    levels, overlapping copies under every statistic, everything outside, point requests on the cell edges."""
    from dask_geomodeling_amd import raster
    source = raster.MemorySource(np.full((1, 10, 10), 7, dtype=np.uint8), 255, "EPSG:28992", 10, (135000, 456000))
    center = (135050, 455950)
    requests = []
    for box in ((0, 0, 100, 80), (0, 0, 50, 40), (0, 0, 200, 160)):
        requests.append(dict(mode="vals", bbox=box, projection="EPSG:28992", width=int(box[2] / 10), height=int(box[3] / 10)))

    def values(place, request):
        return np.asarray(place.get_data(**request)["values"])[:, ::-1, :]  # (south up: readable indices)

    for request in requests:
        v = values(raster.Place(source, "EPSG:28992", center, []), request)
        assert (v[:, :10, :10] == 255).all()
        v = values(raster.Place(source, "EPSG:28992", center, [(50, 50)]), request)
        assert (v[:, :10, :10] == 7).all() and v.shape == (1, request["height"], request["width"])
        v = values(raster.Place(source, "EPSG:28992", center, [(60, 50)]), request)  # one cell to the right
        assert (v[:, :10, 1:11] == 7).all() and (v[:, :, 0] == 255).all()
        v = values(raster.Place(source, "EPSG:28992", center, [(50, 60)]), request)  # one cell up
        assert (v[:, 1:11, :10] == 7).all() and (v[:, 0, :] == 255).all()
        # two copies of which only the left and the bottom ridge show: (nothing, first, second, both)
        table = {"first": (255, 7, 7, 7), "last": (255, 7, 7, 7), "count": (0, 1, 1, 2), "sum": (0, 7, 7, 14),
                 "mean": (255, 7, 7, 7), "min": (255, 7, 7, 7), "max": (255, 7, 7, 7), "argmin": (255, 0, 1, 0),
                 "argmax": (255, 0, 1, 0), "std": (255, 0, 0, 0), "var": (255, 0, 0, 0), "median": (255, 7, 7, 7),
                 "p99": (255, 7, 7, 7)}
        for statistic, (none, first, second, both) in table.items():
            v = values(raster.Place(source, "EPSG:28992", center, [(-40, 50), (50, -40)], statistic), request)
            assert (v[:, 1:, 1:] == none).all(), statistic
            assert (v[:, 1:10, 0] == first).all() and (v[:, 0, 1:10] == second).all(), statistic
            assert (v[:, 0, 0] == both).all(), statistic
        x1, y1, x2, y2 = request["bbox"]
        away = raster.Place(source, "EPSG:28992", center, [(x1 - 50, y1), (x1, y1 - 50), (x2 + 50, y2), (x2, y2 + 50)])
        assert (values(away, request) == 255).all()
    place = raster.Place(source, "EPSG:28992", center, [(150, 50)])
    assert place.get_data(mode="time") == source.get_data(mode="time")
    assert place.get_data(mode="meta") == source.get_data(mode="meta")
    # point requests: a cell owns its top-left corner and its top and left edges
    two = raster.Place(source, "EPSG:28992", anchor=center, coordinates=[(60, -40), (-40, 60)])
    for point, expected in (((5, 15), 7), ((15, 15), 255), ((5, 5), 255), ((15, 5), 7), ((10, 15), 255), ((5, 10), 255),
                            ((15, 10), 7), ((10, 5), 7), ((10, 10), 7), ((1000, 1000), 255)):
        got = np.asarray(two.get_data(mode="vals", bbox=point * 2, projection="EPSG:28992", width=1, height=1)["values"])
        assert got.shape == (1, 1, 1) and got.item() == expected, point
    empty = raster.MemorySource(np.full((0, 0, 0), 7, dtype=np.uint8), 255, "EPSG:28992", 20, (0, 0))
    nothing = raster.Place(empty, "EPSG:28992", center, [(50, 50)])
    assert nothing.geometry is None and nothing.extent is None and nothing.get_data(**requests[0]) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["?", "u1", "i1", "u2", "i2", "u4", "i4", "u8", "i8", "f4", "f8"])
def test_frame_kernels_against_numpy(gpu, dtype):
    """gm_frames_sum against np.sum(axis=0) (accumulator dtype, wrap-around of 64-bit sums, frame-by-frame float order;
    a raster of ONE cell is summed pairwise, in pieces of 8192) and gm_frames_take against fancy indexing: vector and
    scalar cell counts, more frames than the unroll."""
    from dask_geomodeling_amd.raster import temporal as T
    rng = np.random.default_rng(abs(hash(dtype)) % 2**32)
    for shape in ((1, 3, 5), (7, 16, 64), (19, 37, 53), (130, 1, 1), (20000, 1, 1), (9, 1, 4096), (3, 128, 256)):
        dt = np.dtype(dtype)
        if dt.kind == "f":
            v = (rng.random(shape) * 2000 - 700).astype(dt)
        elif dt.kind == "b":
            v = rng.random(shape) < 0.5
        else:
            info = np.iinfo(dt)
            v = rng.integers(info.min, info.max, size=shape, dtype=dt, endpoint=True)
        with np.errstate(all="ignore"):
            want = O.frames_sum(v)
        got = np.asarray(T.sum_frames(v))
        assert_values_equal(got, want, 0, "sum of %s %s" % (shape, dt))
        index = rng.integers(0, shape[0], size=min(2 * shape[0] + 1, 300))
        assert_values_equal(np.asarray(T.take_frames(v, index)), O.take_frames(v, index), 0, "take of %s %s" % (shape, dt))
    assert np.asarray(T.take_frames(np.zeros((3, 2, 2), dtype), [])).shape == (0, 2, 2)
    zeros = np.asarray(T.sum_frames(np.zeros((0, 2, 3), dtype)))
    assert zeros.shape == (1, 2, 3) and not zeros.any() and zeros.dtype == np.zeros((0, 2, 3), dtype).sum(axis=0).dtype


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", ["1", "0"], ids=["block-pipeline", "wave-per-label"])
def test_zonal_block_pipeline(gpu, blocks):
    """Sums of long labels, block-parallel (csrc/gm_zonal.hip: every 256-cell block as "sum += D[sum mod 2]" while the
    running sum stays inside its binade, literal addition otherwise) against the oracle's np.bincount order: 240 label
    sums on adversarial rasters (tests/zonal_blocks_driver.py) with the threshold lowered so that labels of 512 cells
    already take the pipeline -- and the same cases with the pipeline switched off."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_LONG="512", GM_ZONAL_BLOCKS=blocks, GM_ZONAL_SORTED="1")
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_blocks_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "BLOCKS ok 240" in res.stdout
    took = routes_of(res.stdout)["zonal"]   # the route this test names is the route that ran (gm_route_stats)
    assert took["ordered_sums"] > 0 and took["chunk_attempts"] == 0 and took["exact_all"] == 0, took
    assert (took["block_pipeline"] > 0) == (blocks == "1"), took


@pytest.mark.gpu
@pytest.mark.parametrize("part,select", [("1", "1"), ("0", "1"), ("1", "0")], ids=["partition+select", "radix-sort", "partition+lexsort"])
def test_zonal_label_partition(gpu, part, select):
    """The stable sort by label as one counting pass (csrc/gm_zonal.hip: zonal_part_*_kernel, up to 2047 labels) and as
    this library's own radix sort; medians / percentiles by radix select per label on the partitioned cells (zonal_select_*_kernel) and
    by the two sorts of lexsort -- against the oracle: 240 statistics over label rasters of every texture
    (tests/zonal_partition_driver.py), sums forced through the sort."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_PART=part, GM_ZONAL_SELECT=select, GM_ZONAL_SORTED="1")
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_partition_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "PARTITION ok 240" in res.stdout
    took = routes_of(res.stdout)["zonal"]
    if part == "1":
        assert took["sort_partition"] > 0, took
    else:
        assert took["sort_radix"] > 0 and took["sort_partition"] == 0 and took["select"] == 0, took
    assert (took["select"] > 0) == (part == "1" and select == "1"), took
    if part == "0" or select == "0":   # (with both on, only label counts beyond the partition's 2047 take the two sorts)
        assert took["lexsort"] > 0, took


@pytest.mark.gpu
@pytest.mark.parametrize("chunks,lazy", [("2", "1"), ("2", "0"), ("1", "1"), ("0", "1")],
                         ids=["chunks-always", "chunks-always-every-sum-in-order", "product", "chunks-off"])
def test_zonal_chunk_records(gpu, chunks, lazy):
    """Sums and means without a sort (csrc/gm_zonal.hip: zonal_chunk_kernel leaves a record per (stretch of the raster, label),
    zonal_chunk_combine_kernel walks a label's records in order -- one addition per chunk that cannot round, the chunk's cells
    again from the raster otherwise) against np.bincount's order: 80 statistics over label layouts of many chunks per label,
    every label ending in cells that cancel the sequential sum before them (tests/zonal_chunks_driver.py).  With every label
    kept on the route -- with and without the float32 certificate of round 6, which answers the labels whose tails do not cancel
    --, with the product's choice (labels that are replayed too often take the sorting route), and without the route."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_CHUNKS=chunks, GM_ZONAL_LAZY=lazy)
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_chunks_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "CHUNKS ok 80" in res.stdout
    took = routes_of(res.stdout)["zonal"]
    if chunks == "0":
        assert took["chunk_attempts"] == 0 and took["chunk_answered"] == 0, took
    else:
        assert took["chunk_answered"] > 0, took
        if chunks == "2":   # no label is handed on because it was replayed too often: what gives up gave up for its layout
            assert took["chunk_answered"] >= took["chunk_gave_up"], took


@pytest.mark.gpu
def test_zonal_exactness_certificates_through_the_chunks(gpu):
    """The adversarial rasters of the block pipeline's certificates (labels of 450 k - 1 M cells whose tails cancel the sum
    before them) answered by the chunk records alone (GM_ZONAL_CHUNKS=2: no label is handed to the sorting route), every sum
    in bincount's order (GM_ZONAL_LAZY=0: the cancelled sums fail the float32 certificate anyway, the others would not)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_CHUNKS="2", GM_ZONAL_LAZY="0")
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_certificate_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "CERTIFICATES ok 56" in res.stdout
    took = routes_of(res.stdout)["zonal"]
    assert took["chunk_answered"] == took["calls"] > 0 and took["ordered_sums"] == 0 and took["block_pipeline"] == 0, took


@pytest.mark.gpu
@pytest.mark.parametrize("certify", ["1", "0"])
def test_zonal_exactness_certificates(gpu, certify):
    """Labels of 7 - 15 whole groups of 256 blocks whose tails cancel the sequential float64 sum before them, so that the
    float32 result shows every bit of it (tests/zonal_certificate_driver.py): the block and group certificates of round 4
    ("no addition of this block / group rounds for the sum entering it": one addition instead of a walk) against np.bincount's
    order, on data made for them and data made to break them; and the same cases with the certificates switched off."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_CERTIFY=certify, GM_ZONAL_VERBOSE="1", GM_ZONAL_CHUNKS="0")
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_certificate_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "CERTIFICATES ok 56" in res.stdout
    took = routes_of(res.stdout)["zonal"]
    assert took["chunk_attempts"] == 0 and took["block_pipeline"] > 0, took
    groups = [int(line.split("certified exact, ")[1].split(" groups")[0]) for line in res.stderr.splitlines() if "groups of 256 certified" in line]
    assert groups and (sum(groups) > 100 if certify == "1" else sum(groups) == 0), groups[:20]


@pytest.mark.gpu
@pytest.mark.parametrize("lazy", ["1", "0"], ids=["product", "every-sum-in-order"])
def test_zonal_float32_certificate(gpu, lazy):
    """Round 6: a label whose float64 sum -- in any order -- lies far enough from every float32 rounding boundary is answered
    without bincount's order (csrc/gm_zonal.hip: the first pass of zonal_chunk_combine_kernel over the label's chunk records;
    the bound is derived there).  tests/zonal_lazy_driver.py: zero-mean noise under long labels (which the certificate must
    answer: no label of those rasters reaches the sorting route), and what must NOT be answered that way -- exact sums on
    float32 ties and a hair beside them, sums that cancel to zeros of either sign and to float32 subnormals, cells beyond
    float32's range, int64 cells, non-finite cells -- against np.bincount's order; and all of it with the certificate off."""
    import os
    import subprocess
    import sys
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_lazy_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=dict(os.environ, GM_ZONAL_LAZY=lazy), timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "LAZY ok 104" in res.stdout, res.stdout[-300:]
    took = routes_of(res.stdout)["zonal"]
    assert (took["lds_lazy"] > 0) == (lazy == "1") and (took["exact_lazy_labels"] > 0) == (lazy == "1"), took
    assert (took["parallel_lazy"] >= 6) == (lazy == "1"), took   # (the rasters of thousands of small polygons: one pass)
    noise = subprocess.run([sys.executable, driver, "noise"], capture_output=True, text=True, env=dict(os.environ, GM_ZONAL_LAZY=lazy), timeout=900)
    assert noise.returncode == 0 and "LAZY ok 20" in noise.stdout, (noise.stdout[-500:], noise.stderr[-2500:])
    took = routes_of(noise.stdout)["zonal"]
    if lazy == "1":   # noise under long labels used to send the frame to the sorting route (`hard`): nothing does now
        # (a sum may land within the bound of a float32 tie -- a few per cent of the labels of millions of cells: those frames go on)
        # (up to 85 labels the certificate first runs over per-wave tables, zonal_tables_lazy_kernel: "lds_lazy")
        assert took["chunk_answered"] + took["lds_lazy"] >= 22 and took["ordered_sums"] <= 2, took
    else:
        assert took["ordered_sums"] > 0, took


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", ["0", "1"], ids=["block-pipeline", "product"])
def test_zonal_labels_beyond_64_groups(gpu, chunks):
    """Labels of 4 - 12 M cells (tests/zonal_long_label_driver.py): the composing wave of the block pipeline holds the
    descriptions of 64 groups of 256 blocks at a time, so these labels make it move on to the next 64 -- twice and more -- on
    cells that only grow the sum, on zero-mean noise with and without very fine cells, and on integers; every label with a tail
    that cancels the sequential sum before it.  With the chunk records off (every label through the pipeline) and on."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GM_ZONAL_CHUNKS=chunks)
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_long_label_driver.py")
    res = subprocess.run([sys.executable, driver], capture_output=True, text=True, env=env, timeout=1800)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "LONG LABELS ok 16" in res.stdout
    took = routes_of(res.stdout)["zonal"]
    assert (took["chunk_attempts"] > 0) == (chunks == "1"), took
    if chunks == "0":
        assert took["block_pipeline"] > 0, took


@pytest.mark.gpu
def test_zonal_statistics_at_the_raster_limit(gpu):
    """3072 x 4096 cells (tests/zonal_full_size_driver.py; all eleven layouts, 174 statistics, when run by hand): 16 x 16 blocks
    (49 152 labels) and 48 x 40 blocks off the grid (6 592: the one-pass route of the small labels and its few labels added in
    order); vertical stripes 25 wide (the tables after the chunk records' first attempt) and one label -- sum, mean, count, min,
    max (and medians of the noise) of float32 noise, positive cells and int32 against the oracle.  Two driver processes side by
    side (most of their time is the oracle's, on the host: round 6); the other layouts at this size in the campaign logs."""
    import os
    import subprocess
    import sys
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zonal_full_size_driver.py")
    runs = [(layouts, cases, subprocess.Popen([sys.executable, driver, layouts], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
            for layouts, cases in (("0,2", 30), ("3,9", 32))]
    for layouts, cases, proc in runs:
        try:
            out, err = proc.communicate(timeout=1800)
        except subprocess.TimeoutExpired:
            for _, _, other in runs:
                other.kill()
            raise
        assert proc.returncode == 0, (layouts, out[-500:], err[-2500:])
        assert "FULL SIZE ok %d" % cases in out, (layouts, out[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("parts,checks", [(["stencils"], 120), (["resample"], 75), (["temporal,stacks", "quick"], 17)],
                         ids=["stencils", "resample", "temporal-stacks"])
def test_kernels_of_the_next_rows_at_the_sizes_they_are_timed_on(gpu, parts, checks):
    """tests/full_size_ops_driver.py: MovingMax (3, 7, 15, 17), HillShade, Smooth and Dilate on 8192^2 tiles and bilinear results of
    8192^2 cells at six zooms -- windows of the device result (corners, tile seams far from the origin, random places) against
    the oracle on host copies of those windows --, nearest warps of 8192^2 cells, temporal statistics of 24 frames and stack
    statistics of 8 rasters whole (a quarter of the timed sizes here; the timed sizes when the driver is run by hand)."""
    import os
    import subprocess
    import sys
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_size_ops_driver.py")
    res = subprocess.run([sys.executable, driver] + parts, capture_output=True, text=True, timeout=1800)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-2500:])
    assert "FULL SIZE OPS ok %d" % checks in res.stdout, res.stdout[-300:]


@pytest.mark.gpu
@pytest.mark.parametrize("part,checks", [("temporal", 40), ("stencils", 24), ("resample", 8), ("stacks", 15)])
def test_tiles_of_more_than_2_to_the_31_cells_through_the_next_rows(gpu, part, checks):
    """tests/beyond_2_31_driver.py: 40 frames of 8192^2 cells through the temporal statistics and Cumulative, 9 bands of 16384^2
    through MovingMax / HillShade / Smooth / Dilate, bilinear and nearest results of 3 bands of 32768^2, reduce_rasters over 40
    rasters of 8192^2 -- windows at the far end
    of the results against the oracle.  (Needs ~45 GB of host memory: skipped on smaller hosts.)"""
    import os
    import subprocess
    import sys
    import psutil
    if psutil.virtual_memory().available < 96 * 2 ** 30:
        pytest.skip("needs 96 GB of free host memory")
    if part == "stencils" and os.environ.get("GM_SUITE") != "full":
        # (a minute, nearly all of it the host making and rolling 9 bands of 16384^2: part of the campaign run -- GM_SUITE=full,
        #  profiles/r06_fuzz_campaign.log --, not of the driver's time-limited run; the stencils' 64-bit indexing is also covered
        #  at 8192^2 by test_kernels_of_the_next_rows_at_the_sizes_they_are_timed_on and through the other three parts' windows)
        pytest.skip("GM_SUITE=full runs the stencils over 2.4 G cells")
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "beyond_2_31_driver.py")
    res = subprocess.run([sys.executable, driver, part], capture_output=True, text=True, timeout=1800)
    assert res.returncode == 0, (res.stdout[-800:], res.stderr[-2500:])
    assert "BEYOND 2^31 ok %d" % checks in res.stdout, res.stdout[-300:]


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8"])
def test_signed_zeros_through_the_temporal_statistics(gpu, dt):
    """What the temporal fuzzing of round 4 found: NumPy's sums over the frame axis start from add's identity, so a cell whose
    every frame is -0.0 sums (and averages, and -- through np.mean of the middle pair -- has a median of) +0.0; minima and
    maxima hand a tie between -0.0 and +0.0 to the later frame.  Every pattern of zeros and no data over 1-3 frames, every
    statistic, against the oracle (= the NumPy functions of the reference's table); TemporalSum and reduce_rasters likewise."""
    from dask_geomodeling_amd.raster import temporal as T
    from dask_geomodeling_amd.raster import reduction
    nd = np.dtype(dt).type(np.finfo(dt).max)
    patterns = [[-0.0], [-0.0, -0.0], [-0.0, 0.0], [0.0, -0.0], [-0.0, nd], [nd, -0.0], [-0.0] * 3, [nd, nd], [-0.0, 1.0, -1.0],
                [-0.0] * 9, [0.0, -0.0, -0.0], [nd, -0.0, 0.0]]

    def same(got, want, what):
        got, want = np.asarray(got), np.asarray(want)
        assert got.dtype == want.dtype and got.shape == want.shape, what
        assert np.array_equal(got, want, equal_nan=True) and np.array_equal(np.signbit(got), np.signbit(want)), (what, got.ravel()[:3], want.ravel()[:3])

    for p in patterns:
        v = np.array(p, dtype=dt).reshape(len(p), 1, 1) * np.ones((1, 5, 7), dtype=dt)
        runs = [(0, len(p))]
        mixed = any(x == 0 and np.signbit(x) for x in p) and any(x == 0 and not np.signbit(x) for x in p)
        for stat in ("sum", "mean", "min", "max", "median", "std", "count", "p90"):
            if stat == "p90" and mixed:
                continue  # (a percentile picks from the SORTED frames: where +0.0 lands among -0.0s is the sort's business)
            dtype = np.dtype("i4") if stat == "count" else np.dtype(dt)
            same(T.aggregate_frames(v, nd, runs, stat, dtype), O.temporal_aggregate_values(v, nd, runs, stat, dtype), (p, stat))
        same(T.sum_frames(v), O.frames_sum(v), (p, "TemporalSum"))
        one = v[:, :1, :1]
        same(T.sum_frames(one), O.frames_sum(one), (p, "TemporalSum of one cell"))
        stack = [{"values": v[k:k + 1], "no_data_value": nd} for k in range(len(p))]
        for stat in ("sum", "mean", "median", "min", "max"):
            # (min / max: a tie between the zeros goes to the later raster, the rule of the temporal kernels -- np.nanmin /
            #  np.nanmax over axis 0 on this host, whose SIMD minimum hands back its second operand)
            got = reduction.reduce_rasters(stack, stat)
            want = O.reduce_rasters(stack, stat)
            same(got["values"], want["values"], (p, "reduce_rasters " + stat))


@pytest.mark.gpu
def test_frames_that_do_not_line_up(gpu, tiler_route):
    """More of what the temporal fuzzing found.  (1) A raster of one frame next to a series in a ufunc step is NumPy
    broadcasting -- also when the step is planned / fused (the remembered program used to be handed the one frame as if it were
    n).  (2) Where the reference indexes with a boolean array (Clip, FillNoData, Max) frames that differ are NumPy's IndexError,
    except for an EMPTY index, which selects nothing; an all-no-data store leaves Clip before it gets there.  (3) An empty stack
    (no frame in the window) passes through the stencils.  (4) uint64 series (TemporalSum of a uint8 raster) are counted."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import raster
    from oracle_walk import walk
    rng = np.random.default_rng(11)
    day = timedelta(days=1)

    def source(frames, dt="f4", ndv=-9999.0, first=datetime(2021, 3, 1)):
        v = (rng.random((frames, 24, 40)) * 40).astype(dt)
        v[rng.random(v.shape) < 0.1] = ndv
        return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, 24.0), time_first=first, time_delta=day if frames > 1 else None)

    box = dict(mode="vals", bbox=(0.0, 0.0, 40.0, 24.0), width=40, height=24, projection="EPSG:28992")
    everything = dict(box, start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))
    series, one = source(3), raster.TemporalSum(source(5))
    for view in (series / one, one / series, series * 0.5 / one, raster.Snap(series, series) - one):
        for plans in (True, False):
            with gmconfig.config.set({"geomodeling.plans": plans}):
                assert_result_equal(view.get_data(**everything), walk(view, everything), 0, "1 frame against 3, plans %s" % plans)
    for view in (raster.Clip(series, one > 5), raster.FillNoData(series, one), raster.Max(series, one), raster.Clip(series * 2, one > 5)):
        for plans in (True, False):
            with gmconfig.config.set({"geomodeling.plans": plans}):
                with pytest.raises(IndexError):
                    walk(view, everything)
                with pytest.raises(IndexError):
                    view.get_data(**everything)
    # an empty index: the second source has no frame between its days
    gap = dict(box, start=datetime(2021, 3, 1, 3), stop=datetime(2021, 3, 1, 9))
    late = source(3, first=datetime(2021, 3, 1, 6))
    assert walk(late, gap)["values"].shape[0] == 1 and walk(series, gap)["values"].shape[0] == 0
    for view in (raster.Clip(raster.TemporalSum(late), series > 5), raster.Mask(raster.Clip(raster.TemporalSum(late) * 2, series > 5), 1)):
        for plans in (True, False):
            with gmconfig.config.set({"geomodeling.plans": plans}):
                want = walk(view, gap)
                assert want["values"].shape[0] == 1
                assert_result_equal(view.get_data(**gap), want, 0, "empty mask, plans %s" % plans)
    # an empty stack through the stencils
    between = dict(box, start=datetime(2021, 3, 1, 15), stop=datetime(2021, 3, 1, 22))
    for view in (raster.Dilate(source(2, "i4", -9999), [0, 7]), raster.MovingMax(series, 3), raster.Smooth(series, 1.5), raster.HillShade(series),
                 raster.Smooth(series, 30.0)):
        want, got = walk(view, between), view.get_data(**between)
        assert want["values"].shape[0] == 0 and np.asarray(got["values"]).shape == want["values"].shape, type(view).__name__
    # uint64 series
    counts = raster.TemporalSum(source(4, "u1", 255))
    assert np.asarray(counts.get_data(**everything)["values"]).dtype == np.dtype("u8")  # (np.sum's accumulator)
    # (the Block still declares uint8 -- the reference's TemporalSum inherits its store's dtype --, so statistics whose result
    #  dtype follows the declared one are refused rather than cast; the counting ones do not depend on it)
    for view in (raster.Cumulative(counts, "count"), raster.TemporalAggregate(counts, "D", "count"), raster.TemporalAggregate(counts, None, "count")):
        assert_result_equal(view.get_data(**everything), walk(view, everything), 0, type(view).__name__ + " of uint64")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f4", "f8", "i2", "u1", "i4"])
def test_temporal_statistics_of_one_cell(gpu, dt):
    """The time series of a point (a 1 x 1 request): with no axis beside the frames NumPy's sums run their contiguous loop --
    pairwise over pieces of 8192 frames -- not frame by frame; sum / mean / std (and the rest, for completeness) of 1 to 9000
    frames in bins of several lengths, with no data and NaN frames, bit for bit against the oracle."""
    from dask_geomodeling_amd.raster import temporal as T
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 24, 129, 1000, 9000):
        if np.dtype(dt).kind == "f":
            v = (rng.random((n, 1, 1)) * 1000 - 300).astype(dt)
            nd = np.dtype(dt).type(np.finfo(dt).max)
            if n > 8:
                v[5] = np.nan
        else:
            v = rng.integers(0, 120, size=(n, 1, 1)).astype(dt)
            nd = np.dtype(dt).type(7)
        v[rng.random(n) < 0.1] = nd
        runs = [(0, n)] if n < 24 else [(0, 9), (9, n // 2 - 9), (n // 2, 0), (n // 2, n - n // 2)]
        for stat in ("sum", "mean", "std", "min", "max", "count", "median"):
            source = np.dtype(dt)
            narrow = source.itemsize <= 2 or source == np.dtype("f4")
            dtype = {"count": np.dtype("i4"), "min": source, "max": source,
                     "sum": source if source.kind == "f" else (np.dtype("i4") if source.itemsize < 4 else source)
                     }.get(stat, np.dtype("f4") if narrow else np.dtype("f8"))  # (utils.py:826-845)
            want = O.temporal_aggregate_values(v, nd, runs, stat, dtype)
            got = np.asarray(T.aggregate_frames(v, nd, runs, stat, dtype))
            assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (dt, n, stat, got.ravel(), want.ravel())


@pytest.mark.gpu
@pytest.mark.parametrize("dt,ndv", [("u4", np.uint32(4294967295)), ("u8", np.uint64(2 ** 64 - 1)), ("u2", np.uint16(65535)), ("i8", -9999),
                                    ("u4", np.uint32(7))])
def test_merges_of_wide_unsigned_rasters(gpu, dt, ndv, tiler_route):
    """FillNoData / Group / Clip / Max of uint32 and uint64 rasters -- values beyond 2^31 and 2^63, the dtype's maximum as
    no data: nothing is computed in a merge, so the values travel in the int64 class (uint64 as its bit pattern; the fuzzing
    of round 4 found these refused), and a float64 result above 2^63 is stored as uint64, not through int64 (Max of uint64
    rasters works in float64: reduction.py:99-107)."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import raster
    from oracle_walk import walk
    rng = np.random.default_rng(2)
    box = dict(mode="vals", bbox=(0.0, 0.0, 40.0, 24.0), width=40, height=24, projection="EPSG:28992",
               start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))

    def source(first_hour=0):
        v = rng.integers(0, 2 ** 31, size=(3, 24, 40)).astype(dt) * (2 if dt in ("u4", "u8") else 1)
        if dt == "u8":
            v = v * np.uint64(2 ** 32 + 3)
        v[rng.random(v.shape) < 0.3] = ndv
        return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, 24.0), time_first=datetime(2021, 3, 1, first_hour),
                                   time_delta=timedelta(days=1))

    a, b = source(), source()
    for view in (raster.FillNoData(a, b), raster.Group(a, b), raster.Group(a, source(6)), raster.Clip(a, b), raster.Max(a, b)):
        assert_result_equal(view.get_data(**box), walk(view, box), 0, "%s of %s" % (type(view).__name__, dt))


@pytest.mark.gpu
def test_point_requests_on_resident_series(gpu, tiler_route):
    """The time series of a location from a source that lives in HBM: the cell of every requested frame comes back in one
    strided copy (raster/sources.py) -- every time window, points on cell edges and outside the raster, against the same
    source held on the host and against the walking oracle; and through a view."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip.array import DeviceArray
    from oracle_walk import walk
    rng = np.random.default_rng(4)
    v = (rng.random((9, 31, 45)) * 100).astype("f4")
    v[rng.random(v.shape) < 0.2] = -9999.0
    kw = dict(time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))
    host = raster.MemorySource(v, -9999.0, "EPSG:28992", 2.0, (10.0, 100.0), **kw)
    resident = raster.MemorySource(DeviceArray.from_numpy(v), -9999.0, "EPSG:28992", 2.0, (10.0, 100.0), **kw)
    windows = [dict(), dict(start=datetime(2021, 3, 2)), dict(start=datetime(2021, 3, 1, 5), stop=datetime(2021, 3, 2, 13)),
               dict(start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1)), dict(start=datetime(2021, 3, 1, 1), stop=datetime(2021, 3, 1, 2))]
    for x, y in ((11.0, 99.0), (10.0, 100.0), (12.0, 98.0), (99.9, 38.1), (100.0, 38.0), (5.0, 50.0), (50.0, 120.0), (57.3, 61.9)):
        for when in windows:
            request = dict(mode="vals", bbox=(x, y, x, y), width=1, height=1, projection="EPSG:28992", **when)
            want = walk(host, request)
            for source in (host, resident):
                assert_result_equal(source.get_data(**request), want, 0, "%s at (%s, %s) %s" % (type(source.data).__name__, x, y, when))
            view = raster.TemporalAggregate(resident * 2, "D", "mean")
            assert_result_equal(view.get_data(**request), walk(raster.TemporalAggregate(host * 2, "D", "mean"), request), 0, "view")


@pytest.mark.gpu
@pytest.mark.parametrize("plans", [True, False])
def test_sums_of_integer_series_under_blocks_that_cast_them(gpu, plans, tiler_route):
    """TemporalSum of an integer series is an int64 / uint64 tile under a Block that declares its store's dtype
    (temporal.py:548-602); the Blocks above it ask NumPy for their declared float32 / int32 results, and NumPy casts the
    tile first (`np.divide(int64 tile, 100, dtype=float32)`).  No register class holds an int64 tile next to float32
    arithmetic, so the device answers such views only when the sums lie within +-2^24 (a pass that looks, then the tile as
    int32: program.run_expr) -- the usual case, rain in 0.01 mm summed over days -- and still refuses beyond.  Against the
    walking oracle."""
    from datetime import datetime, timedelta
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd import raster
    from oracle_walk import walk
    rng = np.random.default_rng(9)

    def series(dt, ndv, high, frames=6):
        v = rng.integers(0, high, size=(frames, 40, 56)).astype(dt)
        v[rng.random(v.shape) < 0.1] = ndv
        return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, 40.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=6))

    rain, cover = series("i2", -1, 3000), series("u1", 255, 200)
    high = rng.integers(30000, 32767, size=(600, 8, 8)).astype("i2")  # (sums of 18 million: beyond 2^24)
    wide = raster.MemorySource(high, -1, "EPSG:28992", 1.0, (0.0, 8.0), time_first=datetime(2021, 3, 1), time_delta=timedelta(hours=1))
    f = (rng.random((1, 40, 56)) * 10).astype("f4")
    F = raster.MemorySource(f, -3.5, "EPSG:28992", 1.0, (0.0, 40.0))
    request = dict(mode="vals", bbox=(0.0, 0.0, 56.0, 40.0), width=56, height=40, projection="EPSG:28992",
                   start=datetime(2021, 1, 1), stop=datetime(2021, 6, 1))
    total, shade = raster.TemporalSum(rain), raster.TemporalSum(cover)
    views = [total / 100, shade / 7, total * 0.5, shade + 1.5, total - shade, raster.FillNoData(total, shade),
             raster.Max(total, shade), raster.Group(total, shade), total > 5000, raster.Clip(total, shade > 300),
             # (the temporal kernels: the reference works on values.astype(result_type(float32, declared dtype)) -- int32, float32,
             #  int16 / uint8 frames of the same numbers, whichever gives the declared dtype: temporal._sums_as_int32)
             raster.TemporalAggregate(total, None, "mean"), raster.TemporalAggregate(shade, None, "min"),
             raster.TemporalAggregate(total, None, "std"), raster.TemporalAggregate(shade, None, "median"),
             raster.Cumulative(total), raster.TemporalAggregate(total, None, "sum")]
    answered = 0
    with gmconfig.config.set({"geomodeling.plans": plans}):
        for view in views:
            try:
                want = walk(view, request)
            except Exception:  # noqa: BLE001 -- (Group: the one frame of a sum under a period of six; the reference raises)
                with pytest.raises(Exception):
                    view.get_data(**request)
                continue
            try:
                got = view.get_data(**request)
            except NotImplementedError as refusal:
                print("refused:", type(view).__name__, refusal)
                continue  # (none may be WRONG)
            answered += 1
            assert_result_equal(got, want, 0, type(view).__name__)
        assert answered >= 14, answered
        # sums beyond 2^24: refused, loudly
        with pytest.raises(NotImplementedError):
            (raster.TemporalSum(wide) / 3).get_data(**dict(request, bbox=(0.0, 0.0, 8.0, 8.0), width=8, height=8))
    del F
