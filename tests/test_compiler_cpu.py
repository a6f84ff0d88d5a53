"""Host logic on the CPU: expression -> gm_program compiler + per-op process bodies,
executed by the oracle's NumPy interpreter of the program (``cpu_vm`` fixture) and
compared with the reference-generated golden vectors."""
import numpy as np
import pytest

from helpers import assert_result_equal, load_family
from dask_geomodeling_amd.raster import devops

# NumPy's float32 exp/log SIMD loops are themselves 2 ulp off the correctly rounded value on these vectors
# (measured: tests/test_gpu_parity.py); the program interpreter evaluates in float64 and rounds once like the
# device does, hence 2 ulp from NumPy's (Power stays within 1 ulp)
ULP_OPS = {"power": 1, "exp": 2, "log": 2, "log10": 2}


@pytest.mark.parametrize("case", load_family("math"), ids=lambda c: c["id"] + "-" + c["op"])
def test_math(cpu_vm, case):
    if "raises" in case:
        pytest.skip("reference raises %s" % case["raises"])
    out = devops.math_process(case["op"], case["kwargs"], *case["args"])
    assert_result_equal(out, case["expect"], ULP_OPS.get(case["op"], 0), case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("n1") if c["op"] in devops.P.OPCODES],
                         ids=lambda c: c["id"] + "-" + c["op"])
def test_n1_wrapper_ops(cpu_vm, case):
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
    assert_result_equal(out, case["expect"], ULP_OPS.get(op, 0), case["id"])


@pytest.mark.parametrize("case", load_family("clip"), ids=lambda c: c["id"])
def test_clip(cpu_vm, case, monkeypatch):
    monkeypatch.setattr(devops, "all_equal",
                        lambda values, ndv: bool(np.all(np.asarray(values) == ndv)))
    out = devops.clip_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", load_family("mask"), ids=lambda c: c["id"])
def test_mask(cpu_vm, case):
    if "raises" in case:
        with pytest.raises(Exception) as ei:
            devops.mask_process(*case["args"])
        assert type(ei.value).__name__ == case["raises"]
        return
    out = devops.mask_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


# ---------------------------------------------------------------------------------
# Classify / Reclassify as program ops (GM_OP_DIGITIZE / GM_OP_LOOKUP): every golden case that
# fits a program must come out of the compiler + interpreter exactly as the reference computed it
# ---------------------------------------------------------------------------------
def _program_result(node):
    from dask_geomodeling_amd.hip import program as P
    values, ndv = P.evaluate(node)
    return {"values": values, "no_data_value": ndv}


@pytest.mark.parametrize("case", load_family("classify"), ids=lambda c: c["id"])
def test_classify_as_program_op(cpu_vm, case):
    from dask_geomodeling_amd.hip import program as P
    from dask_geomodeling_amd.raster import tables
    data, bins, right = case["args"]
    if data is None or "values" not in data:
        pytest.skip("control-flow case")
    leaf = P.Leaf(None, data["values"].dtype, data["no_data_value"], values=data["values"])
    try:
        node = tables.classify_node(leaf, bins, right)
        out = _program_result(node)
    except (P.FusionLimit, NotImplementedError):
        pytest.skip("runs as its own kernel")
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("reclassify") if c["op"] == "reclassify"],
                         ids=lambda c: c["id"])
def test_reclassify_as_program_op(cpu_vm, case):
    from dask_geomodeling_amd.hip import program as P
    from dask_geomodeling_amd.raster import tables
    data, kwargs = case["args"]
    if data is None or "values" not in data:
        pytest.skip("control-flow case")
    leaf = P.Leaf(None, data["values"].dtype, data["no_data_value"], values=data["values"])
    try:
        node = tables.reclassify_node(leaf, kwargs)
        out = _program_result(node)
    except (P.FusionLimit, NotImplementedError):
        pytest.skip("runs as its own kernel")
    assert_result_equal(out, case["expect"], 0, case["id"])


def _table_chain_inputs():
    rng = np.random.default_rng(3)
    fill = np.float32(np.finfo("f4").max)
    a = (rng.random((1, 40, 50)) * 200 - 100).astype("f4")
    b = (rng.random((1, 40, 50)) * 50).astype("f4")
    a[0, ::7, ::3] = fill
    a[0, 3, 3] = np.nan
    bins = [-50.0, -10.0, 0.0, 10.0, 50.0]
    table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False,
             "data": [[0, 10], [1, 11], [3, 13], [5, 15]]}
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    return a, b, fill, bins, table, kw


@pytest.mark.parametrize("select", [False, True])
def test_mask_folds_into_the_lookup_and_the_chain_is_one_program(cpu_vm, select):
    """float32 arithmetic -> Classify -> Reclassify -> Mask: the Mask is folded into the table on the
    host, the folded table holds small integers, so the whole chain is ONE program of the F32 class
    (4 f32 leaves in, uint8 out) and still equals the reference op by op."""
    from dask_geomodeling_amd.hip import program as P
    from dask_geomodeling_amd.raster import tables
    from oracle import ref_numpy as O
    a, b, fill, bins, table, kw = _table_chain_inputs()
    table = dict(table, select=select)
    A, B = {"values": a, "no_data_value": fill}, {"values": b, "no_data_value": fill}
    want = O.mask_process(O.reclassify_process(O.classify_process(O.math_process("add", kw, A, B), bins, False), table), 1)
    la, lb = P.Leaf(None, "f4", fill, values=a), P.Leaf(None, "f4", fill, values=b)
    x = devops.math_node("add", kw, [la, lb])
    x = tables.classify_node(x, bins, False)
    x = tables.reclassify_node(x, table)
    x = devops.mask_node(x, 1)
    stages, shape = P.plan_staged(x)
    assert len(stages) == 1 and stages[0][0].reg_class == P.CLASS_F32
    assert stages[0][0].out_dtype == np.dtype("u1")
    out = P.run_staged(stages, shape)
    assert_result_equal({"values": out, "no_data_value": stages[-1][0].out_ndv}, want, 0, "folded chain")


def test_table_ops_cut_at_the_register_class_boundary(cpu_vm):
    """float32 arithmetic + Classify | Reclassify + integer arithmetic: no single register class
    serves the chain (int64 targets); the planner cuts it once (F32 program -> uint8 tile -> I64
    program) instead of op by op."""
    from dask_geomodeling_amd.hip import program as P
    from dask_geomodeling_amd.raster import tables
    from oracle import ref_numpy as O
    a, b, fill, bins, table, kw = _table_chain_inputs()
    A, B = {"values": a, "no_data_value": fill}, {"values": b, "no_data_value": fill}
    kwi = {"dtype": "int64", "fillvalue": int(np.iinfo("i8").max)}
    want = O.math_process("add", kwi, O.reclassify_process(O.classify_process(O.math_process("add", kw, A, B), bins, False), table), 7)
    la, lb = P.Leaf(None, "f4", fill, values=a), P.Leaf(None, "f4", fill, values=b)
    x = devops.math_node("add", kw, [la, lb])
    x = tables.classify_node(x, bins, False)
    x = tables.reclassify_node(x, table)
    x = devops.math_node("add", kwi, [x, P.Lit(7)])
    stages, shape = P.plan_staged(x)
    assert len(stages) == 2
    assert [c.reg_class for c, _ in stages] == [P.CLASS_F32, P.CLASS_I64]
    assert stages[0][0].out_dtype == np.dtype("u1") and stages[1][0].out_dtype == np.dtype("i8")
    out = P.run_staged(stages, shape)
    assert_result_equal({"values": out, "no_data_value": stages[-1][0].out_ndv}, want, 0, "two-stage chain")


@pytest.mark.parametrize("case", load_family("reduce"), ids=lambda c: c["id"])
def test_reduce_max(cpu_vm, case):
    """Max.process = reduce_rasters(stack, "max") as one program (NaN-coded no data + np.fmax)."""
    args = case["args"]  # (incl. 64-bit integer rasters: through float64 like the reference)
    out = devops.reduce_max_process(*args)
    assert_result_equal(out, case["expect"], 0, case["id"])


def test_max_block_fuses_with_its_neighbours(cpu_vm):
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.fuse import fuse_graph, FusedTask
    from blocks_util import ArrayLeaf
    from oracle import ref_numpy as O
    rng = np.random.default_rng(8)
    fill = np.float32(np.finfo("f4").max)
    leaves, tiles = [], []
    for k in range(3):
        v = (rng.random((1, 32, 48)) * 100).astype("f4")
        v[0, k::5, ::3] = fill
        ArrayLeaf.ARRAYS["mx/%d" % k] = {"values": v, "no_data_value": fill}
        leaves.append(ArrayLeaf("mx/%d" % k, "<f4", fill))
        tiles.append({"values": v, "no_data_value": fill})
    view = raster.Max(leaves[0] + 1.5, leaves[1], leaves[2]) * 2.0
    assert view.dtype == np.dtype("f4")
    req = dict(mode="vals", bbox=(0.0, 0.0, 48.0, 32.0), width=48, height=32, projection="EPSG:28992")
    graph, name = view.get_compute_graph(**req)
    fused = fuse_graph(graph, name)
    assert sum(isinstance(v[0], FusedTask) for v in fused.values()) == 1 and len(fused) == 4  # 3 leaves + 1 program
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    want = O.math_process("multiply", kw, O.reduce_max_process(kw, O.math_process("add", kw, tiles[0], 1.5), tiles[1], tiles[2]), 2.0)
    assert_result_equal(view.get_data(**req), want, 0, "Max inside a fused chain")


@pytest.mark.parametrize("case", [c for c in load_family("reducestats") if c["op"] in ("first", "last", "count", "max")],
                         ids=lambda c: c["id"] + "-" + c["op"])
def test_reduce_rasters_program_statistics(cpu_vm, case):
    """reduce_rasters: the statistics that run as one element-wise program (first / last / count / max)
    against the reference's vectors (the nan-function statistics need the device: tests/test_gpu_parity.py)."""
    from dask_geomodeling_amd.raster.reduction import reduce_rasters
    out = reduce_rasters([dict(d) for d in case["stack"]], case["op"], **case["kwargs"])
    assert_result_equal(out, case["expect"], 0, case["id"])


def test_reduce_rasters_errors():
    from dask_geomodeling_amd.raster.reduction import check_statistic, reduce_rasters
    with pytest.raises(ValueError, match="zero-length"):
        reduce_rasters([], "first")
    with pytest.raises(KeyError):
        reduce_rasters([{"values": np.zeros((1, 2, 2), "f4"), "no_data_value": 1.0}], "mode")
    with pytest.raises(ValueError):
        check_statistic("p101")
    check_statistic("p99.5")
    check_statistic("median")


def test_integer_power_with_negative_raster_exponent_raises(cpu_vm):
    """np.power(integer array, integer array) refuses negative exponents anywhere in the array -- no-data cells
    included (the reference's Power only guards scalar exponents, elemwise.py:383-406): same error here."""
    a = {"values": np.full((1, 4, 4), 3, "i4"), "no_data_value": -9999}
    b = {"values": np.arange(16, dtype="i4").reshape(1, 4, 4) % 5, "no_data_value": -9999}
    kw = {"dtype": "int32", "fillvalue": 2147483647}
    out = devops.math_process("power", kw, a, b)
    assert np.array_equal(np.asarray(out["values"]), np.power(a["values"], b["values"]))
    b["values"][0, 2, 1] = -9999  # a no-data cell: NumPy still raises
    with pytest.raises(ValueError, match="negative integer powers"):
        devops.math_process("power", kw, a, b)
    with pytest.raises(ValueError, match="negative integer powers"):
        np.power(a["values"], b["values"])


@pytest.mark.parametrize("const_shape", [(3, 1, 1), (1, 5, 7), (5, 7), (7,), (3, 5, 1), (1,), ()])
def test_ndarray_operands_broadcast_like_numpy(cpu_vm, const_shape):
    """An ndarray operand of wrap_math_process_func (elemwise.py:257-258) broadcasts against the raster by
    NumPy's rule: one constant per band, per row, per cell ..."""
    from oracle import ref_numpy as O
    rng = np.random.default_rng(5)
    fill = np.float32(np.finfo("f4").max)
    tile = {"values": (rng.random((3, 5, 7)) * 10).astype("f4"), "no_data_value": fill}
    tile["values"][1, 2, 3] = fill
    const = (rng.random(const_shape) * 4 + 1).astype("f4")
    kw = {"dtype": "float32", "fillvalue": float(fill)}
    for op, args in (("multiply", (tile, const)), ("subtract", (const, tile))):
        assert_result_equal(devops.math_process(op, kw, *args), O.math_process(op, kw, *args), 0, "%s %s" % (op, const_shape))


def test_ndarray_operand_that_does_not_broadcast(cpu_vm):
    tile = {"values": np.zeros((3, 5, 7), "f4"), "no_data_value": np.float32(1)}
    with pytest.raises(ValueError, match="broadcast"):
        devops.math_process("add", {"dtype": "float32", "fillvalue": 1.0}, tile, np.zeros((2, 5, 7), "f4"))


def test_a_leaf_that_answers_with_a_bare_array_is_not_a_fusable_tile():
    """What the round-5 temporal fuzzing found (seed 6001061, 23 x 39 rasters): a fused group one of whose leaves answered with a
    bare ndarray made FusedTask._signature raise numpy's IndexError ("only integers, slices ...") out of get_data instead
    of declining the fusion -- the group then runs Block by Block."""
    import numpy as np
    from dask_geomodeling_amd.fuse import FusedTask
    task = FusedTask.__new__(FusedTask)
    assert task._signature([np.zeros((1, 2, 2), dtype="f4")]) is None
    assert task._signature([{"values": np.zeros((1, 2, 2), dtype="f4"), "no_data_value": None}]) is not None
    assert task._signature([None]) is None
