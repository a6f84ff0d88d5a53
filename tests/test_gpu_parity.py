"""GPU parity: the HIP path (through the C-ABI) against the reference-generated golden
vectors and against the CPU oracle on seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest

from helpers import assert_result_equal, assert_values_equal, load_family
from dask_geomodeling_amd.raster import devops
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu

# Power: float64 pow rounded once vs NumPy's faithful powf -> <= 1 ulp (north_star tolerance).
# Exp/Log (outside the north-star op list): NumPy's own SIMD loops are ~2.5 ulp off -> 3 ulp.
ULP_OPS = {"power": 1, "exp": 3, "log": 3, "log10": 3}


def _int_power_with_raster_exponent(case):
    if case["op"] == "power" and np.dtype(case["kwargs"]["dtype"]).kind in "iu":
        b = case["args"][1] if len(case["args"]) > 1 else None
        return isinstance(b, dict)
    return False


@pytest.mark.parametrize("case", load_family("math"), ids=lambda c: c["id"] + "-" + c["op"])
def test_math_golden(gpu, case):
    if "raises" in case:
        pytest.skip("reference raises %s" % case["raises"])
    if _int_power_with_raster_exponent(case):
        with pytest.raises(NotImplementedError):
            devops.math_process(case["op"], case["kwargs"], *case["args"])
        return
    out = devops.math_process(case["op"], case["kwargs"], *case["args"])
    assert_result_equal(out, case["expect"], ULP_OPS.get(case["op"], 0), case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("n1") if c["op"] in devops.P.OPCODES],
                         ids=lambda c: c["id"] + "-" + c["op"])
def test_n1_wrapper_golden(gpu, case):
    op = case["op"]
    if op in ("invert", "isdata", "isnodata"):
        out = devops.unary_bool_process(op, case["args"][0])
    else:
        out = devops.math_process(op, case["kwargs"], *case["args"])
    assert_result_equal(out, case["expect"], ULP_OPS.get(op, 0), case["id"])


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
    for xcd in (0, 1):
        for nt in (0, 3):
            program.set_tuning(vec=vec, blocks_per_cu=8, xcd_chunking=xcd, nontemporal=nt)
            try:
                got = devops.math_process("multiply", kw, {"values": a, "no_data_value": fill},
                                          {"values": b, "no_data_value": fill})
            finally:
                program.set_tuning()
            assert_result_equal(got, want, 0, "n=%d vec=%d xcd=%d nt=%d" % (n, vec, xcd, nt))


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


@pytest.mark.parametrize("case", load_family("classify"), ids=lambda c: c["id"])
def test_classify_golden(gpu, case):
    from dask_geomodeling_amd.raster import tables
    out = tables.classify_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])


@pytest.mark.parametrize("case", [c for c in load_family("reclassify") if c["op"] == "reclassify"],
                         ids=lambda c: c["id"])
def test_reclassify_golden(gpu, case):
    from dask_geomodeling_amd.raster import tables
    if "raises" in case:
        pytest.skip("reference raises %s" % case["raises"])
    out = tables.reclassify_process(*case["args"])
    assert_result_equal(out, case["expect"], 0, case["id"])
