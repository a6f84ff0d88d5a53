"""The C-ABI library loads without a GPU and exports every symbol include/gmhip.h declares
(no compute calls here); its struct layout matches the ctypes mirror."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("gm_init", "gm_malloc", "gm_elemwise_run", "gm_classify", "gm_classify2", "gm_reclassify",
                 "gm_resample_nearest", "gm_resample_bilinear", "gm_all_equal", "gm_last_error", "gm_sizeof"):
        assert must in names


@pytest.mark.parametrize("name", declared_symbols())
def test_library_exports(name):
    from dask_geomodeling_amd.hip import runtime
    lib = runtime.lib()
    assert getattr(lib, name) is not None


def test_struct_layout_and_no_device_behaviour():
    from dask_geomodeling_amd.hip import program, runtime
    L = program._lib()  # raises on sizeof mismatch
    assert L.gm_sizeof(0) == ctypes.sizeof(program.gm_program) <= 3072
    assert L.gm_version().decode() == "0.1.0"
    if runtime.device_count() == 0:
        # the product path fails loudly without a device: no silent CPU fallback
        with pytest.raises(runtime.GmHipError):
            runtime.require_device()
        import numpy as np
        from dask_geomodeling_amd.raster import devops
        tile = {"values": np.ones((1, 2, 2), "f4"), "no_data_value": np.float32(9)}
        with pytest.raises(runtime.GmHipError):
            devops.math_process("add", {"dtype": "float32", "fillvalue": 1.0}, tile, tile)
        from dask_geomodeling_amd import geometry
        from dask_geomodeling_amd.raster import stencil
        with pytest.raises(runtime.GmHipError):
            geometry.label_statistics(tile["values"], tile["no_data_value"], np.zeros((2, 2), "i4"), 1, "sum")
        with pytest.raises(runtime.GmHipError):
            stencil.hillshade(np.ones((1, 5, 5), "f4"), np.float32(9), 0, (1.0, 1.0), 45.0, 315.0)
        assert L.gm_init() != 0 and b"no HIP device" in L.gm_last_error()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "dask_geomodeling_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(base, f)
