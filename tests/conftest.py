import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from dask_geomodeling_amd.hip import runtime

        return runtime.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; when selected on a box without a GPU they
    # must fail loudly (never silently pass), so no skip is added here.
    pass


@pytest.fixture(scope="session")
def gpu():
    from dask_geomodeling_amd.hip import runtime

    n = runtime.device_count()
    assert n > 0, "no HIP device visible: GPU tests must run on the MI355X box"
    runtime.set_device(0)
    return runtime


TILER_ROUTES = ("product-default", "batched-tiles")


def pytest_generate_tests(metafunc):
    # every test that evaluates a RasterTiler names the route its tiles take (the `tiler_route` fixture) and runs once per route
    if "tiler_route" in metafunc.fixturenames:
        metafunc.parametrize("tiler_route", TILER_ROUTES, indirect=True)


@pytest.fixture
def tiler_route(request, monkeypatch):
    """The route of a RasterTiler over an element-wise store in tiles of fewer than 2048^2 cells: "product-default" answers it
    as one request for the whole box where that changes no cell (plan._whole_instead_of_tiles) -- what a user of the product
    gets for rasters as small as the suites' --, "batched-tiles" makes those tiles take the batched launch into the mosaic
    (fuse.TiledAssembly) that the bench's 4096^2 tiles take."""
    from dask_geomodeling_amd import plan
    if request.param == "batched-tiles":
        monkeypatch.setattr(plan, "_BATCHED_TILE_CELLS", 0)
    return request.param


@pytest.fixture
def cpu_vm(monkeypatch):
    """Run the product's host logic (Blocks, fusion, program compiler) on the CPU by
    plugging the oracle's NumPy interpreter of gm_program in place of the HIP launch.
    Test double only -- the product itself has no CPU path."""
    import numpy as np

    from dask_geomodeling_amd.hip import program, runtime
    from oracle import vm_numpy

    from dask_geomodeling_amd.raster import tables
    monkeypatch.setattr(runtime, "require_device", lambda: None)
    monkeypatch.setattr(tables, "_PREFER_PROGRAM", True)  # lone table ops too run as program steps here
    from dask_geomodeling_amd.raster import combine
    monkeypatch.setattr(combine, "_stack_bands", lambda bands, dtype, fill, shape_yx: np.concatenate(
        [np.full((1,) + tuple(shape_yx), fill, dtype) if b is None else np.asarray(b) for b in bands], axis=0))
    from dask_geomodeling_amd.raster import temporal
    from oracle import ref_numpy
    monkeypatch.setattr(temporal, "aggregate_frames", ref_numpy.temporal_aggregate_values)
    monkeypatch.setattr(temporal, "accumulate_frames", ref_numpy.temporal_cumulative_values)
    monkeypatch.setattr(temporal, "take_frames", ref_numpy.take_frames)
    monkeypatch.setattr(temporal, "sum_frames", ref_numpy.frames_sum)
    from dask_geomodeling_amd.raster import devops
    monkeypatch.setattr(devops, "_all_false", lambda tile: not np.any(np.asarray(tile)))
    monkeypatch.setattr(devops, "as_device", lambda v: np.asarray(v))
    monkeypatch.setattr(program, "as_device", lambda v: np.asarray(v))
    monkeypatch.setattr(program, "as_device_view", lambda v: np.asarray(v))
    monkeypatch.setattr(program, "launch", vm_numpy.launch)
    monkeypatch.setattr(program, "launch_batch",
                        lambda compiled, tiles, shape, outs=None: [vm_numpy.launch(compiled, t, shape) for t in tiles])
    program._CACHE.clear()
    yield
    program._CACHE.clear()
