import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
side = 4096
F = bench.F32MAX
da = generate_window((side, side), "f4", 1, 0.0, 100.0, F, bench.LATTICE, 0, 0, side, side)
db = generate_window((side, side), "f4", 2, 0.0, 100.0, F, bench.LATTICE, 0, 0, side, side)
A = raster.MemorySource(da, np.float32(F), "EPSG:28992", 1.0, (0.0, float(side)))
B = raster.MemorySource(db, np.float32(F), "EPSG:28992", 1.0, (0.0, float(side)))
view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
req = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")
with gmconfig.config.set({"geomodeling.materialize": False}):
    for tile in (256, 1024):
        tiler = raster.RasterTiler(view, tile)
        tiler.get_data(**req); runtime.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            tiler.get_data(**req)
        runtime.sync()
        dt = (time.perf_counter() - t0) / 5
        print("tile %4d: %.2f ms per get_data (%d tiles) = %.0f Mpix/s" % (tile, dt * 1e3, (side // tile) ** 2, side * side / dt / 1e6))
    tiler = raster.RasterTiler(view, 256)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        tiler.get_data(**req)
    runtime.sync(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
