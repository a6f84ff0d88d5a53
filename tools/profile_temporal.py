"""cProfile of get_data for tiled temporal views over a resident series of 8 frames of 4096^2 (GPU box): where the host time of
`tiler(TemporalAggregate(T, "D", "mean"), 1024)` and `tiler(T * 2, 1024)` goes (windows of several bands)."""
import cProfile, io, os, pstats, sys
from datetime import datetime, timedelta
import numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import config as gmconfig, raster
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
frames = 8
dt = generate_window((frames * 4096, 4096), "f4", 7004, 0.0, 100.0, F32MAX, (97, 89), 0, 0, frames * 4096, 4096).reshape((frames, 4096, 4096))
T = raster.MemorySource(dt, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, 4096.0), time_first=datetime(2020, 1, 1), time_delta=timedelta(hours=6))
point = dict(mode="vals", bbox=(2000.5, 2000.5, 2000.5, 2000.5), width=1, height=1, projection="EPSG:28992", start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
tile = dict(mode="vals", bbox=(1024.0, 1024.0, 2048.0, 2048.0), width=1024, height=1024, projection="EPSG:28992", start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
whole = dict(mode="vals", bbox=(0.0, 0.0, 4096.0, 4096.0), width=4096, height=4096, projection="EPSG:28992", start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
gmconfig.config.set({"geomodeling.materialize": False}).__enter__()
for name, view, req in (("tiler TAgg", raster.RasterTiler(raster.TemporalAggregate(T, "D", "mean"), 1024), whole), ("tiler T*2", raster.RasterTiler(T * 2, 1024), whole)):
    for _ in range(5):
        view.get_data(**req)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        view.get_data(**req)
    runtime.sync(); pr.disable()
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(40)
    print("=====", name); print("\n".join(l[:150] for l in out.getvalue().splitlines()[4:50]))
