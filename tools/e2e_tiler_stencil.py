import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import raster, config as gmconfig
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
side, big = 4096, 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
dc = generate_window((big, big), "u1", 7003, 0.0, 12.0, 255, (97, 89), 0, 0, big, big)
C = raster.MemorySource(dc, np.uint8(255), "EPSG:28992", 1.0, (0.0, float(big)))
views = {"tiler(HillShade(A))": raster.RasterTiler(raster.HillShade(A), side),
         "tiler(Dilate(C, [3, 5]))": raster.RasterTiler(raster.Dilate(C, [3, 5]), side),
         "tiler(MovingMax(A, 5))": raster.RasterTiler(raster.MovingMax(A, 5), side),
         "tiler(Smooth(A, 2))": raster.RasterTiler(raster.Smooth(A, 2.0), side),
         "tiler(MovingMax(A + B, 5))": raster.RasterTiler(raster.MovingMax(A + B, 5), side),
         "tiler(Smooth(A * 2, 2))": raster.RasterTiler(raster.Smooth(A * 2, 2.0), side),
         "tiler(HillShade(A * 2 + B))": raster.RasterTiler(raster.HillShade(A * 2 + B), side),
         "tiler(HillShade(A) * 0.5 + B)": raster.RasterTiler(raster.HillShade(A) * 0.5 + B, side),
         # 254 tiles: one request where a mosaic of tiles is the whole (plan._whole_instead_of_tiles: the stencils but Smooth)
         "tiler(HillShade(A), 1024)": raster.RasterTiler(raster.HillShade(A), 1024),
         "tiler(Dilate(C, [3, 5]), 1024)": raster.RasterTiler(raster.Dilate(C, [3, 5]), 1024),
         "tiler(Smooth(A, 2), 1024)": raster.RasterTiler(raster.Smooth(A, 2.0), 1024)}
inner = 64.0
req = dict(mode="vals", bbox=(inner, inner, big - inner, big - inner), width=big - 128, height=big - 128, projection="EPSG:28992")
for name, view in views.items():
    for plans in (True, False):
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": plans}):
            view.get_data(**req); view.get_data(**req); runtime.sync()
            t0 = time.perf_counter(); n = 10
            for _ in range(n):
                view.get_data(**req)
            runtime.sync()
            ms = (time.perf_counter() - t0) / n * 1e3
        print("%-32s %-18s %7.2f ms per 16256^2 request  %8.0f Mpix/s" % (name, "plans" if plans else "graph per request", ms, (big - 128) ** 2 / ms / 1e3), flush=True)
