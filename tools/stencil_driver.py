import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import stencil
from dask_geomodeling_amd.raster.sources import generate_window
F32MAX = float(np.finfo("f4").max)
side = 8192
which = sys.argv[1] if len(sys.argv) > 1 else "hillshade"
src = [generate_window((side, side), "f4", 90 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(2)]
for rep in range(5):
    for t in src:
        if which == "hillshade":
            stencil.hillshade(t, np.float32(F32MAX), 0, (0.5, 0.5), 45.0, 315.0)
        elif which.startswith("mm"):
            stencil.moving_max(t, np.float32(F32MAX), int(which[2:]))
        elif which == "bilinear":
            from dask_geomodeling_amd.raster import resample
            resample.bilinear(t, 2 * side, 2 * side, np.float32(F32MAX))
        elif which.startswith("smooth"):
            stencil.gaussian_smooth(t, np.float32(F32MAX), 0, (float(which[6:]), float(which[6:])))
runtime.sync()
