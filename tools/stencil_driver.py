import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import stencil
from dask_geomodeling_amd.raster.sources import generate_window
F32MAX = float(np.finfo("f4").max)
side = 8192
which = sys.argv[1] if len(sys.argv) > 1 else "hillshade"
src = [generate_window((side, side), "f4", 90 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(2)]
if which.startswith("zonal"):
    from dask_geomodeling_amd import geometry
    from dask_geomodeling_amd.hip.array import DeviceArray
    h, w = 3072, 4096
    rng = np.random.default_rng(5)
    values = (rng.standard_normal((1, h, w)) * 100).astype("f4")
    values[rng.random(values.shape) < 0.1] = -9999.0
    blocks = ((np.arange(h)[:, None] // 96) * 32 + np.arange(w)[None, :] // 128).astype(np.int32)
    dv, dl = DeviceArray.from_numpy(values), DeviceArray.from_numpy(blocks)
    for rep in range(10):
        geometry.label_statistics(dv, np.float32(-9999.0), dl, 1024, which[5:] or "sum")
    runtime.sync()
    sys.exit(0)
u8 = [generate_window((side, side), "u1", 95 + i, 0.0, 6.0, 255, (50, 50), 0, 0, side, side) for i in range(2)] if which == "dilate8" else []
for rep in range(5):
    for t in u8:
        stencil.dilate(t, [1, 3])
    for t in (src if not u8 else []):
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
