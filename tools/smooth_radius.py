import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from scipy import ndimage
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.hip.array import DeviceArray
from dask_geomodeling_amd.raster import stencil
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
if "--parity" in sys.argv:
    rng = np.random.default_rng(5)
    bad = 0
    for dtype in ("f4", "f8", "i2", "u1", "i8"):
        for (h, w), sig, crop in (((300, 517), (13.07, 13.07), (40, 40)), ((300, 517), (30.0, 2.0), (0, 0)), ((97, 1030), (0.7, 50.3), (3, 7)),
                                  ((700, 130), (80.0, 80.0), (0, 0)), ((64, 64), (100.0, 100.0), (0, 0)), ((1200, 70), (200.0, 3.0), (5, 5))):
            a = (rng.random((2, h, w)) * 200).astype(dtype)
            ndv = np.array(200 if dtype != "u1" else 199).astype(dtype)[()]
            a[:, ::7, ::5] = ndv
            fill = 3
            exp = a.copy(); exp[exp == ndv] = fill
            exp = ndimage.gaussian_filter(exp, (0, sig[0], sig[1]), mode="constant", cval=fill)
            exp = exp[:, crop[0]:h - crop[0], crop[1]:w - crop[1]]
            got = stencil.gaussian_smooth(DeviceArray.from_numpy(a), ndv, fill, sig, crop, fused=False).to_numpy()
            ok = np.array_equal(got, exp)
            bad += not ok
            print(dtype, (h, w), sig, crop, "ok" if ok else "MISMATCH %d" % (got != exp).sum())
    print("parity:", "all equal" if not bad else "%d FAILED" % bad)
    sys.exit(bad)
t = generate_window((4096 + 80, 4096 + 80), "f4", 1, 0.0, 100.0, F32MAX, (97, 89), 0, 0, 4176, 4176)
t16 = DeviceArray.from_numpy((np.random.default_rng(1).random((1, 4176, 4176)) * 100).astype("i2"))
for name, tile, ndv in (("f4", t, np.float32(F32MAX)), ("i2", t16, np.int16(-1))):
    for sigma in (13.07, 26.0, 6.0, 3.0, 1.0):
        for _ in range(3):
            stencil.gaussian_smooth(tile, ndv, 0.0, (sigma, sigma), (40, 40), fused=False)
        runtime.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            stencil.gaussian_smooth(tile, ndv, 0.0, (sigma, sigma), (40, 40), fused=False)
        runtime.sync()
        print("%s sigma %.2f radius %d: %.1f us" % (name, sigma, int(4 * sigma + 0.5), (time.perf_counter() - t0) / 10 * 1e6))
