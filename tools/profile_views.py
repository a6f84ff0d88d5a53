"""cProfile of get_data on the plan path (PLANS=0: the graph path) for the views tools/e2e_views.py found slow (GPU box):
    python tools/profile_views.py [group fill max tagg resample cumulative place dilate]"""
import cProfile
import io
import os
import pstats
import sys
from datetime import datetime, timedelta

import numpy as np

sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd import config as gmconfig  # noqa: E402
from dask_geomodeling_amd import raster  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.raster.sources import generate_window  # noqa: E402

runtime.set_device(0)
F32MAX = float(np.finfo("f4").max)
side, big = 4096, 16384
da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big)
dc = generate_window((big, big), "u1", 7003, 0.0, 12.0, 255, (97, 89), 0, 0, big, big)
A = raster.MemorySource(da, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
B = raster.MemorySource(db, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
C = raster.MemorySource(dc, np.uint8(255), "EPSG:28992", 1.0, (0.0, float(big)))
frames = 8
dt = generate_window((frames * 4096, 4096), "f4", 7004, 0.0, 100.0, F32MAX, (97, 89), 0, 0, frames * 4096, 4096).reshape((frames, 4096, 4096))
T = raster.MemorySource(dt, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, 4096.0), time_first=datetime(2020, 1, 1), time_delta=timedelta(hours=6))
boxes = [(float(i * side + 64), float(j * side + 64), float((i + 1) * side + 64), float((j + 1) * side + 64)) for i in range(3) for j in range(3)]
req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")
tbox = dict(mode="vals", bbox=(0.0, 0.0, 4096.0, 4096.0), width=4096, height=4096, projection="EPSG:28992",
            start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
spatial = [dict(req, bbox=b) for b in boxes]
views = {"group": (raster.Group(A, B), spatial),
         "fill": (raster.FillNoData(A, B), spatial),
         "max": (raster.Max(A, B, A * 2), spatial),
         "dilate": (raster.Dilate(raster.Reclassify(C, [[1, 3], [2, 3], [5, 9]]), [3]), spatial),
         "tagg": (raster.TemporalAggregate(T, "D", "mean"), [tbox]),
         "resample": (raster.Resample(T, "3h", "nearest"), [tbox]),
         "cumulative": (raster.Cumulative(T), [tbox]),
         "place": (raster.Place(A, "EPSG:28992", (8192.0, 8192.0), [(8192.0, 8192.0), (8292.0, 8092.0)]), spatial)}
small = [dict(mode="vals", bbox=(float(i * 256), float(j * 256), float(i * 256 + 256), float(j * 256 + 256)), width=256, height=256,
              projection="EPSG:28992") for i in range(1, 4) for j in range(1, 4)]
views.update({"classify_hs": (raster.Classify(raster.HillShade(A), [20, 50, 80, 110, 140, 170, 200, 230]), spatial),
              "mask_dilate": (raster.Mask(raster.Dilate(C, [3, 5]), 1), spatial)})
views.update({"hillshade256": (raster.HillShade(A), small), "smooth256": (raster.Smooth(A, 2.0), small),
              "add256": (A + B, small)})
whole = [dict(mode="vals", bbox=(64.0, 64.0, 64.0 + 16256, 64.0 + 16256), width=16256, height=16256, projection="EPSG:28992")]
views.update({"tiler_hs_mix": (raster.RasterTiler(raster.HillShade(A) * 0.5 + B, 4064), whole),
              "tiler_hs": (raster.RasterTiler(raster.HillShade(A), 4064), whole)})
for name in sys.argv[1:] or views:
    view, requests = views[name]
    with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": os.environ.get("PLANS", "1") == "1"}):
        for k in range(5):
            view.get_data(**requests[k % len(requests)])
        runtime.sync()
        pr = cProfile.Profile()
        pr.enable()
        for k in range(40 if len(requests) == 1 else 400):
            view.get_data(**requests[k % len(requests)])
        pr.disable()
        runtime.sync()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
    print("=====", name, "(400 requests)")
    print("\n".join(line for line in out.getvalue().splitlines() if line.strip())[:7000])
