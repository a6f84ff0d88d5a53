"""Which kernels one request of a view launches, and for how long (GPU box).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vk -- python3 tools/view_kernels.py place      # run the view 12 times
  python3 tools/view_kernels.py --read gpurun_out/vk 12                                       # per-request kernel table
  PLANS=0 CPROFILE=1 python3 tools/view_kernels.py classify200                                # cProfile of 50 requests (graph path)

Views: the series probes of tools/e2e_views.py that sit well above their kernels' time (MEASUREMENTS.md section 8, series seams)."""
import csv, glob, sys
from collections import OrderedDict

RUNS = 12


def read(folder, runs):
    for f in glob.glob(folder + "/**/*kernel_trace.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = OrderedDict()
        for r in rows:
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:90]
            e = per.setdefault(name, [0, 0])
            e[0] += 1
            e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        total = 0
        for name, (n, ns) in per.items():
            print("%6.2f launches  %9.1f us per request  %s" % (n / runs, ns / 1e3 / runs, name))
            total += ns
        print("kernels per request (uploads of the setup included in the first lines): %.1f us" % (total / 1e3 / runs))


def build(name):
    """(view, request) of one of the probes."""
    sys.path.insert(0, "/root/repo")
    from datetime import datetime, timedelta
    import numpy as np
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.hip import runtime
    from dask_geomodeling_amd.raster.sources import generate_window
    runtime.set_device(0)
    F32MAX = float(np.finfo("f4").max)
    frames, side = 8, 4096
    if name in ("smooth40", "classify200", "maskdilate", "hillclass", "groupab", "hillmix", "tilesmooth"):
        big = 16384
        A = raster.MemorySource(generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big),
                                np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
        C = raster.MemorySource(generate_window((big, big), "u1", 7003, 0.0, 12.0, 255, (97, 89), 0, 0, big, big),
                                np.uint8(255), "EPSG:28992", 1.0, (0.0, float(big)))
        views = {"smooth40": raster.Smooth(A, 40.0), "classify200": raster.Classify(A, [float(x) / 2 for x in range(200)]),
                 "maskdilate": raster.Mask(raster.Dilate(C, [3, 5]), 1),
                 "hillclass": raster.Classify(raster.HillShade(A), [20, 50, 80, 110, 140, 170, 200, 230])}
        if name == "tilesmooth":  # 254 tiles, tile by tile (a Smooth's tiles are not windows of the whole)
            return raster.RasterTiler(raster.Smooth(A, 2.0), 1024), dict(mode="vals", bbox=(64.0, 64.0, big - 64.0, big - 64.0),
                                                                         width=big - 128, height=big - 128, projection="EPSG:28992")
        if name in ("groupab", "hillmix"):
            B = raster.MemorySource(generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, (97, 89), 0, 0, big, big),
                                    np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(big)))
            views.update(groupab=raster.Group(A, B), hillmix=raster.HillShade(A + B) * 0.5)
        return views[name], dict(mode="vals", bbox=(4160.0, 4160.0, 8256.0, 8256.0), width=side, height=side, projection="EPSG:28992")
    dt = generate_window((frames * side, side), "f4", 7004, 0.0, 100.0, F32MAX, (97, 89), 0, 0, frames * side, side).reshape((frames, side, side))
    T = raster.MemorySource(dt, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(side)), time_first=datetime(2020, 1, 1),
                            time_delta=timedelta(hours=6))
    tbox = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992",
                start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
    views = {
        "place": raster.Place(T, "EPSG:28992", (2048.0, 2048.0), [(2100.0, 2000.0)]),
        "group": raster.TemporalAggregate(raster.Group(T, T * 0.5), "D", "mean"),
        "fillnodata": raster.FillNoData(T, T * 0.5),
        "clip": raster.Clip(raster.Cumulative(T), T > 50),
        "pointmean": raster.TemporalAggregate(T, "D", "mean"),
    }
    if name == "pointmean":
        tbox = dict(tbox, bbox=(2000.5, 2000.5, 2000.5, 2000.5), width=1, height=1)
    return views[name], tbox


def main():
    if sys.argv[1] == "--read":
        return read(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else RUNS)
    import os, time
    view, tbox = build(sys.argv[1])
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd.hip import runtime
    # (results stay on the device, like tools/e2e_views.py; PLANS=0: the fused graph per request through the scheduler)
    gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": os.environ.get("PLANS", "1") == "1"}).__enter__()
    if os.environ.get("CPROFILE"):  # where the host time of a request goes
        import cProfile, io, pstats
        for _ in range(8):
            view.get_data(**tbox)
        runtime.sync()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(50):
            view.get_data(**tbox)
        runtime.sync()
        pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
        print("\n".join(line[:160] for line in out.getvalue().splitlines()[4:60]))
        return
    for i in range(RUNS):
        if i == RUNS - 5:
            runtime.sync()
            t0 = time.perf_counter()
        view.get_data(**tbox)
    runtime.sync()
    print("%s: %.1f us per request (last 5 of %d)" % (sys.argv[1], (time.perf_counter() - t0) / 5 * 1e6, RUNS))


if __name__ == "__main__":
    main()
