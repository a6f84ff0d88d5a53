#!/usr/bin/env python3
"""Per-kernel roofline check of the table / resample kernels (GPU box only).
usage: kbench.py [reclass] [classify] [mask] [bilinear] [nearest] [side=8192]
Prints average launch duration (HIP events) and algorithmic GB/s per kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.raster import devops, resample, stencil, tables  # noqa: E402
from dask_geomodeling_amd.raster.sources import generate_window  # noqa: E402

F32MAX = float(np.finfo("f4").max)
I32MAX = 2147483647


# What BINDS the kernels that are not bound by HBM (round 6): wave instructions per launch on the pipe that limits them, from
# the SQ counters of profiles/r05_pmc_stencils_before.log / r05_pmc_power.log (SQ_INSTS_VALU, SQ_INSTS_LDS per launch over
# 8192^2 cells; re-collected for round 6 by tools/pmc_kbench.sh: profiles/r06_pmc_stencils.log), priced at what the part issues:
# a compute unit issues one wave64 vector instruction per clock (4 SIMDs of 16 lanes: /opt/skills/guides/MI355X_MICROARCH.md),
# float64 ones at half that; its LDS moves 128 bytes per clock, so a ds_read2_b32 of a wave (512 bytes) holds it for 4 clocks.
# label prefix -> (pipe, wave instructions per cell, clocks per instruction)
CUS, CLOCK_HZ = 256, 2.4e9
BINDING = {
    "hillshade f32 -> u8 (5": ("VALU issue", 3.42e7 / 8192 ** 2, 1.0),
    "hillshade f32 -> u8 with": ("VALU issue", 3.42e7 / 8192 ** 2, 1.0),
    "moving max f32, size 15 (": ("LDS pipe", 1.294e7 / 8192 ** 2, 4.0),
    "moving max f32, size 11 (": ("LDS pipe", 1.135e7 / 8192 ** 2, 4.0),
    "dilate u8, 2 values": ("VALU issue", 1.679e7 / 8192 ** 2, 1.0),
    "smooth f32, sigma 2 px, radius 8 (one": ("VALU issue (float64 taps: between 1 and 2 clocks each)", 7.691e7 / 8192 ** 2, 1.5),
    "smooth f32, sigma 1 px, radius 4 (one": ("VALU issue (float64 taps: between 1 and 2 clocks each)", 4.639e7 / 8192 ** 2, 1.5),
}


def binding_note(label, cells, us):
    for prefix, (pipe, per_cell, clocks) in BINDING.items():
        if label.startswith(prefix):
            floor_us = per_cell * cells * clocks / (CUS * CLOCK_HZ) * 1e6
            return "  | %.0f us of %s = %.2f of that bound" % (floor_us, pipe, floor_us / us)
    return ""


def timeit(label, bytes_per_call, calls, reps=3, cells=None):
    for f in calls:
        f()
    runtime.sync()
    e0, e1 = runtime.Event(), runtime.Event()
    e0.record()
    for _ in range(reps):
        for f in calls:
            f()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_ms(e1) / (reps * len(calls)) * 1e3
    print("%-44s %9.1f us  %7.1f GB/s  (%4.1f%% of 8 TB/s)%s" % (label, us, bytes_per_call / us / 1e3, bytes_per_call / us / 1e3 / 80,
                                                                    binding_note(label, cells, us) if cells else ""), flush=True)


def main():
    which = [a for a in sys.argv[1:] if "=" not in a] or ["reclass", "classify", "mask", "bilinear", "nearest", "stencil", "power", "temporal", "zonal"]
    opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    side = int(opts.get("side", 8192))
    runtime.set_device(0)
    n = side * side
    T = 4
    if "reclass" in which:
        table = [[k, (7 * k) % 1000] for k in range(0, 256, 2)]
        kw = {"dtype": "int64", "fillvalue": 9223372036854775807, "data": table, "select": False}
        tiles = [generate_window((side, side), "i4", 20 + i, 0.0, 256.0, I32MAX, (50, 50), 0, 0, side, side) for i in range(T)]
        timeit("reclassify i32->i64 dense LUT (12 B/px)", 12.0 * n,
               [lambda t=t: tables.reclassify_process({"values": t, "no_data_value": I32MAX}, kw) for t in tiles])
        kw2 = dict(kw, data=[[k * 1000, k] for k in range(100)])
        timeit("reclassify i32->i64 sparse keys (12 B/px)", 12.0 * n,
               [lambda t=t: tables.reclassify_process({"values": t, "no_data_value": I32MAX}, kw2) for t in tiles])
        u8 = [generate_window((side, side), "u1", 30 + i, 0.0, 6.0, 255, (50, 50), 0, 0, side, side) for i in range(T)]
        kw3 = dict(kw, data=[[0, 10], [1, 11], [3, 13], [5, 15]])
        timeit("reclassify u8->i64 dense LUT (9 B/px)", 9.0 * n,
               [lambda t=t: tables.reclassify_process({"values": t, "no_data_value": 255}, kw3) for t in u8])
        del tiles, u8
    if "classify" in which:
        tiles = [generate_window((side, side), "f4", 40 + i, -100.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(T)]
        bins5 = [-50.0, -10.0, 0.0, 10.0, 50.0]
        timeit("classify f32->u8, 5 edges (5 B/px)", 5.0 * n,
               [lambda t=t: tables.classify_process({"values": t, "no_data_value": np.float32(F32MAX)}, bins5, False) for t in tiles])
        bins200 = list(np.linspace(-100, 100, 200))
        timeit("classify f32->u8, 200 edges (5 B/px)", 5.0 * n,
               [lambda t=t: tables.classify_process({"values": t, "no_data_value": np.float32(F32MAX)}, bins200, True) for t in tiles])
        del tiles
    if "mask" in which:
        i8 = [generate_window((side, side), "i8", 50 + i, 0.0, 1000.0, 9223372036854775807, (50, 50), 0, 0, side, side) for i in range(T)]
        timeit("mask i64->u8 (VM, 9 B/px)", 9.0 * n,
               [lambda t=t: devops.mask_process({"values": t, "no_data_value": 9223372036854775807}, 1) for t in i8])
        del i8
        i4 = [generate_window((side, side), "i4", 60 + i, 0.0, 1000.0, I32MAX, (50, 50), 0, 0, side, side) for i in range(T)]
        timeit("mask i32->u8 (VM, 5 B/px)", 5.0 * n,
               [lambda t=t: devops.mask_process({"values": t, "no_data_value": I32MAX}, 1) for t in i4])
        del i4
        from dask_geomodeling_amd.hip import program as P
        u8 = [generate_window((side, side), "u1", 30 + i, 0.0, 12.0, 255, (50, 50), 0, 0, side, side) for i in range(T)]
        table = {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": False, "data": [[0, 10], [1, 11], [3, 13], [5, 15]]}
        leaf = P.Leaf(0, "u1", 255)
        for label, root, bpp in (("u8 program: mask -> u8 (table path, 2 B/px)", devops.mask_node(leaf, 1), 2.0),
                                 ("u8 program: reclassify + mask -> u8 (table path, 2 B/px)", devops.mask_node(tables.reclassify_node(leaf, table), 1), 2.0),
                                 ("u8 program: reclassify -> i64 (table path, 9 B/px)", tables.reclassify_node(leaf, table), 9.0)):
            comp = P.compile_expr(root, 1)
            timeit(label, bpp * n, [lambda t=t: P.launch(comp, [t], (1, side, side)) for t in u8])
        del u8
    if "bilinear" in which:
        src = [generate_window((side, side), "f4", 70 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(2)]
        timeit("bilinear f32 2x upsample (5 B/out px)", 5.0 * 4 * n,
               [lambda t=t: resample.bilinear(t, 2 * side, 2 * side, np.float32(F32MAX)) for t in src])
        timeit("bilinear f32 1.5x upsample (5.78 B/out px)", (4 + 4 / 2.25) * 2.25 * n,
               [lambda t=t: resample.bilinear(t, side * 3 // 2, side * 3 // 2, np.float32(F32MAX)) for t in src])
        half = [generate_window((side // 2, side // 2), "f4", 72 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side // 2, side // 2) for i in range(2)]
        timeit("bilinear f32 3x upsample (4.44 B/out px)", (4 + 4 / 9.0) * 2.25 * n,
               [lambda t=t: resample.bilinear(t, side * 3 // 2, side * 3 // 2, np.float32(F32MAX)) for t in half])
        timeit("bilinear f32 1.25x upsample (6.56 B/out px)", (4 + 4 / 1.5625) * 1.5625 * n,
               [lambda t=t: resample.bilinear(t, side * 5 // 4, side * 5 // 4, np.float32(F32MAX)) for t in src])
        timeit("bilinear f32 1.5x upsample onto %d columns (rows not 16-byte aligned)" % (side * 3 // 2 - 3), (4 + 4 / 2.25) * 2.25 * n,
               [lambda t=t: resample.bilinear(t, side * 3 // 2 - 3, side * 3 // 2 - 3, np.float32(F32MAX)) for t in src])
        quarter = [generate_window((side * 3 // 8, side * 3 // 8), "f4", 74 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side * 3 // 8, side * 3 // 8) for i in range(2)]
        timeit("bilinear f32 4x upsample (4.25 B/out px)", (4 + 4 / 16.0) * 2.25 * n,
               [lambda t=t: resample.bilinear(t, side * 3 // 2, side * 3 // 2, np.float32(F32MAX)) for t in quarter])
        del src, half, quarter
    if "stencil" in which:
        src = [generate_window((side, side), "f4", 90 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(T)]
        m = (side - 2) * (side - 2)
        timeit("hillshade f32 -> u8 (5 B/px)", 5.0 * m,
               [lambda t=t: stencil.hillshade(t, np.float32(F32MAX), 0, (0.5, 0.5), 45.0, 315.0) for t in src], cells=m)
        srcm = [generate_window((side + 2, side + 2), "f4", 90 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side + 2, side + 2) for i in range(2)]
        timeit("hillshade f32 -> u8 with margin (%d wide, 5 B/px)" % (side + 2), 5.0 * n,
               [lambda t=t: stencil.hillshade(t, np.float32(F32MAX), 0, (0.5, 0.5), 45.0, 315.0) for t in srcm], cells=n)
        del srcm
        for size in (3, 5, 7, 11, 15):
            timeit("moving max f32, size %d (8 B/px)" % size, 8.0 * (side - size + 1) ** 2,
                   [lambda t=t: stencil.moving_max(t, np.float32(F32MAX), size) for t in src], cells=(side - size + 1) ** 2)
        # the zero-sign pass behind a float launch at work: 2 % of the cells -0.0, 2 % +0.0, the rest negative (every
        # window's maximum a zero, most windows with zeros of both signs)
        from dask_geomodeling_amd.hip.array import DeviceArray
        rng = np.random.default_rng(5)
        zeros = -1.0 - rng.random((1, side, side), dtype=np.float32)
        pick = rng.random(zeros.shape)
        zeros[pick < 0.02] = -0.0
        zeros[pick > 0.98] = 0.0
        zeros = [DeviceArray.from_numpy(zeros)]
        for size in (3, 7):
            timeit("moving max f32, size %d, zeros of both signs in most windows (8 B/px)" % size, 8.0 * (side - size + 1) ** 2,
                   [lambda t=t: stencil.moving_max(t, np.float32(F32MAX), size) for t in zeros])
        del zeros
        u8 = [generate_window((side, side), "u1", 95 + i, 0.0, 6.0, 255, (50, 50), 0, 0, side, side) for i in range(T)]
        timeit("dilate u8, 2 values (2 B/px)", 2.0 * m, [lambda t=t: stencil.dilate(t, [1, 3]) for t in u8], cells=m)
        u8m = [generate_window((side + 2, side + 2), "u1", 95 + i, 0.0, 6.0, 255, (50, 50), 0, 0, side + 2, side + 2) for i in range(T)]
        timeit("dilate u8 with margin (%d wide), 2 values (2 B/px)" % (side + 2), 2.0 * n, [lambda t=t: stencil.dilate(t, [1, 3]) for t in u8m])
        del u8m
        timeit("dilate f32, 2 values (8 B/px)", 8.0 * m, [lambda t=t: stencil.dilate(t, [1.0, 3.0]) for t in src])
        for size_px in (3.0, 6.0):
            timeit("smooth f32, sigma %.0f px, radius %d (one launch, 8 B/px)" % (size_px / 3, int(4 * size_px / 3 + 0.5)), 8.0 * n,
                   [lambda t=t: stencil.gaussian_smooth(t, np.float32(F32MAX), 0, (size_px / 3, size_px / 3)) for t in src[:2]], cells=n)
            timeit("smooth f32, sigma %.0f px, radius %d (two launches, 16 B/px)" % (size_px / 3, int(4 * size_px / 3 + 0.5)), 16.0 * n,
                   [lambda t=t: stencil.gaussian_smooth(t, np.float32(F32MAX), 0, (size_px / 3, size_px / 3), fused=False) for t in src[:2]])
        for size_px in (18.0, 39.0):  # (radii 24 and 52: what a Smooth of 40 pixels in zoom mode asks for)
            timeit("smooth f32, sigma %.0f px, radius %d (two launches: the sliding-window passes, 16 B/px)"
                   % (size_px / 3, int(4 * size_px / 3 + 0.5)), 16.0 * n,
                   [lambda t=t: stencil.gaussian_smooth(t, np.float32(F32MAX), 0, (size_px / 3, size_px / 3)) for t in src[:2]])
        del src, u8
    if "power" in which:
        tiles = [generate_window((side, side), "f4", 50 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(T)]
        kw = {"dtype": "float32", "fillvalue": F32MAX}
        # (a bare math_process call builds and compiles its one-step program anew: ~200 us of host time against a kernel of
        #  ~130, so the events around the calls time the host.  The kernel's own duration is what the kernel stamps at its
        #  begin and end -- csrc/gm_spec.hip, the figure bench.py reports as kernel_only_us -- once the program's specialised
        #  kernel has been compiled.)
        from dask_geomodeling_amd.hip import program as P
        for op, args in (("power", (2.5,)), ("exp", ()), ("log", ())):
            calls = [lambda t=t: devops.math_process(op, kw, {"values": t, "no_data_value": np.float32(F32MAX)}, *args) for t in tiles]
            for _ in range(3):
                for f in calls:
                    f()
            if P.spec_pending():
                P.spec_wait()
            timeit("%s f32 (VM, 8 B/px), per call incl. the host's program build" % op, 8.0 * n, calls)
            P.spec_timing(True)
            for _ in range(3):
                for f in calls:
                    f()
            runtime.sync()
            P.spec_timing(False)
            total_us, timed = P.spec_timing_read()
            if timed:
                us = total_us / timed
                print("%-44s %9.1f us  %7.1f GB/s  (%4.1f%% of 8 TB/s)" % ("%s f32 (VM, 8 B/px), kernel only" % op, us, 8.0 * n / us / 1e3, 8.0 * n / us / 1e3 / 80), flush=True)
        del tiles
    if "temporal" in which:
        from dask_geomodeling_amd.hip.array import DeviceArray
        from dask_geomodeling_amd.raster import temporal
        tside, frames = 4096, 24
        stack = DeviceArray.empty((frames, tside, tside), np.dtype("f4"))
        for k in range(frames):
            tile = generate_window((tside, tside), "f4", 60 + k, 0.0, 100.0, F32MAX, (97, 89), 0, 0, tside, tside)
            runtime.check(runtime.lib().gm_d2d(stack.band_slice(k, k + 1).ptr, tile.ptr, tside * tside * 4), "gm_d2d")
        px = tside * tside
        hourly = [(6 * i, 6) for i in range(4)]
        for stat in ("sum", "mean", "max", "count", "std", "median"):
            out_dt = temporal.dtype_for_statistic("f4", stat)
            reads = 2 if stat == "std" else 1
            timeit("temporal %s f32, 24 frames -> 4 labels (%d B/px)" % (stat, reads * 96 + 16), (reads * 96.0 + 16.0) * px,
                   [lambda: temporal.aggregate_frames(stack, np.float32(F32MAX), hourly, stat, out_dt)])
        timeit("temporal sum f32, 24 frames -> 1 label (100 B/px)", 100.0 * px,
               [lambda: temporal.aggregate_frames(stack, np.float32(F32MAX), [(0, 24)], "sum", np.dtype("f4"))])
        timeit("cumulative sum f32, 24 frames (192 B/px)", 192.0 * px,
               [lambda: temporal.accumulate_frames(stack, np.float32(F32MAX), hourly, (0, 24), "sum", np.dtype("f4"))])
        del stack
    if "nearest" in which:
        src = [generate_window((side, side), "f4", 80 + i, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side) for i in range(T)]
        gt = (0.0, 1.0, 0.0, float(side), 0.0, -1.0)
        timeit("nearest warp f32 identity grid (8 B/px)", 8.0 * n,
               [lambda t=t: resample.nearest_warp(t, gt, (0.0, 0.0, float(side), float(side)), side, side, F32MAX) for t in src])
        timeit("nearest warp f32 2x upsample (5 B/out px)", 5.0 * 4 * n,
               [lambda t=t: resample.nearest_warp(t, gt, (0.0, 0.0, float(side), float(side)), 2 * side, 2 * side, F32MAX) for t in src[:2]])
        odd = side * 3 // 2 - 3
        timeit("nearest warp f32 1.5x upsample onto %d columns (rows not 16-byte aligned, 5.78 B/out px)" % odd, (4 + 4 / 2.25) * 2.25 * n,
               [lambda t=t: resample.nearest_warp(t, gt, (0.0, 0.0, float(side), float(side)), odd, odd, F32MAX) for t in src[:2]])
        half = side // 2
        timeit("nearest warp f32 window, quarter (8 B/out px)", 8.0 * n / 4,
               [lambda t=t: resample.nearest_warp(t, gt, (100.0, 100.0, 100.0 + half, 100.0 + half), half, half, F32MAX) for t in src])

    if "zonal" in which:
        # AggregateRaster's numeric core on the reference's default raster limit (12 * 1024^2 cells): 8 B/cell
        # algorithmic (value + label); device-resident raster and labels, results copied back (as the caller needs them)
        import time
        from dask_geomodeling_amd import geometry
        from dask_geomodeling_amd.hip.array import DeviceArray
        h, w = 3072, 4096
        rng = np.random.default_rng(5)
        values = (rng.standard_normal((1, h, w)) * 100).astype("f4")
        values[rng.random(values.shape) < 0.1] = -9999.0
        blocks = ((np.arange(h)[:, None] // 96) * 32 + np.arange(w)[None, :] // 128).astype(np.int32)  # 1024 labels of 12 k cells
        one = np.zeros((h, w), dtype=np.int32)                                                       # one label of 12.6 M cells
        dev_values = DeviceArray.from_numpy(values)
        for name, labels, n_labels in (("1024 labels", blocks, 1024), ("1 label", one, 1)):
            dev_labels = DeviceArray.from_numpy(labels)
            for stat, q in (("sum", None), ("mean", None), ("count", None), ("max", None), ("median", None), ("percentile", 95.0)):
                for _ in range(3):  # (a route's scratch buffers are sized by its first call's verdict and taken by its second)
                    geometry.label_statistics(dev_values, np.float32(-9999.0), dev_labels, n_labels, stat, q)
                runtime.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    geometry.label_statistics(dev_values, np.float32(-9999.0), dev_labels, n_labels, stat, q)
                us = (time.perf_counter() - t0) / 5 * 1e6
                print("zonal %-10s f32 %d x %d, %-12s %9.1f us  %7.1f GB/s" % (stat + ("" if q is None else str(q)), h, w, name, us, 8.0 * h * w / us / 1e3), flush=True)
        # the rasters above are zero-mean noise: a sum that wanders through the binades and changes sign is the worst case of
        # the in-order float sums.  An elevation-like raster (all cells positive: the sum only grows) for comparison:
        positive = np.abs(values) + np.float32(3.0)
        positive[values == -9999.0] = -9999.0
        dev_positive = DeviceArray.from_numpy(positive)
        for name, labels, n_labels in (("1024 labels", blocks, 1024), ("1 label", one, 1)):
            dev_labels = DeviceArray.from_numpy(labels)
            for _ in range(3):
                geometry.label_statistics(dev_positive, np.float32(-9999.0), dev_labels, n_labels, "sum", None)
            runtime.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                geometry.label_statistics(dev_positive, np.float32(-9999.0), dev_labels, n_labels, "sum", None)
            us = (time.perf_counter() - t0) / 5 * 1e6
            print("zonal %-10s f32 %d x %d, %-12s %9.1f us  %7.1f GB/s  (positive cells)" % ("sum", h, w, name, us, 8.0 * h * w / us / 1e3), flush=True)
        ints = DeviceArray.from_numpy(np.round(values).astype("i4"))  # an integer raster: sums need no order, hence no sort
        for name, labels, n_labels in (("1024 labels", blocks, 1024), ("1 label", one, 1)):
            dev_labels = DeviceArray.from_numpy(labels)
            for _ in range(3):
                geometry.label_statistics(ints, -9999, dev_labels, n_labels, "sum")
            runtime.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                geometry.label_statistics(ints, -9999, dev_labels, n_labels, "sum")
            us = (time.perf_counter() - t0) / 5 * 1e6
            print("zonal %-10s i32 %d x %d, %-12s %9.1f us  %7.1f GB/s" % ("sum", h, w, name, us, 8.0 * h * w / us / 1e3), flush=True)
        if opts.get("cpu", "1") == "1":
            from scipy import ndimage
            frame = values[0]
            active = frame != np.float32(-9999.0)
            for stat, fn in (("sum", ndimage.sum), ("median", ndimage.median)):
                t0 = time.perf_counter()
                fn(frame[active], labels=blocks[active], index=np.arange(1024))
                print("  scipy.ndimage.%s on the host, 1024 labels: %.1f ms" % (stat, (time.perf_counter() - t0) * 1e3), flush=True)


if __name__ == "__main__":
    main()
