#!/usr/bin/env python3
"""Launch-shape sweeps of the round-3 kernels (GPU box only): every configuration runs in a child process (the
kernels read their GM_* tuning variable once), is checked bit for bit against the baseline configuration's output on
ragged / windowed inputs, and timed with HIP events.
usage: sweep_r03.py bl2x | lookup32 | child-bl2x | child-lookup32"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F32MAX = float(np.finfo("f4").max)
I32MAX = 2147483647


def timed(calls, reps=5):
    from dask_geomodeling_amd.hip import runtime
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
    return e0.elapsed_ms(e1) / (reps * len(calls)) * 1e3


def child_bl2x():
    from dask_geomodeling_amd.hip import runtime
    from dask_geomodeling_amd.hip.array import DeviceArray
    from dask_geomodeling_amd.raster import resample
    from dask_geomodeling_amd.raster.sources import generate_window
    runtime.set_device(0)
    rng = np.random.default_rng(7)
    h = hashlib.sha1()
    ndv = np.float32(F32MAX)
    for bands, hs, ws in ((1, 36, 40), (2, 101, 260), (1, 130, 516), (1, 33, 1028), (3, 64, 256), (1, 257, 772), (1, 5, 4)):
        a = (rng.random((bands, hs, ws)) * 100).astype("f4")
        a[rng.random(a.shape) < 0.01] = ndv
        a[rng.random(a.shape) < 0.002] = np.nan
        a[rng.random(a.shape) < 0.002] = np.inf
        for nd in (ndv, None):
            out = np.asarray(resample.bilinear(a, 2 * hs, 2 * ws, nd))
            h.update(out.tobytes())
    # a strided window of a resident raster
    big = (rng.random((1, 300, 1200)) * 100).astype("f4")
    big[rng.random(big.shape) < 0.01] = ndv
    dev = DeviceArray.from_numpy(big)
    for (r0, c0, hh, ww) in ((0, 0, 128, 512), (17, 64, 200, 1028), (100, 1000, 200, 200), (3, 4, 64, 260)):
        out = np.asarray(resample.bilinear(dev.window(r0, c0, hh, ww), 2 * hh, 2 * ww, ndv))
        h.update(out.tobytes())
    side = int(os.environ.get("SWEEP_SIDE", "16384"))
    src = generate_window((side, side), "f4", 70, 0.0, 100.0, F32MAX, (97, 89), 0, 0, side, side)
    us = timed([lambda: resample.bilinear(src, 2 * side, 2 * side, ndv)])
    chk = resample.bilinear(src, 2 * side, 2 * side, ndv).checksum()
    print("RESULT %s %s %.1f us %.1f GB/s (%.3f of 8 TB/s)" % (h.hexdigest()[:12], chk, us, 20.0 * side * side / us / 1e3,
                                                             20.0 * side * side / us / 1e3 / 8000), flush=True)


def child_lookup32():
    from dask_geomodeling_amd import raster
    from dask_geomodeling_amd.core.graphs import get_sync
    from dask_geomodeling_amd.fuse import fuse_graph
    from dask_geomodeling_amd.hip import runtime
    from dask_geomodeling_amd.raster.sources import generate_window
    runtime.set_device(0)
    h = hashlib.sha1()
    rng = np.random.default_rng(3)
    table = [[k, (7 * k) % 1000] for k in range(0, 256, 2)]
    for shape in ((1, 64, 64), (1, 130, 272), (1, 1000, 1008), (1, 77, 33)):
        for tab, ndv in ((table, I32MAX), ([[1, 5], [2, 0], [700, 3], [-5, 9]], 3), (table, 4)):
            a = rng.integers(-10, 300, shape).astype("i4")
            a[rng.random(shape) < 0.02] = ndv
            src = raster.MemorySource(a, ndv, "EPSG:28992", 1.0, (0.0, float(shape[1])))
            for select in (False, True):
                view = raster.Mask(raster.Reclassify(src, tab, select), 1)
                out = view.get_data(mode="vals", bbox=(0.0, 0.0, float(shape[2]), float(shape[1])), width=shape[2], height=shape[1],
                                    projection="EPSG:28992")
                h.update(np.asarray(out["values"]).tobytes())
    side, tiles = 8192, 4
    request = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side, projection="EPSG:28992")
    graph, names = {}, []
    for i in range(tiles):
        d = generate_window((side, side), "i4", 2000 + i, 0.0, 256.0, I32MAX, (50, 50), 0, 0, side, side)
        src = raster.MemorySource(d, I32MAX, "EPSG:28992", 1.0, (0.0, float(side)))
        view = raster.Mask(raster.Reclassify(src, table), 1)
        graph, name = view.get_compute_graph(cached_compute_graph=graph, **request)
        names.append(name)
    fused = fuse_graph(graph, names)
    us = timed([lambda: get_sync(fused, names)], reps=10)
    byts = 5.0 * side * side * tiles
    print("RESULT %s %.1f us %.1f GB/s (%.3f of 8 TB/s)" % (h.hexdigest()[:12], us, byts / us / 1e3, byts / us / 1e3 / 8000), flush=True)


def sweep(kind, var, configs):
    for cfg in configs:
        env = dict(os.environ)
        env[var] = cfg
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child-" + kind], env=env, capture_output=True, text=True,
                           timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print("%s=%-10s %s" % (var, cfg, lines[0] if lines else "FAILED rc=%d %s" % (p.returncode, (p.stderr or "")[-300:])), flush=True)


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "child-bl2x":
        child_bl2x()
    elif what == "child-lookup32":
        child_lookup32()
    elif what == "bl2x":
        sweep("bl2x", "GM_BL2X", sys.argv[2:] or ["0,0,0", "1,1,1", "1,0,1", "1,1,0", "2,1,1", "3,1,1", "4,1,1", "5,1,1", "6,1,1", "7,1,1", "8,1,1", "5,0,1"])
    elif what == "lookup32":
        sweep("lookup32", "GM_LOOKUP32", sys.argv[2:] or ["0,0,0", "1,0,0", "2,0,0", "4,1,0", "2,1,0", "8,1,0", "1,0,4", "4,1,4", "4,0,4"])
