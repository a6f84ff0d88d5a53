#!/usr/bin/env python3
"""Bilinear up-sampling at a range of zooms (GPU box): the general kernel of csrc/gm_resample.hip with its source window staged in
LDS or read through the caches (GM_BL_STAGED), in tiles of 32 or 64 destination rows (GM_BL_H).
usage: bilinear_zoom_sweep.py            -- runs itself once per configuration and prints a table
       bilinear_zoom_sweep.py one        -- one configuration (the environment's)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ZOOMS = [(5, 4), (3, 2), (7, 4), (2001, 1000), (5, 2), (3, 1), (4, 1), (6, 1)]
OUT = 12288


def one():
    import numpy as np
    from dask_geomodeling_amd.hip import runtime
    from dask_geomodeling_amd.raster import resample
    from dask_geomodeling_amd.raster.sources import generate_window
    runtime.set_device(0)
    fmax = float(np.finfo("f4").max)
    for num, den in ZOOMS:
        side = OUT * den // num // 4 * 4
        out = side * num // den
        src = [generate_window((side, side), "f4", 70 + i, 0.0, 100.0, fmax, (97, 89), 0, 0, side, side) for i in range(2)]
        calls = [lambda t=t: resample.bilinear(t, out, out, np.float32(fmax)) for t in src]
        for f in calls:
            f()
        runtime.sync()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record()
        for _ in range(3):
            for f in calls:
                f()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_ms(e1) / 6 * 1e3
        nbytes = 4.0 * out * out + 4.0 * side * side
        print("%.3f %d %d %.1f %.3f" % (num / den, side, out, us, nbytes / us / 1e3 / 8000), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
        sys.exit(0)
    table = {}
    configs = [("default", {}), ("staged 32", {"GM_BL_STAGED": "1", "GM_BL_H": "32"}), ("staged 64", {"GM_BL_STAGED": "1", "GM_BL_H": "64"}),
               ("direct 32", {"GM_BL_STAGED": "0", "GM_BL_H": "32"}), ("direct 64", {"GM_BL_STAGED": "0", "GM_BL_H": "64"})]
    for name, env in configs:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], capture_output=True, text=True, env=dict(os.environ, **env))
        for line in res.stdout.splitlines():
            z, side, out, us, frac = line.split()
            table.setdefault((z, side, out), {})[name] = (float(us), float(frac))
        if res.returncode:
            print(name, "failed:", res.stderr[-300:])
    print("%-8s %-14s " % ("zoom", "side -> out") + " ".join("%-18s" % n for n, _ in configs))
    for (z, side, out), row in table.items():
        print("%-8s %-14s " % (z, "%s -> %s" % (side, out)) + " ".join("%7.1f us (%.2f) " % row.get(n, (0, 0)) for n, _ in configs))
