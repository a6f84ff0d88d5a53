"""Stencils and tables at parameters tools/kbench.py does not reach (GPU box): large MovingMax disks, wide Gaussians, many Dilate
values, long Reclassify tables, sparse keys, thousands of Classify edges; 4096^2 tiles, us per call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import stencil, tables
from dask_geomodeling_amd.raster.sources import generate_window
runtime.set_device(0)
F = float(np.finfo("f4").max)
I = 2147483647
side = 4096
f4 = generate_window((side, side), "f4", 90, 0.0, 100.0, F, (97, 89), 0, 0, side, side)
u1 = generate_window((side, side), "u1", 91, 0.0, 250.0, 255, (97, 89), 0, 0, side, side)
i4 = generate_window((side, side), "i4", 92, 0.0, 100000.0, I, (97, 89), 0, 0, side, side)


def t(label, f, n=3):
    try:
        f(); runtime.sync()
        e0, e1 = runtime.Event(), runtime.Event()
        e0.record()
        for _ in range(n):
            f()
        e1.record(); e1.synchronize()
        print("%-60s %10.1f us" % (label, e0.elapsed_ms(e1) / n * 1e3), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-60s %s: %s" % (label, type(e).__name__, str(e)[:80]), flush=True)


for size in (15, 31, 63):
    t("moving max f32, size %d" % size, lambda: stencil.moving_max(f4, np.float32(F), size))
for sigma in (2.0, 6.0, 20.0, 50.0):
    t("smooth f32, sigma %g" % sigma, lambda: stencil.gaussian_smooth(f4, np.float32(F), 0.0, (sigma, sigma)))
for nv in (2, 16, 17, 100):
    t("dilate u8, %d values" % nv, lambda: stencil.dilate(u1, list(range(1, nv + 1))))
for nk in (10, 1000, 50000):
    table = [[int(k), int((7 * k) % 1000)] for k in range(0, 2 * nk, 2)]
    kw = {"dtype": "int64", "fillvalue": 9223372036854775807, "data": table, "select": False}
    t("reclassify i32 -> i64, %d dense keys" % nk, lambda: tables.reclassify_process({"values": i4, "no_data_value": I}, kw))
    sparse = [[int(k) * 20011, int(k)] for k in range(nk)]
    kw2 = dict(kw, data=sparse)
    t("reclassify i32 -> i64, %d keys 20011 apart" % nk, lambda: tables.reclassify_process({"values": i4, "no_data_value": I}, kw2))
for ne in (5, 200, 5000, 60000):
    edges = list(np.linspace(0, 100, ne))
    t("classify f32, %d edges" % ne, lambda: tables.classify_process({"values": f4, "no_data_value": np.float32(F)}, edges, False))
edges = sorted(set(np.concatenate([np.linspace(0, 100, 20), 50.0 + np.arange(400) * 1e-5]).tolist()))
t("classify f32, %d clustered edges" % len(edges), lambda: tables.classify_process({"values": f4, "no_data_value": np.float32(F)}, edges, False))
