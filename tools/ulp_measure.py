#!/usr/bin/env python3
"""Measured distances (GPU box) of the device's Exp / Log / Log10 / Power to (a) the reference-generated golden
vectors (NumPy's own loops) and (b) the correctly rounded value (the function in extended precision, rounded once):
the numbers behind the budgets in tests/test_gpu_parity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_family, ulp_distance  # noqa: E402
from dask_geomodeling_amd.hip import runtime  # noqa: E402
from dask_geomodeling_amd.raster import devops  # noqa: E402

runtime.set_device(0)
FUNCS = {"exp": np.exp, "log": np.log, "log10": np.log10}
worst = {}
for fam in ("n1", "math"):
    for c in load_family(fam):
        op = c.get("op")
        if op not in ("exp", "log", "log10", "power") or "raises" in c:
            continue
        out = devops.math_process(op, c["kwargs"], *c["args"])
        got, exp = np.asarray(out["values"]), c["expect"]["values"]
        if got.dtype.kind != "f":
            continue
        fin = np.isfinite(got) & np.isfinite(exp) & (exp != c["expect"]["no_data_value"])
        d_gold = int(ulp_distance(got[fin], exp[fin]).max()) if fin.any() else 0
        d_true = -1
        if op in FUNCS:
            v = c["args"][0]["values"]
            with np.errstate(all="ignore"):
                true = FUNCS[op](v.astype(np.longdouble)).astype(got.dtype)
            fin2 = fin & np.isfinite(true)
            d_true = int(ulp_distance(got[fin2], true[fin2]).max()) if fin2.any() else 0
        key = (op, str(got.dtype))
        w = worst.setdefault(key, [0, 0])
        w[0], w[1] = max(w[0], d_gold), max(w[1], d_true)
        print(fam, c["id"], op, got.dtype, "vs golden", d_gold, "vs true", d_true, flush=True)
# a wider sweep than the golden vectors hold
rng = np.random.default_rng(0)
for dt in ("f4", "f8"):
    x = np.concatenate([rng.uniform(-80, 80, 200000), rng.uniform(0, 1e-3, 50000), 10.0 ** rng.uniform(-30, 30, 200000)]).astype(dt)
    x = x.reshape(1, 450, 1000)
    fill = float(np.finfo(dt).max)
    kw = {"dtype": np.dtype(dt).name, "fillvalue": fill}
    for op, f in FUNCS.items():
        with np.errstate(all="ignore"):
            true = f(x.astype(np.longdouble)).astype(dt)
            host = f(x)
        got = np.asarray(devops.math_process(op, kw, {"values": x, "no_data_value": fill})["values"])
        fin = np.isfinite(true) & (got != np.dtype(dt).type(fill))
        print("sweep", op, dt, "device vs true", int(ulp_distance(got[fin], true[fin]).max()),
              "| host NumPy vs true", int(ulp_distance(host[fin], true[fin]).max()),
              "| device vs host NumPy", int(ulp_distance(got[fin], host[fin]).max()), flush=True)
print("WORST (op, dtype): [vs golden, vs true]", worst)
