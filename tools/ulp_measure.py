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
# power, float32: wide sweeps against the extended-precision value rounded once, and the special values of C99 F.9.4.4
FILL = np.float32(np.finfo("f4").max)


def power_on_device(x, y):
    fill = float(FILL)
    kw = {"dtype": "float32", "fillvalue": fill}
    shape = (1, 1, x.size)
    out = devops.math_process("power", kw, {"values": x.reshape(shape), "no_data_value": fill},
                              {"values": y.reshape(shape), "no_data_value": fill})
    return np.asarray(out["values"]).reshape(-1)


n = 1 << 20
sweeps = {
    "x in 1e-10..1e10, y in -8..8": (10.0 ** rng.uniform(-10, 10, n), rng.uniform(-8, 8, n)),
    "x in 0.5..2, y in -120..120": (rng.uniform(0.5, 2, n), rng.uniform(-120, 120, n)),
    "x within 1e-4 of 1, y up to 1e6": (1 + rng.uniform(-1e-4, 1e-4, n), rng.uniform(-1e6, 1e6, n)),
    "x within 64 ulp of 1, y up to 1e9": (np.float32(1) + rng.integers(-64, 65, n).astype("f4") * np.float32(2 ** -24), rng.uniform(-1e9, 1e9, n)),
    "x subnormal, y in -1..1": (rng.uniform(1e-45, 1e-38, n), rng.uniform(-1, 1, n)),
    "x < 0, y integer": (-(10.0 ** rng.uniform(-3, 3, n)), rng.integers(-20, 21, n).astype("f8")),
    "results around the subnormal range": (rng.uniform(1e-3, 1e-1, n), rng.uniform(15, 45, n)),
    "x in 0..100, y = 2.5": (rng.uniform(0, 100, n), np.full(n, 2.5)),
}
for name, (x, y) in sweeps.items():
    x, y = x.astype("f4"), y.astype("f4")
    with np.errstate(all="ignore"):
        true = np.power(x.astype(np.longdouble), y.astype(np.longdouble)).astype("f4")
        host = np.power(x, y)
    got = power_on_device(x, y)
    fin = np.isfinite(true) & (got != FILL)
    same_class = np.array_equal(~np.isfinite(true), got == FILL)  # (the wrapper turns every non-finite result into the fill value)
    d = ulp_distance(got[fin], true[fin])
    print("power sweep [%s]: device vs true max %d ulp, %.4f %% not correctly rounded | host NumPy vs true max %d | device vs host max %d | "
          "non-finite results where the true value is non-finite, only there: %s" % (name, int(d.max()), 100.0 * float((d > 0).mean()), int(ulp_distance(host[fin], true[fin]).max()),
                                      int(ulp_distance(got[fin], host[fin]).max()), same_class), flush=True)
special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 2.5, -2.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.0e38, -3.0e38,
                    16777216.0, 16777217.0, -16777216.0, 33554432.0, 1e10, -1e10, 0.99999994, 1.0000001, 7.0, -7.0, 1e-20, 1e20], dtype="f4")
xs, ys = [a.reshape(-1) for a in np.meshgrid(special, special)]
with np.errstate(all="ignore"):
    host = np.power(xs, ys)
got = power_on_device(xs, ys)
finite = np.isfinite(host)
bad = np.where(finite, ~((got != FILL) & (np.signbit(host) == np.signbit(got)) & (ulp_distance(got, np.where(finite, host, 0).astype("f4")) <= 1)),
               got != FILL)
print("power specials: %d pairs (zeros, ones, infinities, NaN, subnormals, odd / even / huge integers ...), %d differ from host NumPy "
      "(value beyond 1 ulp, sign of a zero, finite vs non-finite)" % (xs.size, int(bad.sum())), flush=True)
for k in np.flatnonzero(bad)[:20]:
    print("   pow(%r, %r): device %r host %r" % (float(xs[k]), float(ys[k]), float(got[k]), float(host[k])))
print("WORST (op, dtype): [vs golden, vs true]", worst)
