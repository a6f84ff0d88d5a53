"""Random shapes / dtypes through the stencil kernels against the oracle (GPU box): python tests/stencil_stress.py [n].
Lives under tests/ because it checks against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.raster import stencil
from oracle import ref_numpy as O
runtime.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)  # [n] [seed]
bad = 0
for it in range(N):
    dt = ["u1", "i1", "i2", "u2", "i4", "f4", "f8"][int(rng.integers(7))]
    bands = int(rng.integers(1, 4))
    size = [3, 3, 5, 7, 9, 11, 13, 15, 17][int(rng.integers(9))]
    h, w = int(rng.integers(size + 1, 80)), int(rng.integers(size + 1, 700))
    if it % 7 == 0:  # now and then a tile that spans several workgroup windows both ways
        h, w = int(rng.integers(100, 220)), int(rng.integers(1000, 4300))
    if np.dtype(dt).kind == "f":
        v = (rng.random((bands, h, w)) * 20 - 5).astype(dt); ndv = np.dtype(dt).type(-9999.0)
    else:
        v = rng.integers(0, 9, size=(bands, h, w)).astype(dt); ndv = np.dtype(dt).type(7)
    if np.dtype(dt).kind == "f" and it % 4 == 1:
        # zeros of both signs under a negative neighbourhood: the sign of a zero maximum is SciPy's scan's
        v = -np.abs(v) - 1
        pick = rng.random(v.shape)
        v[pick < 0.1] = -0.0
        v[pick > 0.9] = 0.0
    v[rng.random(v.shape) < 0.2] = ndv
    if np.dtype(dt).kind == "f" and it % 5 == 0:
        v[rng.random(v.shape) < 0.02] = np.nan
    try:
        want = O.movingmax_process({"values": v, "no_data_value": ndv}, size)["values"]
        got = stencil.moving_max(v, ndv, size).to_numpy()
        assert np.array_equal(got, want, equal_nan=True), "movingmax"
        assert got.dtype.kind != "f" or np.array_equal(np.signbit(got), np.signbit(want)), "movingmax: the sign of a zero"
        vals = [int(x) for x in rng.integers(0, 9, size=int(rng.integers(1, 5)))]
        want = O.dilate_process({"values": v, "no_data_value": ndv}, vals)["values"]
        got = stencil.dilate(v, vals).to_numpy()
        assert np.array_equal(got, want, equal_nan=True), "dilate"
        kw = dict(resolution=(float(rng.choice([0.5, 1.0, 2.0, 3.0])), float(rng.choice([0.5, 1.0, 0.7]))), altitude=45.0, azimuth=315.0, fill=int(rng.integers(0, 3)))
        with np.errstate(all="ignore"):
            want = O.hillshade_process({"values": v, "no_data_value": ndv}, kw, trig="rounded")["values"]
        got = stencil.hillshade(v, ndv, kw["fill"], kw["resolution"], kw["altitude"], kw["azimuth"]).to_numpy()
        assert np.array_equal(got, want), "hillshade: %d cells differ" % (got != want).sum()
        if np.dtype(dt).kind == "f" or it % 3 == 0:
            sigma = (float(rng.choice([0.4, 1.0, 1.7, 2.0, 3.3, 4.0])),) * 2 if it % 2 else (float(rng.random() * 3), float(rng.random() * (30 if it % 4 == 0 else 3)))  # (now and then a radius of up to 120)
            crop = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
            fill = float(rng.choice([0, 0.1, 2.5]))
            clean = np.where(np.isnan(v), 0, v).astype(dt) if np.dtype(dt).kind == "f" else v
            work = clean.copy()
            work[work == ndv] = fill
            for axis, sg in ((1, sigma[0]), (2, sigma[1])):
                if sg > 1e-15:
                    work = O.gaussian_axis(work, sg, axis, fill)
            want = work[:, crop[0]:work.shape[1] - crop[0], crop[1]:work.shape[2] - crop[1]]
            got = stencil.gaussian_smooth(clean, ndv, fill, sigma, crop).to_numpy()
            assert np.array_equal(got, want), "smooth %s %s" % (sigma, crop)
    except AssertionError as e:
        bad += 1
        print("FAIL", it, dt, (bands, h, w), size, e)
print("done", N, "failures", bad)
print("guard violations", runtime.guard_violations(), "guard mode", runtime.lib().gm_guard_enabled())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import print_routes  # noqa: E402
print_routes("stencils")
