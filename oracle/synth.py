"""CPU ORACLE (test infrastructure) -- NumPy restatement of the device-side procedural
tile generator ``gm_generate`` (dask_geomodeling_amd/csrc/gm_util.hip), used to build
host copies of synthetic tiles for parity tests and for the CPU baseline of bench.py."""
import numpy as np

_M1, _M2, _GOLD = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB), np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = (np.asarray(x, dtype=np.uint64) + _GOLD)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def generate_window(shape, dtype, seed, lo, hi, ndv, lattice, row0, col0, height, width):
    """(1, height, width) window of the procedural (H, W) mosaic (see gmhip.h: gm_generate)."""
    H, W = shape
    dtype = np.dtype(dtype)
    gr = (row0 + np.arange(height, dtype=np.int64))[:, None]
    gc = (col0 + np.arange(width, dtype=np.int64))[None, :]
    key = splitmix64(np.uint64(seed))
    with np.errstate(over="ignore"):
        idx = gr.astype(np.uint64) * np.uint64(W) + gc.astype(np.uint64)
    bits = splitmix64(key ^ idx)
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    with np.errstate(all="ignore"):
        vals = (lo + u * (hi - lo)).astype(dtype)
    ly, lx = lattice
    if lx:
        ly = ly or 1
        vals[((gr % ly) == 0) & ((gc % lx) == 0)] = ndv
    inside = (gr >= 0) & (gr < H) & (gc >= 0) & (gc < W)
    vals[~np.broadcast_to(inside, vals.shape)] = ndv
    return vals[np.newaxis]
