"""Random views for differential testing: a seeded generator of Block trees over the element-wise /
table / reduction Blocks on the path, and an evaluator that walks the same tree with the CPU
oracle (oracle/ref_numpy.py), one reference ``process`` body per Block -- i.e. what the reference
would compute op by op, against which the fused / cut / batched device evaluation is compared."""
import os

import numpy as np

from blocks_util import ArrayLeaf
from dask_geomodeling_amd import raster
from oracle import ref_numpy as O

BINARY = {"Add": "add", "Subtract": "subtract", "Multiply": "multiply", "Divide": "divide", "Equal": "equal",
          "NotEqual": "not_equal", "Greater": "greater", "GreaterEqual": "greater_equal", "Less": "less",
          "LessEqual": "less_equal", "And": "logical_and", "Or": "logical_or", "Xor": "logical_xor"}
LEAF_KINDS = [("f4", np.float32(np.finfo("f4").max)), ("f4", -9999.0), ("u1", np.uint8(255)), ("i2", np.int16(-1)),
              ("i4", -9999), ("i4", 2147483647), ("f8", float(np.finfo("f8").max))]
# (GM_FUZZ_SHAPE=HxW: the campaigns of tests/fuzz_more.py on rasters of another shape -- odd ones found what 24 x 40 never met)
SHAPE = (1,) + tuple(int(x) for x in os.environ.get("GM_FUZZ_SHAPE", "24x40").split("x"))
REQ = dict(mode="vals", bbox=(0.0, 0.0, float(SHAPE[2]), float(SHAPE[1])), width=SHAPE[2], height=SHAPE[1], projection="EPSG:28992")


def make_leaf(rng, tag, k, memory=False, kinds=None):
    """memory=True: a MemorySource (the real source Block; answers any window) instead of an ArrayLeaf."""
    kinds = kinds or LEAF_KINDS
    dt, ndv = kinds[int(rng.integers(len(kinds)))]
    if np.dtype(dt).kind == "f":
        v = (rng.random(SHAPE) * 40 - 10).astype(dt)
        v[0, 0, 0] = np.nan if rng.random() < 0.3 else 3.0
    else:
        v = rng.integers(0, 12, size=SHAPE).astype(dt)
    v[rng.random(SHAPE) < 0.15] = ndv
    if memory:
        if np.dtype(dt).kind == "f":
            v[~np.isfinite(v)] = ndv  # (a source turns non-finite cells into no data itself, sources.py:148)
        return raster.MemorySource(v, ndv, "EPSG:28992", 1.0, (0.0, float(SHAPE[1])))
    key = "fuzz/%s/%d" % (tag, k)
    ArrayLeaf.ARRAYS[key] = {"values": v, "no_data_value": ndv}
    return ArrayLeaf(key, np.dtype(dt).str, ndv)


def is_bool(b):
    return b.dtype == np.dtype("bool")


def is_int(b):
    return b.dtype.kind in "iu"


def wide_view(rng, tag, n):
    """A view over n leaves of one register class (float32, or int32 / int16), every leaf folded into the running view by a
    random Block -- arithmetic, FillNoData, Clip, Max --: ONE program of n inputs where n <= 16 (round 6; the random trees below
    never use more than three leaves), so that generic steps, fast micro-ops and no-data rules all meet inputs 8 .. 15."""
    kinds = [LEAF_KINDS[0], LEAF_KINDS[1]] if rng.random() < 0.7 else [LEAF_KINDS[3], LEAF_KINDS[4]]
    leaves = [make_leaf(rng, tag, k, kinds=kinds) for k in range(n)]
    view = leaves[0]
    for leaf in leaves[1:]:
        choice = rng.random()
        if choice < 0.35:
            op = ["Add", "Subtract", "Multiply"][int(rng.integers(3))]
            view = getattr(raster, op)(view, leaf) if rng.random() < 0.7 else getattr(raster, op)(leaf, view)
        elif choice < 0.6:
            view = raster.FillNoData(view, leaf) if rng.random() < 0.5 else raster.FillNoData(leaf, view)
        elif choice < 0.75:
            view = raster.Clip(view, leaf)
        elif choice < 0.9:
            view = raster.Max(view, leaf)
        else:
            view = raster.Clip(view, raster.Greater(leaf, [3, 5.5, 0][int(rng.integers(3))]))
    view.dtype, view.fillvalue
    return view


def random_view(rng, tag, depth=4, memory=False, smooth=False):
    """A random Block tree (and the leaves it uses).  ``smooth``: Smooth among the stencils (not for tiled == whole
    comparisons, see below)."""
    leaves = [make_leaf(rng, tag, k, memory) for k in range(int(rng.integers(1, 4)))]

    def numeric(d):
        if d == 0 or rng.random() < 0.2:
            return leaves[int(rng.integers(len(leaves)))]
        choice = rng.random()
        a = numeric(d - 1)
        if memory and rng.random() < 0.12:
            # stencils answer through request enlargement: only meaningful on real sources
            # (Smooth is left out of the tiled == whole comparison: the reference's own margin, round(size),
            #  is smaller than scipy's kernel radius, int(4 * size / 3 + 0.5), so its result depends on the
            #  request box by construction -- spatial.py:255-295.  The golden 'smooth' family covers it.)
            kind = int(rng.integers(4))
            if kind == 0 and a.dtype != np.dtype("bool"):
                return raster.MovingMax(a, [3, 5, 7][int(rng.integers(3))])
            if kind == 1 and a.dtype != np.dtype("bool"):
                return raster.Dilate(a, [int(x) for x in rng.integers(0, 12, size=int(rng.integers(1, 4)))])
            if kind == 3 and a.dtype != np.dtype("bool") and not (a.dtype.kind == "u" and a.dtype.itemsize == 8):
                return raster.HillShade(a, 45, 315, int(rng.integers(0, 3)))
            if kind == 2 and smooth and a.dtype != np.dtype("bool"):
                return raster.Smooth(a, [1.0, 2.5, 9.0][int(rng.integers(3))], int(rng.integers(0, 3)))
        if choice < 0.45:
            op = ["Add", "Subtract", "Multiply", "Divide"][int(rng.integers(4))]
            b = numeric(d - 1) if rng.random() < 0.5 else [2.5, 3, -1.5, 0.5, 7][int(rng.integers(5))]
            return getattr(raster, op)(a, b) if rng.random() < 0.8 or not isinstance(b, raster.RasterBlock) else getattr(raster, op)(b, a)
        if choice < 0.55:
            return raster.Clip(a, boolean(d - 1) if rng.random() < 0.6 else numeric(d - 1))
        if choice < 0.62:
            return raster.MaskBelow(a, [0.0, 3, 5.5][int(rng.integers(3))])
        if choice < 0.70:
            return raster.FillNoData(a, numeric(d - 1))
        if choice < 0.78:
            return raster.Max(a, numeric(d - 1))
        if choice < 0.86:
            return raster.Mask(a, [1, 0, 7, 300, 2.5][int(rng.integers(5))])
        if choice < 0.93:
            bins = sorted(set(np.round(rng.random(int(rng.integers(2, 7))) * 30 - 5, 1).tolist()))
            if len(bins) < 2:
                bins = [0.0, 10.0]
            return raster.Classify(a, bins, bool(rng.random() < 0.5))
        if is_int(a) or is_bool(a):
            table = [[int(k), int(rng.integers(0, 50))] for k in rng.choice(14, size=int(rng.integers(1, 8)), replace=False)]
            if rng.random() < 0.3:
                table = [[k, t + 0.5] for k, t in table]
            return raster.Reclassify(a, table, bool(rng.random() < 0.4))
        return raster.Step(a, 1, 5, [2.0, 4][int(rng.integers(2))], 3)

    def boolean(d):
        a = numeric(d)
        if is_bool(a):
            return a
        choice = rng.random()
        if choice < 0.6:
            op = ["Equal", "NotEqual", "Greater", "GreaterEqual", "Less", "LessEqual"][int(rng.integers(6))]
            return getattr(raster, op)(a, numeric(max(d - 1, 0)) if rng.random() < 0.4 else [3, 5.5, 0][int(rng.integers(3))])
        if choice < 0.8:
            return raster.IsData(a) if rng.random() < 0.5 else raster.IsNoData(a)
        return raster.Invert(raster.IsNoData(a))

    for _ in range(20):
        try:
            view = numeric(depth) if rng.random() < 0.85 else boolean(depth - 1)
            view.dtype, view.fillvalue  # attribute algebra must hold
            return view
        except (TypeError, ValueError):
            continue  # e.g. Reclassify of a float raster, logic op on non-bool: try another tree
    return leaves[0] + 1


def oracle_eval(block):
    """Evaluate the Block tree op by op with the reference restatement."""
    if isinstance(block, ArrayLeaf):
        return dict(ArrayLeaf.ARRAYS[block.args[0]])
    if isinstance(block, raster.MemorySource):
        # (the whole source on its own grid: what a request for its full extent is answered with)
        ndv = block.no_data_value
        return {"values": np.asarray(block.data), "no_data_value": ndv.item() if hasattr(ndv, "item") else ndv}
    if not isinstance(block, raster.RasterBlock):
        return block
    name = type(block).__name__
    args = [oracle_eval(a) if isinstance(a, raster.RasterBlock) else a for a in block.args]
    kw = {"dtype": block.dtype.name, "fillvalue": block.fillvalue}
    if name in BINARY:
        return O.math_process(BINARY[name], kw, *args)
    if name == "Clip":
        return O.clip_process(args[0], args[1])
    if name == "Mask":
        return O.mask_process(args[0], args[1])
    if name == "MaskBelow":
        return O.maskbelow_process(args[0], args[1])
    if name == "Step":
        return O.step_process(args[0], *args[1:5])
    if name == "FillNoData":
        return O.fillnodata_process(kw, *args)
    if name == "Max":
        return O.reduce_max_process(kw, *args)
    if name == "Classify":
        return O.classify_process(args[0], args[1], args[2])
    if name == "Reclassify":
        return O.reclassify_process(args[0], {"dtype": block.dtype.str, "fillvalue": block.fillvalue, "data": args[1], "select": args[2]})
    if name == "IsData":
        return O.isdata_process(args[0])
    if name == "IsNoData":
        return O.isnodata_process(args[0])
    if name == "Invert":
        return O.invert_process(args[0])
    raise NotImplementedError(name)


def describe(block, depth=0):
    if isinstance(block, np.ndarray):
        return "array<%s%s>" % (block.dtype, list(block.shape))
    if not isinstance(block, raster.RasterBlock):
        return repr(block)
    if isinstance(block, ArrayLeaf):
        return "%s<%s>" % (block.args[0].split("/")[-1], block.dtype)
    return "%s(%s)" % (type(block).__name__, ", ".join(describe(a, depth + 1) for a in block.args))


def contains(block, name):
    """Does the Block tree contain a Block of class ``name``?"""
    if not isinstance(block, raster.RasterBlock):
        return False
    return type(block).__name__ == name or any(contains(a, name) for a in block.args)


def unsigned_zero(result, view=None):
    """-0.0 -> +0.0: the sign of a zero that wins a Max tie is NumPy-build dependent (np.fmax).  With ``view``
    given the normalisation is applied only when the view contains a ``Max``; every other view is compared
    with its zeros signed."""
    if view is not None and not contains(view, "Max"):
        return result
    if isinstance(result, dict) and "values" in result and np.asarray(result["values"]).dtype.kind == "f":
        result = dict(result)
        result["values"] = np.asarray(result["values"]) + 0.0
    return result
