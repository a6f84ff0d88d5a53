#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own ``process()`` code.

Run in the build container only (``/root/reference`` is mounted there):

    python tests/golden/make_golden.py

It imports nens/dask-geomodeling from ``/root/reference`` (third-party GIS
dependencies stubbed, see ``_refimport.py``), feeds seeded inputs through the
reference's hot-path functions and stores *inputs + outputs* under
``tests/golden/*.json|*.npz``.  Only those vectors are committed; no reference
source is copied.  NumPy used for generation: see ``manifest.json``.

Families:
  math        A1-A4  wrap_math_process_func x {add,subtract,multiply,divide,power}
  n1          same wrapper for comparisons/logic/exp/log + Invert/IsData/IsNoData/
              FillNoData/MaskBelow/Step
  clip        A6     Clip.process
  mask        A7     Mask.process
  classify    A8     Classify.process
  reclassify  A9     Reclassify.process
  tiler       P1     RasterTiler.get_sources_and_requests / process
  views       full get_data() of small views built from reference Blocks on a
              probe source (graph core + dtype/fill algebra + chain semantics)
  smoothzoom  R1     ndimage.affine_transform(order=0) call of Smooth's zoom branch
  spatial     N3     Dilate / MovingMax / HillShade process() + expand_request_pixels
  reduce      N4     Max.process = reduce_rasters(stack, "max")
  reducestats N4     reduce_rasters(stack, statistic, no_data_value, dtype) for every statistic
  group       N4     Group: whole get_data() on temporal probe sources (by time / by bands)
  temporal    N4     TemporalAggregate / Cumulative: whole get_data() + the process() arguments
  round4      Snap / Shift / TemporalSum / Resample: whole get_data() + the process() arguments; Place.process
"""
import json
import os
import sys
from datetime import datetime

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import _refimport  # noqa: E402
from codec import Writer  # noqa: E402

dg = _refimport.import_reference()
from dask_geomodeling.raster import elemwise as R_el  # noqa: E402
from dask_geomodeling.raster import misc as R_misc  # noqa: E402
from dask_geomodeling.raster import parallelize as R_par  # noqa: E402
from dask_geomodeling.raster.base import RasterBlock  # noqa: E402
from dask_geomodeling import utils as R_utils  # noqa: E402

F32MAX = float(np.finfo("f4").max)
F64MAX = float(np.finfo("f8").max)


class Probe(RasterBlock):
    """Probe-only leaf: hands a fixed array to the reference graph machinery."""

    def __init__(self, name, dtype_str, ndv):
        super().__init__(name, dtype_str, ndv)

    # populated outside of args so that tokenising stays cheap
    ARRAYS = {}

    @property
    def dtype(self):
        return np.dtype(self.args[1])

    @property
    def fillvalue(self):
        return self.args[2]

    period = (datetime(1970, 1, 1), datetime(1970, 1, 1))
    timedelta = None
    temporal = False
    extent = None
    geometry = None
    projection = "EPSG:28992"
    geo_transform = None

    def get_sources_and_requests(self, **request):
        return [(self.args[0], None), (request.get("mode", "vals"), None)]

    @staticmethod
    def process(name, mode):
        if mode == "time":
            return {"time": [datetime(1970, 1, 1)]}
        if mode == "meta":
            return {"meta": [None]}
        values, ndv = Probe.ARRAYS[name]
        if values is None:
            return None
        out = {"values": values}
        if values.dtype != bool:
            out["no_data_value"] = ndv
        else:
            out["no_data_value"] = None
        return out


def dtype_max(dt):
    dt = np.dtype(dt)
    return float(np.finfo(dt).max) if dt.kind == "f" else int(np.iinfo(dt).max)


def make_tile(rng, dtype, shape, ndv, specials=True):
    """Seeded tile with injected no-data, non-finite and edge values."""
    dt = np.dtype(dtype)
    if dt == bool:
        return rng.random(shape) < 0.5
    if dt.kind == "f":
        v = (rng.random(shape) * 200 - 100).astype(dt)
        flat = v.reshape(-1)
        if specials:
            n = flat.size
            pos = rng.choice(n, size=min(n, 24), replace=False)
            inject = [np.nan, np.inf, -np.inf, 0.0, -0.0, np.finfo(dt).max / 2,
                      -np.finfo(dt).max / 2, np.finfo(dt).tiny / 4, np.finfo(dt).max,
                      1.0, -1.0, 2.0, 0.5, 1e-30 if dt == np.float32 else 1e-300,
                      np.nextafter(dt.type(ndv), dt.type(0)) if ndv is not None else 3.0,
                      7.0, 100.0, -100.0, 1e30 if dt == np.float32 else 1e300]
            for p, x in zip(pos, inject):
                flat[p] = x
    else:
        info = np.iinfo(dt)
        lo, hi = max(info.min, -50), min(info.max, 200)
        v = rng.integers(lo, hi, size=shape, endpoint=True).astype(dt)
        flat = v.reshape(-1)
        if specials:
            n = flat.size
            pos = rng.choice(n, size=min(n, 10), replace=False)
            inject = [0, 1, info.max, info.min, info.max - 1, 2, 3, 7, info.max // 2 + 1, 10]
            for p, x in zip(pos, inject):
                flat[p] = x
    if ndv is not None:
        mask = rng.random(shape) < 0.1
        v[mask] = ndv
    return v


SHAPES = [(1, 24, 40), (3, 9, 21)]


# --------------------------------------------------------------------------
# family: math  (reference: raster/elemwise.py:235-406)
# --------------------------------------------------------------------------
def gen_math(w):
    ops = {
        "add": R_el.Add, "subtract": R_el.Subtract, "multiply": R_el.Multiply,
        "divide": R_el.Divide, "power": R_el.Power,
    }
    # (kind_a, kind_b): kind = ("r", dtype, ndv_kind) raster | ("s", python scalar) | ("n", ndarray)
    f32n = np.float32(F32MAX)
    combos = [
        (("r", "f4", f32n), ("r", "f4", F32MAX)),
        (("r", "f4", f32n), ("s", 2.5)),
        (("s", 2.5), ("r", "f4", F32MAX)),
        (("r", "f4", F32MAX), ("s", 3)),
        (("r", "f4", F32MAX), ("s", 0)),
        (("r", "f8", F64MAX), ("r", "f8", np.float64(F64MAX))),
        (("r", "f8", F64MAX), ("r", "f4", f32n)),
        (("r", "i4", 2147483647), ("r", "i4", np.int32(2147483647))),
        (("r", "i4", 2147483647), ("s", 3)),
        (("r", "i4", 2147483647), ("s", 2.5)),
        (("r", "i4", 2147483647), ("r", "f4", f32n)),
        (("r", "u1", np.uint8(255)), ("r", "u1", np.uint8(255))),
        (("r", "u1", np.uint8(255)), ("s", 5)),
        (("r", "u1", np.uint8(255)), ("s", 0.5)),
        (("r", "u1", np.uint8(255)), ("r", "f4", f32n)),
        (("r", "i2", np.int16(-32768)), ("r", "u1", np.uint8(255))),
        (("r", "i8", 9223372036854775807), ("r", "i4", 2147483647)),
        (("r", "i8", 9223372036854775807), ("s", 2)),
        (("r", "?", None), ("s", 2)),
        (("r", "?", None), ("r", "u1", np.uint8(255))),
        (("r", "f4", f32n), ("n", np.array([1.5], dtype="f4"))),
        (("r", "u2", np.uint16(65535)), ("r", "i2", np.int16(32767))),
        (("r", "f4", np.float32(-9999.0)), ("r", "f4", np.float32(-9999.0))),
    ]
    cid = 0
    for opname, cls in ops.items():
        for ci, (ka, kb) in enumerate(combos):
            for si, shape in enumerate(SHAPES):
                rng = np.random.default_rng(1000 * ci + 10 * si + 1)
                args, blocks = [], []
                for slot, k in enumerate((ka, kb)):
                    if k[0] == "r":
                        arr = make_tile(rng, k[1], shape, k[2])
                        if opname == "power" and np.dtype(k[1]).kind in "iu" and slot == 1:
                            # integer ** negative integer raises in NumPy: keep exponents >= 0
                            arr = np.abs(arr.astype("i8")).astype(k[1]) % 5
                        d = {"values": arr, "no_data_value": k[2]}
                        args.append(d)
                        blocks.append(Probe("p%d" % slot, np.dtype(k[1]).str, k[2]))
                    elif k[0] == "s":
                        args.append(k[1])
                        blocks.append(k[1])
                    else:
                        args.append(k[1])
                        blocks.append(k[1])
                try:
                    block = cls(*blocks)
                    pk = {"dtype": block.dtype.name, "fillvalue": block.fillvalue}
                except Exception as e:  # e.g. unsupported type
                    print("skip", opname, ci, e)
                    continue
                try:
                    out = cls.process(pk, *args)
                except Exception as e:
                    w.add(id="math%04d" % cid, op=opname, kwargs=pk, args=args,
                          raises=type(e).__name__)
                    cid += 1
                    continue
                w.add(id="math%04d" % cid, op=opname, kwargs=pk, args=args, expect=out)
                cid += 1
    # None / time / meta passthrough (elemwise.py:254-262)
    pk = {"dtype": "float32", "fillvalue": F32MAX}
    tile = {"values": np.ones((1, 2, 2), "f4"), "no_data_value": F32MAX}
    for args in ([tile, None], [None, tile], [{"time": [datetime(2000, 1, 1).isoformat()]}, tile],
                 [tile, {"meta": ["x"]}], []):
        w.add(id="math%04d" % cid, op="add", kwargs=pk, args=args,
              expect=R_el.Add.process(pk, *args))
        cid += 1


# --------------------------------------------------------------------------
# family: n1  (same wrapper + the stand-alone element-wise blocks)
# --------------------------------------------------------------------------
def gen_n1(w):
    cid = 0
    f32n = np.float32(F32MAX)
    cmp_ops = {"equal": R_el.Equal, "not_equal": R_el.NotEqual, "greater": R_el.Greater,
               "greater_equal": R_el.GreaterEqual, "less": R_el.Less,
               "less_equal": R_el.LessEqual}
    combos = [
        (("r", "f4", f32n), ("r", "f4", F32MAX)),
        (("r", "f4", f32n), ("s", 7.0)),
        (("s", 7), ("r", "u1", np.uint8(255))),
        (("r", "i4", 2147483647), ("r", "u1", np.uint8(255))),
        (("r", "u1", np.uint8(255)), ("s", 7)),
    ]
    for opname, cls in cmp_ops.items():
        for ci, (ka, kb) in enumerate(combos):
            rng = np.random.default_rng(77 + ci)
            args = []
            for k in (ka, kb):
                if k[0] == "r":
                    arr = make_tile(rng, k[1], SHAPES[1], k[2])
                    if np.dtype(k[1]).kind in "iu":
                        arr[arr != k[2]] = arr[arr != k[2]] % 12
                    else:
                        sel = rng.random(arr.shape) < 0.3
                        arr[sel & (arr != k[2])] = 7.0
                    args.append({"values": arr, "no_data_value": k[2]})
                else:
                    args.append(k[1])
            pk = {"dtype": "bool", "fillvalue": None}
            w.add(id="n1%04d" % cid, op=opname, kwargs=pk, args=args,
                  expect=cls.process(pk, *args))
            cid += 1
    logic = {"logical_and": R_el.And, "logical_or": R_el.Or, "logical_xor": R_el.Xor}
    for opname, cls in logic.items():
        rng = np.random.default_rng(5)
        a = {"values": make_tile(rng, "?", SHAPES[1], None), "no_data_value": None}
        b = {"values": make_tile(rng, "?", SHAPES[1], None), "no_data_value": None}
        pk = {"dtype": "bool", "fillvalue": None}
        for args in ([a, b], [a, True], [False, b]):
            w.add(id="n1%04d" % cid, op=opname, kwargs=pk, args=args,
                  expect=cls.process(pk, *args))
            cid += 1
    unary = {"exp": R_el.Exp, "log": R_el.Log, "log10": R_el.Log10}
    for opname, cls in unary.items():
        for dt, ndv in (("f4", f32n), ("f8", F64MAX), ("u1", np.uint8(255)), ("i4", 2147483647)):
            rng = np.random.default_rng(9)
            arr = make_tile(rng, dt, SHAPES[0], ndv)
            a = {"values": arr, "no_data_value": ndv}
            blk = cls(Probe("p", np.dtype(dt).str, ndv))
            pk = {"dtype": blk.dtype.name, "fillvalue": blk.fillvalue}
            w.add(id="n1%04d" % cid, op=opname, kwargs=pk, args=[a],
                  expect=cls.process(pk, a))
            cid += 1
    # stand-alone blocks
    rng = np.random.default_rng(11)
    b = {"values": make_tile(rng, "?", SHAPES[1], None), "no_data_value": None}
    w.add(id="n1%04d" % cid, op="invert", args=[b], expect=R_el.Invert.process(b)); cid += 1
    for dt, ndv in (("f4", f32n), ("u1", np.uint8(255)), ("i4", 2147483647)):
        a = {"values": make_tile(rng, dt, SHAPES[1], ndv), "no_data_value": ndv}
        w.add(id="n1%04d" % cid, op="isdata", args=[a], expect=R_el.IsData.process(a)); cid += 1
        w.add(id="n1%04d" % cid, op="isnodata", args=[a], expect=R_el.IsNoData.process(a)); cid += 1
        for value in (0, 7, 2.5):
            if isinstance(value, float) and np.dtype(dt).kind != "f":
                continue
            w.add(id="n1%04d" % cid, op="maskbelow", args=[a, value],
                  expect=R_misc.MaskBelow.process(a, value)); cid += 1
        for (left, right, loc, at) in ((0, 1, 0, 0.5), (2, 5, 7, 3), (-1.5, 1.5, 2.0, 0.0)):
            if np.dtype(dt).kind != "f" and any(isinstance(x, float) and x != int(x) for x in (left, right, at)):
                # NumPy raises / truncates on float into int arrays; keep integer cases integral
                continue
            try:
                out = R_misc.Step.process(a, left, right, loc, at)
            except Exception as e:
                print("skip step", dt, e)
                continue
            w.add(id="n1%04d" % cid, op="step", args=[a, left, right, loc, at], expect=out); cid += 1
    # FillNoData (elemwise.py:724-757)
    for dts in (("f4", "f4"), ("u1", "u1", "u1"), ("u1", "f4"), ("i4", "i4")):
        tiles, blocks = [], []
        for i, dt in enumerate(dts):
            ndv = {"f4": f32n, "u1": np.uint8(255), "i4": 2147483647}[dt]
            arr = make_tile(rng, dt, SHAPES[1], ndv)
            arr[rng.random(arr.shape) < 0.4] = ndv
            tiles.append({"values": arr, "no_data_value": ndv})
            blocks.append(Probe("p%d" % i, np.dtype(dt).str, ndv))
        blk = R_el.FillNoData(*blocks)
        pk = {"dtype": blk.dtype.name, "fillvalue": blk.fillvalue}
        w.add(id="n1%04d" % cid, op="fillnodata", kwargs=pk, args=tiles,
              expect=R_el.FillNoData.process(pk, *tiles)); cid += 1
        w.add(id="n1%04d" % cid, op="fillnodata", kwargs=pk, args=[None] + tiles[1:],
              expect=R_el.FillNoData.process(pk, None, *tiles[1:])); cid += 1


# --------------------------------------------------------------------------
# family: clip  (reference: raster/misc.py:98-123)
# --------------------------------------------------------------------------
def gen_clip(w):
    cid = 0
    f32n = np.float32(F32MAX)
    rng = np.random.default_rng(21)
    for sdt, sndv in (("f4", f32n), ("f4", F32MAX), ("u1", np.uint8(255)), ("i4", 2147483647),
                      ("f8", F64MAX), ("i8", 9223372036854775807)):
        for mdt, mndv in (("?", None), ("f4", f32n), ("u1", np.uint8(255)), ("i4", -1)):
            for shape in SHAPES:
                store = {"values": make_tile(rng, sdt, shape, sndv), "no_data_value": sndv}
                src = {"values": make_tile(rng, mdt, shape, mndv), "no_data_value": mndv}
                w.add(id="clip%04d" % cid, op="clip", args=[store, src],
                      expect=R_misc.Clip.process(store, src)); cid += 1
    # early outs
    tile = {"values": make_tile(rng, "f4", SHAPES[0], f32n), "no_data_value": f32n}
    empty = {"values": np.full(SHAPES[0], F32MAX, "f4"), "no_data_value": F32MAX}
    msk = {"values": make_tile(rng, "?", SHAPES[0], None), "no_data_value": None}
    for args in ([None, msk], [tile, None], [empty, None], [empty, msk],
                 [{"time": ["t"]}, msk], [{"meta": ["m"]}, None]):
        w.add(id="clip%04d" % cid, op="clip", args=args,
              expect=R_misc.Clip.process(*args)); cid += 1


# --------------------------------------------------------------------------
# family: mask  (reference: raster/misc.py:208-222, utils.py:61-64)
# --------------------------------------------------------------------------
def gen_mask(w):
    cid = 0
    rng = np.random.default_rng(31)
    f32n = np.float32(F32MAX)
    stores = [("u1", np.uint8(255)), ("i4", 2147483647), ("i8", 9223372036854775807),
              ("f4", f32n), ("f4", F32MAX), ("f8", F64MAX), ("f4", np.float32(-9999.0)),
              ("f4", -9999.0), ("f8", np.float64(1e-9)), ("i2", np.int16(-1))]
    for sdt, sndv in stores:
        arr = make_tile(rng, sdt, SHAPES[0], sndv)
        if np.dtype(sdt).kind == "f":
            # values just inside / outside the isclose() band around the ndv
            flat = arr.reshape(-1)
            nd = np.dtype(sdt).type(sndv)
            band = [nd * (1 - 0.5e-5), nd * (1 - 2e-5), nd * (1 + 0.5e-5), nd * (1 + 2e-5),
                    nd * (1 - 1.0e-5), nd - np.dtype(sdt).type(5e-9), nd + np.dtype(sdt).type(2e-8)]
            with np.errstate(all="ignore"):
                for i, x in enumerate(band):
                    flat[40 + i] = x
        data = {"values": arr, "no_data_value": sndv}
        for value in (0, 1, 8, 255, 256, 1000, -1000, 3.14159, 0.0, -40000, 70000, 65536, -129):
            try:
                out = R_misc.Mask.process(data, value)
            except Exception as e:  # value == 256: uint8 chosen, NumPy 2 refuses the store
                w.add(id="mask%04d" % cid, op="mask", args=[data, value],
                      raises=type(e).__name__); cid += 1
                continue
            w.add(id="mask%04d" % cid, op="mask", args=[data, value], expect=out); cid += 1
    for args in ([None, 1], [{"time": ["t"]}, 1]):
        w.add(id="mask%04d" % cid, op="mask", args=args, expect=R_misc.Mask.process(*args)); cid += 1


# --------------------------------------------------------------------------
# family: classify  (reference: raster/misc.py:387-399)
# --------------------------------------------------------------------------
def gen_classify(w):
    cid = 0
    rng = np.random.default_rng(41)
    f32n = np.float32(F32MAX)
    bins_sets = [
        [3, 8], [0.5], [-10.5, 0.0, 7.0, 7.5, 100.0], [1, 2, 3, 4, 5, 6, 7],
        list(np.linspace(-100, 100, 253)), list(np.linspace(-100, 100, 254)),
        list(np.linspace(-100, 100, 255)), list(range(-50, 200, 2)),
        [0.1, 0.2, 0.30000000000000004], [-1e30, 1e30],
    ]
    stores = [("f4", f32n), ("f8", F64MAX), ("u1", np.uint8(255)), ("i4", 2147483647),
              ("i8", 9223372036854775807), ("i2", np.int16(-32768))]
    for sdt, sndv in stores:
        for bi, bins in enumerate(bins_sets):
            for right in (False, True):
                arr = make_tile(rng, sdt, SHAPES[bi % 2], sndv)
                flat = arr.reshape(-1)
                # hit bin edges exactly
                edges = np.asarray(bins)
                with np.errstate(all="ignore"):
                    cast = edges.astype(arr.dtype)
                k = min(len(cast), 20)
                flat[100:100 + k] = cast[:k]
                data = {"values": arr, "no_data_value": sndv}
                bins_l = np.asarray(bins).tolist()
                w.add(id="cls%04d" % cid, op="classify", args=[data, bins_l, right],
                      expect=R_misc.Classify.process(data, bins_l, right)); cid += 1
    # the reference's own KAT: tests/test_raster.py:1447-1467
    arr = np.array([[[1, 5], [7, 10], [255, 255]]], dtype="u1")
    data = {"values": arr, "no_data_value": 255}
    for right in (False, True):
        w.add(id="cls%04d" % cid, op="classify", args=[data, [3, 8], right],
              expect=R_misc.Classify.process(data, [3, 8], right)); cid += 1
    w.add(id="cls%04d" % cid, op="classify", args=[None, [3, 8], False],
          expect=R_misc.Classify.process(None, [3, 8], False)); cid += 1


# --------------------------------------------------------------------------
# family: reclassify  (reference: raster/misc.py:421-515)
# --------------------------------------------------------------------------
def gen_reclassify(w):
    cid = 0
    rng = np.random.default_rng(51)
    tables = []
    tables.append([[7, 1000]])
    tables.append([[7, 8.2]])
    tables.append([[1, 10], [7, 70]])
    tables.append([[k, (7 * k) % 1000] for k in range(0, 126, 2)])           # 63
    tables.append([[k, (7 * k) % 1000] for k in range(0, 128, 2)])           # 64
    tables.append([[k, (7 * k) % 1000] for k in range(0, 130, 2)])           # 65
    tables.append([[int(k), int(v)] for k, v in zip(rng.permutation(1000) - 300,
                                                     rng.integers(-5000, 5000, 1000))])  # unsorted, negative
    tables.append([[k, k * 0.25 - 3.0] for k in range(-20, 180, 3)])          # float targets
    tables.append([[255, 5], [3, 9]])                                         # ndv (u1) is a key
    tables.append([[2147483647, 11], [-2147483648, -11], [0, 1]])             # i32 extremes
    tables.append([[k, (7 * k) % 1000] for k in range(0, 256, 2)])           # cfg-3 table (128)
    tables.append([[-1000000, 1], [5, 2], [1000000, 3], [99, 4]])            # sparse keys
    stores = [("u1", np.uint8(255)), ("i4", 2147483647), ("i4", np.int32(-1)),
              ("i8", 9223372036854775807), ("i2", np.int16(-32768)), ("?", None),
              ("u2", np.uint16(65535))]
    for ti, table in enumerate(tables):
        for sdt, sndv in stores:
            for select in (False, True):
                arr = make_tile(rng, sdt, SHAPES[ti % 2], sndv)
                data = {"values": arr, "no_data_value": sndv}
                blk = R_misc.Reclassify(Probe("p", np.dtype(sdt).str, sndv), table, select)
                pk = {"dtype": blk.dtype.str, "fillvalue": blk.fillvalue,
                      "data": blk.data, "select": blk.select}
                try:
                    out = R_misc.Reclassify.process(data, pk)
                except Exception as e:
                    w.add(id="rcl%04d" % cid, op="reclassify", args=[data, pk],
                          raises=type(e).__name__); cid += 1
                    continue
                w.add(id="rcl%04d" % cid, op="reclassify", args=[data, pk], expect=out); cid += 1
    # bool keys on a bool store (tests/test_raster_misc.py:212-218)
    arr = make_tile(rng, "?", SHAPES[1], None)
    data = {"values": arr, "no_data_value": None}
    blk = R_misc.Reclassify(Probe("p", "|b1", None), [[True, 1000]])
    pk = {"dtype": blk.dtype.str, "fillvalue": blk.fillvalue, "data": blk.data, "select": False}
    w.add(id="rcl%04d" % cid, op="reclassify", args=[data, pk],
          expect=R_misc.Reclassify.process(data, pk)); cid += 1
    w.add(id="rcl%04d" % cid, op="reclassify", args=[None, pk],
          expect=R_misc.Reclassify.process(None, pk)); cid += 1
    # construction-time errors (tests/test_raster_misc.py:232-248)
    errs = []
    for args in ((Probe("p", "<f4", 1.0), [[7.0, 1000]], False),
                 (Probe("p", "|u1", 255), [[7.4, 1000]], False),
                 (Probe("p", "|u1", 255), [[[7, 1000]], [1, 100]], False),
                 (Probe("p", "|u1", 255), [[7, 1], [7, 2]], False),
                 (Probe("p", "|u1", 255), [[7, 1]], 1)):
        try:
            R_misc.Reclassify(*args)
            errs.append(None)
        except Exception as e:
            errs.append(type(e).__name__)
    w.add(id="rcl_init_errors", op="reclassify_init_errors", expect=errs)


# --------------------------------------------------------------------------
# family: tiler  (reference: raster/parallelize.py:43-125)
# --------------------------------------------------------------------------
def gen_tiler(w):
    cid = 0
    src = Probe("p", "<f4", F32MAX)
    for bbox, width, height, tile in (
        ((0.0, 0.0, 14.0, 7.0), 14, 7, 7),
        ((10.0, -10.0, 20.0, 2.0), 10, 12, 7),
        ((3.0, 3.0, 15.0, 15.0), 6, 4, 24),
        ((0.0, 0.0, 64.0, 48.0), 128, 96, [32, 16]),
        ((5.0, 5.0, 5.0, 5.0), 1, 1, 4),
    ):
        blk = R_par.RasterTiler(src, tile)
        req = dict(mode="vals", bbox=bbox, width=width, height=height, projection="EPSG:28992")
        sr = blk.get_sources_and_requests(**req)
        out = []
        for s, r in sr:
            if isinstance(s, dict):
                s = dict(s)
                s["dtype"] = np.dtype(s["dtype"]).str
                out.append(["kwargs", s])
            elif s is None:
                out.append(["none", None])
            else:
                r = dict(r)
                r["bbox"] = tuple(float(x) for x in r["bbox"])
                out.append(["store", r])
        w.add(id="tiler%04d" % cid, op="tiler_requests", tile_size=tile, request=req, expect=out)
        cid += 1
    # process: assemble tiles incl. leftovers and None tiles (parallelize.py:93-125)
    rng = np.random.default_rng(61)
    for (H, W, ty, tx) in ((12, 10, 7, 7), (96, 128, 16, 32), (5, 9, 2, 4)):
        count_x, count_y = -(-W // tx), -(-H // ty)
        kw = {"dtype": np.dtype("f4"), "fillvalue": F32MAX, "shape_yx": (H, W),
              "count_xy": (count_x, count_y), "tilesize_xy": [tx, ty]}
        tiles = []
        for i in range(count_x):
            for j in range(count_y):
                w_ = min(tx, W - i * tx)
                h_ = min(ty, H - j * ty)
                if (i + j) % 5 == 4:
                    tiles.append(None)
                else:
                    tiles.append({"values": rng.random((2, h_, w_)).astype("f4"),
                                  "no_data_value": F32MAX})
        out = R_par.RasterTiler.process(kw, *tiles)
        kw_enc = dict(kw)
        kw_enc["dtype"] = kw["dtype"].str
        w.add(id="tiler%04d" % cid, op="tiler_process", kwargs=kw_enc, tiles=tiles, expect=out)
        cid += 1


# --------------------------------------------------------------------------
# family: views  (graph core + attribute algebra + chains through get_data)
# --------------------------------------------------------------------------
def build(expr, leaves):
    """expr: nested list ["Op", arg, ...]; str args name leaves; numbers are literals."""
    if isinstance(expr, str):
        return leaves[expr]
    if isinstance(expr, (int, float, bool)) or expr is None:
        return expr
    if isinstance(expr, dict):  # literal payloads (lists as args)
        return expr["lit"]
    head, rest = expr[0], [build(x, leaves) for x in expr[1:]]
    cls = getattr(R_el, head, None) or getattr(R_misc, head)
    return cls(*rest)


def gen_views(w):
    cid = 0
    f32n = np.float32(F32MAX)
    shape = (1, 32, 48)

    def leafset(rng, spec):
        arrays, blocks = {}, {}
        for name, (dt, ndv) in spec.items():
            arr = make_tile(rng, dt, shape, ndv)
            Probe.ARRAYS[name] = (arr, ndv)
            arrays[name] = {"values": arr, "no_data_value": ndv}
            blocks[name] = Probe(name, np.dtype(dt).str, ndv)
        return arrays, blocks

    views = [
        # canonical 5-op chain
        (["Clip", ["Add", ["Subtract", ["Multiply", ["Add", "A", "B"], 2.5], "B"], 1.5], "A"],
         {"A": ("f4", f32n), "B": ("f4", f32n)}),
        # BASELINE.md variant
        (["Clip", ["Add", ["Subtract", ["Multiply", ["Add", "A", "B"], 2.0], "B"], 1.5], "B"],
         {"A": ("f4", f32n), "B": ("f4", f32n)}),
        # F/2 + F/2 trap: finite result equal to fill becomes no-data downstream
        (["Multiply", ["Add", "A", "A"], 0.0], {"A": ("f4", f32n)}),
        (["Divide", ["Subtract", ["Multiply", ["Add", "A", "B"], 0.5], "C"], 4.0],
         {"A": ("f4", f32n), "B": ("f4", f32n), "C": ("f4", f32n)}),
        (["Multiply", ["Add", "A", "B"], ["Subtract", "A", "B"]],
         {"A": ("f4", f32n), "B": ("f4", f32n)}),
        (["Add", ["Multiply", "U", 3], "V"], {"U": ("u1", np.uint8(255)), "V": ("u1", np.uint8(255))}),
        (["Divide", "U", 10], {"U": ("u1", np.uint8(255))}),
        (["Multiply", ["Add", "I", 1000], 2.5], {"I": ("i4", 2147483647)}),
        (["Add", ["Multiply", "I", 65536], ["Multiply", "I", 65536]], {"I": ("i4", 2147483647)}),
        (["Power", ["Add", "A", 1.0], 2], {"A": ("f4", f32n)}),
        (["Power", "A", 0.5], {"A": ("f4", f32n)}),
        (["Power", "A", -1], {"A": ("f4", f32n)}),
        (["Power", "U", 2], {"U": ("u1", np.uint8(255))}),
        (["Clip", "A", ["Greater", "A", 0.0]], {"A": ("f4", f32n)}),
        (["Clip", ["Multiply", "U", 2], ["Equal", "U", 7]], {"U": ("u1", np.uint8(255))}),
        (["Mask", ["Add", "A", "B"], 1], {"A": ("f4", f32n), "B": ("f4", f32n)}),
        (["Mask", ["Reclassify", "I", {"lit": [[k, (7 * k) % 1000] for k in range(0, 256, 2)]}], 1],
         {"I": ("i4", 2147483647)}),
        (["Reclassify", ["Classify", ["Clip", ["Divide", ["Subtract", ["Multiply", ["Add", "A", "B"], 0.5], "C"], 2.0], "M"],
                         {"lit": [-50.0, -10.0, 0.0, 10.0, 50.0]}],
          {"lit": [[0, 10], [1, 11], [3, 13], [5, 15]]}],
         {"A": ("f4", f32n), "B": ("f4", f32n), "C": ("f4", f32n), "M": ("?", None)}),
        (["Mask", ["Reclassify", ["Classify", ["Clip", ["Divide", ["Subtract", ["Multiply", ["Add", "A", "B"], 0.5], "C"], 2.0], "M"],
                                  {"lit": [-50.0, -10.0, 0.0, 10.0, 50.0]}],
                   {"lit": [[0, 10], [1, 11], [3, 13], [5, 15]]}, True], 1],
         {"A": ("f4", f32n), "B": ("f4", f32n), "C": ("f4", f32n), "M": ("?", None)}),
        (["Add", ["Multiply", "D", 2.0], "A"], {"D": ("f8", F64MAX), "A": ("f4", f32n)}),
        (["Subtract", "L", ["Multiply", "I", 3]], {"L": ("i8", 9223372036854775807), "I": ("i4", 2147483647)}),
        (["And", ["Greater", "A", 0.0], ["Less", "B", 50.0]], {"A": ("f4", f32n), "B": ("f4", f32n)}),
        (["Invert", ["Or", ["IsNoData", "A"], ["LessEqual", "A", "B"]]], {"A": ("f4", f32n), "B": ("f4", f32n)}),
        (["FillNoData", "A", ["Multiply", "B", 2.0]], {"A": ("f4", f32n), "B": ("f4", f32n)}),
        (["Log", ["Exp", "A"]], {"A": ("f4", f32n)}),
        (["Step", ["MaskBelow", "A", 0.0], 0, 1, 50.0], {"A": ("f4", f32n)}),
    ]
    for vi, (expr, spec) in enumerate(views):
        rng = np.random.default_rng(500 + vi)
        arrays, blocks = leafset(rng, spec)
        view = build(expr, blocks)
        req = dict(mode="vals", bbox=(0.0, 0.0, 48.0, 32.0), width=48, height=32,
                   projection="EPSG:28992")
        out = view.get_data(**req)
        graph, name = view.get_compute_graph(**req)
        w.add(id="view%04d" % cid, op="view", expr=expr, leaves=arrays, request=req,
              expect=out, dtype=np.dtype(view.dtype).str, fillvalue=view.fillvalue,
              n_keys=len(graph), key_prefix=name.rsplit("_", 1)[0])
        cid += 1
    # None leaf propagation through a chain
    rng = np.random.default_rng(600)
    arrays, blocks = leafset(rng, {"A": ("f4", f32n), "B": ("f4", f32n)})
    Probe.ARRAYS["B"] = (None, None)
    for expr in (["Add", ["Multiply", "A", 2.0], "B"], ["Clip", "A", "B"], ["Clip", "B", "A"],
                 ["Mask", "B", 1]):
        view = build(expr, blocks)
        req = dict(mode="vals", bbox=(0.0, 0.0, 48.0, 32.0), width=48, height=32,
                   projection="EPSG:28992")
        out = view.get_data(**req)
        lv = dict(arrays)
        lv["B"] = None
        w.add(id="view%04d" % cid, op="view", expr=expr, leaves=lv, request=req, expect=out,
              dtype=np.dtype(view.dtype).str, fillvalue=view.fillvalue)
        cid += 1


# --------------------------------------------------------------------------
# family: dtype rules  (reference: elemwise.py:134-152, 377-380; misc.py:195-206, 380-385, 464-471)
# --------------------------------------------------------------------------
def gen_dtypes(w):
    rows = []
    dts = ["?", "u1", "i1", "u2", "i2", "u4", "i4", "u8", "i8", "f2", "f4", "f8"]
    others = [("s", 5), ("s", 300), ("s", -3), ("s", 2.5), ("s", 70000), ("s", 2**40)] + [("r", d) for d in dts]
    for cls_name in ("Add", "Subtract", "Multiply", "Divide", "Power"):
        cls = getattr(R_el, cls_name)
        for da in dts:
            for kind, ob in others:
                a = Probe("a", np.dtype(da).str, 0)
                b = Probe("b", np.dtype(ob).str, 0) if kind == "r" else ob
                for order in (0, 1):
                    try:
                        blk = cls(a, b) if order == 0 else cls(b, a)
                        rows.append([cls_name, da, [kind, ob], order, blk.dtype.str, blk.fillvalue])
                    except Exception as e:
                        rows.append([cls_name, da, [kind, ob], order, "ERR:" + type(e).__name__, None])
    w.add(id="dtype_math", op="dtype_math", expect=rows)
    rows = []
    for value in (0, 1, 8, 255, 256, 257, 1000, 65535, 65536, 65537, -1, -128, -129, -1000, -32768,
                  -32769, 3.14159, 0.0, 2**32, 2**32 + 1, -2**31, -2**31 - 1):
        blk = R_misc.Mask(Probe("a", "<f4", 0.0), value)
        rows.append([value, blk.dtype.str, blk.fillvalue])
    w.add(id="dtype_mask", op="dtype_mask", expect=rows)
    rows = []
    for n in (1, 2, 253, 254, 255, 256, 65533, 65534, 65535):
        bins = list(range(n))
        try:
            blk = R_misc.Classify(Probe("a", "<f4", 0.0), bins)
            rows.append([n, blk.dtype.str, blk.fillvalue])
        except Exception as e:  # single-edge bins are rejected by the reference (empty diff)
            rows.append([n, "ERR:" + type(e).__name__, None])
    w.add(id="dtype_classify", op="dtype_classify", expect=rows)
    rows = []
    for n in (-2**63, -2**31 - 1, -2**31, -32769, -32768, -129, -128, -1, 0, 1, 127, 128, 129, 255, 256,
              257, 32767, 32768, 65535, 65536, 65537, 2**31, 2**32, 2**32 + 1, 2**63):
        row = [n]
        for f in (R_utils.get_int_dtype, R_utils.get_uint_dtype):
            try:
                row.append(f(n).str)
            except Exception as e:
                row.append("ERR:" + type(e).__name__)
        rows.append(row)
    w.add(id="dtype_intuint", op="dtype_intuint", expect=rows)
    rows = []
    for dt in ("u1", "i1", "u2", "i2", "u4", "i4", "u8", "i8", "f2", "f4", "f8"):
        rows.append([dt, R_utils.get_dtype_max(dt), R_utils.get_dtype_min(dt)])
    w.add(id="dtype_maxmin", op="dtype_maxmin", expect=rows)


# --------------------------------------------------------------------------
# family: smoothzoom  (reference: raster/spatial.py:296-305 -- SciPy nearest)
# --------------------------------------------------------------------------
def gen_smoothzoom(w):
    from scipy import ndimage
    rng = np.random.default_rng(71)
    cid = 0
    for shape, size_px in (((1, 16, 24), (2.0, 3.0)), ((2, 9, 7), (1.5, 1.0)),
                           ((1, 40, 40), (10.0, 10.0)), ((1, 12, 5), (0.25, 2.4)),
                           ((1, 64, 64), (16.0, 8.0))):
        values = rng.random(shape).astype("f4")
        _, ny, nx = values.shape
        zy, zx = [1 - 2 * size_px[0] / ny, 1 - 2 * size_px[1] / nx]
        # literal call pattern of the reference's zoom-back step (spatial.py:300-305)
        out = ndimage.affine_transform(values, order=0, matrix=np.diag([1, zy, zx]),
                                       offset=[0, size_px[0], size_px[1]])
        w.add(id="zoom%04d" % cid, op="affine_nearest",
              args=[values, [zy, zx], [float(size_px[0]), float(size_px[1])]], expect=out)
        cid += 1


# --------------------------------------------------------------------------
# family: spatial  (reference: raster/spatial.py:27-47, 111-213, 320-440)
# --------------------------------------------------------------------------
def gen_spatial(w):
    from dask_geomodeling.raster import spatial as R_sp
    rng = np.random.default_rng(81)
    cid = 0
    # Dilate: integer and float tiles, several bands (scipy's default structure also spans the band axis)
    for dt, ndv in (("u1", np.uint8(255)), ("i4", -9999), ("f4", np.float32(F32MAX)), ("i8", 9223372036854775807)):
        for shape in ((1, 14, 19), (3, 9, 12), (2, 3, 3)):
            v = rng.integers(0, 6, size=shape).astype(dt)
            v[rng.random(shape) < 0.1] = ndv
            for values in ([1], [2, 3], [5, 1, 2], [dt == "f4" and 2.0 or 2, 4, 0]):
                data = {"values": v, "no_data_value": ndv}
                w.add(id="sp%04d" % cid, op="dilate", args=[data, values],
                      expect=R_sp.Dilate.process(data, values)); cid += 1
    for args in ([None, [1]], [{"time": ["t"]}, [1]], [{"values": np.zeros((1, 4, 4), "u1"), "no_data_value": 255}, None]):
        w.add(id="sp%04d" % cid, op="dilate", args=args, expect=R_sp.Dilate.process(*args)); cid += 1
    # MovingMax: finite values only (SciPy's NaN behaviour depends on the filter path it picks)
    for dt, ndv in (("f4", np.float32(F32MAX)), ("f4", np.float32(-1.0)), ("u1", np.uint8(255)), ("i4", -9999),
                    ("i2", np.int16(-32768)), ("f8", F64MAX)):
        for size in (3, 5, 7, 11):
            r = size // 2
            for shape in ((1, 12 + 2 * r, 17 + 2 * r), (2, 5 + 2 * r, 6 + 2 * r)):
                if np.dtype(dt).kind == "f":
                    v = (rng.random(shape) * 100 - 50).astype(dt)
                else:
                    info = np.iinfo(dt)
                    v = rng.integers(max(info.min, -100), min(info.max, 100), size=shape, endpoint=True).astype(dt)
                v[rng.random(shape) < 0.3] = ndv
                v[:, :r + 1, :r + 1] = ndv  # a region whose whole footprint is no-data
                data = {"values": v, "no_data_value": ndv}
                w.add(id="sp%04d" % cid, op="movingmax", args=[data, size],
                      expect=R_sp.MovingMax.process(data, size)); cid += 1
    for args in ([None, 3], [{"meta": ["m"]}, 3], [{"values": np.zeros((1, 4, 4), "u1"), "no_data_value": 255}, None]):
        w.add(id="sp%04d" % cid, op="movingmax", args=args, expect=R_sp.MovingMax.process(*args)); cid += 1
    # HillShade: float and integer elevation tiles
    for dt, ndv in (("f4", np.float32(F32MAX)), ("f8", F64MAX), ("i4", -9999), ("u1", np.uint8(255)), ("i2", np.int16(-1))):
        for shape in ((1, 16, 21), (2, 7, 9)):
            for res, alt, az, fill in (((0.5, 0.5), 45.0, 315.0, 0), ((5.0, 2.0), 30.0, 90.0, 10), ((1.0, 1.0), 80.0, 200.5, 3)):
                if np.dtype(dt).kind == "f":
                    v = (np.cumsum(rng.random(shape), axis=2) * 3 + rng.random(shape)).astype(dt)
                else:
                    info = np.iinfo(dt)
                    v = rng.integers(max(info.min, 0), min(info.max, 120), size=shape, endpoint=True).astype(dt)
                v[rng.random(shape) < 0.05] = ndv
                data = {"values": v, "no_data_value": ndv}
                kw = dict(resolution=res, altitude=alt, azimuth=az, fill=fill)
                w.add(id="sp%04d" % cid, op="hillshade", args=[data, kw],
                      expect=R_sp.HillShade.process(data, kw)); cid += 1
    w.add(id="sp%04d" % cid, op="hillshade", args=[{"time": ["t"]}, None],
          expect=R_sp.HillShade.process({"time": ["t"]}, None)); cid += 1
    # request expansion
    for request in (dict(mode="vals", bbox=(0.0, 0.0, 10.0, 20.0), width=20, height=10, projection="EPSG:28992"),
                    dict(mode="vals", bbox=(5.0, -3.0, 6.5, 0.0), width=3, height=12, projection="EPSG:28992"),
                    dict(mode="time", bbox=(0.0, 0.0, 1.0, 1.0), width=1, height=1),
                    dict(mode="vals", bbox=(1.0, 1.0, 1.0, 1.0), width=1, height=1, projection="EPSG:28992")):
        for radius in (1, 2, 5):
            w.add(id="sp%04d" % cid, op="expand", args=[request, radius],
                  expect=R_sp.expand_request_pixels(dict(request), radius)); cid += 1


# --------------------------------------------------------------------------
# family: reduce  (reference: raster/reduction.py:38-119, 189-226)
# --------------------------------------------------------------------------
def gen_reduce(w):
    from dask_geomodeling.raster import reduction as R_red
    rng = np.random.default_rng(91)
    cid = 0
    f32n = np.float32(F32MAX)
    for dts in (("f4", "f4"), ("f4", "f4", "f4"), ("u1", "u1"), ("i4", "u1"), ("i4", "i4", "i4", "i4"),
                ("f8", "f4"), ("i2", "f4"), ("i8", "i8")):
        for shape in SHAPES:
            stack = []
            for dt in dts:
                ndv = {"f4": f32n, "f8": F64MAX, "u1": np.uint8(255), "i4": -9999, "i2": np.int16(-1),
                       "i8": 9223372036854775807}[dt]
                stack.append({"values": make_tile(rng, dt, shape, ndv, specials=(dt in ("f4", "f8"))), "no_data_value": ndv})
            out_dt = np.result_type(*[np.dtype(d) for d in dts])
            kw = {"dtype": out_dt.name, "fillvalue": dtype_max(out_dt)}
            w.add(id="red%04d" % cid, op="max", args=[kw] + stack, expect=R_red.Max.process(kw, *stack)); cid += 1
    tile = {"values": make_tile(rng, "f4", SHAPES[0], f32n), "no_data_value": f32n}
    kw = {"dtype": "float32", "fillvalue": F32MAX}
    for args in ([kw, {"time": ["t"]}, tile], [kw, tile, {"meta": ["m"]}], [kw]):
        w.add(id="red%04d" % cid, op="max", args=args, expect=R_red.Max.process(*args)); cid += 1


# --------------------------------------------------------------------------
# family: reducestats  (reference: raster/reduction.py:38-119) -- reduce_rasters with every statistic
# --------------------------------------------------------------------------
REDUCE_STATISTICS = ("first", "last", "count", "sum", "mean", "min", "max", "argmin", "argmax", "std", "var",
                     "median", "product", "p25", "p50", "p99.5")


def gen_reducestats(w):
    from dask_geomodeling.raster import reduction as R_red
    rng = np.random.default_rng(191)
    cid = 0
    ndvs = {"f4": np.float32(F32MAX), "f8": F64MAX, "u1": np.uint8(255), "i4": -9999, "i2": np.int16(-1), "i1": np.int8(-128)}
    shape = (2, 11, 19)
    # (stack dtypes, output dtype or None = the first element's, inject non-finite values)
    plans = [(("f4", "f4", "f4", "f4", "f4"), None, False), (("f4", "f4", "f4"), None, True),
             (("u1", "u1", "u1", "u1"), None, False), (("i1", "i1", "i1"), None, False),
             (("i2", "i2", "i2", "i2", "i2", "i2"), None, False), (("i4", "i4", "i4"), None, False),
             (("f8", "f8", "f8", "f8"), None, False), (("f4", "f8", "f4"), None, False),
             (("i2", "u1", "i2"), None, False), (("i4", "i4", "i4"), "f4", False), (("f4", "f4", "f4"), "i4", False),
             (("u1", "u1", "u1"), "f4", False), (("f4", "f4"), "f8", False), (("f4",), None, False)]
    for dts, out_dt, specials in plans:
        stack = []
        for dt in dts:
            ndv = ndvs[dt]
            v = make_tile(rng, dt, shape, ndv, specials=specials)
            if np.dtype(dt).kind != "f":  # small values: sums and products stay inside every dtype involved
                v = np.where(v == ndv, v, (v.astype("i8") % 7 + 1).astype(dt))
            v[0, 0, :4] = ndv   # a run of cells that are no data in every raster
            stack.append({"values": v, "no_data_value": ndv})
        for stat in REDUCE_STATISTICS:
            kwargs = {}
            if out_dt is not None:
                kwargs = {"dtype": out_dt, "no_data_value": dtype_max(np.dtype(out_dt))}
            try:
                expect = R_red.reduce_rasters([dict(d) for d in stack], stat, **kwargs)
            except Exception as e:  # noqa: BLE001 -- what the reference raises is part of its behaviour
                w.add(id="rs%04d" % cid, op=stat, stack=stack, kwargs=kwargs, raises=type(e).__name__); cid += 1
                continue
            w.add(id="rs%04d" % cid, op=stat, stack=stack, kwargs=kwargs, expect=expect); cid += 1


# --------------------------------------------------------------------------
# family: group  (reference: raster/combine.py:143-466) -- whole get_data() of Group views on
# temporal probe sources (period / timedelta / frames of their own)
# --------------------------------------------------------------------------
class TemporalProbe(RasterBlock):
    """Probe source with ``n`` frames ``delta`` apart from ``first`` (delta None: free time stamps)."""

    ARRAYS = {}

    def __init__(self, name, dtype_str, ndv, times_iso, delta_s):
        super().__init__(name, dtype_str, ndv, times_iso, delta_s)

    dtype = property(lambda self: np.dtype(self.args[1]))
    fillvalue = property(lambda self: self.args[2])
    extent = None
    geometry = None
    projection = "EPSG:28992"
    geo_transform = None

    @property
    def times(self):
        return [datetime.fromisoformat(t) for t in self.args[3]]

    @property
    def period(self):
        t = self.times
        return (t[0], t[-1]) if t else None

    @property
    def timedelta(self):
        from datetime import timedelta
        return None if self.args[4] is None else timedelta(seconds=self.args[4])

    @property
    def temporal(self):
        return self.args[4] is not None or len(self.args[3]) > 1

    def get_sources_and_requests(self, **request):
        return [({"name": self.args[0], "times": self.args[3], "mode": request["mode"],
                  "start": request.get("start"), "stop": request.get("stop")}, None)]

    @staticmethod
    def process(kw):
        times = [datetime.fromisoformat(t) for t in kw["times"]]
        start, stop = kw["start"], kw["stop"]
        if not times:
            return None
        if start is None:
            sel = [len(times) - 1]
        elif stop is None:
            sel = [min(range(len(times)), key=lambda i: abs(times[i] - start))]
        else:
            sel = [i for i, t in enumerate(times) if start <= t <= stop]
        if kw["mode"] == "time":
            return {"time": [times[i] for i in sel]}
        if kw["mode"] == "meta":
            return {"meta": ["%s/%d" % (kw["name"], i) for i in sel]}
        if not sel:
            return None
        data = TemporalProbe.ARRAYS[kw["name"]]
        return {"values": data["values"][sel], "no_data_value": data["no_data_value"]}


def gen_group(w):
    from dask_geomodeling.raster import combine as R_comb
    rng = np.random.default_rng(101)
    cid = 0
    f32n = np.float32(F32MAX)
    day = 86400

    def src(name, dt, ndv, times, delta):
        shape = (len(times), 12, 16)
        TemporalProbe.ARRAYS[name] = {"values": make_tile(rng, dt, shape, ndv, specials=False), "no_data_value": ndv}
        return TemporalProbe(name, np.dtype(dt).str, ndv, times, delta)

    t = ["2020-01-%02dT00:00:00" % d for d in range(1, 11)]
    setups = {
        "static3": [("a", "f4", f32n, t[:1], None), ("b", "f4", f32n, t[:1], None), ("c", "f4", f32n, t[:1], None)],
        "mixeddtype": [("a", "u1", np.uint8(255), t[:1], None), ("b", "i4", -9999, t[:1], None)],
        "aligned": [("a", "f4", f32n, t[0:4], day), ("b", "f4", f32n, t[2:8], day), ("c", "f4", f32n, t[6:10], day)],
        "freetimes": [("a", "f4", f32n, [t[0], t[3], t[4]], None), ("b", "f4", f32n, [t[3], t[7]], None)],
        "misaligned": [("a", "i4", -9999, t[0:3], day), ("b", "i4", -9999, ["2020-01-02T12:00:00", "2020-01-03T12:00:00"], day)],
    }
    reqs = [dict(), dict(start=datetime(2020, 1, 3)), dict(start=datetime(2020, 1, 2), stop=datetime(2020, 1, 8)),
            dict(start=datetime(2019, 1, 1)), dict(start=datetime(2020, 1, 4, 7)), dict(start=datetime(2021, 1, 1), stop=datetime(2021, 2, 1)),
            dict(start=datetime(2020, 1, 7), stop=datetime(2020, 3, 1))]
    for sname, specs in setups.items():
        leaves, spec_out = [], []
        for (name, dt, ndv, times, delta) in specs:
            key = "%s/%s" % (sname, name)
            leaves.append(src(key, dt, ndv, times, delta))
            spec_out.append({"name": key, "dtype": np.dtype(dt).str, "ndv": ndv, "times": times, "delta": delta,
                             "data": TemporalProbe.ARRAYS[key]["values"]})
        view = R_comb.Group(*leaves)
        attrs = {"dtype": view.dtype.str, "fillvalue": view.fillvalue, "period": list(view.period) if view.period else None,
                 "timedelta": None if view.timedelta is None else view.timedelta.total_seconds(), "temporal": view.temporal}
        for extra in reqs:
            for mode in ("vals", "time", "meta"):
                request = dict(mode=mode, bbox=(0.0, 0.0, 16.0, 12.0), width=16, height=12, projection="EPSG:28992", **extra)
                out = view.get_data(**request)
                w.add(id="grp%04d" % cid, setup=sname, sources=spec_out, attrs=attrs, request=request, expect=out)
                cid += 1


# --------------------------------------------------------------------------
# family: smooth  (reference: raster/spatial.py:276-308; SciPy gaussian_filter + zoom)
# --------------------------------------------------------------------------
def gen_smooth(w):
    from dask_geomodeling.raster import spatial as R_sp
    rng = np.random.default_rng(111)
    cid = 0
    for dt, ndv in (("f4", np.float32(F32MAX)), ("f8", F64MAX), ("f4", np.float32(-9999.0))):
        for shape, size, mode in (((1, 30, 41), (2.0, 3.0), "exact"), ((2, 17, 19), (1.0, 1.0), "exact"),
                                  ((1, 24, 24), (4.6, 0.5), "exact"), ((1, 40, 48), (3.75, 6.0), "zoom"),
                                  ((1, 33, 20), (5.2, 2.9), "zoom"), ((1, 12, 12), (0.2, 0.2), "exact")):
            for fill in (0, 2.5):
                v = (rng.random(shape) * 50).astype(dt)
                v[rng.random(shape) < 0.1] = ndv
                data = {"values": v, "no_data_value": ndv}
                kw = dict(smooth_mode=mode, fill=fill, size=list(size))
                w.add(id="sm%04d" % cid, op="smooth", args=[data, kw], expect=R_sp.Smooth.process(data, kw)); cid += 1
    for args in ([None, dict(smooth_mode="exact", fill=0, size=[1.0, 1.0])], [{"time": ["t"]}, None]):
        w.add(id="sm%04d" % cid, op="smooth", args=args, expect=R_sp.Smooth.process(*args)); cid += 1
    # integer rasters (elevation models in int16 ...): SciPy filters in float64 and stores with a C cast
    for dt, ndv in (("i2", np.int16(-9999)), ("u1", np.uint8(255)), ("i4", -9999), ("i8", -9999), ("u2", np.uint16(65535))):
        for shape, size, mode in (((1, 30, 41), (2.0, 3.0), "exact"), ((2, 17, 19), (1.0, 1.0), "exact"),
                                  ((1, 40, 48), (3.75, 6.0), "zoom")):
            for fill in (0, 2.5):
                v = rng.integers(0, 200, size=shape).astype(dt)
                v[rng.random(shape) < 0.1] = ndv
                data = {"values": v, "no_data_value": ndv}
                kw = dict(smooth_mode=mode, fill=fill, size=list(size))
                w.add(id="sm%04d" % cid, op="smooth", args=[data, kw], expect=R_sp.Smooth.process(data, kw)); cid += 1


# --------------------------------------------------------------------------
# family: temporal  (reference: raster/temporal.py:476-1003) -- TemporalAggregate / Cumulative: whole
# get_data() on temporal probe sources plus the process() arguments the reference built on the way
# --------------------------------------------------------------------------
def gen_temporal(w):
    from datetime import timedelta
    from dask_geomodeling.raster import temporal as R_tmp
    rng = np.random.default_rng(131)
    cid = 0
    f32n = np.float32(F32MAX)

    def stamps(first, n, seconds):
        t0 = datetime.fromisoformat(first)
        return [(t0 + timedelta(seconds=seconds * i)).isoformat() for i in range(n)]

    def src(name, dt, ndv, times, delta, shape_yx):
        shape = (len(times),) + shape_yx
        v = make_tile(rng, dt, shape, ndv, specials=np.dtype(dt).kind == "f")
        if ndv is not None:
            v[:, 0, 1] = ndv          # a cell that never has data
            v[: len(times) // 2, 1, 0] = ndv
        TemporalProbe.ARRAYS[name] = {"values": v, "no_data_value": ndv}
        return TemporalProbe(name, np.dtype(dt).str, ndv, times, delta)

    free = ["2020-03-01T06:00:00", "2020-03-01T07:30:00", "2020-03-01T23:59:59", "2020-03-02T00:00:00",
            "2020-03-04T12:00:00", "2020-03-04T12:00:01", "2020-03-07T01:00:00"]
    sources = {
        "tenmin_f4": ("f4", f32n, stamps("2020-01-01T00:00:00", 30, 600), 600, (6, 8)),
        "daily_u1": ("u1", np.uint8(255), stamps("2020-01-25T00:00:00", 16, 86400), 86400, (5, 7)),
        "free_i4": ("i4", -9999, free, None, (6, 8)),
        "hourly_f8": ("f8", F64MAX, stamps("2019-12-31T20:00:00", 10, 3600), 3600, (5, 7)),
        "hourly_i2": ("i2", np.int16(-1), stamps("2020-06-01T00:00:00", 12, 3600), 3600, (6, 8)),
        "daily_bool": ("?", None, stamps("2020-01-01T00:00:00", 6, 86400), 86400, (6, 8)),
        "tenmin_f4neg": ("f4", np.float32(-9999.0), stamps("2020-01-01T00:00:00", 13, 600), 600, (5, 7)),
    }
    probes, specs = {}, {}
    for name, (dt, ndv, times, delta, shape_yx) in sources.items():
        probes[name] = src(name, dt, ndv, times, delta, shape_yx)
        specs[name] = {"name": name, "dtype": np.dtype(dt).str, "ndv": ndv, "times": times, "delta": delta,
                       "data": TemporalProbe.ARRAYS[name]["values"]}

    def window(name):
        h, wd = specs[name]["data"].shape[1:]
        return dict(bbox=(0.0, 0.0, float(wd), float(h)), width=wd, height=h, projection="EPSG:28992")

    all_stats = ["sum", "count", "min", "max", "mean", "std", "var", "median", "p50", "p95"]
    aggregates = [
        ("tenmin_f4", dict(frequency="h"), all_stats),
        ("tenmin_f4", dict(frequency="30min", closed="right", label="right"), ["sum", "mean"]),
        ("tenmin_f4", dict(frequency=None), ["sum", "max", "std"]),
        ("tenmin_f4", dict(frequency="D"), ["mean", "count"]),
        ("daily_u1", dict(frequency="W"), ["sum", "min", "max", "mean", "var", "count", "median", "p25", "p90"]),
        ("daily_u1", dict(frequency="ME"), ["sum", "mean", "max"]),
        ("daily_u1", dict(frequency="2D", closed="left", label="right"), ["sum", "std"]),
        ("free_i4", dict(frequency="D"), ["sum", "count", "min", "max", "mean", "std", "var", "median", "p10", "p66.6"]),
        ("free_i4", dict(frequency="D", timezone="Europe/Amsterdam"), ["sum", "mean"]),
        ("free_i4", dict(frequency=None), ["mean", "min"]),
        ("hourly_f8", dict(frequency="D", timezone="Europe/Amsterdam"), ["sum", "mean", "std", "min"]),
        ("hourly_f8", dict(frequency="3h"), ["sum", "count", "min", "max", "mean", "std", "var", "median", "p100", "p33"]),
        ("hourly_i2", dict(frequency="4h"), ["sum", "count", "min", "max", "mean", "std", "var", "median"]),
        ("daily_bool", dict(frequency="2D"), ["sum", "count", "mean"]),
        ("tenmin_f4neg", dict(frequency="h", closed="right"), ["sum", "min", "max", "mean", "var", "p75", "p99.9"]),
        ("tenmin_f4", dict(frequency=None), ["median", "p5", "p50", "p87.5"]),
    ]
    day = timedelta(days=1)

    def requests_for(name):
        times = [datetime.fromisoformat(t) for t in specs[name]["times"]]
        first, last = times[0], times[-1]
        mid = first + (last - first) / 2
        return [dict(), dict(start=mid), dict(start=first - day, stop=last + day), dict(start=mid, stop=last),
                dict(start=first + (last - first) / 4, stop=mid), dict(start=last + 40 * day, stop=last + 50 * day),
                dict(start=first - 3 * day)]

    def record(kind, view, source, request, extra):
        nonlocal cid
        plan = view.get_sources_and_requests(**request)
        args = [plan[0][0]] + [src_.get_data(**req) for src_, req in plan[1:]]
        w.add(id="tmp%04d" % cid, kind=kind, source=specs[source], request=request, process_args=args,
              attrs={"dtype": np.dtype(view.dtype).str, "fillvalue": view.fillvalue,
                     "period": list(view.period) if view.period else None,
                     "timedelta": None if view.timedelta is None else view.timedelta.total_seconds(),
                     "temporal": view.temporal},
              expect=view.get_data(**request), **extra)
        cid += 1

    for name, kw, stats in aggregates:
        for k, stat in enumerate(stats):
            view = R_tmp.TemporalAggregate(probes[name], kw.get("frequency"), stat, kw.get("closed"), kw.get("label"),
                                           kw.get("timezone", "UTC"))
            for extra in requests_for(name):
                for mode in (("vals", "time", "meta") if k == 0 else ("vals",)):
                    record("aggregate", view, name, dict(mode=mode, **window(name), **extra),
                           dict(block_args=[kw.get("frequency"), stat, kw.get("closed"), kw.get("label"), kw.get("timezone", "UTC")]))

    cumulatives = [
        ("tenmin_f4", None, "UTC"), ("tenmin_f4", "h", "UTC"), ("daily_u1", "W", "UTC"), ("free_i4", "D", "Europe/Amsterdam"),
        ("hourly_f8", "3h", "UTC"), ("hourly_i2", None, "UTC"), ("daily_bool", "2D", "UTC"), ("tenmin_f4neg", "30min", "UTC"),
    ]
    for name, freq, tz in cumulatives:
        for k, stat in enumerate(("sum", "count")):
            view = R_tmp.Cumulative(probes[name], stat, freq, tz)
            for extra in requests_for(name):
                for mode in (("vals", "time", "meta") if k == 0 else ("vals",)):
                    record("cumulative", view, name, dict(mode=mode, **window(name), **extra),
                           dict(block_args=[stat, freq, tz]))


# --------------------------------------------------------------------------
# family: zonal  (reference: geometry/aggregate.py:113-203 aggregate_polygons) -- the real function with the
# real scipy.ndimage / measurements.percentile; only the rasterization (GDAL, absent here) is replaced: the
# "geometries" are masks, and utils.rasterize_geoseries paints the masks of the requested bucket in order
# --------------------------------------------------------------------------
class MaskGeometries:
    """Stand-in for the GeoSeries aggregate_polygons receives: a list of boolean masks (the cells GDAL would
    paint for each polygon).  Supports exactly what the function uses: len(), .bounds.values, .iloc[select]."""

    def __init__(self, masks, picked=None):
        self.masks, self.picked = masks, picked

    def __len__(self):
        return len(self.masks)

    @property
    def bounds(self):
        boxes = []
        for m in self.masks:
            ys, xs = np.nonzero(m)
            if len(ys) == 0:  # a polygon too small to cover a cell centre: a sliver of a bbox
                boxes.append((0.25, 0.25, 0.3, 0.3))
            else:
                h = m.shape[0]
                boxes.append((float(xs.min()), float(h - 1 - ys.max()), float(xs.max() + 1), float(h - ys.min())))
        return type("Bounds", (), {"values": np.array(boxes, dtype="f8")})()

    @property
    def iloc(self):
        outer = self
        return type("ILoc", (), {"__getitem__": lambda _, select: MaskGeometries(outer.masks, list(select))})()


def gen_zonal(w):
    from dask_geomodeling.geometry import aggregate as R_agg
    rng = np.random.default_rng(733)
    label_ndv = np.iinfo(np.int32).max
    painted = []

    def rasterize(geoseries, bbox, projection, height, width, values=None):
        labels = np.full((1, height, width), label_ndv, dtype=np.int32)
        for index in geoseries.picked:
            labels[0][geoseries.masks[index]] = index
        painted.append(([int(i) for i in geoseries.picked], labels[0].copy()))
        return {"values": labels, "no_data_value": label_ndv}

    R_agg.utils.rasterize_geoseries = rasterize
    ndvs = {"f4": np.float32(F32MAX), "f8": F64MAX, "u1": np.uint8(255), "i4": -9999, "i2": np.int16(-1), "i1": np.int8(-128),
            "i8": np.int64(-1), "u2": np.uint16(65535), "?": None}
    cid = 0
    h, wd = 37, 53
    for dt in ("f4", "f8", "u1", "i1", "i2", "u2", "i4", "i8", "?"):
        for layout in ("tiles", "overlap"):
            n = 9
            masks = []
            for g in range(n):
                m = np.zeros((h, wd), dtype=bool)
                if layout == "tiles":   # disjoint blobs in a 3 x 3 grid: one bucket
                    y0, x0 = (g // 3) * 12, (g % 3) * 17
                    m[y0 + 1:y0 + 11, x0 + 1:x0 + 15] = rng.random((10, 14)) < 0.8
                else:                   # overlapping rectangles: several buckets
                    y0, x0 = rng.integers(0, h - 8), rng.integers(0, wd - 8)
                    m[y0:y0 + rng.integers(1, 20), x0:x0 + rng.integers(1, 25)] = True
                masks.append(m)
            masks[4][:] = False  # a geometry that covers no cell
            ndv = ndvs[dt]
            if dt == "?":
                values = rng.random((2, h, wd)) < 0.6
            elif np.dtype(dt).kind == "f":
                values = (rng.standard_normal((2, h, wd)) * 50).astype(dt)
                values[rng.random(values.shape) < 0.05] = 0.0
            else:
                info = np.iinfo(dt)
                values = rng.integers(max(info.min, -120), min(info.max, 120), size=(2, h, wd)).astype(dt)
            if ndv is not None:
                values[rng.random(values.shape) < 0.15] = ndv
                values[1][masks[2]] = ndv  # a geometry without an active cell in one frame
            thresholds = (rng.standard_normal(n) * 20).astype("f4")
            thresholds[1] = np.nan
            runs, buckets = [], None
            for stat, q in (("sum", None), ("count", None), ("min", None), ("max", None), ("mean", None), ("median", None),
                            ("percentile", 0.0), ("percentile", 37.5), ("percentile", 95.0), ("percentile", 100.0)):
                for thr in (None, thresholds):
                    if thr is not None and stat in ("count", "max", "percentile") and q not in (None, 37.5):
                        continue
                    del painted[:]
                    run = dict(statistic=stat, percentile=q, thresholds=thr is not None)
                    try:
                        with np.errstate(all="ignore"):
                            agg, no_cells = R_agg.aggregate_polygons(MaskGeometries(masks), values.copy(), ndv, (0, 0, wd, h),
                                                                     "EPSG:28992", None if thr is None else thr.copy(), stat, q)
                        run.update(expect=agg, no_cells=sorted(int(i) for i in no_cells))
                    except Exception as e:  # noqa: BLE001 -- what the reference raises is part of its behaviour
                        run.update(raises=type(e).__name__)
                    if "expect" in run:  # (the buckets do not depend on the statistic: kept once per case)
                        assert buckets is None or [b[0] for b in buckets] == [b[0] for b in painted]
                        buckets = list(painted)
                    runs.append(run)
            w.add(id="zn%04d" % cid, n=n, values=values, no_data_value=ndv, threshold_values=thresholds, buckets=buckets, runs=runs)
            cid += 1


# --------------------------------------------------------------------------
# family: round3  (what the device refused until round 3, each produced by the reference itself)
#   dilate with more than 16 values (spatial.py:147-154), HillShade / Smooth on bool and uint64 rasters
#   (spatial.py:276-308, 354-417), reduce_rasters into a 1-byte dtype from wider rasters (reduction.py:99-107:
#   the work dtype is float16 and the assignment into the stack rounds)
# --------------------------------------------------------------------------
def gen_round3(w):
    from dask_geomodeling.raster import reduction as R_red
    from dask_geomodeling.raster import spatial as R_sp
    rng = np.random.default_rng(303)
    cid = 0
    # Dilate, 17 .. 40 values, in an order that makes later values overwrite earlier ones
    for dt, ndv in (("u1", np.uint8(255)), ("i4", -9999), ("f4", np.float32(F32MAX)), ("i8", 9223372036854775807), ("u2", np.uint16(65535))):
        for shape in ((1, 20, 31), (2, 9, 12)):
            v = rng.integers(0, 48, size=shape).astype(dt)
            v[rng.random(shape) < 0.1] = ndv
            for count in (17, 33, 40):
                values = [int(x) for x in rng.permutation(48)[:count]]
                if dt == "f4":
                    values = [float(x) for x in values]
                data = {"values": v, "no_data_value": ndv}
                w.add(id="r3%04d" % cid, op="dilate", args=[data, values], expect=R_sp.Dilate.process(data, values)); cid += 1
    # HillShade on bool and uint64 elevation tiles
    u64max = np.uint64(18446744073709551615)
    for dt, ndv in (("bool", None), ("u8", u64max), ("u8", np.uint64(7))):
        for shape in ((1, 16, 21), (2, 7, 9)):
            for res, alt, az, fill in (((0.5, 0.5), 45.0, 315.0, 0), ((5.0, 2.0), 30.0, 90.0, 1), ((1.0, 1.0), 80.0, 200.5, 3)):
                if dt == "bool":
                    v = rng.random(shape) < 0.4
                else:
                    v = rng.integers(0, 120, size=shape).astype(dt)
                    v[rng.random(shape) < 0.05] = ndv
                    v.reshape(-1)[:3] = [u64max - np.uint64(1), np.uint64(2 ** 63), np.uint64(2 ** 53 + 1)]  # sums wrap modulo 2^64
                data = {"values": v, "no_data_value": ndv}
                kw = dict(resolution=res, altitude=alt, azimuth=az, fill=fill)
                w.add(id="r3%04d" % cid, op="hillshade", args=[data, kw], expect=R_sp.HillShade.process(data, kw)); cid += 1
    # Smooth on bool and uint64 tiles (SciPy filters in float64 and stores with a C cast)
    for dt, ndv in (("bool", None), ("u8", u64max)):
        for shape, size, mode in (((1, 30, 41), (2.0, 3.0), "exact"), ((2, 17, 19), (1.0, 1.0), "exact"), ((1, 40, 48), (3.75, 6.0), "zoom")):
            for fill in (0, 1):
                if dt == "bool":
                    v = rng.random(shape) < 0.6
                    v[:, 10:16, 8:16] = True  # (a region whose filtered value reaches 1.0 and survives the cast)
                else:
                    v = rng.integers(0, 200, size=shape).astype(dt)
                    v[rng.random(shape) < 0.1] = ndv
                    v.reshape(-1)[5] = np.uint64(2 ** 62)
                data = {"values": v, "no_data_value": ndv}
                kw = dict(smooth_mode=mode, fill=fill, size=list(size))
                w.add(id="r3%04d" % cid, op="smooth", args=[data, kw], expect=R_sp.Smooth.process(data, kw)); cid += 1
    # reduce_rasters into a 1-byte output dtype from wider rasters: work dtype float16
    ndvs = {"f4": np.float32(F32MAX), "f8": F64MAX, "u1": np.uint8(255), "i4": -9999, "i2": np.int16(-1), "i8": -9999}
    shape = (2, 11, 19)
    plans = [(("i4", "i4", "i4"), "u1"), (("f4", "f4"), "u1"), (("f8", "f4", "i2"), "i1"), (("f8", "f8", "f8"), "u1"),
             (("i2", "u1", "i4"), "u1"), (("i8", "i8"), "i1")]
    for dts, out_dt in plans:
        stack = []
        for dt in dts:
            ndv = ndvs[dt]
            if np.dtype(dt).kind == "f":
                v = (rng.random(shape) * 12).astype(dt)
                flat = v.reshape(-1)
                # float16 traps: a tie that only the float64 value breaks (rounding through float32 would lose it),
                # values that need rounding, one beyond float16's range (-> inf) and one just inside
                trap = 1.0 + 2.0 ** -11 + 2.0 ** -30
                v[1, 3, :8] = [trap, 1.0 + 2.0 ** -11, 2049.0, 0.1, 65519.9, 70000.0, 3.0001, 1e-8]
            else:
                v = rng.integers(0, 9, size=shape).astype(dt)
                if np.dtype(dt).itemsize >= 4:
                    v[1, 3, :4] = [2049, 4099, 65519, 70000]  # not float16 numbers; the last one overflows
            drop = rng.random(shape) < 0.1
            drop[1, 3, :8] = False  # (the traps stay data in every raster)
            v[drop] = ndv
            v[0, 0, :4] = ndv
            stack.append({"values": v, "no_data_value": ndv})
        for stat in REDUCE_STATISTICS:
            kwargs = {"dtype": out_dt, "no_data_value": dtype_max(np.dtype(out_dt))}
            try:
                expect = R_red.reduce_rasters([dict(d) for d in stack], stat, **kwargs)
            except Exception as e:  # noqa: BLE001
                w.add(id="r3%04d" % cid, op="reduce:" + stat, stack=stack, kwargs=kwargs, raises=type(e).__name__); cid += 1
                continue
            w.add(id="r3%04d" % cid, op="reduce:" + stat, stack=stack, kwargs=kwargs, expect=expect); cid += 1
    # long stacks: the array the reference reduces, stack_array[:, not_all_nan], has the rasters along its FAST axis, so
    # NumPy sums them pairwise (8 interleaved accumulators from 8 rasters on, halves beyond 128) -- not one after the other
    small = (1, 6, 9)
    for n, dt, out_dt in ((8, "f4", None), (9, "f4", None), (17, "f4", None), (40, "f8", None), (130, "f4", None),
                          (300, "f4", None), (23, "i4", "f4"), (9, "f4", "u1"), (40, "u1", None)):
        ndv = ndvs[dt]
        stack = []
        for _ in range(n):
            if np.dtype(dt).kind == "f":
                v = (rng.random(small) * 100 - 20).astype(dt)
            else:
                v = rng.integers(0, 7, size=small).astype(dt)
            v[rng.random(small) < 0.15] = ndv
            v[0, 0, 0] = ndv
            stack.append({"values": v, "no_data_value": ndv})
        for stat in ("sum", "mean", "var", "std", "product", "median", "max", "count"):
            kwargs = {} if out_dt is None else {"dtype": out_dt, "no_data_value": dtype_max(np.dtype(out_dt))}
            expect = R_red.reduce_rasters([dict(d) for d in stack], stat, **kwargs)
            w.add(id="r3%04d" % cid, op="reduce:" + stat, stack=stack, kwargs=kwargs, expect=expect); cid += 1


# --------------------------------------------------------------------------
# family: round4  (reference: raster/temporal.py Snap 42-182, Shift 185-236, TemporalSum 239-255, Resample 1008-1214:
# whole get_data() on temporal probe sources + the arguments their process() got; raster/spatial.py Place.process
# 620-731 called directly -- Place's request planning needs GDAL / PROJ (Extent, get_sr), absent here)
# --------------------------------------------------------------------------
def gen_round4(w):
    from datetime import timedelta
    from dask_geomodeling.raster import spatial as R_sp
    from dask_geomodeling.raster import temporal as R_tmp
    rng = np.random.default_rng(404)
    cid = 0
    f32n = np.float32(F32MAX)

    def stamps(first, n, seconds):
        t0 = datetime.fromisoformat(first)
        return [(t0 + timedelta(seconds=seconds * i)).isoformat() for i in range(n)]

    free = ["2020-01-01T00:20:00", "2020-01-01T00:55:00", "2020-01-01T01:00:00", "2020-01-01T03:10:00",
            "2020-01-01T03:15:00", "2020-01-02T12:00:00", "2020-01-05T00:00:00"]
    sources = {
        "r4_tenmin_f4": ("f4", f32n, stamps("2020-01-01T00:00:00", 30, 600), 600, (6, 8)),
        "r4_hourly_f8": ("f8", F64MAX, stamps("2019-12-31T20:00:00", 10, 3600), 3600, (6, 8)),
        "r4_free_i4": ("i4", -9999, free, None, (6, 8)),
        "r4_daily_u1": ("u1", np.uint8(255), stamps("2019-12-30T00:00:00", 9, 86400), 86400, (6, 8)),
        "r4_single_i2": ("i2", np.int16(-1), ["2020-01-01T02:00:00"], None, (6, 8)),
        "r4_hourly_bool": ("?", None, stamps("2020-01-01T00:00:00", 7, 3600), 3600, (6, 8)),
        "r4_halfhour_u4": ("u4", np.uint32(4294967295), stamps("2020-01-01T00:15:00", 9, 1800), 1800, (6, 8)),
        "r4_onecell_f4": ("f4", f32n, stamps("2020-01-01T00:00:00", 40, 600), 600, (1, 1)),
        "r4_empty_f4": ("f4", f32n, [], None, (6, 8)),
    }
    probes, specs = {}, {}
    for name, (dt, ndv, times, delta, shape_yx) in sources.items():
        shape = (len(times),) + shape_yx
        v = make_tile(rng, dt, shape, ndv, specials=False) if len(times) else np.zeros(shape, dt)
        if np.dtype(dt).kind == "f" and len(times):
            v = (rng.random(shape) * 2000 - 500).astype(dt)  # sums whose order of addition shows
        if ndv is not None and len(times) and shape_yx != (1, 1):
            v[:, 0, 1] = ndv
            v[: len(times) // 2, 1, 0] = ndv
        TemporalProbe.ARRAYS[name] = {"values": v, "no_data_value": ndv}
        probes[name] = TemporalProbe(name, np.dtype(dt).str, ndv, times, delta)
        specs[name] = {"name": name, "dtype": np.dtype(dt).str, "ndv": ndv, "times": times, "delta": delta, "data": v}

    def window(name):
        h, wd = specs[name]["data"].shape[1:]
        return dict(bbox=(0.0, 0.0, float(wd), float(h)), width=wd, height=h, projection="EPSG:28992")

    day, hour = timedelta(days=1), timedelta(hours=1)

    def requests_around(times_iso):
        times = [datetime.fromisoformat(t) for t in times_iso] or [datetime(2020, 1, 1)]
        first, last = times[0], times[-1]
        mid = first + (last - first) / 2
        return [dict(), dict(start=mid), dict(start=first - day, stop=last + day), dict(start=mid, stop=last),
                dict(start=first + (last - first) / 4, stop=mid), dict(start=last + 40 * day, stop=last + 50 * day),
                dict(start=first - 3 * day), dict(start=mid, stop=mid)]

    def record(kind, view, names, block_args, request):
        nonlocal cid
        try:
            plan = view.get_sources_and_requests(**request)
            args = [src_.get_data(**req) if isinstance(src_, RasterBlock) and req is not None else src_ for src_, req in plan]
            period = view.period
            expect = view.get_data(**request)
        except Exception as e:  # noqa: BLE001 -- (what the reference itself does with this request)
            w.add(id="r4%04d" % cid, kind=kind, sources=[specs[n] for n in names], block_args=block_args, request=request,
                  raises=type(e).__name__)
            cid += 1
            return
        w.add(id="r4%04d" % cid, kind=kind, sources=[specs[n] for n in names], block_args=block_args, request=request,
              process_args=args,
              attrs={"dtype": np.dtype(view.dtype).str, "fillvalue": view.fillvalue,
                     "period": list(period) if period else None,
                     "timedelta": None if view.timedelta is None else view.timedelta.total_seconds(),
                     "temporal": view.temporal, "len": len(view)},
              expect=expect)
        cid += 1

    # Snap(store, index)
    for store, index in (("r4_hourly_f8", "r4_tenmin_f4"), ("r4_tenmin_f4", "r4_hourly_f8"), ("r4_free_i4", "r4_tenmin_f4"),
                         ("r4_single_i2", "r4_tenmin_f4"), ("r4_tenmin_f4", "r4_free_i4"), ("r4_daily_u1", "r4_free_i4"),
                         ("r4_hourly_bool", "r4_halfhour_u4"), ("r4_tenmin_f4", "r4_single_i2"),
                         ("r4_empty_f4", "r4_tenmin_f4"), ("r4_tenmin_f4", "r4_empty_f4")):
        view = R_tmp.Snap(probes[store], probes[index])
        for extra in requests_around(specs[index]["times"]):
            for mode in ("vals", "time", "meta"):
                record("snap", view, [store, index], [], dict(mode=mode, **window(store), **extra))
    # Shift(store, milliseconds)
    for store, ms in (("r4_tenmin_f4", 3600 * 1000), ("r4_tenmin_f4", -90 * 1000), ("r4_free_i4", 86400 * 1000),
                      ("r4_daily_u1", -12 * 3600 * 1000), ("r4_single_i2", 1)):
        view = R_tmp.Shift(probes[store], ms)
        shifted = [(datetime.fromisoformat(t) + timedelta(milliseconds=ms)).isoformat() for t in specs[store]["times"]]
        for extra in requests_around(shifted):
            for mode in ("vals", "time", "meta"):
                record("shift", view, [store], [ms], dict(mode=mode, **window(store), **extra))
    # TemporalSum(store)
    for store in ("r4_tenmin_f4", "r4_hourly_f8", "r4_free_i4", "r4_daily_u1", "r4_single_i2", "r4_hourly_bool",
                  "r4_halfhour_u4", "r4_onecell_f4"):
        view = R_tmp.TemporalSum(probes[store])
        for extra in requests_around(specs[store]["times"]):
            for mode in ("vals", "time", "meta"):
                record("temporalsum", view, [store], [], dict(mode=mode, **window(store), **extra))
    # Resample(source, frequency, direction, timezone)
    resamples = [("r4_tenmin_f4", "h", "nearest", "UTC"), ("r4_tenmin_f4", "h", "forward", "UTC"), ("r4_tenmin_f4", "h", "backward", "UTC"),
                 ("r4_tenmin_f4", "5min", "nearest", "UTC"), ("r4_tenmin_f4", "25min", "backward", "UTC"),
                 ("r4_hourly_f8", "D", "nearest", "Europe/Amsterdam"), ("r4_hourly_f8", "20min", "forward", "UTC"),
                 ("r4_free_i4", "h", "nearest", "UTC"), ("r4_free_i4", "D", "backward", "Europe/Amsterdam"),
                 ("r4_free_i4", "30min", "forward", "UTC"), ("r4_daily_u1", "ME", "nearest", "UTC"), ("r4_daily_u1", "W", "forward", "UTC"),
                 ("r4_daily_u1", "12h", "backward", "UTC"), ("r4_single_i2", "h", "nearest", "UTC"), ("r4_hourly_bool", "90min", "nearest", "UTC"),
                 ("r4_halfhour_u4", "h", "nearest", "UTC"), ("r4_empty_f4", "h", "nearest", "UTC")]
    for store, freq, direction, tz in resamples:
        view = R_tmp.Resample(probes[store], freq, direction, tz)
        for extra in requests_around(specs[store]["times"]):
            for mode in ("vals", "time", "meta"):
                record("resample", view, [store], [freq, direction, tz], dict(mode=mode, **window(store), **extra))

    # Place.process, called directly
    def place(kw, *multi):
        nonlocal cid
        try:
            expect = R_sp.Place.process(dict(kw), *[None if d is None else dict(d) for d in multi])
        except Exception as e:  # noqa: BLE001
            w.add(id="r4%04d" % cid, kind="place", process_args=[kw] + list(multi), raises=type(e).__name__)
            cid += 1
            return
        w.add(id="r4%04d" % cid, kind="place", process_args=[kw] + list(multi), expect=expect)
        cid += 1

    statistics = ["last", "first", "count", "sum", "mean", "min", "max", "argmin", "argmax", "product", "std", "var", "median", "p30"]
    tiles = [("u1", np.uint8(255), (1, 6, 8)), ("f4", f32n, (2, 5, 7)), ("i4", -9999, (1, 7, 5)), ("f8", -1.5, (1, 4, 6)),
             ("i2", np.int16(-32768), (3, 4, 4))]
    for dt, ndv, shape in tiles:
        v = make_tile(rng, dt, shape, ndv, specials=False)
        v[rng.random(shape) < 0.2] = ndv
        v[:, 0, :] = ndv                          # a whole row without data: copies that only show this row are left out
        if np.dtype(dt).kind == "f":
            v[0, 1, 1] = ndv * (1 + 1e-7) if dt == "f8" else np.nextafter(ndv, np.float32(0))  # close to no data
        data = {"values": v, "no_data_value": ndv}
        _, src_h, src_w = shape
        for cell, dst_hw in (((1.0, 1.0), (9, 11)), ((0.5, 0.5), (6, 6)), ((2.0, 1.0), (5, 9)), ((1.0, 1.0), (3, 3))):
            size_x, size_y = cell
            dst_h, dst_w = dst_hw
            src_bbox = (100.0, 200.0, 100.0 + src_w * size_x, 200.0 + src_h * size_y)
            dst_bbox = (40.0, 60.0, 40.0 + dst_w * size_x, 60.0 + dst_h * size_y)
            anchor = (100.0 + 0.5 * src_w * size_x, 200.0 + 0.5 * src_h * size_y)
            cx, cy = dst_bbox[0] + 0.5 * dst_w * size_x, dst_bbox[1] + 0.5 * dst_h * size_y
            layouts = [
                [(cx, cy)],
                [(cx - 2 * size_x, cy + size_y), (cx + 2 * size_x, cy - 2 * size_y), (cx, cy)],
                [(dst_bbox[0], dst_bbox[1]), (dst_bbox[2], dst_bbox[3]), (cx + 0.4 * size_x, cy - 0.6 * size_y)],
                [(dst_bbox[0] - src_w * size_x, cy), (cx, dst_bbox[3] + src_h * size_y), (cx, dst_bbox[1] - 0.5 * (src_h - 1) * size_y)],
                [(cx - 1e4, cy), (cx, cy + 1e4)],
                [],
            ]
            for k, coordinates in enumerate(layouts):
                for stat in (statistics if k in (1, 3) and cell == (1.0, 1.0) else ["last", "sum", "argmax"]):
                    place({"mode": "warp", "anchor": anchor, "coordinates": coordinates, "src_bbox": src_bbox, "dst_bbox": dst_bbox,
                           "cellsize": cell, "statistic": stat}, data)
        place({"mode": "warp", "anchor": (0.0, 0.0), "coordinates": [(1.0, 1.0)], "src_bbox": (0.0, 0.0, 8.0, 6.0),
               "dst_bbox": (0.0, 0.0, 8.0, 6.0), "cellsize": (1.0, 1.0), "statistic": "last"}, None)
        nothing = {"values": np.full(shape, ndv, dt), "no_data_value": ndv}
        place({"mode": "warp", "anchor": (104.0, 203.0), "coordinates": [(44.0, 63.0)], "src_bbox": (100.0, 200.0, 100.0 + shape[2], 200.0 + shape[1]),
               "dst_bbox": (40.0, 60.0, 50.0, 70.0), "cellsize": (1.0, 1.0), "statistic": "sum"}, nothing)
        # already shifted rasters (one request per coordinate)
        others = []
        for _ in range(3):
            o = make_tile(rng, dt, shape, ndv, specials=False)
            o[rng.random(shape) < 0.4] = ndv
            others.append({"values": o, "no_data_value": ndv})
        for stat in statistics:
            place({"mode": "group", "statistic": stat}, others[0], None, others[1], others[2])
        place({"mode": "group", "statistic": "last"}, None, None)
        place({"mode": "empty", "dtype": np.dtype(dt), "fillvalue": dtype_max(np.dtype(dt)), "width": 5, "height": 4, "statistic": "last"},
              {"time": [datetime(2020, 1, 1), datetime(2020, 1, 2)]})
        place({"mode": "empty", "dtype": np.dtype(dt), "fillvalue": dtype_max(np.dtype(dt)), "width": 5, "height": 4, "statistic": "sum"}, None)
    place({"mode": "time"}, {"time": [datetime(2020, 1, 1)]})
    place({"mode": "meta"}, {"meta": ["a"]})
    place({"mode": "null"})


def main():
    fams = {
        "math": gen_math, "n1": gen_n1, "clip": gen_clip, "mask": gen_mask,
        "classify": gen_classify, "reclassify": gen_reclassify, "tiler": gen_tiler,
        "views": gen_views, "dtypes": gen_dtypes, "smoothzoom": gen_smoothzoom,
        "spatial": gen_spatial, "reduce": gen_reduce, "reducestats": gen_reducestats, "group": gen_group, "smooth": gen_smooth,
        "temporal": gen_temporal, "zonal": gen_zonal, "round3": gen_round3, "round4": gen_round4,
    }
    only = sys.argv[1:]
    counts = {}
    for name, fn in fams.items():
        if only and name not in only:
            continue
        w = Writer()
        with np.errstate(all="ignore"):
            fn(w)
        w.save(os.path.join(HERE, name))
        counts[name] = len(w.cases)
        print(name, len(w.cases), "cases")
    import scipy
    manifest_path = os.path.join(HERE, "manifest.json")
    manifest = {}
    if os.path.exists(manifest_path):
        manifest = json.load(open(manifest_path))
    manifest.update({
        "generator": "tests/golden/make_golden.py",
        "reference": "nens/dask-geomodeling @ /root/reference (v2.8.3.dev0)",
        "python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
    })
    manifest.setdefault("cases", {}).update(counts)
    json.dump(manifest, open(manifest_path, "w"), indent=1)


if __name__ == "__main__":
    main()
