"""CPU ORACLE (test infrastructure) -- NumPy interpreter of a ``gm_program``.

It restates, step by step, what the HIP kernel ``vm_kernel`` (dask_geomodeling_amd/
csrc/gm_vm.h) does with a program, so that CPU-only tests can check the *host*
compiler (expression -> program, register class choice, scalar conversion) against
the reference-generated golden vectors without a GPU, and so that GPU tests can
tell compiler bugs from kernel bugs.  Only tests may import it; it is also what the
``cpu_vm`` test fixture plugs in place of ``program.launch``.

Every step re-applies raster/elemwise.py:263-296 literally (no-data by exact ==,
compute in the step dtype, non-finite -> fill, no-data -> fill).
"""
import ctypes

import numpy as np

CLASS_DTYPE = {0: np.dtype("f4"), 1: np.dtype("f8"), 2: np.dtype("i4"), 3: np.dtype("i8")}
CODE_DTYPE = {0: "bool", 1: "u1", 2: "i1", 3: "u2", 4: "i2", 5: "u4", 6: "i4", 7: "u8", 8: "i8",
              9: "f4", 10: "f8"}
OPD_NONE, OPD_ACC, OPD_INPUT, OPD_SCALAR, OPD_TEMP = range(5)


def _scalar(slot, cls):
    return {0: slot.f32, 1: slot.f64, 2: slot.i32, 3: slot.i64}[cls]


def _table(prog, index):
    """The 8-byte words of table ``index`` of the program (host memory owned by the Compiled object)."""
    t = prog.tables[index]
    if t.n_words == 0:
        return np.zeros(0, dtype=np.uint64)
    return np.ctypeslib.as_array((ctypes.c_uint64 * t.n_words).from_address(t.data)).copy()


def run_program(prog, inputs, n):
    """Interpret ``prog`` over flat ndarrays ``inputs`` (n elements each, or 1 if broadcast)."""
    cls = prog.reg_class
    rc = CLASS_DTYPE[cls]
    regs_in = []
    for k in range(prog.n_inputs):
        d = prog.inputs[k]
        if k >= len(inputs) or inputs[k] is None:
            regs_in.append(None)
            continue
        x = np.asarray(inputs[k]).reshape(-1)
        assert x.dtype == np.dtype(CODE_DTYPE[d.dtype]), (x.dtype, d.dtype)
        if d.broadcast:
            x = np.broadcast_to(x[:1], (n,))
        with np.errstate(all="ignore"):
            if x.dtype == np.dtype("u8") and rc == np.dtype("i8"):
                regs_in.append(x.view("i8") if x.flags.c_contiguous else x.astype("i8"))
            else:
                r = x.astype(rc)
                if d.native_ndv:  # 64-bit integer tile in the F64 class: tag no-data cells
                    r = np.where(x.astype("u8").view("i8") == np.int64(d.ndv.i64), rc.type(0.5), r)
                regs_in.append(r)
    acc = np.zeros(n, dtype=rc)
    temps = [None] * 4

    def fetch(o):
        if o.kind == OPD_SCALAR:
            return np.full(n, _scalar(o.value, cls), dtype=rc), np.zeros(n, dtype=bool)
        if o.kind in (OPD_ACC, OPD_TEMP, OPD_INPUT):
            v = acc if o.kind == OPD_ACC else (temps[o.index] if o.kind == OPD_TEMP else regs_in[o.index])
            nd = (v == rc.type(_scalar(o.value, cls))) if o.has_ndv else np.zeros(n, dtype=bool)
            return v, nd
        return np.zeros(n, dtype=rc), np.zeros(n, dtype=bool)

    for s in range(prog.n_steps):
        S = prog.steps[s]
        a, nda = fetch(S.a)
        b, ndb = fetch(S.b)
        nd = nda | ndb
        fill = rc.type(_scalar(S.fill, cls))
        op = S.opcode
        with np.errstate(all="ignore"):
            if op == 1:
                r = a + b
            elif op == 2:
                r = a - b
            elif op == 3:
                r = a * b
            elif op == 4:
                r = a / b
            elif op == 5:
                if rc.kind == "f":
                    r = np.power(a.astype("f8"), b.astype("f8")).astype(rc)
                else:
                    r = np.power(a, b)
            elif op == 6:
                r = (a == b).astype(rc)
            elif op == 7:
                r = (a != b).astype(rc)
            elif op == 8:
                r = (a > b).astype(rc)
            elif op == 9:
                r = (a >= b).astype(rc)
            elif op == 10:
                r = (a < b).astype(rc)
            elif op == 11:
                r = (a <= b).astype(rc)
            elif op == 12:
                r = ((a != 0) & (b != 0)).astype(rc)
            elif op == 13:
                r = ((a != 0) | (b != 0)).astype(rc)
            elif op == 14:
                r = ((a != 0) ^ (b != 0)).astype(rc)
            elif op in (15, 16, 17):
                f = {15: np.exp, 16: np.log, 17: np.log10}[op]
                r = f(a.astype("f8")).astype(rc)
            elif op == 18:
                r = (a == 0).astype(rc)
                nd = np.zeros(n, dtype=bool)
            elif op == 19:
                r = (~nda).astype(rc)
                nd = np.zeros(n, dtype=bool)
            elif op == 20:
                r = nda.astype(rc)
                nd = np.zeros(n, dtype=bool)
            elif op == 21:  # clip
                drop = ndb.copy()
                if S.b.is_bool:
                    drop |= b == 0
                r = a.copy()
                nd = drop
            elif op == 22:  # mask
                value = rc.type(_scalar(S.aux0, cls))
                if S.flags & 4:
                    nodata = nda
                else:
                    y, tol = S.aux2.f64, S.aux1.f64
                    y_finite = bool(S.flags & 2)
                    if S.flags & 1:
                        x64 = a.astype("f8")
                        nodata = ((np.abs(x64 - y) <= tol) & y_finite) | (x64 == y)
                    else:
                        x32 = a.astype("f4")
                        nodata = ((np.abs(x32 - np.float32(y)) <= np.float32(tol)) & y_finite) | (x32 == np.float32(y))
                r = np.full(n, value, dtype=rc)
                nd = nodata
            elif op == 25:  # fillnodata: b where b is data, else a
                if S.flags & 4:
                    nodata = ndb
                else:
                    y, tol = S.aux2.f64, S.aux1.f64
                    y_finite = bool(S.flags & 2)
                    if S.flags & 1:
                        x64 = b.astype("f8")
                        nodata = ((np.abs(x64 - y) <= tol) & y_finite) | (x64 == y)
                    else:
                        x32 = b.astype("f4")
                        nodata = ((np.abs(x32 - np.float32(y)) <= np.float32(tol)) & y_finite) | (x32 == np.float32(y))
                r = np.where(nodata, a, b)
                nd = np.zeros(n, dtype=bool)
            elif op == 23:  # maskbelow
                r = a.copy()
                nd = a < rc.type(_scalar(S.aux0, cls))
            elif op == 24:  # step
                loc = rc.type(_scalar(S.aux0, cls))
                r = a.copy()
                r = np.where(a > loc, rc.type(_scalar(S.aux3, cls)), r)
                r = np.where(a == loc, rc.type(_scalar(S.aux2, cls)), r)
                r = np.where(a < loc, rc.type(_scalar(S.aux1, cls)), r)
                nd = nda
            elif op == 27:  # digitize (Classify.process, raster/misc.py:391-398)
                nb = int(S.aux1.u64)
                words = _table(prog, int(S.aux0.u64))[:nb]
                right = bool(S.flags & 1)
                if S.flags & 2:  # integer tile, integer edges: exact int64 compares
                    edges, x = words.view("i8"), a.astype("i8")
                else:            # float64 compares of the (exactly converted) value
                    edges, x = words.view("f8"), a.astype("f8")
                r = np.searchsorted(edges, x, side="left" if right else "right").astype(rc)
                if rc.kind == "f":
                    r = np.where(np.isnan(a), rc.type(nb), r)
                nd = nda
            elif op == 28:  # lookup (Reclassify.process, raster/misc.py:505-514)
                kmin, slots, n_exc = int(S.aux1.i64), int(S.aux2.u64), int(S.aux3.u64)
                words = _table(prog, int(S.aux0.u64))
                lut_words = (slots * rc.itemsize + 7) // 8
                lut = words[:lut_words].view(rc)[:slots]
                key = a.astype("i8")
                idx = key - np.int64(kmin)
                inside = (idx >= 0) & (idx < slots)
                miss = np.full(n, fill, dtype=rc) if (S.flags & 1) else a.copy()
                for x in range(n_exc):
                    ek = words[lut_words + 2 * x:lut_words + 2 * x + 1].view("i8")[0]
                    ev = words[lut_words + 2 * x + 1:lut_words + 2 * x + 2].view(rc)[0]
                    miss = np.where(key == ek, ev, miss)
                r = np.where(inside, lut[np.where(inside, idx, 0)], miss)
                nd = np.zeros(n, dtype=bool)
            elif op == 29:  # nanifnodata
                if S.flags & 4:
                    nodata = nda
                else:
                    y, tol = S.aux2.f64, S.aux1.f64
                    y_finite = bool(S.flags & 2)
                    if S.flags & 1:
                        x64 = a.astype("f8")
                        nodata = ((np.abs(x64 - y) <= tol) & y_finite) | (x64 == y)
                    else:
                        x32 = a.astype("f4")
                        nodata = ((np.abs(x32 - np.float32(y)) <= np.float32(tol)) & y_finite) | (x32 == np.float32(y))
                r = np.where(nodata, rc.type(np.nan), a)
                if S.flags & 8:  # float16 work dtype over a wider raster: the assignment rounds (reduction.py:99-107)
                    with np.errstate(over="ignore"):
                        r = r.astype(np.float16).astype(rc)
                nd = np.zeros(n, dtype=bool)
            elif op == 30:  # fmax
                r = np.where((a > b) | np.isnan(b), a, b)  # (ties: the later operand, as maxps does)
                nd = np.zeros(n, dtype=bool)
            elif op == 31:  # nantofill
                r = a.copy()
                nd = np.isnan(a)
            else:  # copy
                r = a.copy()
                if rc.kind == "f" and (S.flags & 1):  # float -> integer store like x86's cvttss2si (gm_vm.h GM_OP_COPY)
                    inside = (a >= rc.type(-2147483648.0)) & (a < rc.type(2147483648.0))
                    r = np.where(inside, a, rc.type(-2147483648.0))
            if cls == 1 and S.round_f32:
                r = r.astype("f4").astype("f8")
            if cls == 3 and S.wrap_i32:
                r = r.astype("i4").astype("i8")
            if rc.kind == "f" and S.check_finite:
                nd = nd | ~np.isfinite(r)
        acc = np.where(nd, fill, r).astype(rc)
        if S.save_temp:
            temps[S.save_temp - 1] = acc.copy()
    out_dt = np.dtype(CODE_DTYPE[prog.out_dtype])
    with np.errstate(all="ignore"):
        if out_dt == np.dtype("u8") and rc == np.dtype("i8"):
            return acc.view("u8")
        if out_dt == np.dtype("i8") and rc == np.dtype("f8"):
            # gm_vm.h to_stored(): 2^63 (the int64 fill value as float64) is stored as INT64_MAX
            top = acc >= 9223372036854775807.0
            return np.where(top, np.int64(9223372036854775807), np.where(top, 0.0, acc).astype("i8"))
        return acc.astype(out_dt)


def launch(compiled, inputs, shape, out=None):
    """Drop-in for ``dask_geomodeling_amd.hip.program.launch`` on host arrays."""
    n = int(np.prod(shape, dtype=np.int64))
    flat = run_program(compiled.program, [np.asarray(x) for x in inputs], n)
    return flat.reshape(shape)
