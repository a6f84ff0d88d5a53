"""Host side of K1: expression trees -> gm_program (include/gmhip.h) -> one launch.

An expression tree mirrors a sub-graph of element-wise Blocks.  ``compile_expr``
picks the register class, linearises the tree into the accumulator program the
kernel interprets and converts every literal / no-data value into the class'
representation *using NumPy's own scalar rules* (so that e.g. ``f32_tile == ndv``
and ``np.add(tile, 2.5, dtype="float32")`` mean exactly what they mean in the
reference, raster/elemwise.py:263-296).
"""
import ctypes

import numpy as np
from math import prod as _prod

from . import runtime
from .array import DeviceArray, as_device, as_device_view
from .dtypes import code_of
from .runtime import gm_scalar

GM_MAX_INPUTS, GM_MAX_STEPS, GM_MAX_TEMPS = 16, 24, 4
GM_MAX_TABLES, GM_MAX_LINEAR_EDGES = 4, 32
LOOKUP_MAX_WINDOW, LOOKUP_MAX_EXCEPTIONS = 8192, 8  # the window lives in LDS: 8192 x 8 B = 64 KiB
CLASS_F32, CLASS_F64, CLASS_I32, CLASS_I64 = 0, 1, 2, 3
CLASS_DTYPE = {CLASS_F32: np.dtype("f4"), CLASS_F64: np.dtype("f8"),
               CLASS_I32: np.dtype("i4"), CLASS_I64: np.dtype("i8")}
CLASS_COST = {CLASS_F32: 0, CLASS_I32: 1, CLASS_F64: 2, CLASS_I64: 3}
OPD_NONE, OPD_ACC, OPD_INPUT, OPD_SCALAR, OPD_TEMP = 0, 1, 2, 3, 4

OPCODES = {
    "add": 1, "subtract": 2, "multiply": 3, "divide": 4, "power": 5,
    "equal": 6, "not_equal": 7, "greater": 8, "greater_equal": 9, "less": 10, "less_equal": 11,
    "logical_and": 12, "logical_or": 13, "logical_xor": 14,
    "exp": 15, "log": 16, "log10": 17,
    "invert": 18, "isdata": 19, "isnodata": 20, "clip": 21, "mask": 22,
    "maskbelow": 23, "step": 24, "fillnodata": 25, "copy": 26, "digitize": 27, "lookup": 28,
    "nanifnodata": 29, "fmax": 30, "nantofill": 31,
}
MATH_OPS = ("add", "subtract", "multiply", "divide", "power", "exp", "log", "log10")
COMPARE_OPS = ("equal", "not_equal", "greater", "greater_equal", "less", "less_equal")
LOGIC_OPS = ("logical_and", "logical_or", "logical_xor")
HEAVY_OPS = ("power", "exp", "log", "log10")


class gm_operand(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("value", gm_scalar)]

    def set(self, kind, index=0, has_ndv=0, is_bool=0):
        self.bits = kind | (index << 8) | (int(bool(has_ndv)) << 16) | (int(bool(is_bool)) << 17)

    kind = property(lambda self: self.bits & 0xff)
    index = property(lambda self: (self.bits >> 8) & 0xff)
    has_ndv = property(lambda self: (self.bits >> 16) & 1)
    is_bool = property(lambda self: (self.bits >> 17) & 1)


class gm_step(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("a", gm_operand), ("b", gm_operand),
                ("fill", gm_scalar), ("aux0", gm_scalar), ("aux1", gm_scalar), ("aux2", gm_scalar),
                ("aux3", gm_scalar)]

    def _get(self, shift, mask):
        return (self.bits >> shift) & mask

    def _put(self, shift, mask, value):
        self.bits = (self.bits & ~(mask << shift)) | ((int(value) & mask) << shift)

    opcode = property(lambda s: s._get(0, 0xff), lambda s, v: s._put(0, 0xff, v))
    save_temp = property(lambda s: s._get(8, 0xff), lambda s, v: s._put(8, 0xff, v))
    flags = property(lambda s: s._get(16, 0xff), lambda s, v: s._put(16, 0xff, v))
    round_f32 = property(lambda s: s._get(24, 1), lambda s, v: s._put(24, 1, v))
    wrap_i32 = property(lambda s: s._get(25, 1), lambda s, v: s._put(25, 1, v))
    check_finite = property(lambda s: s._get(26, 1), lambda s, v: s._put(26, 1, v))


class gm_input_desc(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("ndv", gm_scalar)]

    def set(self, dtype_code, broadcast=0, native_ndv=0):
        self.bits = dtype_code | (int(bool(broadcast)) << 8) | (int(bool(native_ndv)) << 9)

    dtype = property(lambda self: self.bits & 0xff)
    broadcast = property(lambda self: (self.bits >> 8) & 1)
    native_ndv = property(lambda self: (self.bits >> 9) & 1)


class gm_table(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("n_words", ctypes.c_uint64)]


class gm_program(ctypes.Structure):
    _fields_ = [("n_steps", ctypes.c_uint32), ("n_inputs", ctypes.c_uint32), ("reg_class", ctypes.c_uint32),
                ("out_dtype", ctypes.c_uint32),
                ("inputs", gm_input_desc * GM_MAX_INPUTS), ("steps", gm_step * GM_MAX_STEPS),
                ("n_tables", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("tables", gm_table * GM_MAX_TABLES)]


class gm_vm_tuning(ctypes.Structure):
    _fields_ = [("vec", ctypes.c_int32), ("max_batch", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2)]


class FusionLimit(Exception):
    """The expression does not fit one program (class conflict, too many inputs/steps/temps)."""


# ----------------------------------------------------------------------------
# NumPy-exact scalar handling
# ----------------------------------------------------------------------------

_EFFECTIVE_NDV = {}  # (dtype, what the no-data value is) -> effective_ndv: asked with every stencil / plan call


def effective_ndv(dtype, ndv):
    """The value v (of ``dtype``) with ``tile == ndv``  <=>  ``tile == v``, or None if no
    element of a ``dtype`` tile can ever compare equal to ``ndv``.  NumPy itself decides
    (weak python scalars, promotion of NumPy scalars) on a 1-element array."""
    if ndv is None:
        return None
    dtype = np.dtype(dtype)
    try:
        key = (dtype.num, _scalar_key(ndv))
        hit = _EFFECTIVE_NDV.get(key, _EFFECTIVE_NDV)
    except TypeError:
        key, hit = None, _EFFECTIVE_NDV
    if hit is not _EFFECTIVE_NDV:
        return hit
    value = _effective_ndv(dtype, ndv)
    if key is not None:
        if len(_EFFECTIVE_NDV) > 1024:
            _EFFECTIVE_NDV.clear()
        _EFFECTIVE_NDV[key] = value
    return value


def _effective_ndv(dtype, ndv):
    with np.errstate(all="ignore"):
        try:
            cand = np.array(ndv).astype(dtype)
        except (OverflowError, ValueError, TypeError):
            return None
        try:
            hit = bool((np.array([cand], dtype=dtype) == ndv)[0])
        except (OverflowError, TypeError):
            hit = False
    return cand[()] if hit else None


def cast_scalar(value, dtype):
    """A python/NumPy scalar operand as NumPy casts it for ``ufunc(..., dtype=dtype)``."""
    with np.errstate(all="ignore"):
        return np.array(value).astype(np.dtype(dtype))[()]


def _set_scalar(slot, cls, value):
    """Write ``value`` (already of a dtype exactly representable in the class) into a gm_scalar."""
    if cls == CLASS_F32:
        slot.f32 = float(np.float32(value))
    elif cls == CLASS_F64:
        slot.f64 = float(value)
    elif cls == CLASS_I32:
        slot.i32 = int(np.array(int(value)).astype("i8").astype("i4"))
    else:
        v = int(value)
        if v >= 2 ** 63:  # uint64 bit pattern
            v -= 2 ** 64
        slot.i64 = v


def exact_in_class(dtype, cls):
    """Every value of ``dtype`` is exactly representable in the register class."""
    dt = np.dtype(dtype)
    if dt == bool:
        return True
    size = dt.itemsize
    if cls == CLASS_F32:
        return (dt.kind == "f" and size <= 4) or (dt.kind in "iu" and size <= 2)
    if cls == CLASS_F64:
        return dt.kind == "f" or (dt.kind in "iu" and size <= 4)
    if cls == CLASS_I32:
        return (dt.kind == "i" and size <= 4) or (dt.kind == "u" and size <= 2)
    if cls == CLASS_I64:
        return dt.kind in "iu"  # uint64 travels as its bit pattern (move-only)
    return False


ALL_CLASSES = frozenset((CLASS_F32, CLASS_F64, CLASS_I32, CLASS_I64))


def classes_exact(dtype):
    return frozenset(c for c in ALL_CLASSES if exact_in_class(dtype, c))


def classes_for_compare(dtype):
    """Classes in which a comparison carried out in ``dtype`` is exact."""
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return frozenset((CLASS_F64,)) if dt.itemsize == 8 else frozenset((CLASS_F32, CLASS_F64))
    return frozenset(c for c in ALL_CLASSES if exact_in_class(dt, c))


def classes_for_compute(dtype):
    """Classes able to *compute* in ``dtype`` bit-exactly (narrower step types are emulated:
    f32 inside F64 by rounding after each op, i32 inside I64 by wrapping)."""
    dt = np.dtype(dtype)
    if dt == np.dtype("f4"):
        return frozenset((CLASS_F32, CLASS_F64))
    if dt == np.dtype("f8"):
        return frozenset((CLASS_F64,))
    if dt == np.dtype("i4"):
        return frozenset((CLASS_I32, CLASS_I64))
    if dt == np.dtype("i8"):
        return frozenset((CLASS_I64,))
    raise NotImplementedError("no register class computes in %s" % dt)


# ----------------------------------------------------------------------------
# expression nodes
# ----------------------------------------------------------------------------

class Node:
    dtype = None   # result dtype (np.dtype)
    ndv = None     # result no_data_value as the reference would put in the dict

    @property
    def is_bool(self):
        return self.dtype == np.dtype("bool")


class Leaf(Node):
    """A tile that arrives from outside the program (an input slot).  ``values`` is the
    device tile (bound at evaluation time); ``slot``/``broadcast`` are assigned by ``evaluate``."""

    def __init__(self, slot, dtype, ndv, broadcast=False, values=None):
        self.slot, self.dtype, self.ndv, self.broadcast = slot, np.dtype(dtype), ndv, broadcast
        self.values = values

    def key(self):
        return ("leaf", self.slot, self.dtype.str, _scalar_key(self.ndv), self.broadcast)


class Lit(Node):
    """A python / NumPy scalar operand."""

    def __init__(self, value):
        self.value = value
        self.dtype = np.result_type(value) if isinstance(value, np.generic) else None

    def key(self):
        return ("lit", _scalar_key(self.value))


class Op(Node):
    def __init__(self, op, args, dtype, ndv, fill=None, extra=None):
        self.op, self.args, self.dtype, self.ndv = op, list(args), np.dtype(dtype), ndv
        self.fill, self.extra = fill, extra

    def key(self):
        return ("op", self.op, self.dtype.str, _scalar_key(self.ndv), _scalar_key(self.fill),
                _scalar_key(self.extra), tuple(a.key() for a in self.args))


def _scalar_key(v):
    if v is None:
        return None
    if isinstance(v, (list, tuple)):
        return tuple(_scalar_key(x) for x in v)
    if isinstance(v, dict):
        return tuple(sorted((k, _scalar_key(x)) for k, x in v.items()))
    if isinstance(v, np.ndarray):
        return ("nd", v.dtype.str, v.shape, v.tobytes())
    if isinstance(v, np.generic):
        return (v.dtype.str, v.tobytes())
    if isinstance(v, float):
        return ("f", v.hex())
    return (type(v).__name__, v)


def compute_dtype(node):
    """The dtype an Op's arithmetic / comparison is carried out in."""
    if node.op in MATH_OPS or node.op == "fillnodata":
        return node.dtype
    if node.op in COMPARE_OPS:
        parts = []
        for a in node.args:
            parts.append(a.value if isinstance(a, Lit) and not isinstance(a.value, np.generic) else a.dtype)
        try:
            dt = np.result_type(*parts)
        except OverflowError:  # python int outside the raster dtype: NumPy compares exactly
            dt = np.dtype("i8")
        if dt == bool:
            return np.dtype("i4")
        if dt.kind in "iu":
            return np.dtype("i4") if exact_in_class(dt, CLASS_I32) else np.dtype("i8")
        if dt.kind == "f" and dt.itemsize < 4:
            return np.dtype("f4")
        return dt
    return None


def node_classes(node):
    """Admissible register classes of one node (not counting its children)."""
    if isinstance(node, Lit):
        return ALL_CLASSES
    if isinstance(node, Leaf):
        return ALL_CLASSES  # constrained by its consumers below
    op = node.op
    allowed = ALL_CLASSES
    if op in MATH_OPS or op in COMPARE_OPS:
        cd = compute_dtype(node)
        if cd.kind == "u" and cd.itemsize == 8:
            raise FusionLimit("uint64 arithmetic is not supported")
        allowed = classes_for_compute(cd)
        if op in COMPARE_OPS and cd.kind in "iu":
            # small integers compare exactly in the float classes too
            allowed = allowed | frozenset(c for c in (CLASS_F64, CLASS_F32, CLASS_I64)
                                          if all(_operand_exact(a, c) for a in node.args))
    elif op in ("maskbelow", "step"):
        # the tile travels through unchanged and is compared with a literal in NumPy's promoted dtype
        allowed = classes_exact(node.args[0].dtype) & classes_for_compare(node.extra["compare_dtype"])
        for v in node.extra.get("values", ()):
            allowed = frozenset(c for c in allowed if _value_exact(v, node.dtype, c))
    elif op == "fillnodata":
        # (integer results: nothing is computed -- a cell is taken from one raster or the other --, so any class that holds
        #  the values will do: uint32 in I64 / F64, uint64 as its bit pattern in I64; round 4, found refused by the fuzzing)
        allowed = classes_for_compute(node.dtype) if node.dtype.kind == "f" else classes_exact(node.dtype)
        if node.dtype.kind != "f" and any(not isinstance(a, Lit) and a.dtype.kind == "f" for a in node.args):
            # float tiles pasted into an integer result (reduce_rasters 'first' / 'last' with an integer output
            # dtype: out[index] = values[index] casts on assignment): nothing is computed on the way, so the values
            # travel as float64 and the truncating cast happens once, at the store
            allowed = frozenset((CLASS_F64,))
        for a in node.args:
            if not isinstance(a, Lit):
                # (a 64-bit integer tile pasted into a float64 result is cast like NumPy casts it; its no-data
                #  cells are tagged at load time: gm_input_desc.native_ndv)
                wide_into_f64 = _wide_int(a.dtype) and node.dtype == np.dtype("f8")
                allowed = allowed & (classes_exact(a.dtype) | (frozenset((CLASS_F64,)) if wide_into_f64 else frozenset()))
    elif op in ("nanifnodata", "fmax", "nantofill"):
        # reduce_rasters works in result_type(dtype, float16) (reduction.py:99-101): values are exact
        # there and so they are in the float class that holds the work dtype
        work = node.extra["work"]
        allowed = frozenset((CLASS_F32, CLASS_F64)) if work.itemsize <= 4 else frozenset((CLASS_F64,))
    elif op == "digitize":
        # Classify: the value is compared with the edges in NumPy's promoted dtype -- float tiles in a
        # float class (float64 edges; the F32 class uses directed-rounded float32 thresholds),
        # integer tiles in an integer class (int64 edges, or float64 edges against (double)value)
        src = node.args[0].dtype
        if src == np.dtype("bool") or (src.kind == "u" and src.itemsize == 8):
            raise FusionLimit("Classify of %s tiles runs as its own kernel" % src)
        float_side = frozenset((CLASS_F32, CLASS_F64)) if src.kind == "f" else frozenset((CLASS_I32, CLASS_I64))
        allowed = classes_exact(src) & float_side
    elif op == "lookup":
        # every target (and the value of a miss) must be exact in the class: plain Reclassify tables hold
        # int64 / float64 targets incl. the dtype's max (I64 / F64); a table folded with a following
        # Mask holds a few small integers and fits any class -- also the float32 class of the
        # arithmetic that feeds a Classify in front of it
        ex = node.extra
        entries = np.concatenate([np.ravel(ex["lut"]), np.ravel(np.asarray(ex["exc_vals"], dtype=ex["lut"].dtype)),
                                  np.asarray([ex.get("miss", node.ndv)], dtype=ex["lut"].dtype)])
        ok = []
        for c in ALL_CLASSES:
            with np.errstate(all="ignore"):
                back = entries.astype(CLASS_DTYPE[c])
                same = (back.astype(np.float64) == entries.astype(np.float64)) & (back.astype(entries.dtype) == entries)
            if not (bool(np.all(same)) and (entries.dtype.kind != "f" or c in (CLASS_F32, CLASS_F64))):
                continue
            if c in (CLASS_F32, CLASS_I32):
                # key arithmetic happens in int32 there: the window must lie inside the keys the class can hold
                limit = 2 ** 24 if c == CLASS_F32 else 2 ** 31 - 1
                if not (-limit <= ex["kmin"] and ex["kmin"] + len(ex["lut"]) - 1 <= limit):
                    continue
            ok.append(c)
        allowed = frozenset(ok) & classes_exact(node.args[0].dtype)
    elif op in ("clip", "copy"):
        allowed = classes_exact(node.args[0].dtype)
    elif op == "mask":
        store = node.args[0]
        if store.dtype.kind == "f":
            allowed = classes_exact(store.dtype)
        value = node.extra["value"]
        vdt = node.dtype
        allowed = frozenset(c for c in allowed if _value_exact(value, vdt, c))
    # Tiles are converted to the class type on load and tested for no-data on the converted
    # value, so the class must hold every leaf exactly -- except 64-bit integer tiles feeding
    # a float64 op: NumPy casts them (rounding) too, and their no-data cells are tagged at
    # load time (gm_input_desc.native_ndv)
    cd = compute_dtype(node)
    casts_to_f64 = (cd is not None and cd == np.dtype("f8")) or \
        op in ("mask", "isdata", "isnodata") or \
        (op == "nanifnodata" and np.dtype(node.extra["work"]) == np.dtype("f8"))
    # (mask / isdata / isnodata only ask "is the cell no data?": the tag suffices; nanifnodata into a float64
    #  work dtype is the reference's own values.astype(float64), reduction.py:99-107)
    for a in node.args:
        if isinstance(a, Leaf):
            allowed = frozenset(c for c in allowed if exact_in_class(a.dtype, c)
                                or (c == CLASS_F64 and casts_to_f64 and _wide_int(a.dtype)))
    return allowed


def _wide_int(dtype):
    dt = np.dtype(dtype)
    return dt.kind in "iu" and dt.itemsize == 8


def _operand_exact(a, cls):
    if isinstance(a, Lit):
        try:
            v = a.value
            return bool(np.array(v).astype(CLASS_DTYPE[cls]) == v)
        except (OverflowError, TypeError):
            return False
    return exact_in_class(a.dtype, cls)


def _value_exact(value, dtype, cls):
    v = cast_scalar(value, dtype)
    with np.errstate(all="ignore"):
        try:
            return bool(np.array(v).astype(CLASS_DTYPE[cls]) == v)
        except (OverflowError, TypeError):
            return False


def _leaf_use_exact(consumer, index):
    """Does this use of a Leaf need its exact value (as opposed to a cast or its no-data bits)?"""
    op = consumer.op
    if op in ("clip", "copy"):
        return index == 0
    if op == "mask":
        return consumer.args[0].dtype.kind == "f"
    if op in ("isdata", "isnodata"):
        return False
    return False  # math / compare casts are handled in node_classes


def tree_classes(root):
    """Intersection of admissible classes over the whole tree (empty -> cannot fuse)."""
    allowed = ALL_CLASSES
    seen = set()
    stack = [root]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        allowed = allowed & node_classes(n)
        if isinstance(n, Op):
            stack.extend(n.args)
    return allowed


# ----------------------------------------------------------------------------
# compiler
# ----------------------------------------------------------------------------

class Compiled:
    __slots__ = ("program", "out_dtype", "out_ndv", "n_inputs", "reg_class", "heavy", "n_steps", "tables")


def compile_expr(root, n_inputs):
    """Linearise ``root`` into a gm_program.  Raises FusionLimit if it does not fit."""
    if isinstance(root, (Leaf, Lit)):
        if isinstance(root, Lit):
            raise FusionLimit("literal root")
        root = Op("copy", [root], root.dtype, root.ndv)
    allowed = tree_classes(root)
    if not allowed:
        if isinstance(root, Op) and all(isinstance(a, (Leaf, Lit)) for a in root.args):
            # a single op that no class can run (e.g. int64 raster combined with a float raster:
            # the int64 tile is not exact in float64 and float64 does not compute in int64)
            raise NotImplementedError("operand dtypes %s of '%s' have no exact register class on the device"
                                      % ([str(a.dtype) for a in root.args], root.op))
        raise FusionLimit("no common register class")
    cls = min(allowed, key=lambda c: CLASS_COST[c])
    if n_inputs > GM_MAX_INPUTS:
        raise FusionLimit("too many inputs")

    prog = gm_program()
    prog.reg_class = cls
    prog.n_inputs = n_inputs
    steps = []
    tables = []  # uint64 word arrays, kept alive by the Compiled object

    def add_table(words):
        if len(tables) >= GM_MAX_TABLES:
            raise FusionLimit("too many tables")
        tables.append(np.ascontiguousarray(words, dtype=np.uint64))
        return len(tables) - 1

    # consumers per shared Op node
    refs = {}
    order = []

    def count(n):
        if isinstance(n, Op):
            refs[id(n)] = refs.get(id(n), 0) + 1
            if refs[id(n)] == 1:
                for a in n.args:
                    count(a)
                order.append(n)

    count(root)
    temp_of = {}           # id(node) -> temp index while it is live
    remaining = dict(refs)  # uses still to come
    free_temps = list(range(GM_MAX_TEMPS))
    leaves = {}

    def alloc_temp():
        if not free_temps:
            raise FusionLimit("out of temps")
        return free_temps.pop(0)

    def release(node):
        if id(node) in temp_of:
            remaining[id(node)] -= 1
            if remaining[id(node)] == 0:
                free_temps.insert(0, temp_of.pop(id(node)))

    def is_simple(n):
        return isinstance(n, (Leaf, Lit)) or id(n) in temp_of

    def reg_operand(kind, index, node):
        o = gm_operand()
        eff = None if node.is_bool else effective_ndv(node.dtype, node.ndv)
        o.set(kind, index, has_ndv=eff is not None, is_bool=node.is_bool)
        if eff is not None:
            _set_scalar(o.value, cls, eff)
        elif cls in (CLASS_F32, CLASS_F64):
            _set_scalar(o.value, cls, float("nan"))  # float classes: "no ndv" == compare with NaN
        return o

    def operand(n, consumer, cd):
        if isinstance(n, Leaf):
            leaves[n.slot] = n
            if exact_in_class(n.dtype, cls):
                return reg_operand(OPD_INPUT, n.slot, n)
            if not (cls == CLASS_F64 and _wide_int(n.dtype)):
                raise FusionLimit("input dtype %s is not exact in register class %d" % (n.dtype, cls))
            o = gm_operand()  # no-data cells arrive as the sentinel 0.5
            has = effective_ndv(n.dtype, n.ndv) is not None
            o.set(OPD_INPUT, n.slot, has_ndv=has)
            o.value.f64 = 0.5 if has else float("nan")
            return o
        if isinstance(n, Lit):
            o = gm_operand()
            o.set(OPD_SCALAR)
            target = cd if cd is not None else CLASS_DTYPE[cls]
            _set_scalar(o.value, cls, cast_scalar(n.value, target))
            return o
        return reg_operand(OPD_TEMP, temp_of[id(n)], n)

    def emit(node, a_opd, b_opd):
        if len(steps) >= GM_MAX_STEPS:
            raise FusionLimit("too many steps")
        s = gm_step()
        s.opcode = OPCODES[node.op]
        s.a = a_opd
        if b_opd is not None:
            s.b = b_opd
        cd = compute_dtype(node)
        if node.op in MATH_OPS:
            s.check_finite = 1 if cls in (CLASS_F32, CLASS_F64) else 0
            s.round_f32 = 1 if (cls == CLASS_F64 and cd == np.dtype("f4")) else 0
            s.wrap_i32 = 1 if (cls == CLASS_I64 and cd == np.dtype("i4")) else 0
            _set_scalar(s.fill, cls, cast_scalar(node.fill, cd))
            if node.op == "power" and cls in (CLASS_I32, CLASS_I64):
                # NumPy raises for a negative integer exponent anywhere in the array; the kernel squares and
                # multiplies with an unsigned exponent.  A literal is checked here, a raster exponent by the caller
                # before the launch (devops.math_process sets `exponent_checked`)
                b = node.args[1]
                checked = bool(node.extra and node.extra.get("exponent_checked"))
                if isinstance(b, Lit):
                    if not (isinstance(b.value, (int, np.integer)) and b.value >= 0):
                        raise NotImplementedError("integer Power needs a non-negative integer exponent")
                elif not checked:
                    raise NotImplementedError("integer Power with a raster exponent runs op by op "
                                              "(its exponents are checked for negative values first)")
        elif node.op in COMPARE_OPS or node.op in LOGIC_OPS:
            _set_scalar(s.fill, cls, 1 if node.op == "not_equal" else 0)
        elif node.op == "clip":
            store = node.args[0]
            eff = effective_ndv(store.dtype, store.ndv)
            if eff is None:
                # nothing can be marked: the reference writes the ndv as is
                eff = cast_scalar(store.ndv, store.dtype)
            _set_scalar(s.fill, cls, eff)
        elif node.op == "fillnodata":
            ex = node.extra
            s.flags = ex["flags"]
            s.aux1.f64 = ex["tol"]
            s.aux2.f64 = ex["y"]
        elif node.op == "copy" and node.extra and node.extra.get("indefinite"):
            s.flags = 1  # float -> integer store with x86's out-of-range result (gm_vm.h GM_OP_COPY)
        elif node.op == "maskbelow":
            ex = node.extra
            store = node.args[0]
            eff = effective_ndv(store.dtype, store.ndv)
            _set_scalar(s.fill, cls, eff if eff is not None else cast_scalar(store.ndv, store.dtype))
            _set_scalar(s.aux0, cls, cast_scalar(ex["value"], ex["compare_dtype"]))
        elif node.op == "step":
            ex = node.extra
            store = node.args[0]
            eff = effective_ndv(store.dtype, store.ndv)
            _set_scalar(s.fill, cls, eff if eff is not None else cast_scalar(store.ndv, store.dtype))
            _set_scalar(s.aux0, cls, cast_scalar(ex["value"], ex["compare_dtype"]))
            left, at, right = ex["values"]
            _set_scalar(s.aux1, cls, cast_scalar(left, node.dtype))
            _set_scalar(s.aux2, cls, cast_scalar(at, node.dtype))
            _set_scalar(s.aux3, cls, cast_scalar(right, node.dtype))
        elif node.op == "nanifnodata":
            ex = node.extra
            s.flags = ex["flags"]
            s.aux1.f64 = ex["tol"]
            s.aux2.f64 = ex["y"]
        elif node.op == "nantofill":
            # (extra["fill"]: the value NaN becomes when the node itself declares no no-data value -- an intermediate
            #  that a cast follows)
            value = node.extra["fill"] if node.extra and "fill" in node.extra else node.ndv
            _set_scalar(s.fill, cls, cast_scalar(value, node.dtype))
        elif node.op == "digitize":
            ex = node.extra
            edges = ex["edges"]
            if cls in (CLASS_F32, CLASS_F64) and edges.dtype != np.dtype("f8"):
                raise FusionLimit("integer edges in a float class")
            _set_scalar(s.fill, cls, node.ndv)
            s.aux0.u64 = add_table(edges.view(np.uint64))
            s.aux1.u64 = int(edges.size)
            s.flags = (1 if ex["right"] else 0) | (2 if edges.dtype == np.dtype("i8") else 0)
        elif node.op == "lookup":
            if any(OPCODES["lookup"] == t.opcode for t in steps):
                raise FusionLimit("one lookup per program (its window is staged in LDS)")
            ex = node.extra
            rc = CLASS_DTYPE[cls]
            lut = np.ascontiguousarray(ex["lut"].astype(rc))
            window = np.zeros((lut.nbytes + 7) // 8, dtype=np.uint64)
            window.view(np.uint8)[:lut.nbytes] = lut.view(np.uint8)
            exc_keys = np.asarray(ex["exc_keys"], dtype=np.int64)
            exc_vals = np.asarray(ex["exc_vals"]).astype(rc)
            if rc.itemsize == 4:  # keys the accumulator cannot hold can never match: drop them
                keep = (exc_keys >= -2 ** 31) & (exc_keys <= 2 ** 31 - 1)
                exc_keys, exc_vals = exc_keys[keep], exc_vals[keep]
            exc = np.zeros(2 * len(exc_keys), dtype=np.uint64)
            exc[0::2] = exc_keys.view(np.uint64)
            vals = np.zeros(len(exc_keys), dtype=np.uint64)
            vals.view(rc)[::8 // rc.itemsize] = exc_vals
            exc[1::2] = vals
            _set_scalar(s.fill, cls, ex.get("miss", node.ndv))
            s.aux0.u64 = add_table(np.concatenate([window, exc]))
            s.aux1.i64 = int(ex["kmin"])
            s.aux2.u64 = int(lut.size)
            s.aux3.u64 = len(exc_keys)
            s.flags = 1 if ex["select"] else 0
        elif node.op == "mask":
            ex = node.extra
            _set_scalar(s.fill, cls, ex["fill"])
            _set_scalar(s.aux0, cls, cast_scalar(ex["value"], node.dtype))
            s.flags = ex["flags"]
            s.aux1.f64 = ex["tol"]
            s.aux2.f64 = ex["y"]
        steps.append(s)
        return s

    def eval_to_acc(node):
        """Emit the steps that leave ``node`` in the accumulator (and in a temp if shared)."""
        if not isinstance(node, Op):
            cp = Op("copy", [node], node.dtype, node.ndv)
            emit(cp, operand(node, cp, None), None)
            return
        cd = compute_dtype(node)
        args = node.args
        pending = [a for a in args if not is_simple(a)]
        if len(pending) == 2 and pending[0] is not pending[1]:
            # both operands are sub-trees: the RIGHT one is evaluated first and parked in a temp, so that
            # the left one ends up in the accumulator -- op(acc, temp) is a fast route of the kernel for
            # every arithmetic op and for Clip (whose mask typically is such a sub-tree: Greater(D, 20))
            x, y = pending
            eval_to_acc(y)
            if not is_simple(y):  # park the right value around the left sub-tree
                t = alloc_temp()
                steps[-1].save_temp = t + 1
                temp_of[id(y)] = t
                remaining[id(y)] = 1
            eval_to_acc(x)
        elif pending:
            eval_to_acc(pending[0])
        opds = []
        for a in args:
            if is_simple(a):
                opds.append(operand(a, node, cd))
            else:
                opds.append(reg_operand(OPD_ACC, 0, a))
        s = emit(node, opds[0], opds[1] if len(opds) > 1 else None)
        for a in args:
            release(a)
        if refs.get(id(node), 1) > 1 and id(node) not in temp_of:
            t = alloc_temp()
            s.save_temp = t + 1
            temp_of[id(node)] = t
            remaining[id(node)] = refs[id(node)]

    eval_to_acc(root)
    prog.n_steps = len(steps)
    for i, s in enumerate(steps):
        prog.steps[i] = s
    for slot, leaf in leaves.items():
        d = prog.inputs[slot]
        native = False
        if not exact_in_class(leaf.dtype, cls):
            eff = effective_ndv(leaf.dtype, leaf.ndv)
            if eff is not None:
                native = True
                v = int(eff)
                d.ndv.i64 = v - 2 ** 64 if v >= 2 ** 63 else v
        d.set(code_of(leaf.dtype), leaf.broadcast, native)
    for slot in range(n_inputs):
        if slot not in leaves:  # unused input: still needs a valid dtype code
            prog.inputs[slot].set(code_of(np.dtype("u1")))
    prog.out_dtype = code_of(root.dtype)
    prog.n_tables = len(tables)
    for t, words in enumerate(tables):
        prog.tables[t].data = words.ctypes.data
        prog.tables[t].n_words = words.size
    out = Compiled()
    out.tables = tables
    out.program, out.out_dtype, out.out_ndv = prog, root.dtype, root.ndv
    out.n_inputs, out.reg_class, out.n_steps = n_inputs, cls, len(steps)
    out.heavy = any(OPCODES[k] == s.opcode for s in steps for k in HEAVY_OPS)
    return out


_CACHE = {}


def compile_cached(root, n_inputs):
    key = (root.key(), n_inputs)
    hit = _CACHE.get(key)
    if hit is None:
        hit = compile_expr(root, n_inputs)
        if len(_CACHE) > 512:
            _CACHE.clear()
        _CACHE[key] = hit
    return hit


# ----------------------------------------------------------------------------
# launch
# ----------------------------------------------------------------------------

_declared = False


def _lib():
    global _declared
    L = runtime.lib()
    if not _declared:
        L.gm_elemwise_run.argtypes = [ctypes.POINTER(gm_program), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.c_void_p, ctypes.c_size_t]
        L.gm_elemwise_run.restype = ctypes.c_int
        L.gm_elemwise_run_batch.argtypes = [ctypes.POINTER(gm_program), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        L.gm_elemwise_run_batch.restype = ctypes.c_int
        L.gm_elemwise_run_batch_2d.argtypes = [ctypes.POINTER(gm_program), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_size_t]
        L.gm_elemwise_run_batch_2d.restype = ctypes.c_int
        L.gm_elemwise_run_batch_bands.argtypes = [ctypes.POINTER(gm_program), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                                  ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32),
                                                  ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t]
        L.gm_elemwise_run_batch_bands.restype = ctypes.c_int
        L.gm_vm_set_tuning.argtypes = [ctypes.POINTER(gm_vm_tuning)]
        L.gm_vm_get_tuning.argtypes = [ctypes.POINTER(gm_vm_tuning)]
        L.gm_sizeof.argtypes = [ctypes.c_int]
        L.gm_sizeof.restype = ctypes.c_size_t
        L.gm_spec_set_mode.argtypes = [ctypes.c_int, ctypes.c_uint64]
        L.gm_spec_get_mode.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
        L.gm_spec_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 4
        L.gm_spec_last_kernel.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.gm_spec_pending.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
        L.gm_elemwise_call_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.gm_spec_timing.argtypes = [ctypes.c_int]
        L.gm_spec_timing_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        L.gm_spec_source.argtypes = [ctypes.POINTER(gm_program), ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.gm_spec_compile_check.argtypes = [ctypes.POINTER(gm_program), ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_size_t)]
        for which, struct in enumerate((gm_program, gm_step, gm_operand, gm_input_desc)):
            if L.gm_sizeof(which) != ctypes.sizeof(struct):
                raise runtime.GmHipError("ABI mismatch: sizeof(%s) is %d in Python, %d in libgmhip" % (
                    struct.__name__, ctypes.sizeof(struct), L.gm_sizeof(which)))
        _declared = True
    return L


def set_tuning(vec=0, max_batch=0):
    """Measurement knobs of the VM launch (defaults = the measured optimum, see gmhip.h)."""
    t = gm_vm_tuning(vec, max_batch)
    runtime.check(_lib().gm_vm_set_tuning(ctypes.byref(t)), "gm_vm_set_tuning")


def get_tuning():
    t = gm_vm_tuning()
    runtime.check(_lib().gm_vm_get_tuning(ctypes.byref(t)), "gm_vm_get_tuning")
    return {"vec": t.vec, "max_batch": t.max_batch}


def set_spec_mode(mode, hot_elements=0):
    """The program specialiser (csrc/gm_spec.hip): 0 = off (interpreter kernels only), 1 = programs that
    have been asked for ``hot_elements`` pixels get a kernel of their own (default), 2 = every program at its
    first launch (tests)."""
    runtime.check(_lib().gm_spec_set_mode(int(mode), int(hot_elements)), "gm_spec_set_mode")


def get_spec_mode():
    mode, hot = ctypes.c_int(0), ctypes.c_uint64(0)
    runtime.check(_lib().gm_spec_get_mode(ctypes.byref(mode), ctypes.byref(hot)), "gm_spec_get_mode")
    return mode.value, hot.value


def spec_stats():
    v = [ctypes.c_uint64(0) for _ in range(4)]
    runtime.check(_lib().gm_spec_stats(*[ctypes.byref(x) for x in v]), "gm_spec_stats")
    return dict(zip(("compiled", "disk_hits", "launches", "failed"), (x.value for x in v)))


def spec_last_kernel():
    buf = ctypes.create_string_buffer(64)
    runtime.check(_lib().gm_spec_last_kernel(buf, 64), "gm_spec_last_kernel")
    return buf.value.decode()


def spec_available():
    return bool(_lib().gm_spec_available())


def spec_pending():
    """Number of kernels the library's worker thread is still compiling / loading."""
    n = ctypes.c_uint64(0)
    runtime.check(_lib().gm_spec_pending(ctypes.byref(n)), "gm_spec_pending")
    return n.value


def call_stats(reset=False):
    """-> (program launches so far, the longest time one of them kept its caller inside the library, in ms)."""
    calls, longest = ctypes.c_uint64(0), ctypes.c_double(0.0)
    runtime.check(_lib().gm_elemwise_call_stats(ctypes.byref(calls), ctypes.byref(longest), 1 if reset else 0),
                  "gm_elemwise_call_stats")
    return calls.value, longest.value


def spec_wait():
    """Block until no kernel is being compiled in the background any more."""
    runtime.check(_lib().gm_spec_wait(), "gm_spec_wait")


def spec_timing(enable):
    """Bracket every specialised launch with events stamped by the kernel itself (kernel-only durations)."""
    runtime.check(_lib().gm_spec_timing(1 if enable else 0), "gm_spec_timing")


def spec_timing_read():
    """-> (sum of the kernel durations in microseconds, number of launches) since the last call."""
    total, n = ctypes.c_double(0.0), ctypes.c_uint64(0)
    runtime.check(_lib().gm_spec_timing_read(ctypes.byref(total), ctypes.byref(n)), "gm_spec_timing_read")
    return total.value, n.value


def spec_source(compiled, strided=False, group=2):
    """The HIP source of the kernel the specialiser generates for a compiled program (no device needed)."""
    L = _lib()
    needed = ctypes.c_size_t(0)
    runtime.check(L.gm_spec_source(ctypes.byref(compiled.program), int(strided), int(group), None, 0,
                                   ctypes.byref(needed)), "gm_spec_source")
    buf = ctypes.create_string_buffer(needed.value)
    runtime.check(L.gm_spec_source(ctypes.byref(compiled.program), int(strided), int(group), buf, needed.value,
                                   ctypes.byref(needed)), "gm_spec_source")
    return buf.value.decode()


def spec_compile_check(compiled, strided=False, group=2):
    """Generate + compile (hiprtc, gfx950) the specialised kernel of a program without a device ->
    size of the code object in bytes."""
    size = ctypes.c_size_t(0)
    runtime.check(_lib().gm_spec_compile_check(ctypes.byref(compiled.program), int(strided), int(group),
                                               ctypes.byref(size)), "gm_spec_compile_check")
    return size.value


def launch(compiled, inputs, shape, out=None):
    """Run a compiled program over device tiles ``inputs`` -> DeviceArray of ``shape``."""
    return launch_batch(compiled, [inputs], shape, outs=None if out is None else [out])[0]


def launch_batch(compiled, tiles, shape, outs=None):
    """Run one compiled program over several tiles at once: ``tiles`` is a list of per-tile input
    lists (all tiles of ``shape``) -> list of DeviceArrays.  One launch per 16 tiles.  Inputs may be
    zero-copy windows (DeviceArray.pitch) of larger resident rasters, and so may the ``outs``
    (tiles written straight into their mosaic)."""
    L = _lib()
    n = _prod(shape)
    n_in = len(tiles[0]) if tiles else 0
    if any(len(inputs) != n_in for inputs in tiles):
        raise ValueError("every tile of a batch must bring the same number of inputs")
    flat = [x for inputs in tiles for x in inputs]
    if any(x.band_pitch is not None for x in flat):
        # windows of several bands (the frames of a series inside a tile of its raster): an element-wise program treats
        # every band alike, so each band goes as a single-band window of its own -- bands x tiles in one batch, nothing
        # compacted first (round 4; a tiled `T * 2` over 8 frames copied every tile's window before: 2 x the traffic)
        if len(shape) == 3 and shape[0] > 1 and (outs is None or all(tuple(o.shape) == tuple(shape) for o in outs)):
            bands, rows, cols = (int(d) for d in shape)
            # (a width that is no multiple of 4 runs over the next one where every operand has the cells to its right -- see
            #  _padded_launch: the 16-byte kernels instead of one cell per lane, 2.3 x on `T * 2` over 8 frames of 2047 columns)
            pad = (-cols) % 4
            if outs is None and pad and cols >= 4 and all(
                    (x.size == 1 and n != 1) or (x.pitch is not None and tuple(x.shape) == tuple(shape) and x.room_right() >= pad) for x in flat):
                wide = [DeviceArray.empty((bands, rows, cols + pad), compiled.out_dtype) for _ in tiles]
                results = [DeviceArray(o.ptr, shape, o.dtype, o.device, owner=o, pitch=cols + pad, band_pitch=rows * (cols + pad)) for o in wide]
                cols += pad
            else:
                results = [DeviceArray.empty(shape, compiled.out_dtype) for _ in tiles] if outs is None else outs
            count = max(len(flat), 1)
            ptrs = (ctypes.c_void_p * count)(*[x.ptr for x in flat])
            pitches = (ctypes.c_uint32 * count)(*[x.pitch or 0 for x in flat])
            # cells between the bands of an operand: a window's, a compact tile's plane, nothing for one value for all
            apart = (ctypes.c_uint64 * count)(*[0 if x.size == 1 else (x.band_pitch or rows * cols) for x in flat])
            out_ptrs = (ctypes.c_void_p * len(results))(*[r.ptr for r in results])
            out_pitches = (ctypes.c_uint32 * len(results))(*[r.pitch or 0 for r in results])
            out_apart = (ctypes.c_uint64 * len(results))(*[r.band_pitch or rows * cols for r in results])
            runtime.check(L.gm_elemwise_run_batch_bands(ctypes.byref(compiled.program), len(tiles), ptrs, pitches, apart, out_ptrs,
                                                        out_pitches, out_apart, bands, rows, cols), "gm_elemwise_run_batch_bands")
            return results
        tiles = [[x.contiguous() if x.band_pitch is not None else x for x in inputs] for inputs in tiles]
        flat = [x for inputs in tiles for x in inputs]
    if outs is not None and any(o.band_pitch is not None for o in outs):
        # results that are windows of several bands (tiles of a mosaic of frames) while every operand is compact
        bands, rows, cols = (int(d) for d in shape)
        count = max(len(flat), 1)
        ptrs = (ctypes.c_void_p * count)(*[x.ptr for x in flat])
        pitches = (ctypes.c_uint32 * count)(*[x.pitch or 0 for x in flat])
        apart = (ctypes.c_uint64 * count)(*[0 if x.size == 1 else rows * cols for x in flat])
        out_ptrs = (ctypes.c_void_p * len(outs))(*[r.ptr for r in outs])
        out_pitches = (ctypes.c_uint32 * len(outs))(*[r.pitch or 0 for r in outs])
        out_apart = (ctypes.c_uint64 * len(outs))(*[r.band_pitch or rows * cols for r in outs])
        runtime.check(L.gm_elemwise_run_batch_bands(ctypes.byref(compiled.program), len(tiles), ptrs, pitches, apart, out_ptrs,
                                                    out_pitches, out_apart, bands, rows, cols), "gm_elemwise_run_batch_bands")
        return outs
    if outs is None:
        padded = _padded_launch(L, compiled, tiles, flat, shape, n)
        if padded is not None:
            return padded
        outs = [DeviceArray.empty(shape, compiled.out_dtype) for _ in tiles]
    ptrs = (ctypes.c_void_p * max(len(flat), 1))(*[x.ptr for x in flat])
    out_ptrs = (ctypes.c_void_p * max(len(outs), 1))(*[o.ptr for o in outs])
    if any(x.pitch is not None for x in flat) or any(o.pitch is not None for o in outs):
        rows, cols = _prod(shape[:-1]), int(shape[-1])
        pitches = (ctypes.c_uint32 * max(len(flat), 1))(*[x.pitch or 0 for x in flat])
        out_pitches = (ctypes.c_uint32 * max(len(outs), 1))(*[o.pitch or 0 for o in outs])
        runtime.check(L.gm_elemwise_run_batch_2d(ctypes.byref(compiled.program), len(tiles), ptrs, pitches, out_ptrs,
                                                 out_pitches, rows, cols), "gm_elemwise_run_batch_2d")
    else:
        runtime.check(L.gm_elemwise_run_batch(ctypes.byref(compiled.program), len(tiles), ptrs, out_ptrs, n),
                      "gm_elemwise_run_batch")
    return outs


class PreparedBatch(object):
    """launch_batch with everything that does not change between calls done once: the same program over the same input tiles
    (resident windows: immutable by contract) into FRESH result tiles.  What is left per call is one allocation call for the
    result tiles, one launch call, and the DeviceArray objects that own the results.  Made by ``prepare_batch``."""

    __slots__ = ("compiled", "inputs", "n_tiles", "shape", "nbytes", "ptrs", "pitches", "out_type", "out_pitches", "rows", "cols",
                 "n", "pitched", "program", "device", "dtype")

    def run(self):
        L = _lib()
        n_tiles = self.n_tiles
        out_ptrs = self.out_type()  # (a table of its own per call: scheduler threads may run the same kept launch at the same time)
        runtime.check(L.gm_malloc_many(out_ptrs, n_tiles, self.nbytes), "gm_malloc_many")
        shape, dtype, device = self.shape, self.dtype, self.device
        outs = [DeviceArray(out_ptrs[i], shape, dtype, device) for i in range(n_tiles)]  # (own the blocks from here on)
        if self.pitched:
            runtime.check(L.gm_elemwise_run_batch_2d(self.program, n_tiles, self.ptrs, self.pitches, out_ptrs, self.out_pitches,
                                                     self.rows, self.cols), "gm_elemwise_run_batch_2d")
        else:
            runtime.check(L.gm_elemwise_run_batch(self.program, n_tiles, self.ptrs, out_ptrs, self.n), "gm_elemwise_run_batch")
        return outs


def prepare_batch(compiled, tiles, shape):
    """A PreparedBatch of ``launch_batch(compiled, tiles, shape)`` into fresh result tiles, or None where the launch takes one
    of the rarer routes (windows of several bands, widths padded to 16 bytes): those keep the general path."""
    n = _prod(shape)
    n_in = len(tiles[0]) if tiles else 0
    if not tiles or any(len(inputs) != n_in for inputs in tiles):
        return None
    flat = [x for inputs in tiles for x in inputs]
    if any(x.band_pitch is not None for x in flat):
        return None
    if len(shape) == 3 and shape[0] == 1 and shape[-1] % 4 != 0 and shape[-1] >= 4:
        return None  # (_padded_launch may apply)
    runtime.require_device()
    p = PreparedBatch()
    p.compiled, p.inputs, p.n_tiles, p.shape, p.n = compiled, flat, len(tiles), tuple(int(d) for d in shape), n
    p.dtype = np.dtype(compiled.out_dtype)
    p.nbytes = max(n * p.dtype.itemsize, 1)
    p.device = runtime.get_device()
    p.program = ctypes.byref(compiled.program)
    p.ptrs = (ctypes.c_void_p * max(len(flat), 1))(*[x.ptr for x in flat])
    p.out_type = ctypes.c_void_p * len(tiles)
    p.pitched = any(x.pitch is not None for x in flat)
    p.rows, p.cols = _prod(shape[:-1]), int(shape[-1])
    p.pitches = (ctypes.c_uint32 * max(len(flat), 1))(*[x.pitch or 0 for x in flat]) if p.pitched else None
    p.out_pitches = (ctypes.c_uint32 * len(tiles))(*[0] * len(tiles)) if p.pitched else None
    return p


def _padded_launch(L, compiled, tiles, flat, shape, n):
    """Windows whose width is no multiple of 4 -- the request a stencil Block grows by its radius, 4096 + 2 columns --
    cannot run on the 16-byte vector kernels, whose groups of 4 cells must not straddle rows.  When every raster input
    is a zero-copy window with enough of its parent raster to its right, the program runs over the next multiple of 4
    columns instead (up to 3 columns of neighbouring cells are computed for nothing) into a result tile of that
    width, and the result is handed on as the window of the requested width inside it (row pitch = the padded
    width: what the stencil kernels and this launch path read in place).  None when that is not possible."""
    if len(shape) != 3 or shape[0] != 1 or shape[-1] % 4 == 0 or shape[-1] < 4:
        return None
    rows, cols = int(shape[1]), int(shape[2])
    pad = (-cols) % 4
    pitches = []
    for x in flat:
        if x.size == 1 and n != 1:  # broadcast operand: no rows
            pitches.append(0)
        elif x.pitch is not None and tuple(x.shape) == tuple(shape) and x.room_right() >= pad:
            pitches.append(x.pitch)
        else:
            return None
    wide = (1, rows, cols + pad)
    outs = [DeviceArray.empty(wide, compiled.out_dtype) for _ in tiles]
    ptrs = (ctypes.c_void_p * max(len(flat), 1))(*[x.ptr for x in flat])
    out_ptrs = (ctypes.c_void_p * len(outs))(*[o.ptr for o in outs])
    pitch_arr = (ctypes.c_uint32 * max(len(flat), 1))(*pitches)
    out_pitches = (ctypes.c_uint32 * len(outs))(*[0] * len(outs))
    runtime.check(L.gm_elemwise_run_batch_2d(ctypes.byref(compiled.program), len(tiles), ptrs, pitch_arr, out_ptrs,
                                             out_pitches, rows, cols + pad), "gm_elemwise_run_batch_2d")
    return [DeviceArray(o.ptr, shape, o.dtype, o.device, owner=o, pitch=cols + pad) for o in outs]


def _collect_leaves(root):
    """Distinct Leaf objects of the tree in first-use (DFS) order."""
    seen, out, stack = set(), [], [root]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if isinstance(n, Leaf):
            out.append(n)
        elif isinstance(n, Op):
            stack.extend(reversed(n.args))
    return out


class StageRef(Leaf):
    """Leaf standing for the result of an earlier stage of a staged plan (a full-size tile)."""

    def __init__(self, stage, dtype, ndv, shape):
        super().__init__(None, dtype, ndv)
        self.stage, self.shape = stage, tuple(shape)


def _leaf_shape(leaf):
    return leaf.shape if isinstance(leaf, StageRef) else tuple(leaf.values.shape)


def plan(root):
    """Bind slots/broadcast flags of the tree's leaves and compile it -> (compiled, leaves, shape).
    Raises FusionLimit when the tree does not fit one program."""
    leaves = _collect_leaves(root)
    if not leaves:
        raise TypeError("at least one raster operand is required")
    shapes = [_leaf_shape(leaf) for leaf in leaves]
    try:
        shape = tuple(np.broadcast_shapes(*shapes))  # NumPy's own rule (and its own error for a mismatch)
    except ValueError:
        raise ValueError("operands could not be broadcast together with shapes %s" % " ".join(str(s) for s in shapes))
    n = _prod(shape)
    for slot, (leaf, sh) in enumerate(zip(leaves, shapes)):
        size = _prod(sh)
        if size != n and size != 1:
            # e.g. one constant per band, (bands, 1, 1), against a (bands, H, W) raster: the kernels index
            # tiles linearly, so such an operand is expanded to the full shape first (rare, and small)
            leaf.values = _broadcast_operand(leaf.values, shape)
        elif not isinstance(leaf, StageRef):
            leaf.values = as_device_view(leaf.values)
        leaf.slot, leaf.broadcast = slot, (size == 1 and n != 1)
    return compile_cached(root, len(leaves)), leaves, shape


def _broadcast_operand(values, shape):
    """An operand expanded to ``shape`` by NumPy's broadcasting rule -> contiguous DeviceArray."""
    if isinstance(values, DeviceArray):
        have = tuple(values.shape)
        lead = len(shape) - len(have)
        if values.pitch is None and lead >= 0 and _prod(shape[:lead + 1]) * _prod(have[1:] if have else ()) == _prod(shape) \
                and tuple(shape[lead + 1:]) == have[1:] and (not have or have[0] == 1):
            # only leading axes grow: the tile repeated back to back (device-to-device copies)
            out = DeviceArray.empty(shape, values.dtype)
            for k in range(_prod(shape) // max(values.size, 1)):
                runtime.check(runtime.lib().gm_d2d(ctypes.c_void_p(out.ptr + k * values.nbytes), ctypes.c_void_p(values.ptr),
                                                   values.nbytes), "gm_d2d")
            return out
        values = values.to_numpy()
    return as_device_view(np.ascontiguousarray(np.broadcast_to(np.asarray(values), shape)))


def _subtree_classes(root, memo):
    """Admissible classes of the whole sub-tree under ``root`` (memoised by node identity)."""
    hit = memo.get(id(root))
    if hit is None:
        if isinstance(root, Op):
            # tree_classes(root), composed from the operands' memoised answers: a chain of n steps asked for every suffix of
            # itself walked n^2 / 2 nodes (31 node_classes per request of a 7-Block view: 90 of its 245 us from scratch)
            hit = node_classes(root)
            for a in root.args:
                if not hit:
                    break
                hit = hit & (_subtree_classes(a, memo) if isinstance(a, Op) else node_classes(a))
        else:
            hit = ALL_CLASSES
        memo[id(root)] = hit
    return hit


def _weight(node):
    return 1 + sum(_weight(a) for a in node.args) if isinstance(node, Op) else 0


# Steps the reference computes by BOOLEAN INDEXING rather than by a ufunc -- values[mask] = ... in Clip.process
# (misc.py:119-122), values[index] = data[index] in FillNoData.process (elemwise.py:749-755), stack_array[i, index] in
# reduce_rasters (reduction.py:103-107): NumPy does not broadcast there, a raster of 1 frame against one of n is an
# IndexError, not n copies.  The kernels would broadcast; the shapes are checked here so that the caller sees the
# reference's error (args: which operands must agree).
_INDEXED = {"clip": (0, 1), "fillnodata": (0, 1), "fmax": (0, 1)}


def _check_indexed_shapes(root):
    memo = {}

    def shape_of(node):
        hit = memo.get(id(node))
        if hit is not None:
            return hit
        if isinstance(node, Leaf):
            out = tuple(_leaf_shape(node))
        elif isinstance(node, Op):
            shapes = [shape_of(a) for a in node.args]
            pair = _INDEXED.get(node.op)
            if pair is not None:
                a, b = shapes[pair[0]], shapes[pair[1]]
                if a != () and b != () and a != b:
                    axis = next((k for k, (x, y) in enumerate(zip(a, b)) if x != y), 0) if len(a) == len(b) else 0
                    raise IndexError("boolean index did not match indexed array along axis %d; size of axis is %s but size of "
                                     "corresponding boolean axis is %s" % (axis, a[axis] if axis < len(a) else "?",
                                                                           b[axis] if axis < len(b) else "?"))
            try:
                out = tuple(np.broadcast_shapes(*shapes)) if shapes else ()
            except ValueError:
                out = ()  # (plan() reports it, with every operand's shape)
        else:
            out = ()
        memo[id(node)] = out
        return out

    shape_of(root)


def plan_staged(root):
    """Partition an expression into as few programs as possible -> (stages, shape).

    ``stages`` is a list of (compiled, leaves) in execution order; a leaf is either a Leaf of the
    expression (its tile bound in ``.values``) or a StageRef to an earlier stage's result; the last
    stage computes ``root``.  Cuts are placed bottom-up where no register class serves both sides
    (e.g. float32 arithmetic + Classify | Reclassify + Mask in the int64 class), keeping the largest
    compatible part fused; programs that exceed the input / step / temp limits are split on top."""
    _check_indexed_shapes(root)
    stages = []
    memo = {}
    shape_of = {}

    def any_shape(node):
        for leaf in _collect_leaves(node):
            sh = _leaf_shape(leaf)
            if _prod(sh) != 1:
                return sh
        return _leaf_shape(_collect_leaves(node)[0])

    def cut(node):
        """Return ``node`` with the sub-trees that cannot share its program replaced by StageRefs."""
        if not isinstance(node, Op):
            return node
        args = [cut(a) for a in node.args]
        node = Op(node.op, args, node.dtype, node.ndv, fill=node.fill, extra=node.extra) \
            if any(x is not y for x, y in zip(args, node.args)) else node
        if _subtree_classes(node, memo):
            return node
        own = node_classes(Op(node.op, [a if not isinstance(a, Op) else Leaf(None, a.dtype, a.ndv) for a in node.args],
                              node.dtype, node.ndv, fill=node.fill, extra=node.extra))
        if not own:
            return node  # not even alone: compile_expr reports it
        children = [a for a in node.args if isinstance(a, Op)]
        best = max(own, key=lambda c: (sum(_weight(a) for a in children if c in _subtree_classes(a, memo)),
                                       -CLASS_COST[c]))
        new_args = []
        for a in node.args:
            if isinstance(a, Op) and best not in _subtree_classes(a, memo):
                new_args.append(materialise(a))
            else:
                new_args.append(a)
        return Op(node.op, new_args, node.dtype, node.ndv, fill=node.fill, extra=node.extra)

    def materialise(node):
        node = cut(node)
        try:
            compiled, leaves, shape = plan(node)
        except FusionLimit:
            if not isinstance(node, Op) or all(isinstance(a, (Leaf, Lit)) for a in node.args):
                raise NotImplementedError("expression '%s' does not fit the device program format"
                                          % getattr(node, "op", node))
            # too many inputs / steps / temps: the operand sub-trees become stages of their own
            new_args = [materialise(a) if isinstance(a, Op) else a for a in node.args]
            node = Op(node.op, new_args, node.dtype, node.ndv, fill=node.fill, extra=node.extra)
            compiled, leaves, shape = plan(node)
        stages.append((compiled, leaves))
        return StageRef(len(stages) - 1, node.dtype, compiled.out_ndv, shape)

    if isinstance(root, (Leaf, Lit)):
        compiled, leaves, shape = plan(root)
        return [(compiled, leaves)], shape
    last = materialise(root)
    return stages, last.shape


def run_staged(stages, shape):
    """Execute a staged plan whose expression leaves carry their tiles -> DeviceArray of the last stage."""
    outs = []
    for compiled, leaves in stages:
        inputs = [outs[leaf.stage] if isinstance(leaf, StageRef) else leaf.values for leaf in leaves]
        outs.append(launch(compiled, inputs, shape))
    return outs[-1]


def evaluate(root):
    """Evaluate an expression whose Leafs carry their tiles -> (DeviceArray, ndv).

    One launch when the tree fits one program; otherwise (no common register class, too
    many inputs / steps / temps) it runs as the stages ``plan_staged`` cuts it into."""
    runtime.require_device()
    stages, shape = plan_staged(root)
    return run_staged(stages, shape), stages[-1][0].out_ndv


_EXPR_PLANS = {}  # (what the expression is, what kind of tiles it is applied to) -> (stages, shape)
_EXPR_PLANS_MAX = 512


def _tiles_key(tiles):
    """dtype, shape and no-data value of every tile, or None when the tiles are not all of one shape (operands that
    are expanded per call) or a value does not hash."""
    try:
        key, shape = [], None
        for v, ndv in tiles:
            sh = tuple(v.shape)
            if _prod(sh) != 1:
                if shape is not None and sh != shape:
                    return None
                shape = sh
            key.append((np.dtype(v.dtype).num, sh, None if ndv is None else _scalar_key(ndv)))
        key = tuple(key)
        hash(key)
        return key
    except TypeError:
        return None


_NARROW_LIMIT = 1 << 24  # whole numbers up to here are exact in every register class, float32 included


def _narrowed(values, ndv, to="i4"):
    """A 64-bit integer tile all of whose values lie within +-2^24 as an int32 (``to``: or float32) tile with the same values
    and its no-data value, if an int32 can be equal to it; None for every other tile.  Three passes and two round trips to the
    host: for the views this serves there was no answer at all before (see run_expr)."""
    if not isinstance(values, DeviceArray) or not _wide_int(values.dtype) or values.size == 0:
        return None
    to = np.dtype(to)
    low, high = (-_NARROW_LIMIT if values.dtype.kind == "i" else 0), _NARROW_LIMIT
    fits = (-(1 << 31), (1 << 31) - 1)
    if to.kind in "iu":  # (narrower targets: the minima of such sums under a declared int16 / uint16)
        fits = (int(np.iinfo(to).min), int(np.iinfo(to).max))
        low, high = max(low, fits[0]), min(high, fits[1])
    eff = effective_ndv(values.dtype, ndv)
    if ndv is not None and (eff is None or not fits[0] <= int(eff) <= fits[1]):
        return None  # (a Clip writes its store's no-data value: it has to be one the narrow tile can hold -- a TemporalSum's is)
    tile = values.contiguous()
    bits = DeviceArray(tile.ptr, tile.shape, np.dtype("i8"), tile.device, owner=tile)  # (uint64: in range <=> 0 <= bits <= 2^24)
    for op, bound in (("less", low), ("greater", high)):
        outside, _ = evaluate(Op(op, [Leaf(None, bits.dtype, None, values=bits), Lit(bound)], np.dtype("bool"), None, fill=False))
        none = ctypes.c_int(0)
        runtime.check(runtime.lib().gm_all_equal(ctypes.c_void_p(outside.ptr), code_of(outside.dtype), runtime.scalar_bits(0, outside.dtype),
                                                 outside.size, ctypes.byref(none)), "gm_all_equal")
        if not none.value:
            return None
    narrow, _ = evaluate(Op("copy", [Leaf(None, bits.dtype, None, values=bits)], to, None))
    return narrow, (None if eff is None else to.type(int(eff)))


def run_expr(build, tiles, key=None):
    """``_run_expr``; an expression that no register class can run because of a 64-bit integer tile -- a TemporalSum of an
    integer series under a Block that still declares its store's dtype: `TemporalSum(rain) / 100` asks for
    np.divide(int64 tile, 100, dtype=float32), whose cast of the tile to float32 comes first -- is tried again with the tile
    as int32 when all its values lie within +-2^24: the same values, exact in every class, and NumPy's casts of them
    (to float32, to int32) change nothing.  Tiles with larger values keep the refusal."""
    try:
        return _run_expr(build, tiles, key)
    except NotImplementedError as refusal:
        if "no exact register class" not in str(refusal):
            raise
        retry, changed = [], False
        for values, ndv in tiles:
            narrow = _narrowed(values, ndv) if hasattr(values, "dtype") and _wide_int(values.dtype) else None
            retry.append((values, ndv) if narrow is None else narrow)
            changed = changed or narrow is not None
        if not changed:
            raise
        values, _ = _run_expr(build, retry, key)
        # what the expression over the tiles as they came declares (a Clip hands its store's dtype and no-data value on)
        root = build([Leaf(None, np.asarray(v).dtype if not hasattr(v, "dtype") else v.dtype, ndv, values=v) for v, ndv in tiles])
        if values.dtype != root.dtype:
            values, _ = evaluate(Op("copy", [Leaf(None, values.dtype, None, values=values)], root.dtype, None))
        return values, root.ndv


_KEPT_EXPR = {}   # (expression plan, addresses of its resident tiles) -> 0 (seen once) / PreparedBatch / False


def drop_kept_launches():
    """Forget the marshalled launches kept for resident tiles (they hold the tiles: a source that gives its windows back --
    SyntheticSource.drop_resident_windows -- calls this)."""
    _KEPT_EXPR.clear()


def _kept_expr_launch(cache_key, stage, tiles, shape):
    """The PreparedBatch of a one-stage expression over exactly these tiles when every one of them is resident data as it stands
    (DeviceArray.resident), from the third sighting on; else None."""
    try:
        where = tuple((v.ptr, v.pitch, v.shape) if v.resident else None for v, _ in tiles)
    except AttributeError:
        return None
    if None in where:
        return None
    key = (cache_key, where)
    kept = _KEPT_EXPR.get(key)
    if kept is None:
        if len(_KEPT_EXPR) >= 1024:
            _KEPT_EXPR.clear()
        _KEPT_EXPR[key] = 0
        return None
    if kept == 0:
        compiled, binding = stage
        inputs = [as_device_view(tiles[ref][0]) if kind == 0 else ref for kind, ref in binding]
        kept = _KEPT_EXPR[key] = prepare_batch(compiled, [inputs], shape) or False
    return kept or None


def _run_expr(build, tiles, key=None):
    """Evaluate ``build(leaves)`` over raster tiles given as (values, ndv) pairs.

    ``key``: a hashable description of everything ``build`` puts into the expression besides the tiles (operation,
    dtype, literals ...).  With it the staged plan -- the expression tree, its register class, the compiled programs
    -- is made once per kind of tiles and later calls only bind their tiles and launch: what a Block that runs on its
    own (not inside a fused group) pays per request drops from ~100 microseconds of Python to ~10."""
    cache_key = None
    if key is not None:
        tk = _tiles_key(tiles)
        if tk is not None:
            cache_key = (key, tk)
            hit = _EXPR_PLANS.get(cache_key)
            if hit is not None:
                runtime.require_device()
                stages, shape = hit
                if len(stages) == 1:
                    # (the launch over exactly these resident tiles, marshalled once: fuse.FusedTask._kept_launch)
                    kept = _kept_expr_launch(cache_key, stages[0], tiles, shape)
                    if kept is not None:
                        return kept.run()[0], stages[0][0].out_ndv
                outs = []
                for compiled, binding in stages:
                    inputs = [as_device_view(tiles[ref][0]) if kind == 0 else (outs[ref] if kind == 2 else ref)
                              for kind, ref in binding]
                    outs.append(launch(compiled, inputs, shape))
                return outs[-1], stages[-1][0].out_ndv
    leaves = [Leaf(None, np.asarray(v).dtype if not hasattr(v, "dtype") else v.dtype, ndv, values=v)
              for v, ndv in tiles]
    if cache_key is None:
        return evaluate(build(leaves))
    runtime.require_device()
    arg_of = {id(leaf): i for i, leaf in enumerate(leaves)}
    staged, shape = plan_staged(build(leaves))
    stages = []
    for compiled, stage_leaves in staged:
        binding = []
        for leaf in stage_leaves:
            if isinstance(leaf, StageRef):
                binding.append((2, leaf.stage))
            elif id(leaf) in arg_of:
                binding.append((0, arg_of[id(leaf)]))
            else:
                binding.append((1, leaf.values))  # a constant of the expression, on the device already
        stages.append((compiled, binding))
    if len(_EXPR_PLANS) >= _EXPR_PLANS_MAX:
        _EXPR_PLANS.clear()
    _EXPR_PLANS[cache_key] = (stages, shape)
    return run_staged(staged, shape), staged[-1][0].out_ndv
