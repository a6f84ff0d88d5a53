"""Loader and thin wrappers for the runtime part of the C-ABI (include/gmhip.h).

The product path is HIP-only: there is no NumPy fallback.  If the shared library is
missing, or no MI355X is visible when a kernel is requested, a ``GmHipError`` is
raised -- loudly, never a silent CPU path.
"""
import ctypes
from math import prod as _prod
import os
import atexit
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgmhip.so")


class GmHipError(RuntimeError):
    """A C-ABI call returned non-zero (message from gm_last_error())."""


class gm_scalar(ctypes.Union):
    _fields_ = [("f64", ctypes.c_double), ("f32", ctypes.c_float), ("i64", ctypes.c_int64),
                ("u64", ctypes.c_uint64), ("i32", ctypes.c_int32)]


def scalar_bits(value, dtype):
    """8-byte gm_scalar payload of ``value`` for a tile of ``dtype``: float tiles carry
    the value as float64 bits, integer/bool tiles as (two's complement) int64 bits."""
    import struct

    import numpy as np
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return struct.unpack("<Q", struct.pack("<d", float(value)))[0]
    return int(value) & 0xFFFFFFFFFFFFFFFF


_lib = None
_lock = threading.Lock()


def lib():
    """The loaded CDLL (loads on first use; raises if the extension was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GmHipError(
                "HIP extension not built: %s is missing. Build it with "
                "`make -C dask_geomodeling_amd/csrc` (or __graft_entry__.build()). "
                "There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
        # kernels still being compiled in the background when the interpreter exits: wait for them here, while every
        # library is still whole (the compiler stack's own exit handlers would otherwise run under the worker thread)
        atexit.register(_quiesce)
    return _lib


def _quiesce():
    try:
        if _lib is not None and not os.environ.get("GM_EXIT_NOWAIT"):  # (tools/exit_stress.py exercises the library's own exit path)
            _lib.gm_spec_wait()
    except Exception:  # (exit path: nothing to report to)
        pass


def _declare(L):
    c = ctypes
    vp, sz, i32, u64 = c.c_void_p, c.c_size_t, c.c_int, c.c_uint64
    L.gm_last_error.restype = c.c_char_p
    L.gm_version.restype = c.c_char_p
    sigs = {
        "gm_init": [], "gm_device_count": [], "gm_set_device": [i32],
        "gm_get_device": [c.POINTER(i32)], "gm_device_name": [c.c_char_p, sz],
        "gm_device_pci_bus_id": [i32, c.c_char_p, sz],
        "gm_guard_enabled": [], "gm_guard_check": [c.POINTER(c.c_ulonglong)],
        "gm_route_stats": [i32, c.POINTER(u64), i32],
        "gm_malloc": [c.POINTER(vp), sz], "gm_free": [vp], "gm_pool_trim": [],
        "gm_malloc_many": [c.POINTER(vp), i32, sz],
        "gm_mem_info": [c.POINTER(sz), c.POINTER(sz)],
        "gm_pool_stats": [c.POINTER(sz), c.POINTER(sz)],
        "gm_h2d": [vp, vp, sz], "gm_d2h": [vp, vp, sz], "gm_d2d": [vp, vp, sz],
        "gm_host_malloc": [c.POINTER(vp), sz], "gm_host_free": [vp],
        "gm_host_register": [vp, sz], "gm_host_unregister": [vp], "gm_d2h_2d": [vp, sz, vp, sz, sz, sz],
        "gm_memset": [vp, i32, sz], "gm_sync": [],
        "gm_event_create": [c.POINTER(vp)], "gm_event_destroy": [vp], "gm_event_record": [vp],
        "gm_event_sync": [vp], "gm_event_elapsed_ms": [vp, vp, c.POINTER(c.c_float)],
        "gm_all_equal": [vp, i32, u64, sz, c.POINTER(i32)],
        "gm_fill": [vp, i32, u64, sz],
        "gm_all_finite": [vp, i32, sz, c.POINTER(i32)],
        "gm_generate": [vp, i32, u64, c.c_int64, c.c_int64, i32, i32, c.c_int64, c.c_int64, c.c_double,
                        c.c_double, c.c_uint32, c.c_uint32, u64],
        "gm_checksum": [vp, sz, c.POINTER(u64)],
        "gm_paste_tile": [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = c.c_int


def check(rc, what=""):
    if rc != 0:
        msg = lib().gm_last_error().decode("utf-8", "replace")
        raise GmHipError("%s failed (code %d): %s" % (what or "libgmhip call", rc, msg))


def device_count():
    return int(lib().gm_device_count())


_ok_pid = None       # the pid in which a device was seen (a fork()ed child asks the library again: GM_ERR_FORKED)
_process_device = None  # (pid, index) once set_device has bound this process


def require_device():
    global _ok_pid
    if _ok_pid == os.getpid():  # (asked by every allocation: one library call per process instead of one per tile)
        return
    if device_count() <= 0:
        why = lib().gm_last_error().decode("utf-8", "replace")
        if why.startswith("forked after gm_init"):  # a fork() of a process that had entered the library (GM_ERR_FORKED)
            raise GmHipError(why)
        raise GmHipError("no HIP device visible: the raster compute path needs an MI355X "
                         "(there is no CPU fallback)")
    _ok_pid = os.getpid()


def set_device(index):
    global _process_device
    check(lib().gm_set_device(int(index)), "gm_set_device")
    _process_device = (os.getpid(), int(index))


def get_device():
    bound = _process_device
    if bound is not None and bound[0] == os.getpid():  # (gm_set_device binds the PROCESS: every thread is pointed at it)
        return bound[1]
    out = ctypes.c_int(0)
    check(lib().gm_get_device(ctypes.byref(out)), "gm_get_device")
    return out.value


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(lib().gm_device_name(buf, 256), "gm_device_name")
    return buf.value.decode()


ROUTE_FAMILIES = {"zonal": 0, "bilinear": 1, "tables": 2, "medians": 3, "stencils": 4}


def route_stats(family):
    """{route name: times taken since the library was loaded} of one family of entry points ("zonal", "bilinear", "tables",
    "medians", "stencils"): which kernel route a call took (gm_route_stats).  Tests that force a route assert on the difference
    of two readings."""
    L = lib()
    L.gm_route_name.restype = ctypes.c_char_p
    L.gm_route_name.argtypes = [ctypes.c_int, ctypes.c_int]
    code = ROUTE_FAMILIES[family] if isinstance(family, str) else int(family)
    n = L.gm_route_stats(code, None, 0)
    if n < 0:
        raise ValueError("unknown route family %r" % (family,))
    counters = (ctypes.c_uint64 * n)()
    L.gm_route_stats(code, counters, n)
    return {L.gm_route_name(code, i).decode(): int(counters[i]) for i in range(n)}


def routes_taken(before, after):
    """{route: count} of the routes whose counters moved between two readings of route_stats."""
    return {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}


def guard_violations():
    """Red-zone violations counted so far (GM_DEBUG_GUARD=1), checking every live block now; raises on a new one."""
    n = ctypes.c_ulonglong(0)
    check(lib().gm_guard_check(ctypes.byref(n)), "gm_guard_check")
    return int(n.value)


def device_pci_bus_id(device):
    """"0000:c1:00.0" of a visible device (the device is not bound by asking)."""
    buf = ctypes.create_string_buffer(64)
    check(lib().gm_device_pci_bus_id(int(device), buf, 64), "gm_device_pci_bus_id")
    return buf.value.decode()


def device_local_cpus(device):
    """The host cores next to a device -- /sys/bus/pci/devices/<bus id>/local_cpulist, else the cores of its numa_node --
    or None where the platform does not say (a VM without NUMA information, a one-node host: nothing to choose)."""
    try:
        base = "/sys/bus/pci/devices/" + device_pci_bus_id(device).lower()
    except GmHipError:
        return None
    for name in ("local_cpulist",):
        try:
            with open(os.path.join(base, name)) as f:
                cpus = parse_cpulist(f.read())
            if cpus:
                return cpus
        except OSError:
            pass
    try:
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
        if node >= 0:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                return parse_cpulist(f.read()) or None
    except (OSError, ValueError):
        pass
    return None


def parse_cpulist(text):
    """"0-3,8,10-11" -> {0, 1, 2, 3, 8, 10, 11}."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        low, _, high = part.partition("-")
        try:
            cpus.update(range(int(low), int(high or low) + 1))
        except ValueError:
            return set()
    return cpus


def sync():
    check(lib().gm_sync(), "gm_sync")


def malloc(nbytes):
    ptr = ctypes.c_void_p()
    check(lib().gm_malloc(ctypes.byref(ptr), int(nbytes)), "gm_malloc(%d)" % nbytes)
    return ptr.value


def free(ptr):
    if ptr:
        lib().gm_free(ctypes.c_void_p(ptr))


class _Pinned(object):
    """Owner of one pinned host buffer; goes back to the library's cache when the last ndarray
    viewing it dies."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        check(lib().gm_host_malloc(ctypes.byref(p), nbytes), "gm_host_malloc(%d)" % nbytes)
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                lib().gm_host_free(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """ndarray of ``shape`` / ``dtype`` in pinned host memory (a device-to-host copy into it runs at
    PCIe speed); the buffer is recycled when the array and all views of it are gone."""
    import numpy as np
    dtype = np.dtype(dtype)
    nbytes = _prod(shape) * dtype.itemsize
    owner = _Pinned(max(nbytes, 1))
    raw = (ctypes.c_ubyte * max(nbytes, 1)).from_address(owner.ptr)
    raw._owner = owner  # the ndarray keeps `raw` alive, `raw` keeps the pinned buffer alive
    return np.frombuffer(raw, dtype=dtype, count=_prod(shape)).reshape(shape)


def pool_trim():
    check(lib().gm_pool_trim(), "gm_pool_trim")


def mem_info():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().gm_mem_info(ctypes.byref(f), ctypes.byref(t)), "gm_mem_info")
    return f.value, t.value


def pool_stats():
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().gm_pool_stats(ctypes.byref(a), ctypes.byref(b)), "gm_pool_stats")
    return a.value, b.value


class Event:
    """HIP event on the current device's compute stream."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(lib().gm_event_create(ctypes.byref(h)), "gm_event_create")
        self._h = h

    def record(self):
        check(lib().gm_event_record(self._h), "gm_event_record")
        return self

    def synchronize(self):
        check(lib().gm_event_sync(self._h), "gm_event_sync")

    def elapsed_ms(self, later):
        ms = ctypes.c_float()
        check(lib().gm_event_elapsed_ms(self._h, later._h, ctypes.byref(ms)), "gm_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self._h:
                lib().gm_event_destroy(self._h)
        except Exception:
            pass
