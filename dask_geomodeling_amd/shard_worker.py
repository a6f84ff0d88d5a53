"""The worker side of dask_geomodeling_amd.shard: one process, one GPU.

    python -m dask_geomodeling_amd.shard_worker <fd of the socket to the parent>

A fresh interpreter (the parent starts it with fork + exec before anything of HIP exists in it).  It binds itself to its
device once (gm_set_device), then serves messages until the socket closes: build and fuse the sub-graphs of its units
(fuse.fuse_graph -- what the reference's ``compute`` does for the whole graph, core/graphs.py:42-49), run them, keep the
results on the device until they are fetched or released, copy fetched tiles device-to-host straight to their place of
the parent's shared-memory mosaic (``values[:, rows, columns] = tile``, raster/parallelize.py:118-124).  Jobs stay --
with their compiled programs, specialised kernels and resident leaf windows -- so that the same request costs one small
message the second time.
"""
import ctypes
import io
import mmap
import os
import pickle
import sys
import time
import traceback
import zlib

import numpy as np


class _State(object):
    def __init__(self):
        self.index = self.device = None
        self.on_device = False
        self.arrays = {}     # shared-memory name -> (mmap, ndarray): the large arrays of views and graphs
        self.ranges = {}     # shared-memory name -> (mmap, address, pinned?): result mosaics of the parent
        self.jobs = {}       # job id -> prepared job
        self.results = {}    # run id -> [response per unit]
        self.mark = None     # HIP event recorded by the parent's ("mark",): start of a timed region
        self.last_metas = {} # job id -> metas of its last run (a re-run that answers the same says so in one word)
        self.cpus = None     # the host cores this worker pinned itself to (None: left to the operating system)
        self.host_s = 0.0    # wall time spent in run / rerun messages (unpickle -> reply queued): ("host_stats",)
        self.host_cpu_s = 0.0
        self.host_runs = 0


STATE = _State()


class _Unpickler(pickle.Unpickler):
    def persistent_load(self, pid):
        kind = pid[0]
        if kind == "gm-array":
            _, name, size, dtype, shape = pid
            hit = STATE.arrays.get(name)
            if hit is None:
                fd = os.open(os.path.join("/dev/shm", name), os.O_RDONLY)
                try:
                    m = mmap.mmap(fd, max(size, 1), access=mmap.ACCESS_READ)
                finally:
                    os.close(fd)
                count = int(np.prod(shape, dtype=np.int64))
                hit = STATE.arrays[name] = (m, np.frombuffer(m, dtype=dtype, count=count).reshape(shape))
            return hit[1]  # (the same object every time: what the per-device upload of a source is remembered by)
        if kind == "gm-small-device-array":
            return pid[1]
        raise pickle.UnpicklingError("unknown persistent id %r" % (kind,))


def _loads(data):
    return _Unpickler(io.BytesIO(data)).load()


def _apply_config(values):
    from . import config as _config
    values = {k: v for k, v in dict(values or {}).items() if v is not None or _config.get(k) is not None}
    values.update({"geomodeling.devices": None, "geomodeling.materialize": False})
    _config.config.set(values)


def _init(settings):
    for p in settings.get("path") or ():
        if p not in sys.path:
            sys.path.append(p)
    STATE.index, STATE.device = settings["index"], settings["device"]
    from . import config as _config
    _apply_config(settings.get("config"))
    preload = settings.get("preload")
    if preload:
        # "module:function", called before the device is bound (dask's worker preload): what a deployment uses to import its
        # own Blocks -- and the CPU test suite to plug its NumPy double of the device in
        module, _, function = preload.partition(":")
        hook = __import__(module, fromlist=["_"])
        if function:
            getattr(hook, function)()
    from .hip import runtime
    if runtime.device_count() > 0:
        if settings.get("affinity", True):
            STATE.cpus = _pin_to_device_cores(runtime, int(STATE.index), settings.get("pool_devices") or (STATE.device,))
        runtime.set_device(int(STATE.device))
        STATE.on_device = True
        return runtime.device_name()
    if not preload:
        raise runtime.GmHipError("no HIP device visible in the worker for device %s (there is no CPU fallback)" % (STATE.device,))
    return None


def _physical_core(cpu):
    """What tells the hardware threads of one core from those of another: the core's first sibling (the cpu itself where the
    topology is not exposed)."""
    try:
        with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
            return int(f.read().replace("-", ",").split(",")[0])
    except (OSError, ValueError):
        return cpu


def _pin_to_device_cores(runtime, index, pool_devices):
    """sched_setaffinity to the cores next to this worker's device BEFORE the device is bound, so that the runtime's own
    threads (and the kernel compiler's) inherit the mask.  Workers of the pool whose devices have the same cores next to
    them (the GPUs of one NUMA node, or several workers on one GPU) each take their share of those cores -- every worker
    works that out by itself from the pool's device list --, so that no two workers' submitting threads sit on one core.
    -> the sorted cores, or None when nothing was changed."""
    try:
        allowed = os.sched_getaffinity(0)
        cores = {}
        for d in set(pool_devices):
            local = runtime.device_local_cpus(int(d))
            cores[d] = tuple(sorted(set(local) & set(allowed))) if local else ()
        mine = cores[pool_devices[index]]
        if not mine:
            return None
        group = [i for i, d in enumerate(pool_devices) if cores[d] == mine]
        k, n = group.index(index), len(group)
        mine = list(mine)
        if n > 1 and len(mine) >= 2 * n:   # (at least two cores each: the submitting thread and the runtime's helper)
            # whole PHYSICAL cores each: the hardware threads of a core stay with one worker (cpu 0 and cpu 128 of a
            # 64-core socket are one core: splitting the sorted list in runs would hand them to two workers)
            cores = {}
            for cpu in mine:
                cores.setdefault(_physical_core(cpu), []).append(cpu)
            physical = [cores[c] for c in sorted(cores)]
            per = max(len(physical) // n, 1)
            share = physical[k * per:(k + 1) * per] if len(physical) >= n else [mine[k::n]]
            mine = sorted(cpu for group in share for cpu in group)
        if not mine or len(mine) == len(allowed):
            return None
        os.sched_setaffinity(0, mine)
        return mine
    except Exception:  # placement is an optimisation: never a reason not to start (whatever the topology files hold)
        return None


def _meta(response):
    if response is None:
        return None
    if isinstance(response, dict) and "values" in response:
        values = response["values"]
        if hasattr(values, "shape") and hasattr(values, "dtype"):
            ndv = response.get("no_data_value")
            return ("raster", tuple(int(s) for s in values.shape), np.dtype(values.dtype).str, ndv)
    return ("object", response)


def _prepare(spec, keep=False):
    from . import config as _config
    from .core.graphs import PreparedGraph, get_sync
    from .fuse import fuse_graph
    kind = spec[0]
    fuse = bool(_config.get("geomodeling.fuse", True))
    if kind == "graph":
        _, graph, keys = spec
        return ("graph", fuse_graph(graph, keys) if fuse else graph, keys, get_sync)
    if kind == "tiles":
        # the sub-graphs of OUR tiles only (parallelize.py:82-89 emits them x-major, low y first), as one graph: equal tile
        # programs then run as batched launches, shared keys are computed once
        _, view, request, indices = spec
        pairs = list(view.get_sources_and_requests(**request))
        graph, keys = {}, []
        for t in indices:
            store, tile_request = pairs[1 + t]
            graph, key = store.get_compute_graph(cached_compute_graph=graph, **tile_request)
            keys.append(key)
        graph = fuse_graph(graph, keys) if fuse else graph
        if keep:
            # a job that is kept is run again and again: the look at its graph, its sources' resident windows and the
            # arguments of its batched launches are kept with it (core/graphs.py PreparedGraph)
            return ("prepared", PreparedGraph(graph, keys))
        return ("graph", graph, keys, get_sync)
    if kind == "view":
        _, view, request = spec
        return ("view", view, request)
    raise ValueError("unknown job kind %r" % (kind,))


def _run(job):
    """The job's units, enqueued on this worker's compute stream.  Like a get_data that leaves its result on the device in the
    single-process path, this returns when the launches are queued, not when they have run: the next request's host work
    (leaf windows, launch arguments) overlaps this one's kernels.  Whatever reads the results -- a fetch, a checksum, the
    parent's sync() -- is ordered behind them on the same stream."""
    return _evaluate(job)


def _evaluate(job):
    if job[0] == "graph":
        _, graph, keys, get = job
        results = list(get(graph, keys))
    elif job[0] == "prepared":
        results = list(job[1].run())
    else:
        _, view, request = job
        results = [view.get_data(**request)]  # (this process's config: not materialised, no device list)
    return results


def _range(name, size):
    hit = STATE.ranges.get(name)
    if hit is None:
        fd = os.open(os.path.join("/dev/shm", name), os.O_RDWR)
        try:
            m = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        address = ctypes.addressof((ctypes.c_char * size).from_buffer(m))
        pinned = False
        if STATE.on_device:
            from .hip import runtime
            runtime.check(runtime.lib().gm_host_register(ctypes.c_void_p(address), size), "gm_host_register(%d bytes)" % size)
            pinned = True
        hit = STATE.ranges[name] = (m, address, pinned, size)
    return hit


def _drop_names(names):
    for name in names:
        STATE.arrays.pop(name, None)
        hit = STATE.ranges.pop(name, None)
        if hit is not None and hit[2]:
            from .hip import runtime
            runtime.lib().gm_host_unregister(ctypes.c_void_p(hit[1]))


def _fetch(run_id, name, size, parts):
    m, address, _, _ = _range(name, size)
    responses = STATE.results[run_id]
    alive = []  # device tiles (compacted / cast copies) the asynchronous copies read: kept until the sync below
    for unit, offset, pitch, plane, dtype, bands, rows, cols in parts:
        values = responses[unit]["values"]
        dtype = np.dtype(dtype)
        if type(values).__name__ == "DeviceArray":
            from .hip import runtime
            tile = values.contiguous()
            if tile.dtype != dtype:  # values[...] = tile casts on assignment (parallelize.py:118-124): NumPy's astype, on the device
                from .raster.stencil import _cast_tile
                tile = _cast_tile(tile, dtype)
            lead = tile.size // max(rows * cols, 1)
            if lead * rows * cols != tile.size or lead not in (bands, 1) or tuple(tile.shape[-2:]) != (rows, cols):
                raise ValueError("a result of shape %s does not fit its place of %d x %d x %d" % (tile.shape, bands, rows, cols))
            item = dtype.itemsize
            L = runtime.lib()
            for b in range(bands):
                src = tile.ptr + (b if lead == bands else 0) * rows * cols * item  # (one band is repeated in every band)
                runtime.check(L.gm_d2h_2d(ctypes.c_void_p(address + offset + b * plane), pitch, ctypes.c_void_p(src), cols * item,
                                          cols * item, rows), "gm_d2h_2d")
            alive.append(tile)
        else:
            dest = np.ndarray((bands, rows, cols), dtype=dtype, buffer=m, offset=offset, strides=(plane, pitch, dtype.itemsize))
            dest[...] = np.asarray(values).reshape((-1, rows, cols))  # NumPy's assignment: casts, repeats one band
    if alive:
        from .hip import runtime
        runtime.sync()
    return True


def _checksums(run_id, units):
    out = []
    for unit in units:
        response = STATE.results[run_id][unit]
        if response is None or not isinstance(response, dict) or "values" not in response:
            out.append(None)
            continue
        values = response["values"]
        if type(values).__name__ == "DeviceArray":
            out.append(int(values.checksum()))  # on the device (gm_checksum)
        else:
            out.append(zlib.crc32(np.ascontiguousarray(values).tobytes()))  # host-resident values (no device involved)
    return out


def _handle(message):
    kind = message[0]
    if kind == "run":
        _, job_id, run_id, spec, keep = message
        job = _prepare(spec, keep)
        if keep:
            STATE.jobs[job_id] = job
        results = STATE.results[run_id] = _run(job)
        metas = [_meta(r) for r in results]
        if keep:
            STATE.last_metas[job_id] = metas
        return metas
    if kind == "rerun":
        _, job_id, run_id = message
        job = STATE.jobs.get(job_id)
        if job is None:
            raise KeyError("job %r is not known to this worker" % (job_id,))
        results = STATE.results[run_id] = _run(job)
        metas = [_meta(r) for r in results]
        try:
            if metas == STATE.last_metas.get(job_id):
                return "same"
        except Exception:  # (metas that do not compare: objects of a non-raster answer)
            pass
        STATE.last_metas[job_id] = metas
        return metas
    if kind == "sync":
        if STATE.on_device:
            from .hip import runtime
            runtime.sync()
        return True
    if kind == "mark":
        # start of a timed region on this worker's compute stream (bench.py): ("elapsed",) answers the HIP-event time since
        if STATE.on_device:
            from .hip import runtime
            STATE.mark = runtime.Event().record()
        return True
    if kind == "elapsed":
        if not STATE.on_device or STATE.mark is None:
            return None
        from .hip import runtime
        stop = runtime.Event().record()
        stop.synchronize()
        return STATE.mark.elapsed_ms(stop)
    if kind == "fetch":
        return _fetch(*message[1:])
    if kind == "checksums":
        return _checksums(message[1], message[2])
    if kind == "drop":
        for job_id in message[1]:
            STATE.jobs.pop(job_id, None)
            STATE.last_metas.pop(job_id, None)
        return True
    if kind == "spec_wait":
        # kernels specialised per program are compiled in the background (csrc/gm_spec.hip): wait for them; what ran so far
        from .hip import program as P
        if P.spec_pending():
            P.spec_wait()
        return {"launches": P.spec_stats()["launches"], "kernel": P.spec_last_kernel()}
    if kind == "job_stats":
        # what is kept with a kept job: runs so far, answers of resident sources kept, launches whose arguments are marshalled once
        job = STATE.jobs.get(message[1])
        if job is None or job[0] != "prepared":
            return None
        g = job[1]
        prepared = sum(1 for _, plan, _ in (g.program or ()) if getattr(plan[1], "__name__", "") == "run" and not plan[2])
        return {"runs": g.runs, "kept": len(g.kept), "live": len(g.program or ()), "prepared": prepared}
    if kind == "config":
        _apply_config(message[1])
        for job_id in message[2]:
            STATE.jobs.pop(job_id, None)
            STATE.last_metas.pop(job_id, None)
        return True
    if kind == "host_stats":
        # (reset, ...): host wall / CPU seconds this worker spent answering run / rerun messages, their count, its cores
        out = {"wall_s": STATE.host_s, "cpu_s": STATE.host_cpu_s, "runs": STATE.host_runs, "cpus": STATE.cpus,
               "pid": os.getpid()}
        if len(message) > 1 and message[1]:
            STATE.host_s = STATE.host_cpu_s = 0.0
            STATE.host_runs = 0
        return out
    if kind == "ping":
        return (os.getpid(), STATE.device, len(STATE.jobs), len(STATE.results))
    raise ValueError("unknown message %r" % (kind,))


def _reply(conn, kind, payload):
    try:
        data = pickle.dumps((kind, payload), protocol=pickle.HIGHEST_PROTOCOL)
    except Exception as e:  # an unpicklable result: say so instead of dying with the socket half written
        data = pickle.dumps(("error", ("reply could not be pickled: %r" % (e,), "", None)))
    conn.send_bytes(data)


def _error_payload(e):
    try:
        pickled = pickle.dumps(e)
        pickle.loads(pickled)
    except Exception:
        pickled = None
    return ("%s: %s" % (type(e).__name__, e), traceback.format_exc(), pickled)


def main(fd):
    from multiprocessing.connection import Connection
    conn = Connection(fd)
    try:
        first = _loads(conn.recv_bytes())
        assert first[0] == "init"
        _reply(conn, "ok", _init(first[1]))
    except BaseException as e:  # noqa: BLE001 -- the parent must hear about it
        _reply(conn, "error", _error_payload(e))
        return 1
    while True:
        try:
            data = conn.recv_bytes()
        except (EOFError, OSError):
            return 0  # the parent has gone (or closed the pool)
        try:
            message = _loads(data)
            if message[0] == "stop":
                return 0
            _, work, names, runs = message
            for run_id in runs:
                STATE.results.pop(run_id, None)
            _drop_names(names)
            if work is None:
                continue
            if work[0] in ("run", "rerun"):
                t0, c0 = time.perf_counter(), time.thread_time()
                payload = _handle(work)
                STATE.host_s += time.perf_counter() - t0
                STATE.host_cpu_s += time.thread_time() - c0
                STATE.host_runs += 1
            else:
                payload = _handle(work)
        except Exception as e:  # noqa: BLE001 -- every failure of a request travels to the parent
            _reply(conn, "error", _error_payload(e))
            continue
        _reply(conn, "ok", payload)


if __name__ == "__main__":
    code = main(int(sys.argv[1]))
    sys.stdout.flush()
    sys.stderr.flush()
    # (leave through the interpreter's normal exit: the library's exit handlers wait for kernels still being compiled)
    sys.exit(code)
