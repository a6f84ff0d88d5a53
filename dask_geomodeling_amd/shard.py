"""One worker process per GPU: the data-parallel fan-out of tile tasks over the MI355X of one node.

``RasterTiler.get_sources_and_requests`` emits its tiles x-major, low y first (raster/parallelize.py:82); every
tile is a self-contained sub-graph (stencils get their halo by request enlargement, spatial.py:27-47), so tile
``t`` of that order is evaluated by worker ``t % G`` on its own GPU and nothing is exchanged between GPUs: no
collective on the data path.  What this module adds to the reference's ``compute`` (core/graphs.py:42-49):

``get(graph, keys)``
    a dask-compatible scheduler.  The tile keys under every ``RasterTiler`` end-point -- and the other wanted
    keys, one unit each -- are dealt over the workers; every worker receives the part of the graph its units
    reach, fuses it (fuse.fuse_graph) and runs it on its GPU; the parent assembles the host mosaic
    (``values[:, rows, columns] = tile``, parallelize.py:108-124) from tiles the workers copy device-to-host
    straight to their place of a shared-memory mosaic that every worker has pinned.

``get_data(view, request)``
    what ``Block.get_data`` calls when ``geomodeling.devices`` lists devices: the view travels to the workers once
    (and its large arrays once, through shared memory), later requests are a few hundred bytes; every worker
    builds and fuses the sub-graphs of ITS tiles only, keeps the fused job (compiled programs, specialised
    kernels, resident leaf windows) and re-runs it when the same request comes again.  With
    ``geomodeling.materialize`` off the tiles stay on their GPUs and the response holds a ``ShardedTiles`` handle
    (``to_numpy()``, ``checksums()``); such a request returns when every worker has queued its launches -- as a
    get_data that leaves a DeviceArray behind does in one process -- and ``Pool.sync()`` waits for the streams.

Workers are fresh interpreters (``python -m dask_geomodeling_amd.shard_worker``): started with fork + exec
*before* they touch HIP, never by fork() alone -- a HIP runtime does not survive fork(), the library refuses it
(GM_ERR_FORKED) -- and never by exec() of an initialised process.  The parent never needs a GPU of its own.
A worker that raises fails the request (its exception type, name and traceback) and stays; a worker that dies fails the
request within seconds, naming the worker: the pool is then torn down and the next request starts a fresh one.
"""
import atexit
import itertools
import mmap
import os
import pickle
import selectors
import subprocess
import sys
import threading
import time
import weakref
from multiprocessing.connection import Connection, wait as _wait

import numpy as np

from . import config as _config
from .core.graphs import get_sync

__all__ = ["tile_owner", "owned_tiles", "split_tiler_graph", "run_owned_tiles", "get", "get_data", "pool_for",
           "shutdown", "configured_devices", "ShardError", "WorkerDied", "ShardedTiles"]

SHARE_FROM = 1 << 20       # ndarrays of this many bytes and more travel through shared memory, once
MAX_JOBS = 64              # fused jobs a pool keeps per (view, request)
RESULT_CACHE_BYTES = 8 << 30   # host mosaics kept (pinned by the workers) for the next request of their size


class ShardError(RuntimeError):
    """A worker reported an error (the worker's traceback is part of the message)."""


class WorkerDied(ShardError):
    """A worker process went away in the middle of a request."""


# ---------------------------------------------------------------------------------------------------------
# dealing
# ---------------------------------------------------------------------------------------------------------

def tile_owner(tile_index, world):
    return tile_index % world


def owned_tiles(n_tiles, rank, world):
    """Indices of the tiles rank ``rank`` of ``world`` evaluates."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_tiles, world))


def split_tiler_graph(graph, name):
    """(process_kwargs, [tile keys]) of a RasterTiler end-point ``graph[name]``."""
    task = graph[name]
    return task[1], list(task[2:])


def run_owned_tiles(graph, name, rank, world, sink=None, get=get_sync):
    """Evaluate the tiles of a tiler graph this rank owns; ``sink(tile_index, response)`` gets
    every result (e.g. a checksum sink); returns {tile_index: sink result or response}."""
    _, tile_keys = split_tiler_graph(graph, name)
    out = {}
    for t in owned_tiles(len(tile_keys), rank, world):
        key = tile_keys[t]
        response = get(graph, [key])[0] if isinstance(key, str) and key in graph else key
        out[t] = sink(t, response) if sink is not None else response
    return out


def sharded_here():
    """True when this process hands its work to workers (a scheduler configured in dask, or ``scheduler_get``, owns the
    evaluation otherwise: core/graphs.py:42-49)."""
    if _config.get("scheduler_get") is not None:
        return False
    try:
        from dask.base import get_scheduler
    except ImportError:
        return True
    return get_scheduler() is None


def configured_devices():
    """``geomodeling.devices`` as a tuple of device indices, or None (evaluate in this process)."""
    devices = _config.get("geomodeling.devices")
    if devices is None or devices is False or devices == []:
        return None
    if devices == "all":
        from .hip import runtime
        n = runtime.device_count()  # (counts, does not initialise)
        if n <= 0:
            raise runtime.GmHipError("geomodeling.devices = 'all', but no HIP device is visible")
        return tuple(range(n))
    if isinstance(devices, int):
        devices = [devices]
    return tuple(int(d) for d in devices)


# ---------------------------------------------------------------------------------------------------------
# shared memory: /dev/shm files, mapped by the parent and by every worker
# ---------------------------------------------------------------------------------------------------------

_SHM_DIR = "/dev/shm" if os.path.isdir("/dev/shm") else None
_names = itertools.count()
_segments_lock = threading.RLock()   # (re-entrant: see _result_lock)
_live_segments = {}   # name -> Segment this process created (unlinked at exit at the latest)


class Segment(object):
    """A named shared-memory range.  The creating side owns the name (unlinks it); mappings stay valid after that."""

    def __init__(self, name, size, create):
        if _SHM_DIR is None:
            raise ShardError("no /dev/shm on this host: the sharded scheduler moves arrays through shared memory")
        self.name, self.size, self.owner = name, int(size), create
        path = os.path.join(_SHM_DIR, name)
        if create:
            # (a mapping of a shared-memory file that the file system cannot back dies with SIGBUS on first touch: ask first)
            room = os.statvfs(_SHM_DIR)
            if room.f_bavail * room.f_frsize < self.size + (16 << 20):
                raise ShardError("%s has %.1f MiB free, the scheduler needs %.1f MiB of shared memory there (arrays of views and "
                                 "host mosaics travel through it)" % (_SHM_DIR, room.f_bavail * room.f_frsize / 2.0 ** 20,
                                                                      self.size / 2.0 ** 20))
        fd = os.open(path, (os.O_CREAT | os.O_EXCL | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, max(self.size, 1))
            self.map = mmap.mmap(fd, max(self.size, 1))
        finally:
            os.close(fd)

    @classmethod
    def create(cls, size):
        name = "gm-shard-%d-%d-%s-%s-%s" % (os.getpid(), next(_names), os.urandom(4).hex(), _pid_namespace(),
                                            _start_time(os.getpid()))
        seg = cls(name, size, True)
        with _segments_lock:
            _live_segments[name] = seg
        return seg

    def array(self, dtype, shape, offset=0):
        return np.frombuffer(self.map, dtype=dtype, count=int(np.prod(shape, dtype=np.int64)), offset=offset).reshape(shape)

    def unlink(self):
        if self.owner:
            with _segments_lock:
                if _live_segments.pop(self.name, None) is None:
                    return
            for pool in list(_pools.values()):  # (the workers may drop their mapping: told with the next message)
                pool.forgotten.append(self.name)
            try:
                os.unlink(os.path.join(_SHM_DIR, self.name))
            except OSError:
                pass


def _pid_namespace():
    """Identity of this process's pid namespace (a /dev/shm shared between containers holds files of several)."""
    try:
        return "%x" % os.stat("/proc/self/ns/pid").st_ino
    except OSError:
        return "0"


def _start_time(pid):
    """Start time (clock ticks since boot) of a process, "0" where /proc does not say: tells a creator from a later process
    that got its pid."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def sweep_stale_segments():
    """Remove the shared-memory files of processes that no longer exist (a parent that was killed could not unlink its own:
    the files are named after their creator's pid, pid namespace and start time).  Files of another pid namespace are not
    ours to judge (their creator's pid means nothing here); a pid that lives but started at another time is a later process.
    Called when a pool starts; returns the names removed."""
    removed = []
    namespace = _pid_namespace()
    if _SHM_DIR is None:
        return removed
    try:
        names = [n for n in os.listdir(_SHM_DIR) if n.startswith("gm-shard-")]
    except OSError:
        return removed
    for name in names:
        parts = name.split("-")
        try:
            pid = int(parts[2])
        except (IndexError, ValueError):
            continue
        if pid == os.getpid() or len(parts) < 7 or parts[5] != namespace:
            continue
        try:
            os.kill(pid, 0)
            if parts[6] != "0" and _start_time(pid) not in ("0", parts[6]):
                raise ProcessLookupError  # (the pid lives, its owner is a later process)
        except ProcessLookupError:
            try:
                os.unlink(os.path.join(_SHM_DIR, name))
                removed.append(name)
            except OSError:
                pass
        except OSError:  # (exists, somebody else's: not ours to judge)
            pass
    return removed


def _unlink_all():
    with _segments_lock:
        names = list(_live_segments)
        _live_segments.clear()
    for name in names:
        try:
            os.unlink(os.path.join(_SHM_DIR, name))
        except OSError:
            pass


atexit.register(_unlink_all)

# large arrays of graphs and views: shared once per array object, dropped with it
_shared_arrays = {}   # id(array) -> (Segment, dtype str, shape)


def _share_array(obj, host):
    hit = _shared_arrays.get(id(obj))
    if hit is None:
        seg = Segment.create(host.nbytes)
        np.copyto(seg.array(host.dtype, host.shape), host)
        hit = _shared_arrays[id(obj)] = (seg, host.dtype.str, tuple(host.shape))
        key = id(obj)

        def gone(key=key, seg=seg):
            _shared_arrays.pop(key, None)
            seg.unlink()
        try:
            weakref.finalize(obj, gone)
        except TypeError:  # (not weak-referenceable: lives as long as the process)
            pass
    # (shared by value at the time of the first message that mentions it: like a dask token, an array in a graph is taken to be
    #  immutable -- a later in-place change does not reach the workers)
    return ("gm-array", hit[0].name, hit[0].size, hit[1], hit[2])


try:  # functions of __main__ and lambdas in a graph travel by value, as under dask's own multiprocessing scheduler
    from cloudpickle import CloudPickler as _BasePickler
except ImportError:  # pragma: no cover - plain pickle: importable functions only
    _BasePickler = pickle.Pickler


class _Pickler(_BasePickler):
    """Pickle of messages to the workers: large ndarrays (the data of a MemorySource) and device-resident arrays go
    by name through shared memory -- once, however many requests mention them."""

    def persistent_id(self, obj):
        kind = type(obj)
        if kind is np.ndarray:
            if obj.nbytes >= SHARE_FROM and obj.dtype != object and _SHM_DIR is not None:
                return _share_array(obj, obj)
            return None
        if kind.__name__ == "DeviceArray" and kind.__module__.endswith("hip.array"):
            host = obj.to_numpy()
            if host.nbytes >= SHARE_FROM and _SHM_DIR is not None:
                return _share_array(obj, host)
            return ("gm-small-device-array", host)
        return None


_PLAIN = (int, str, float, bool, type(None), bytes)


def _is_plain(obj, depth=0):
    """Nothing but numbers, strings and tuples / lists of them: a message the C pickler writes as it stands (re-runs of kept
    jobs, syncs, marks: every message of a steady state)."""
    if type(obj) in _PLAIN:
        return True
    if type(obj) in (tuple, list) and depth < 4:
        return all(_is_plain(x, depth + 1) for x in obj)
    return False


def _dumps(obj):
    if _is_plain(obj):  # (the cloudpickle / persistent-id machinery below costs ~10 us per message: a re-run at 8 workers is 8 of them)
        return pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    import io
    buf = io.BytesIO()
    _Pickler(buf, protocol=pickle.HIGHEST_PROTOCOL).dump(obj)
    return buf.getvalue()


# result mosaics: shared-memory ranges the workers have pinned, recycled by size when the ndarray a caller was given
# has gone (pinning a range anew costs more than the copy into it)
_result_lock = threading.RLock()   # (re-entrant: a lease finalised by a GC pass inside the lock comes back in here)
_result_free = {}     # size -> [Segment]
_result_free_bytes = 0


class _Lease(object):
    """Ties a result segment to the ndarray that views it: back to the free list with the array."""

    def __init__(self, segment):
        self.segment = segment

    def __del__(self):
        try:
            _return_result_segment(self.segment)
        except Exception:
            pass


def _result_segment(nbytes):
    global _result_free_bytes
    size = max((int(nbytes) + 4095) // 4096 * 4096, 4096)
    with _result_lock:
        free = _result_free.get(size)
        if free:
            _result_free_bytes -= size
            return free.pop()
    return Segment.create(size)


def _return_result_segment(segment):
    global _result_free_bytes
    with _result_lock:
        if _result_free_bytes + segment.size <= RESULT_CACHE_BYTES and segment.name in _live_segments:
            _result_free.setdefault(segment.size, []).append(segment)
            _result_free_bytes += segment.size
            return
    segment.unlink()


def _leased_array(segment, dtype, shape):
    """ndarray over the segment whose death hands the segment back (views keep it alive through their base)."""
    import ctypes
    raw = (ctypes.c_ubyte * segment.size).from_buffer(segment.map)
    raw._lease = _Lease(segment)
    count = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)


# ---------------------------------------------------------------------------------------------------------
# the pool
# ---------------------------------------------------------------------------------------------------------

class _Worker(object):
    def __init__(self, index, device, settings):
        import socket
        self.index, self.device = index, device
        ours, theirs = socket.socketpair()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env["PYTHONPATH"] = os.pathsep.join([root] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
        env.setdefault("OMP_NUM_THREADS", "1")
        for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):  # (a worker is nobody's rank)
            env.pop(name, None)
        # a fresh interpreter: fork + exec before anything of HIP exists in the child; its stdout is our stderr (stdout
        # may be somebody's protocol, e.g. the one JSON line of bench.py)
        self.proc = subprocess.Popen([sys.executable, "-m", "dask_geomodeling_amd.shard_worker", str(theirs.fileno())],
                                     pass_fds=[theirs.fileno()], env=env, stdout=2, close_fds=True)
        theirs.close()
        self.conn = Connection(ours.detach())
        self.pid = self.proc.pid
        self.conn.send_bytes(_dumps(("init", dict(settings, device=device, index=index))))

    def describe(self):
        return "shard worker %d (device %s, pid %d)" % (self.index, self.device, self.pid)


class Pool(object):
    """G worker processes, worker i bound to device ``devices[i]``."""

    def __init__(self, devices, preload=None, start_timeout=300.0):
        self.devices = tuple(devices)
        if not self.devices:
            raise ValueError("a pool needs at least one device")
        self.lock = threading.RLock()
        self.jobs = {}           # job key -> job id (least recently used first)
        self.job_ids = itertools.count(1)
        self.run_ids = itertools.count(1)
        self.released = []       # run ids whose results the workers may drop (piggybacked on the next message)
        self.forgotten = []      # names of shared-memory ranges that are gone (likewise)
        self.closed = False
        self.config = _forwarded_config()   # what the workers evaluate under: re-sent when it changes (refresh_config)
        settings = {"preload": preload if preload is not None else _config.get("geomodeling.worker-preload"),
                    "path": [p for p in sys.path if p], "config": self.config,
                    "affinity": bool(_config.get("geomodeling.worker-affinity", True)), "pool_devices": self.devices}
        self.workers = []
        sweep_stale_segments()
        try:
            for i, device in enumerate(self.devices):
                self.workers.append(_Worker(i, device, settings))
            # (one selector for the life of the pool: multiprocessing.connection.wait builds and fills one per call -- a third of
            #  what a re-run at eight workers cost the parent)
            self.selector = selectors.DefaultSelector()
            for w in self.workers:
                self.selector.register(w.conn.fileno(), selectors.EVENT_READ, w.conn)
            ready = self._collect({w.index: None for w in self.workers}, start_timeout, "start")
            self.device_names = [ready[w.index] for w in self.workers]
        except BaseException:
            self.close(kill=True)
            raise

    # -- messaging ------------------------------------------------------------------------------------------
    def _housekeeping(self):
        # (swapped, not copied-then-cleared: a run id appended by a finaliser between the two would be lost)
        names, self.forgotten = self.forgotten, []
        runs, self.released = self.released, []
        return names, runs

    def request(self, messages, timeout=None, what="request"):
        """{worker index: message} -> {worker index: reply payload}; all workers work at the same time."""
        with self.lock:
            if self.closed:
                raise ShardError("the worker pool has been shut down")
            names, runs = self._housekeeping()
            for w in self.workers:  # (housekeeping goes to everybody, also to those without work in this request)
                message = messages.get(w.index)
                if message is None and not (names or runs):
                    continue
                try:
                    w.conn.send_bytes(_dumps(("do", message, names, runs)))
                except (OSError, ValueError) as e:
                    self._fail(WorkerDied("%s is gone (%s) before %s" % (w.describe(), e, what)))
            return self._collect({i: None for i in messages}, timeout, what)

    def _collect(self, pending, timeout, what):
        """The replies of the workers in ``pending``.  A worker that DIED fails the request at once and takes the pool with it;
        a worker that RAISED is alive and in step: the other replies of this request are waited for (so that none of them is
        mistaken for the answer to the next one), then the first error is raised and the pool lives on."""
        by_conn = {self.workers[i].conn: self.workers[i] for i in pending}
        replies, errors = {}, []
        deadline = None if timeout is None else time.time() + timeout
        while by_conn:
            selector = getattr(self, "selector", None)
            if selector is not None:
                ready = [key.data for key, _ in selector.select(0.25) if key.data in by_conn]
            else:
                ready = _wait(list(by_conn), 0.25)
            for conn in ready:
                w = by_conn[conn]
                try:
                    kind, payload = pickle.loads(conn.recv_bytes())
                except (EOFError, OSError):
                    code = w.proc.poll()
                    if code is None:
                        try:
                            code = w.proc.wait(timeout=2.0)
                        except subprocess.TimeoutExpired:
                            code = "unknown"
                    self._fail(WorkerDied("%s died (exit code %s) during %s -- the other workers were stopped"
                                          % (w.describe(), code, what)))
                if kind == "error":
                    errors.append(_worker_error(w, payload, what))
                else:
                    replies[w.index] = payload
                del by_conn[conn]
            if not ready:
                for w in list(by_conn.values()):
                    code = w.proc.poll()
                    if code is not None and not w.conn.poll(0):
                        self._fail(WorkerDied("%s died (exit code %s) during %s -- the other workers were stopped"
                                              % (w.describe(), code, what)))
                if deadline is not None and time.time() > deadline:
                    self._fail(ShardError("no answer from %s within %.0f s during %s"
                                          % (", ".join(w.describe() for w in by_conn.values()), timeout, what)))
        if errors:
            raise errors[0]
        return replies

    def _fail(self, error):
        self.close(kill=True)
        raise error

    def close(self, kill=False):
        with self.lock:
            if self.closed:
                return
            self.closed = True
            if _pools.get(self.devices) is self:
                del _pools[self.devices]
            for w in self.workers:
                try:
                    if not kill:
                        w.conn.send_bytes(_dumps(("stop",)))
                except (OSError, ValueError):
                    pass
            deadline = time.time() + (0.0 if kill else 10.0)
            for w in self.workers:
                try:
                    w.proc.wait(timeout=max(0.0, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    w.proc.terminate()
                    try:
                        w.proc.wait(timeout=3.0)
                    except subprocess.TimeoutExpired:
                        w.proc.kill()
                        w.proc.wait()
                try:
                    if getattr(self, "selector", None) is not None:
                        self.selector.unregister(w.conn.fileno())
                except (KeyError, ValueError, OSError):
                    pass
                try:
                    w.conn.close()
                except OSError:
                    pass
            if getattr(self, "selector", None) is not None:
                try:
                    self.selector.close()
                except OSError:
                    pass
                self.selector = None

    def refresh_config(self):
        """The workers evaluate under the parent's geomodeling.* settings AS THEY ARE NOW: a config.set after the pool
        started (raster-limit, fuse, root, ...) travels before the next request, and the jobs kept under the old settings --
        fused or not, checked against the old limits -- are forgotten on both sides."""
        with self.lock:
            now = _forwarded_config()
            if now == self.config or self.closed:
                return
            stale = [job.id for job in self.jobs.values()]
            self.jobs.clear()
            self.request({w.index: ("config", now, stale) for w in self.workers}, what="a change of configuration")
            self.config = now

    # -- jobs -----------------------------------------------------------------------------------------------
    def job_for(self, key):
        """(job record, known to the workers already?) of a job key; the least recently used jobs are dropped (-> their ids)."""
        job = self.jobs.pop(key, None)
        known = job is not None
        if job is None:
            job = _Job(next(self.job_ids))
        self.jobs[key] = job
        dropped = []
        while len(self.jobs) > MAX_JOBS:
            dropped.append(self.jobs.pop(next(iter(self.jobs))).id)
        return job, known, dropped

    def sync(self):
        """Wait until every worker's compute stream is empty (requests return when their launches are queued)."""
        self.request({w.index: ("sync",) for w in self.workers}, what="sync")


class _Job(object):
    """What the parent remembers of a (view, request) the workers keep a fused job for."""

    __slots__ = ("id", "kwargs", "n_tiles", "shares", "owners", "metas")

    def __init__(self, job_id):
        self.id = job_id
        self.kwargs = self.shares = self.owners = None
        self.n_tiles = 0
        self.metas = {}   # worker index -> metas of its last answer


def _worker_error(worker, payload, what):
    text, trace, pickled = payload
    error = None
    if pickled is not None:
        try:
            error = pickle.loads(pickled)
        except Exception:
            error = None
    note = "%s failed during %s: %s\n--- worker traceback ---\n%s" % (worker.describe(), what, text, trace)
    if isinstance(error, Exception):
        try:
            error.args = (note,) + tuple(error.args[1:])
            return error
        except Exception:
            pass
    return ShardError(note)


def _forwarded_config():
    """The geomodeling.* settings the workers evaluate under (never the device list itself)."""
    keys = ("fuse", "plans", "templates", "resident-window-bytes", "raster-limit", "raster-limit-timesteps",
            "geometry-limit", "root", "strict-file-paths")
    return {"geomodeling." + k: _config.get("geomodeling." + k) for k in keys}


_pools = {}
_pools_lock = threading.Lock()


def pool_for(devices=None, preload=None):
    """The (process-wide) pool over ``devices``; started on first use."""
    devices = tuple(devices) if devices is not None else configured_devices()
    if not devices:
        raise ValueError("no devices: set geomodeling.devices or pass them")
    with _pools_lock:
        pool = _pools.get(devices)
        if pool is None or pool.closed:
            pool = _pools[devices] = Pool(devices, preload=preload)
        return pool


def shutdown():
    """Stop every worker pool of this process."""
    with _pools_lock:
        pools = list(_pools.values())
        _pools.clear()
    for pool in pools:
        pool.close()
    # the result segments kept for the next request: nobody has them mapped any more
    global _result_free_bytes
    with _result_lock:
        kept = [seg for segs in _result_free.values() for seg in segs]
        _result_free.clear()
        _result_free_bytes = 0
    for seg in kept:
        seg.unlink()


atexit.register(shutdown)


# ---------------------------------------------------------------------------------------------------------
# results
# ---------------------------------------------------------------------------------------------------------

class _Placement(object):
    """Where the raster results of one run go: a host range per unit (a tile's slot of a mosaic, or a range of its own)."""

    def __init__(self):
        self.nbytes = 0
        self.parts = []    # (worker, unit, offset, row pitch bytes, plane bytes, dtype str, bands, rows, cols)
        self.fills = []    # (offset, dtype, shape, fill value) ranges to fill before the tiles land

    def reserve(self, nbytes):
        offset = (self.nbytes + 4095) // 4096 * 4096
        self.nbytes = offset + int(nbytes)
        return offset


def _tile_slots(kwargs, n_tiles):
    """(row, column) of the top-left cell each tile index may claim, with the slot's size, after parallelize.py:114-124:
    tiles come x-major, low y first; low y is the high row index."""
    height, width = kwargs["shape_yx"]
    count_x, count_y = kwargs["count_xy"]
    tile_x, tile_y = kwargs["tilesize_xy"]
    if n_tiles != count_x * count_y:
        raise ValueError("a mosaic of %d x %d tiles got %d" % (count_x, count_y, n_tiles))
    return [(ix, iy, tile_x, tile_y, height, width) for ix in range(count_x) for iy in range(count_y)]


def _plan_mosaic(placement, kwargs, metas, owners):
    """Reserve the host mosaic of one RasterTiler end-point and say where every tile lands.  ``metas[t]``: None, or
    ("raster", shape, dtype str, no-data) of tile t; ``owners[t]``: (worker, unit).  -> (offset, shape, dtype) or None."""
    live = [m for m in metas if m is not None]
    if not metas:
        return None
    if not live:
        return None
    for m in live:
        if m[0] != "raster" or len(m[1]) != 3:
            raise ValueError("a RasterTiler got a tile that is no (bands, y, x) raster: %r" % (m,))
    height, width = kwargs["shape_yx"]
    bands = live[0][1][0]
    dtype = np.dtype(kwargs["dtype"])
    fill = kwargs["fillvalue"]
    shape = (bands, height, width)
    offset = placement.reserve(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize)
    covered = 0
    for (ix, iy, tile_x, tile_y, _, _), meta, owner in zip(_tile_slots(kwargs, len(metas)), metas, owners):
        if meta is None:
            continue
        tb, th, tw = meta[1]
        row, col = height - (iy * tile_y + th), ix * tile_x
        if tb not in (bands, 1) or row < 0 or col + tw > width or th > height:
            raise ValueError("could not broadcast input array from shape %s into the mosaic of shape %s at row %d, column %d"
                             % (tuple(meta[1]), shape, row, col))
        if th <= tile_y and tw <= tile_x:
            covered += th * tw
        placement.parts.append((owner[0], owner[1], offset + (row * width + col) * dtype.itemsize, width * dtype.itemsize,
                                height * width * dtype.itemsize, dtype.str, bands, th, tw))
    if covered != height * width:
        # np.full(shape, fill) of parallelize.py:108-112 (a boolean view has no fill value: all False)
        placement.fills.append((offset, dtype, shape, False if fill is None else fill))
    return offset, shape, dtype


def _plan_raster(placement, meta, owner):
    _, shape, dtype, _ = meta
    dtype = np.dtype(dtype)
    shape = tuple(shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    offset = placement.reserve(nbytes)
    if len(shape) >= 2:
        rows, cols = shape[-2], shape[-1]
        bands = int(np.prod(shape[:-2], dtype=np.int64))
    else:
        rows, cols, bands = 1, int(np.prod(shape, dtype=np.int64)), 1
    placement.parts.append((owner[0], owner[1], offset, cols * dtype.itemsize, rows * cols * dtype.itemsize, dtype.str,
                            bands, rows, cols))
    return offset, shape, dtype


def _fetch(pool, run_id, placement, what):
    """The raster results of run ``run_id`` on the host: one shared-memory range, every worker copies its units to their
    place.  -> the Segment."""
    segment = _result_segment(placement.nbytes)
    try:
        for offset, dtype, shape, fill in placement.fills:
            segment.array(dtype, shape, offset)[...] = fill
        messages = {}
        for part in placement.parts:
            messages.setdefault(part[0], []).append(part[1:])
        pool.request({w: ("fetch", run_id, segment.name, segment.size, parts) for w, parts in messages.items()},
                     what="the device-to-host copies of " + what)
    except BaseException:
        _return_result_segment(segment)
        raise
    return segment


class ShardedTiles(object):
    """The tiles of one request, left on the GPUs that computed them (``geomodeling.materialize`` off).  Quacks enough like an
    array for a raster response: ``shape``, ``dtype``, ``np.asarray()``.  The workers drop the tiles when this object goes."""

    def __init__(self, pool, run_id, assemble, shape, dtype, what, units):
        self._pool, self._run, self._assemble, self._what = pool, run_id, assemble, what
        self.shape, self.dtype, self.units = tuple(shape), np.dtype(dtype), units
        self.ndim = len(self.shape)
        self.size = int(np.prod(self.shape, dtype=np.int64))

    def to_numpy(self):
        return self._assemble()

    def __array__(self, dtype=None, copy=None):
        out = self.to_numpy()
        return out if dtype is None else out.astype(dtype, copy=False)

    def checksums(self):
        """{unit index: 64-bit checksum of the unit's values, computed on its GPU} (None units are left out)."""
        by_worker = {}
        for worker, unit, index in self.units:
            by_worker.setdefault(worker, []).append((unit, index))
        replies = self._pool.request({w: ("checksums", self._run, [u for u, _ in units]) for w, units in by_worker.items()},
                                     what="the checksums of " + self._what)
        out = {}
        for w, units in by_worker.items():
            for (unit, index), value in zip(units, replies[w]):
                if value is not None:
                    out[index] = value
        return out

    def __repr__(self):
        return "ShardedTiles(shape=%s, dtype=%s, %d units on %d devices)" % (self.shape, self.dtype, len(self.units),
                                                                            len(self._pool.devices))

    def __del__(self):
        try:
            if not self._pool.closed:
                self._pool.released.append(self._run)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------
# get(graph, keys): the dask-compatible entry
# ---------------------------------------------------------------------------------------------------------

def _is_task(x):
    return isinstance(x, tuple) and len(x) > 0 and callable(x[0])


def _is_key(x, graph):
    if isinstance(x, (list, dict, np.ndarray)) or _is_task(x):
        return False
    try:
        return x in graph
    except TypeError:
        return False


def _reach(graph, keys):
    """The part of ``graph`` the keys need."""
    out, stack = {}, list(keys)
    while stack:
        key = stack.pop()
        if key in out:
            continue
        out[key] = task = graph[key]
        todo = [task]
        while todo:
            obj = todo.pop()
            if _is_task(obj):
                todo.extend(obj[1:])
            elif isinstance(obj, list):
                todo.extend(obj)
            elif _is_key(obj, graph) and obj not in out:
                stack.append(obj)
    return out


def _flatten(keys):
    if isinstance(keys, list):
        return [k for item in keys for k in _flatten(item)]
    return [keys]


def _rebuild(keys, values):
    if isinstance(keys, list):
        return [_rebuild(k, values) for k in keys]
    return values[keys]


def get(dsk, keys, devices=None, **kwargs):
    """``get(graph, keys)`` with dask's calling convention over one worker process per GPU.  Raster results come back as
    host ndarrays (a scheduler's results leave the workers), the mosaic of a ``RasterTiler`` key assembled from its tiles."""
    pool = pool_for(devices)
    pool.refresh_config()
    flat = _flatten(keys)
    G = len(pool.devices)
    units, plans = [], {}   # units: graph keys to evaluate, in dealing order; plans: wanted key -> how it is made of units
    for key in dict.fromkeys(flat):
        task = dsk[key]
        if _is_task(task) and getattr(task[0], "gm_tiler", False) and len(task) > 2 and task[1] is not None:
            tiles = []
            for arg in task[2:]:
                if _is_key(arg, dsk):
                    tiles.append(("unit", len(units)))
                    units.append(arg)
                else:
                    tiles.append(("literal", arg))
            plans[key] = ("mosaic", task[0], task[1], tiles)
        else:
            plans[key] = ("unit", len(units))
            units.append(key)
    with pool.lock:
        job = next(pool.job_ids)
        run_id = next(pool.run_ids)
        messages, owners = {}, []
        per_worker = [[] for _ in range(G)]
        for u, key in enumerate(units):
            w = tile_owner(u, G)
            owners.append((w, len(per_worker[w])))
            per_worker[w].append(key)
        for w, mine in enumerate(per_worker):
            if mine:
                messages[w] = ("run", job, run_id, ("graph", _reach(dsk, mine), mine), False)
        what = "get() of %d keys" % len(flat)
        try:
            replies = pool.request(messages, what=what) if messages else {}
            metas = [replies[w][i] for w, i in owners]
            values = _assemble_results(pool, run_id, plans, metas, owners, what)
        finally:
            if not pool.closed:
                pool.released.append(run_id)
                pool.request({w: ("drop", [job]) for w in messages}, what="clean-up")
    return _rebuild(keys, values)


def _assemble_results(pool, run_id, plans, metas, owners, what):
    placement = _Placement()
    made = {}
    for key, plan in plans.items():
        if plan[0] == "unit":
            meta = metas[plan[1]]
            if meta is not None and meta[0] == "raster":
                made[key] = ("raster", _plan_raster(placement, meta, owners[plan[1]]), meta[3])
            else:
                made[key] = ("object", None if meta is None else meta[1])
        else:
            _, process, kwargs, tiles = plan
            if any(kind == "literal" and value is not None for kind, value in tiles):
                raise ShardError("a RasterTiler task with literal tiles cannot be dealt over workers")
            tile_metas = [metas[value] if kind == "unit" else None for kind, value in tiles]
            tile_owners = [owners[value] if kind == "unit" else None for kind, value in tiles]
            spot = _plan_mosaic(placement, kwargs, tile_metas, tile_owners)
            made[key] = ("object", None) if spot is None else ("raster", spot, kwargs["fillvalue"])
    segment = _fetch(pool, run_id, placement, what) if placement.parts or placement.fills else None
    values = {}
    whole = None
    if segment is not None:
        whole = _leased_array(segment, np.uint8, (segment.size,))
    for key, (kind, spot, *rest) in made.items():
        if kind == "object":
            values[key] = spot
            continue
        offset, shape, dtype = spot
        count = int(np.prod(shape, dtype=np.int64))
        arr = whole[offset:offset + count * dtype.itemsize].view(dtype).reshape(shape)
        values[key] = {"values": arr, "no_data_value": rest[0]}
    return values


# ---------------------------------------------------------------------------------------------------------
# get_data(view, request): what Block.get_data calls when geomodeling.devices is set
# ---------------------------------------------------------------------------------------------------------

def _request_key(request):
    try:
        key = tuple(sorted((k, tuple(v) if isinstance(v, list) else v) for k, v in request.items()))
        hash(key)
        return key
    except TypeError:
        return None


def get_data(view, request, devices=None, materialize=None):
    """``view.get_data(**request)`` over the pool.  A ``RasterTiler`` deals its tiles ``t % G``; any other view is one unit
    on one worker.  -> the response, host-resident (``materialize``, default ``geomodeling.materialize``) or a
    ``ShardedTiles`` left on the GPUs."""
    pool = pool_for(devices)
    pool.refresh_config()
    if materialize is None:
        materialize = bool(_config.get("geomodeling.materialize", True))
    G = len(pool.devices)
    rkey = _request_key(request)
    with pool.lock:
        key = (view.token, rkey) if rkey is not None else None
        if key is not None:
            job, known, dropped = pool.job_for(key)
        else:
            job, known, dropped = _Job(next(pool.job_ids)), False, []
        if not known:
            # the dealing: a RasterTiler's tiles t % G (parallelize.py:82 emits them x-major, low y first); any other view is
            # one unit on one worker
            if getattr(type(view).process, "gm_tiler", False) and request.get("mode") == "vals":
                pairs = list(view.get_sources_and_requests(**request))
                if pairs and pairs[0][0] is not None and len(pairs) > 1:
                    job.kwargs, job.n_tiles = pairs[0][0], len(pairs) - 1
            if job.n_tiles:
                shares = {w: owned_tiles(job.n_tiles, w, G) for w in range(G)}
                job.shares = {w: mine for w, mine in shares.items() if mine}
                job.owners = [(tile_owner(t, G), t // G) for t in range(job.n_tiles)]
            else:
                w = int(view.token[:8], 16) % G if isinstance(view.token, str) else 0
                job.shares, job.owners = {w: None}, [(w, 0)]
        kwargs, n_tiles, owners = job.kwargs, job.n_tiles, job.owners
        run_id = next(pool.run_ids)
        messages = {}
        for w, mine in job.shares.items():
            if known:
                messages[w] = ("rerun", job.id, run_id)
            else:
                spec = ("tiles", view, request, mine) if n_tiles else ("view", view, request)
                messages[w] = ("run", job.id, run_id, spec, True)
        what = "get_data of %s" % type(view).__name__
        if dropped:
            pool.request({w.index: ("drop", dropped) for w in pool.workers}, what="clean-up")
        try:
            replies = pool.request(messages, what=what)
        except BaseException:
            pool.jobs.pop(key, None)
            if not pool.closed:  # (a worker raised: the pool lives on, the half-made job is forgotten everywhere)
                pool.released.append(run_id)
                pool.request({w: ("drop", [job.id]) for w in messages}, what="clean-up")
            raise
        for w, reply in replies.items():
            if not isinstance(reply, str):  # ("same": what this worker answered last time)
                job.metas[w] = reply
        metas = [job.metas[w][i] for w, i in owners]
        if key is None:
            pool.request({w: ("drop", [job.id]) for w in messages}, what="clean-up")

        def assemble():
            plans = {"result": ("mosaic", None, kwargs, [("unit", t) for t in range(n_tiles)]) if n_tiles else ("unit", 0)}
            with pool.lock:
                return _assemble_results(pool, run_id, plans, metas, owners, what)["result"]

        if materialize:
            try:
                return assemble()
            finally:
                pool.released.append(run_id)
        # left on the devices: the response's values are a handle
        if n_tiles:
            live = [m for m in metas if m is not None]
            if not live:
                pool.released.append(run_id)
                return None
            shape = (live[0][1][0],) + tuple(kwargs["shape_yx"])
            dtype, ndv = kwargs["dtype"], kwargs["fillvalue"]
        else:
            meta = metas[0]
            if meta is None or meta[0] != "raster":
                pool.released.append(run_id)
                return None if meta is None else meta[1]
            shape, dtype, ndv = meta[1], meta[2], meta[3]
        units = [(w, u, t) for t, ((w, u), m) in enumerate(zip(owners, metas)) if m is not None]
        handle = ShardedTiles(pool, run_id, lambda: assemble()["values"], shape, dtype, what, units)
        return {"values": handle, "no_data_value": ndv}
