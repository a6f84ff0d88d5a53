"""Graph core: Block, compute, construct (API of the reference's core/graphs.py:18-290).

A Block turns a request into a dask-style graph ``{key: (process, *args)}``; keys are
``"<classname lowercase>_<32 hex token>"`` (graphs.py:173).  The graph can be run by
any dask ``get``; without dask a built-in synchronous evaluator is used.  On top of
the reference API, ``get_data`` runs the fusion pass (dask_geomodeling_amd/fuse.py)
that collapses element-wise sub-graphs into one HIP launch and hands back ndarrays.
"""
import hashlib
import inspect
import json
import logging
import pickle
import sys
from datetime import datetime, timedelta

import numpy as np

from .. import config as _config

try:  # pragma: no cover - dask is optional
    from dask.base import get_scheduler as _dask_get_scheduler
    from dask.base import tokenize as _dask_tokenize
    from dask.base import normalize_token as _dask_normalize_token
except ImportError:
    _dask_get_scheduler = _dask_tokenize = _dask_normalize_token = None

logger = logging.getLogger(__name__)

__all__ = ["construct", "construct_multiple", "compute", "Block", "DummyBlock", "tokenize",
           "get_sync", "get_data_many"]


# ---------------------------------------------------------------------------
# tokenising: dask's when available, else md5 over a canonical byte stream
# ---------------------------------------------------------------------------

def _feed(h, obj):
    if obj is None or isinstance(obj, (bool, int, float, complex, str, bytes)):
        h.update(("%s:%r;" % (type(obj).__name__, obj)).encode())
    elif isinstance(obj, np.ndarray):
        h.update(("nd:%s:%s;" % (obj.dtype.str, obj.shape)).encode())
        h.update(np.ascontiguousarray(obj).view(np.uint8).data if obj.dtype != object
                 else pickle.dumps(obj.tolist()))
    elif isinstance(obj, np.generic):
        h.update(("np:%s:" % obj.dtype.str).encode() + obj.tobytes())
    elif isinstance(obj, np.dtype):
        h.update(("dt:%s;" % obj.str).encode())
    elif isinstance(obj, (list, tuple)):
        h.update(("%s[%d]:" % (type(obj).__name__, len(obj))).encode())
        for item in obj:
            _feed(h, item)
    elif isinstance(obj, dict):
        h.update(("dict[%d]:" % len(obj)).encode())
        for key in sorted(obj, key=repr):
            _feed(h, key)
            _feed(h, obj[key])
    elif isinstance(obj, (datetime, timedelta)):
        h.update(repr(obj).encode())
    elif hasattr(obj, "__gm_tokenize__"):
        _feed(h, obj.__gm_tokenize__())
    else:
        try:
            h.update(pickle.dumps(obj, protocol=4))
        except Exception:  # not tokenisable: unique (random) token, like dask
            import uuid
            h.update(uuid.uuid4().bytes)


def tokenize(*args):
    """Deterministic 32-hex-char token of the arguments."""
    if _dask_tokenize is not None:
        return _dask_tokenize(*args)
    h = hashlib.md5()
    _feed(h, args)
    return h.hexdigest()


_PLAIN = (str, int, float, bool, type(None), datetime, timedelta)


def _request_text(request):
    """The request as text for the graph key, or None when it holds more than plain scalars and tuples of them (the general
    tokeniser then)."""
    for value in request.values():
        if not isinstance(value, _PLAIN):
            if not (isinstance(value, (tuple, list)) and all(isinstance(v, _PLAIN) for v in value)):
                return None
    return repr(sorted(request.items()))


def _request_token(block_token, request, text=False):
    """Token of (block, request) for the graph key.  A tiled view asks for it once per Block per
    tile, so requests made of plain scalars and tuples of them -- all raster requests -- take a
    repr + one md5 instead of the general recursive tokeniser (6x cheaper).  ``text``: the request's
    text when the caller has it already (Block.get_compute_graph makes it once per request object)."""
    if text is False:
        text = _request_text(request)
    if text is None:
        return tokenize([block_token, request])
    return hashlib.md5((block_token + text).encode()).hexdigest()


if _dask_normalize_token is not None:  # pragma: no cover
    @_dask_normalize_token.register((datetime, timedelta))
    def _normalize_datetime(value):
        return repr(value)


# ---------------------------------------------------------------------------
# evaluation
# ---------------------------------------------------------------------------

def _is_task(x):
    return isinstance(x, tuple) and len(x) > 0 and callable(x[0])


def _is_key(x, dsk):
    if isinstance(x, (list, dict, np.ndarray)) or _is_task(x):
        return False
    try:
        return x in dsk
    except TypeError:
        return False


# One look at every graph value instead of two (dependencies, then execution): a value is turned into a small plan
# -- (_KEY, key), (_FLAT, func, args, positions of the args that are keys), (_TASK, func, [plans]), (_LIST, [plans])
# or (_CONST, object) -- together with the keys it needs.  Nearly every task is flat: a function over keys and plain
# arguments (a fused launch over its leaf tiles, a source over its request).
_KEY, _FLAT, _TASK, _LIST, _CONST = range(5)
_PLAIN_NUMBERS = frozenset((float, int, np.float64, np.float32, np.int64, np.int32))


def _plan_of(obj, dsk, needs):
    kind = type(obj)
    if kind is str:  # (most arguments: graph keys are strings)
        if obj in dsk:
            needs.append(obj)
            return (_KEY, obj)
        return (_CONST, obj)
    if kind is tuple and len(obj) > 0 and callable(obj[0]):
        args = obj[1:]
        at, flat = [], True
        for i, a in enumerate(args):
            ka = type(a)
            if ka is str:
                if a in dsk:
                    needs.append(a)
                    at.append(i)
            elif ka is list or (ka is tuple and len(a) > 0 and callable(a[0])):
                flat = False
                break
            elif ka is not dict and ka is not np.ndarray and a is not None:
                try:
                    if a in dsk:
                        needs.append(a)
                        at.append(i)
                except TypeError:
                    pass
        if flat:
            return (_FLAT, obj[0], args, tuple(at))
        del needs[len(needs) - len(at):]  # nested arguments: the general form collects the keys itself
        return (_TASK, obj[0], [_plan_of(a, dsk, needs) for a in args])
    if kind is list:
        if len(obj) > 16 and set(map(type, obj)) <= _PLAIN_NUMBERS and dsk.keys().isdisjoint(obj):
            # a long list of numbers none of which is a key (the 200 edges of a Classify): one constant, not 200 plans --
            # the walk cost 50 us of the 106 us such a request takes through this scheduler
            return (_CONST, obj)
        return (_LIST, [_plan_of(a, dsk, needs) for a in obj])
    if _is_key(obj, dsk):
        needs.append(obj)
        return (_KEY, obj)
    return (_CONST, obj)


def _run(plan, cache):
    kind = plan[0]
    if kind == _FLAT:
        at = plan[3]
        if not at:
            return plan[1](*plan[2])
        args = list(plan[2])
        for i in at:
            args[i] = cache[args[i]]
        return plan[1](*args)
    if kind == _KEY:
        return cache[plan[1]]
    if kind == _CONST:
        return plan[1]
    if kind == _TASK:
        return plan[1](*[_run(a, cache) for a in plan[2]])
    return [_run(a, cache) for a in plan[1]]


def get_sync(dsk, keys, pack_exception=None, **kwargs):
    """Built-in synchronous scheduler with ``dask.local.get_sync``'s calling convention
    (used when dask is not installed).  Iterative, so deep chains are fine.  Like dask's
    schedulers it drops an intermediate result as soon as its last consumer has run, which is
    what hands device tiles back to the pool while a large tiled graph is still running."""
    cache = {}
    wanted = list(keys) if isinstance(keys, list) else [keys]
    plans, pending = {}, {}

    # plans and reference counts over the part of the graph the wanted keys reach (every use counts)
    stack = list(wanted)
    while stack:
        key = stack.pop()
        if key in plans:
            continue
        needs = []
        plans[key] = (_plan_of(dsk[key], dsk, needs), needs)
        for d in needs:
            pending[d] = pending.get(d, 0) + 1
            if d not in plans:
                stack.append(d)
    for key in wanted:
        pending[key] = pending.get(key, 0) + 1  # never released

    def evaluate(target):
        stack = [target]
        while stack:
            key = stack[-1]
            if key in cache:
                stack.pop()
                continue
            plan, needs = plans[key]
            missing = [d for d in needs if d not in cache]
            if missing:
                stack.extend(missing)
                continue
            try:
                cache[key] = _run(plan, cache)
            except Exception as e:
                if pack_exception is not None:
                    pack_exception(e, None)  # reads ``key`` from this frame, then re-raises
                raise
            for d in needs:
                pending[d] -= 1
                if pending[d] == 0:
                    del cache[d]
            stack.pop()
        return cache[target]

    if isinstance(keys, list):
        return [evaluate(k) for k in keys]
    return evaluate(keys)


class _RunCache(dict):
    """The results of one run; keys it does not hold are the kept (static) results of the PreparedGraph."""

    __slots__ = ("kept",)

    def __missing__(self, key):
        return self.kept[key]


def _is_resident_answer(value):
    """A source's answer that is resident data as it stands (DeviceArray.resident): the same for the same task every time."""
    if type(value) is dict:
        values = value.get("values")
        return values is not None and getattr(values, "resident", False) is True
    return False


class PreparedGraph(object):
    """``get_sync(dsk, keys)`` for a graph that is evaluated again and again -- the fused job a worker of the one-worker-per-GPU
    scheduler keeps for a (view, request), a prebuilt batch -- with everything that does not depend on the run done once:

    * the look at the graph (plans, dependencies, evaluation order, when a result may be dropped) is kept;
    * a task WITHOUT dependencies that answered with resident data as it stands (a source's zero-copy window, a kept procedural
      window: ``DeviceArray.resident``) is not asked again: its answer is kept.  Nothing a kernel computed for the request is
      ever kept: every launch of the first run happens in every run;
    * a task all of whose dependencies are kept, and whose function offers ``gm_prepare(*arguments)``, is replaced by what that
      returns: the same launches with their arguments marshalled once (fuse.BatchedTask.gm_prepare).

    ``run()`` returns what ``get_sync(dsk, keys)`` returns.  Results are dropped as soon as their last consumer has run, as in
    get_sync.  Not thread-safe: one run at a time (a worker serves one message at a time)."""

    def __init__(self, dsk, keys):
        self.keys = keys
        self.wanted = list(keys) if isinstance(keys, list) else [keys]
        plans, uses = {}, {}
        stack = list(self.wanted)
        while stack:
            key = stack.pop()
            if key in plans:
                continue
            needs = []
            plans[key] = (_plan_of(dsk[key], dsk, needs), needs)
            for d in needs:
                uses[d] = uses.get(d, 0) + 1
                if d not in plans:
                    stack.append(d)
        # evaluation order: get_sync's depth-first walk from the wanted keys, producers first
        order, seen = [], set()
        for target in self.wanted:
            stack = [target]
            while stack:
                key = stack[-1]
                if key in seen:
                    stack.pop()
                    continue
                missing = [d for d in plans[key][1] if d not in seen]
                if missing:
                    stack.extend(missing)
                    continue
                seen.add(key)
                order.append(key)
                stack.pop()
        self.plans, self.uses, self.order = plans, uses, order
        self.kept = {}        # key -> kept answer
        self.program = None   # [(key, plan, releases)] of the tasks that run every time, built by the first run
        self.runs = 0

    def _compile(self, first_results):
        """After the first run: which tasks stay, with their arguments resolved as far as they are kept."""
        plans, kept = self.plans, self.kept
        for key in self.order:
            if not plans[key][1] and _is_resident_answer(first_results.get(key)):
                kept[key] = first_results[key]
        live = [k for k in self.order if k not in kept]
        # when a live result may go: after its last live consumer (wanted keys never)
        remaining = {}
        for key in live:
            for d in plans[key][1]:
                if d not in kept:
                    remaining[d] = remaining.get(d, 0) + 1
        for key in self.wanted:
            remaining[key] = remaining.get(key, 0) + (1 << 30)
        program = []
        for key in live:
            plan, needs = plans[key]
            releases = []
            for d in needs:
                if d in kept:
                    continue
                remaining[d] -= 1
                if remaining[d] == 0:
                    releases.append(d)
            if plan[0] == _FLAT:
                func, args, at = plan[1], list(plan[2]), []
                for i in plan[3]:
                    if args[i] in kept:
                        args[i] = kept[args[i]]
                    else:
                        at.append(i)
                prepare = getattr(func, "gm_prepare", None)
                if not at and prepare is not None:
                    prepared = prepare(*args)
                    if prepared is not None:
                        func, args = prepared, []
                plan = (_FLAT, func, tuple(args), tuple(at))
            program.append((key, plan, tuple(releases)))
        self.program = program

    def run(self):
        self.runs += 1
        if self.program is None:
            # the first run: get_sync's own walk; the answers of tasks without dependencies are remembered until the kept
            # ones are picked, everything else goes when its last consumer has run
            cache, pending = {}, dict(self.uses)
            for key in self.wanted:
                pending[key] = pending.get(key, 0) + 1
            for key in self.order:
                plan, needs = self.plans[key]
                cache[key] = _run(plan, cache)
                for d in needs:
                    pending[d] -= 1
                    if pending[d] == 0 and self.plans[d][1]:
                        del cache[d]
            self._compile(cache)
        else:
            cache = _RunCache()
            cache.kept = self.kept
            for key, plan, releases in self.program:
                cache[key] = _run(plan, cache)
                for d in releases:
                    del cache[d]
        if isinstance(self.keys, list):
            return [cache[k] for k in self.keys]
        return cache[self.keys]


def _configured_get():
    """The scheduler ``compute`` hands a graph to (graphs.py:42-49): dask's configured one; else ``scheduler_get``; else --
    when ``geomodeling.devices`` lists devices -- the one-worker-per-GPU scheduler (shard.get); else the built-in one."""
    get = _dask_get_scheduler() if _dask_get_scheduler is not None else None
    if get is None:
        get = _config.get("scheduler_get")
    if get is None and _config.get("geomodeling.devices"):
        from ..shard import get
    return get or get_sync


def compute(graph, name, *args, **kwargs):
    """Evaluate ``graph[name]`` with the configured scheduler (graphs.py:42-49)."""
    return _configured_get()(graph, [name])[0]


def get_data_many(blocks, requests):
    """Evaluate several (block, request) pairs as ONE graph -> list of results.

    The dask idiom ``dask.compute(*delayed)``: the compute graphs are merged (shared keys are
    computed once), fused as a whole -- so identical element-wise sub-graphs of different views
    or tiles run as one batched launch -- and evaluated with the configured scheduler."""
    if isinstance(requests, dict):
        requests = [requests] * len(blocks)
    graph, names = {}, []
    for block, request in zip(blocks, requests):
        graph, name = block.get_compute_graph(cached_compute_graph=graph, **request)
        names.append(name)
    get = _configured_get()
    if _config.get("geomodeling.fuse", True) and getattr(get, "__module__", "") != "dask_geomodeling_amd.shard":
        # (the sharded scheduler's workers fuse the part of the graph they are dealt)
        from ..fuse import fuse_graph
        graph = fuse_graph(graph, names)
    results = list(get(graph, names))
    if _config.get("geomodeling.materialize", True):
        from ..fuse import materialize
        results = [materialize(r) for r in results]
    return results


def _construct_exc_callback(e, dumps):
    """Prefix the failing graph key to the exception (graphs.py:21-27)."""
    key = inspect.currentframe().f_back.f_locals.get("key")
    e.args = ("{0}: {1}".format(key, str(e)),)
    raise e


def _reconstruct_token(key):
    """'SomeName_<32 hex>' -> token (graphs.py:30-39)."""
    if len(key) < 34 or key[-33] != "_":
        return None
    token = key[-32:]
    try:
        int(token, 16)
    except ValueError:
        return None
    return token.lower()


def construct(graph, name, validate=True):
    return construct_multiple(graph, [name], validate)[0]


def construct_multiple(graph, names, validate=True):
    """Build Blocks from ``{key: (class or import path, *args)}`` (graphs.py:58-82)."""
    new_graph = {}
    for key, value in graph.items():
        cls = value[0]
        if isinstance(cls, str):
            cls = Block.from_import_path(cls)
        if not (isinstance(cls, type) and issubclass(cls, Block)):
            raise TypeError("Cannot construct from object of type '{}'".format(cls))
        args = tuple(value[1:])
        if validate:
            new_graph[key] = (cls,) + args
        else:
            token = _reconstruct_token(key)
            if token is None:
                logger.warning(
                    "Construct received a key with an invalid name ('%s'),"
                    "while validation was turned off", key)
            new_graph[key] = (cls._init_no_validation, token) + args
    return get_sync(new_graph, names, pack_exception=_construct_exc_callback)


class Block(object):
    """Generates dask-like compute graphs for requests (see the reference's Block)."""

    JSON_VERSION = 2

    def __init__(self, *args):
        self.args = args

    @property
    def token(self):
        try:
            return self._cached_token
        except AttributeError:
            pass
        parts = [arg.token if isinstance(arg, Block) else arg for arg in self.args]
        self._cached_token = tokenize(self.get_import_path(), *parts)
        return self._cached_token

    @staticmethod
    def process(data):
        return data

    @classmethod
    def _init_no_validation(cls, token, *args):
        obj = cls.__new__(cls)
        obj.args = args
        if token:
            obj._cached_token = token
        return obj

    def get_sources_and_requests(self, **request):
        return ((source, request) for source in self.args)

    # -- never overridden by subclasses ------------------------------------------
    def get_data(self, **request):
        """Evaluate the request.  Element-wise sub-graphs are fused into single HIP
        launches first; the result is handed back as host ndarrays."""
        if _config.get("geomodeling.devices"):
            # one worker process per GPU (shard.py): a RasterTiler deals its tiles over them, the result is assembled here
            from .. import shard
            if shard.sharded_here():
                return shard.get_data(self, request)
        from ..plan import evaluate
        planned, result = evaluate(self, request)  # a cached plan of this view, if it has one (plan.py)
        if not planned:
            # a real graph for whatever scheduler is configured (core/graphs.py:42-49), already fused, made from the
            # view's template when it has one (template.py) instead of from scratch
            from ..template import fused_compute_graph
            graph, name = fused_compute_graph(self, request)
            result = compute(graph, name)
        if _config.get("geomodeling.materialize", True):
            from ..fuse import materialize
            result = materialize(result)
        return result

    def get_compute_graph(self, cached_compute_graph=None, **request):
        graph = cached_compute_graph or dict()
        return self._gm_compute_graph(graph, request, {})

    def _gm_compute_graph(self, graph, request, texts):
        """get_compute_graph (core/graphs.py:351-395 of the reference) without packing the request into keyword arguments at
        every Block, and with the text of a request made once per request OBJECT: a Block that hands its request on as it got
        it -- the element-wise family, most of a view -- shares it with its sources (``texts``: id -> (request, text); the
        entry keeps the object alive while its id is a key)."""
        known = texts.get(id(request))
        if known is None:
            known = texts[id(request)] = (request, _request_text(request))
        token = _request_token(self.token, request, known[1])
        name = "{}_{}".format(self.__class__.__name__.lower(), token)
        if name in graph:
            return graph, name
        args = [self.process]
        for source, req in self.get_sources_and_requests(**request):
            if isinstance(source, Block) and req is not None:
                if type(source).get_compute_graph is Block.get_compute_graph:
                    graph, compute_name = source._gm_compute_graph(graph, req, texts)
                else:  # (a Block with a get_compute_graph of its own)
                    graph, compute_name = source.get_compute_graph(cached_compute_graph=graph, **req)
                args.append(compute_name)
            else:
                args.append(source)
        graph[name] = tuple(args)
        return graph, name

    def get_graph(self, serialize=False):
        args = [self.get_import_path() if serialize else self.__class__]
        graph = dict()
        for arg in self.args:
            if isinstance(arg, Block):
                sub_graph, sub_name = arg.get_graph(serialize=serialize)
                graph.update(sub_graph)
                args.append(sub_name)
            else:
                args.append(arg)
        name = self.name
        graph[name] = args
        return graph, name

    @property
    def name(self):
        return "{}_{}".format(self.__class__.__name__, self.token)

    def __reduce__(self):
        return construct, self.get_graph() + (False,)

    @classmethod
    def get_import_path(cls):
        name, module = cls.__name__, cls.__module__
        try:
            __import__(module)
            klass = getattr(sys.modules[module], name)
        except (ImportError, KeyError, AttributeError):
            raise Exception("Can't serialize %r: it's not found as %s.%s" % (cls, module, name))
        if klass is not cls:
            raise Exception("Can't serialize %r: it's not the same object as %s.%s" % (cls, module, name))
        return "{}.{}".format(module, name)

    @staticmethod
    def from_import_path(path):
        module, name = path.rsplit(".", 1)
        __import__(module)
        klass = getattr(sys.modules[module], name)
        if isinstance(klass, type) and issubclass(klass, Block):
            return klass
        raise TypeError('"{}" is not valid Block.'.format(path))

    @classmethod
    def from_json(cls, val, **kwargs):
        return cls.deserialize(json.loads(val, **kwargs))

    def to_json(self, **kwargs):
        return json.dumps(self.serialize(), **kwargs)

    def serialize(self):
        graph, name = self.get_graph(serialize=True)
        return {"version": self.JSON_VERSION, "graph": graph, "name": name}

    @classmethod
    def deserialize(cls, val, validate=False):
        return construct(val["graph"], val["name"], validate=validate)

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, ", ".join(repr(x) for x in self.args))


class DummyBlock(Block):
    """Pretends to have the user-supplied name and token (graphs.py:293-309)."""

    def __init__(self, name):
        super().__init__(name)

    @property
    def token(self):
        return self.name.split("_")[1]

    @property
    def name(self):
        return self.args[0]
