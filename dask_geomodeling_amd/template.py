"""Graph templates: the fused compute graph of a request without building and fusing it from scratch.

The reference's ``Block.get_data`` builds the graph of every request anew -- one ``get_sources_and_requests`` and
one token per Block (core/graphs.py:157-190) -- and hands it to whatever scheduler dask is configured with
(core/graphs.py:42-49).  Here the fusion pass then collapses that graph again: 250-270 microseconds of Python per
single-tile request of a five-Block view against a 33 microsecond kernel.  ``plan.py`` avoids the graph altogether,
but steps aside as soon as a scheduler is configured -- the deployment the Block API exists for.  This module keeps
the graph and makes it cheap: ``fused_compute_graph(block, request)`` returns a real dask-style graph, already
fused, that any ``get(graph, keys)`` evaluates -- built from a TEMPLATE that is kept per (view, request up to its
bounding box):

* an element-wise view (the ``wrap_math_process_func`` family, Clip, Mask, Classify, Reclassify ...): its request
  planning only clamps ``start`` / ``stop`` (elemwise.py:64-78, misc.py:75-96), it never looks at the bounding
  box.  The fused graph of such a view is "leaf Blocks -> ONE FusedTask", and the same FusedTask -- with its compiled
  programs -- serves every request that differs in its ``bbox`` only.  The template holds the FusedTask and the
  leaf Blocks with the requests that reach them; per request only the leaves' own tasks (``process`` + their
  ``get_sources_and_requests``) and the graph keys are made anew;
* a ``RasterTiler`` (parallelize.py:43-125): the tile boxes are the Block's own planning, per tile the store's
  template gives the FusedTask; the tiles of one template become one ``BatchedTask`` and the whole a
  ``TiledAssembly`` (fuse.py), both kept per tile layout;
* every other Block is one task over the fused graphs of its sources (what ``get_compute_graph`` does, one level).

Keys keep the reference's form ``<classname lowercase>_<32 hex>`` (core/graphs.py:173) and are the very keys
``get_compute_graph`` gives, so graphs of several requests can be merged and shared sub-graphs are computed once.
A request whose template cannot be built (fused graph of another shape, other modes) takes the full
``get_compute_graph`` + ``fuse_graph`` path; ``geomodeling.templates = False`` switches the templates off.
"""
import threading

from . import config as _config
from .core.graphs import Block, _request_token

__all__ = ["fused_compute_graph", "stats", "clear"]

MAX_TEMPLATES_PER_VIEW = 32
_NONE = object()

# what the real-dask driver and the tests read: templates built, requests served from one, requests that took
# the full path
stats = {"built": 0, "hits": 0, "full": 0}
# get_data runs on the threads of the configured scheduler: templates are built and put into their view's cache, and
# the counters are updated, under this lock (the look-up of an existing template takes no lock)
_lock = threading.RLock()


def _count(name, n=1):
    with _lock:
        stats[name] += n

_SAME = ("bbox", "width", "height", "projection", "mode")


def enabled():
    return bool(_config.get("geomodeling.fuse", True)) and bool(_config.get("geomodeling.templates", True))


def _key_of(block, request):
    return "{}_{}".format(block.__class__.__name__.lower(), _request_token(block.token, request))


_PROBE = (-1.25e300, -2.5e300, 1.25e300, 2.5e300)  # a bounding box no request asks for


class _KeyMaker(object):
    """Graph keys of one (Block, request up to its bbox) without sorting and printing the request every time:
    ``repr(sorted(request.items()))`` -- what core.graphs._request_token hashes -- is the same text for every bounding
    box but for the printed box itself, so it is printed once and cut there.  Checked against ``_key_of`` when made;
    a request that does not print that way (no plain bbox tuple) keeps ``_key_of``."""

    __slots__ = ("block", "head", "tail", "prefix")

    def __init__(self, block, request):
        import hashlib
        self.block = block
        self.head = None
        probe = dict(request)
        probe["bbox"] = _PROBE
        text = repr(sorted(probe.items()))
        cut = text.find(repr(_PROBE))
        if cut < 0 or not isinstance(request.get("bbox"), tuple):
            return
        self.prefix = block.__class__.__name__.lower() + "_"
        head = (block.token + text[:cut]).encode()
        self.tail = text[cut + len(repr(_PROBE)):].encode()
        self.head = hashlib.md5(head)  # (copied per key: the common prefix is hashed once)
        if self(request) != _key_of(block, request):
            self.head = None

    def __call__(self, request):
        if self.head is None or type(request["bbox"]) is not tuple:
            return _key_of(self.block, request)
        h = self.head.copy()
        h.update(repr(request["bbox"]).encode())
        h.update(self.tail)
        return self.prefix + h.hexdigest()


def _request_key(request):
    """Hashable identity of a request up to its bounding box, or None.  The type of every value is part of it:
    256, 256.0 and True are equal and hash alike but print differently, and the template's _KeyMakers hold the
    printed request."""
    try:
        key = tuple(sorted((k, type(v), tuple(v) if isinstance(v, list) else v)
                           for k, v in request.items() if k != "bbox"))
        hash(key)
        return key
    except TypeError:
        return None


def _collect(block, request, out):
    """{graph key: (Block, the request that reaches it)} of the whole view, walked as get_compute_graph walks it."""
    name = _key_of(block, request)
    if name in out:
        return
    out[name] = (block, request)
    for source, sub in block.get_sources_and_requests(**request):
        if isinstance(source, Block) and sub is not None:
            _collect(source, sub, out)


class _ViewTemplate(object):
    """``root = run(*leaves)`` of an element-wise view: valid for every bbox of one request shape."""

    __slots__ = ("run", "pattern", "leaves", "root_key", "leaf_keys")

    def __init__(self, view, request, run, pattern, leaves):
        self.run = run          # fuse.FusedTask (or the process function of a single element-wise Block)
        self.pattern = pattern  # call arguments: int = index into leaves, 1-tuple = literal
        self.leaves = leaves    # [(leaf Block, its request without the bbox)]
        self.root_key = _KeyMaker(view, request)
        self.leaf_keys = [_KeyMaker(block, dict(sub, bbox=request["bbox"])) for block, sub in leaves]

    def instantiate(self, view, request, graph):
        bbox = request["bbox"]
        names = []
        for (block, sub), key in zip(self.leaves, self.leaf_keys):
            sub = dict(sub)
            sub["bbox"] = bbox
            names.append(_graph_of(block, sub, graph, key))
        root = self.root_key(request)
        graph[root] = (self.run,) + tuple(names[slot] if isinstance(slot, int) else slot[0] for slot in self.pattern)
        return root


def _build_view_template(view, request):
    from .fuse import FusedTask, fuse_graph
    graph, name = view.get_compute_graph(**request)
    fused = fuse_graph(graph, name)
    task = fused.get(name)
    if not (isinstance(task, tuple) and task and callable(task[0])):
        return None
    if not isinstance(task[0], FusedTask) and getattr(task[0], "gm_op", None) is not None:
        # a single element-wise operation stays a plain task in the graph (nothing to fuse it with); as a program of one step
        # it can be batched over the tiles of a RasterTiler and written into the mosaic like any other (plan.py does the
        # same): a tiled `T * 2` over 8 frames as a graph per request 670 -> ~250 us
        from .fuse import _deps, _fusable
        if _fusable(task):
            inputs = []
            for dep in _deps(task, fused):
                if dep not in inputs:
                    inputs.append(dep)
            task = (FusedTask({name: task}, [name], name, inputs),) + tuple(inputs)
    leaf_keys, pattern = [], []
    if isinstance(task[0], FusedTask):
        if list(task[1:]) != list(task[0].leaf_keys):
            return None
    elif getattr(task[0], "gm_op", None) is None:
        return None
    for arg in task[1:]:
        is_key = isinstance(arg, str) and arg in fused
        if is_key:
            if arg not in leaf_keys:
                leaf_keys.append(arg)
            pattern.append(leaf_keys.index(arg))
        else:
            pattern.append((arg,))
    blocks = {}
    _collect(view, dict(request), blocks)
    leaves = []
    for key in leaf_keys:
        hit = blocks.get(key)
        if hit is None:
            return None
        block, sub = hit
        # the leaf must be asked for the very box and grid that was asked of the view
        if any(sub.get(k) != request.get(k) for k in _SAME):
            return None
        leaves.append((block, {k: v for k, v in sub.items() if k != "bbox"}))
    return _ViewTemplate(view, request, task[0], pattern, leaves)


def _view_template(view, request):
    """The template of (view, request up to its bbox), built on first use; None when there is none."""
    key = _request_key(request)
    if key is None:
        return None
    cache = view.__dict__.get("_gm_templates")
    hit = cache.get(key) if cache is not None else None
    if hit is None:
        with _lock:
            cache = view.__dict__.setdefault("_gm_templates", {})
            hit = cache.get(key)  # another thread may have built it meanwhile
            if hit is None:
                try:
                    hit = _build_view_template(view, request) or _NONE
                except NotImplementedError:
                    hit = _NONE
                if len(cache) >= MAX_TEMPLATES_PER_VIEW:
                    cache.clear()
                cache[key] = hit
                if hit is not _NONE:
                    stats["built"] += 1
    return None if hit is _NONE else hit


def _one_level(block, request, graph, key=None):
    """``block.process`` over the fused graphs of its sources: Block.get_compute_graph, one level of it."""
    name = key(request) if key is not None else _key_of(block, request)
    if name in graph:
        return name
    args = [block.process]
    for source, sub in block.get_sources_and_requests(**request):
        if isinstance(source, Block) and sub is not None:
            args.append(_graph_of(source, sub, graph))
        else:
            args.append(source)
    graph[name] = tuple(args)
    return name


def mosaic_of_whole(process_kwargs, response):
    """Graph task: the store's answer for the whole box as RasterTiler.process hands its mosaic on (plan._as_mosaic)."""
    from .plan import _as_mosaic
    mosaic = _as_mosaic(response, process_kwargs)
    if mosaic is False:
        raise RuntimeError("the store answered the whole box with %r" % (getattr(response.get("values"), "shape", None),))
    return mosaic


def _tiler(tiler, request, graph):
    """RasterTiler: the tiles through the templates of its store, batched and assembled in place."""
    from .fuse import BatchedTask, FusedTask, TiledAssembly
    pairs = list(tiler.get_sources_and_requests(**request))
    kwargs = pairs[0][0]
    if kwargs is None or len(pairs) < 2:
        return _one_level(tiler, request, graph)  # non-tiled / point requests
    store = tiler.store
    from .plan import _whole_instead_of_tiles
    if _whole_instead_of_tiles(store, request, kwargs, pairs[1:]):
        # a pixel-local view whose tiles cover the box exactly: the store's graph for the whole box, handed on as the mosaic
        # (plan.py: the same cells as the tiles pasted next to each other)
        name = _key_of(tiler, request)
        graph[name] = (mosaic_of_whole, kwargs, _graph_of(store, dict(request), graph))
        return name
    templates = []
    if getattr(type(store).process, "gm_op", None) is not None:
        for _, tile_request in pairs[1:]:
            tpl = _view_template(store, tile_request)
            if tpl is None or not isinstance(tpl.run, FusedTask):
                templates = []
                break
            templates.append(tpl)
    name = _key_of(tiler, request)
    if not templates:
        from .plan import _batches_tiles
        if _batches_tiles(store):
            # a stencil straight over a resident source: ONE task shades / filters all tiles into the mosaic (gm_batch);
            # the scheduler computes the tiles' source windows
            from .fuse import MosaicOfTiles
            counts, args = [], []
            for _, tile_request in pairs[1:]:
                per_tile = list(store.get_sources_and_requests(**tile_request))
                counts.append(len(per_tile))
                for source, sub in per_tile:
                    args.append(_graph_of(source, sub, graph) if isinstance(source, Block) and sub is not None else source)
            graph[name] = (MosaicOfTiles(type(tiler).process, type(store)), kwargs, tuple(counts)) + tuple(args)
            return name
        # a store with a kernel of its own, a single operation: tile by tile, pasted by the Block's own process
        names = [_graph_of(store, tile_request, graph) for _, tile_request in pairs[1:]]
        graph[name] = (type(tiler).process, kwargs) + tuple(names)
        return name
    _count("hits", len(templates))
    index_of, order, members, layout = {}, [], [], []
    for tpl, (_, tile_request) in zip(templates, pairs[1:]):
        b = index_of.get(id(tpl))
        if b is None:
            b = index_of[id(tpl)] = len(order)
            order.append(tpl)
            members.append([])
        layout.append((b, len(members[b])))
        members[b].append(tile_request["bbox"])
    signature = (tuple(id(t) for t in order), tuple(layout))
    cache = tiler.__dict__.get("_gm_assemblies")
    assembly = cache.get(signature) if cache is not None else None
    if assembly is None:
        with _lock:
            cache = tiler.__dict__.setdefault("_gm_assemblies", {})
            assembly = cache.get(signature)
            if assembly is None:
                if len(cache) >= MAX_TEMPLATES_PER_VIEW:
                    cache.clear()
                tasks = [BatchedTask([tpl.run] * len(boxes)) for tpl, boxes in zip(order, members)]
                # (the FusedTasks -- hence the templates' programs -- are kept alive by the assembly)
                assembly = cache[signature] = TiledAssembly(type(tiler).process, layout, tasks)
    leaf_names = []
    for tpl, boxes in zip(order, members):
        for bbox in boxes:
            for (block, sub), key in zip(tpl.leaves, tpl.leaf_keys):
                sub = dict(sub)
                sub["bbox"] = bbox
                leaf_names.append(_graph_of(block, sub, graph, key))
    graph[name] = (assembly, kwargs) + tuple(leaf_names)
    return name


def _graph_of(block, request, graph, key=None):
    """Add the fused graph of (block, request) to ``graph``; -> its key.  ``key``: a _KeyMaker of (block, request up
    to its bbox) when the caller keeps one."""
    process = type(block).process
    if request.get("mode") in ("time", "meta"):
        # time stamps and metadata: no tiles, nothing to fuse -- one task per Block over its sources' graphs, which is
        # what get_compute_graph builds (a Group asks every source for its time stamps with every request)
        return _one_level(block, request, graph, key)
    if request.get("mode") == "vals" and "bbox" in request:
        if getattr(process, "gm_tiler", False) and hasattr(block, "store"):
            return _tiler(block, request, graph)
        if getattr(process, "gm_op", None) is None:
            return _one_level(block, request, graph, key)
        tpl = _view_template(block, request)
        if tpl is not None:
            _count("hits")
            return tpl.instantiate(block, request, graph)
    # everything else: the whole graph, fused as a whole
    from .fuse import fuse_graph
    _count("full")
    full, name = block.get_compute_graph(**request)
    graph.update(fuse_graph(full, name))
    return name


def fused_compute_graph(block, request, graph=None):
    """(graph, name): the compute graph of ``block.get_data(**request)`` with element-wise sub-graphs fused into
    single launches -- ``fuse_graph(*block.get_compute_graph(**request))`` in effect, made from the view's templates."""
    graph = {} if graph is None else graph
    if not enabled():
        from .fuse import fuse_graph
        full, name = block.get_compute_graph(**request)
        if _config.get("geomodeling.fuse", True):
            full = fuse_graph(full, name)
        graph.update(full)
        return graph, name
    return graph, _graph_of(block, request, graph)


def clear(view):
    view.__dict__.pop("_gm_templates", None)
    view.__dict__.pop("_gm_assemblies", None)
