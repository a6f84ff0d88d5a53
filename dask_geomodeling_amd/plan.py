"""Evaluation plans: ``get_data`` of a view without rebuilding its graph for every request.

The reference's ``Block.get_data`` (core/graphs.py:157-190) builds the compute graph of a request from
scratch -- one ``get_sources_and_requests`` + one token per Block -- and the fusion pass then collapses
it again.  For a view of element-wise Blocks over raster sources that is 120-170 microseconds of Python per
request against a 33 microsecond kernel per 4096 x 4096 tile; for a ``RasterTiler`` over 256 tiles it is
more than 100 ms against 12 ms of GPU work.

What makes a plan possible: the request planning of every Block the fusion pass fuses (the
``wrap_math_process_func`` family, Clip, Mask, Classify, Reclassify ...; elemwise.py:64-78, misc.py:75-96)
only clamps ``start`` / ``stop`` -- it never looks at the bounding box.  So when the fused graph of a
request is "leaf Blocks -> ONE fused program", the same program serves every other request that differs
in its ``bbox`` only, and all that has to be redone per request is the leaves' own planning (which window of
which source).  A ``ViewPlan`` keeps the fused task and the leaf Blocks of one (view, request-without-bbox);
a ``RasterTiler`` over such a view evaluates all its tiles through the store's plans and the in-place
mosaic assembly of fuse.py.  Blocks with a kernel of their own over one store (Dilate, MovingMax, HillShade, Smooth,
SpatialResample, TemporalAggregate, Cumulative, Group: ``gm_direct``) are leaves of the plans above them and evaluate their store
through the store's plan (``_direct``).  Everything else -- other modes, configured dask schedulers -- takes the
ordinary graph path, as does a request whose plan cannot be built.
"""

import collections
import os
import threading

import numpy as np

from . import config as _config
from .core.graphs import Block, _is_resident_answer, _request_token

try:  # resolved once: a failing import would search sys.path on every request
    from dask.base import get_scheduler as _dask_get_scheduler
except ImportError:
    _dask_get_scheduler = None

__all__ = ["evaluate", "ViewPlan", "clear"]

MAX_PLANS_PER_VIEW = 32
_NO_PLAN = object()
_lock = threading.RLock()


def _request_key(request):
    """Hashable identity of a request up to its bounding box, or None."""
    try:
        key = tuple(sorted((k, tuple(v) if isinstance(v, list) else v) for k, v in request.items() if k != "bbox"))
        hash(key)
        return key
    except TypeError:
        return None


def _leaf_blocks(block, request, out):
    """{graph key: (leaf Block, the request that reaches it)} of the view under ``block``, walking the
    Blocks exactly like ``Block.get_compute_graph`` does.  A leaf is a Block the fusion pass cannot make part of
    a program: a source, or a Block with a kernel of its own (stencils, temporal statistics ...) -- the view
    below such a Block is its own business (``_direct``)."""
    pairs = list(block.get_sources_and_requests(**request))
    children = [(s, r) for s, r in pairs if isinstance(s, Block) and r is not None]
    if not children or getattr(type(block).process, "gm_op", None) is None:
        name = "{}_{}".format(block.__class__.__name__.lower(), _request_token(block.token, request))
        out[name] = (block, request)
        return
    for source, sub in children:
        _leaf_blocks(source, sub, out)


class _PlainTask(object):
    """``(process, literal or graph key, ...)`` of one element-wise Block as a callable of its leaves' responses."""

    __slots__ = ("function", "template", "leaf_keys")

    def __init__(self, task, graph):
        self.function = task[0]
        self.template, self.leaf_keys = [], []
        for arg in task[1:]:
            is_key = isinstance(arg, str) and arg in graph
            self.template.append(len(self.leaf_keys) if is_key else (arg,))
            if is_key:
                self.leaf_keys.append(arg)

    def __call__(self, *values):
        return self.function(*[values[slot] if isinstance(slot, int) else slot[0] for slot in self.template])


MAX_KEPT_BOXES = 256


class ViewPlan(object):
    """One fused program over the leaf Blocks of a view, valid for every bbox of one request shape."""

    __slots__ = ("task", "leaves", "boxes")

    def __init__(self, task, leaves):
        self.task = task      # fuse.FusedTask: the whole view above its leaves
        self.leaves = leaves  # [(leaf Block, its request without a usable bbox)] in the order of task.leaf_keys
        # bounding box -> the launch of that box with its arguments marshalled (round 6): a box whose leaves all answered with
        # resident data as it stands (zero-copy windows: DeviceArray.resident) asks them, plans and marshals once; the next
        # request for the same box is one allocation and one launch call -- every launch still happens, nothing computed is
        # kept.  Least recently used first; a tile server's boxes repeat, a box never seen before costs what it cost.
        self.boxes = collections.OrderedDict()

    def leaf_values(self, bbox):
        values = []
        for block, request in self.leaves:
            request = dict(request, bbox=bbox)
            short_cut = getattr(block, "gm_plan_response", None)
            response = short_cut(request) if short_cut is not None else None
            if response is None:
                response = _direct(block, request)
            values.append(response)
        return values

    def run(self, bbox):
        boxes = self.boxes
        try:
            kept = boxes.get(bbox)
        except TypeError:  # (a bounding box that does not hash: a list)
            bbox = tuple(bbox)
            kept = boxes.get(bbox)
        if kept is not None:
            return kept()
        values = self.leaf_values(bbox)
        result = self.task(*values)
        prepare = getattr(self.task, "gm_prepare", None)
        if prepare is not None and all(_is_resident_answer(v) for v in values):
            kept = prepare(*values)
            if kept is not None:
                with _lock:
                    boxes[bbox] = kept
                    while len(boxes) > MAX_KEPT_BOXES:
                        boxes.popitem(last=False)
        return result


def _build_view_plan(view, request):
    from .fuse import FusedTask, fuse_graph
    graph, name = view.get_compute_graph(**request)
    fused = fuse_graph(graph, name)
    task = fused.get(name)
    if not (isinstance(task, tuple) and task and callable(task[0])):
        return None
    if isinstance(task[0], FusedTask):
        leaf_keys = list(task[1:])
        if leaf_keys != list(task[0].leaf_keys):
            return None
        run = task[0]
    elif getattr(task[0], "gm_op", None) is not None:
        # a single element-wise operation stays a plain task in the graph (nothing to fuse it with); here it becomes a
        # program of one step, so that its compiled form is kept like everyone else's
        from .fuse import _deps, _fusable
        if _fusable(task):
            leaf_keys = []
            for dep in _deps(task, fused):
                if dep not in leaf_keys:
                    leaf_keys.append(dep)
            run = FusedTask({name: task}, [name], name, leaf_keys)
        else:
            run = _PlainTask(task, fused)
            leaf_keys = run.leaf_keys
    else:
        return None
    # (every input of the one fused program must be a leaf Block of the walk below; what the graph holds
    #  underneath a leaf with a kernel of its own is evaluated by that leaf)
    found = {}
    _leaf_blocks(view, dict(request), found)
    leaves = []
    for key in leaf_keys:
        hit = found.get(key)
        if hit is None:
            return None
        block, sub = hit
        # the leaf must see the very box (and grid) that was asked of the view
        if any(sub.get(k) != request.get(k) for k in ("bbox", "width", "height", "projection", "mode")):
            return None
        leaves.append((block, dict(sub)))
    return ViewPlan(run, leaves)


def view_plan(view, request):
    """The plan of (view, request up to its bbox), built on first use; None when there is none."""
    key = _request_key(request)
    if key is None:
        return None
    plans = view.__dict__.get("_gm_plans")
    hit = plans.get(key) if plans is not None else None
    if hit is None:
        with _lock:  # get_data may run on several threads: one of them builds the plan
            plans = view.__dict__.setdefault("_gm_plans", {})
            hit = plans.get(key)
            if hit is None:
                try:
                    hit = _build_view_plan(view, request) or _NO_PLAN
                except NotImplementedError:
                    hit = _NO_PLAN
                if len(plans) >= MAX_PLANS_PER_VIEW:
                    plans.clear()
                plans[key] = hit
    return None if hit is _NO_PLAN else hit


_STENCILS_WHOLE = True  # (tests of the tile-by-tile route of stencil Blocks under a tiler switch it off)


def _local_everywhere(block, request, stencils=1):
    """Every Block of the view computes a cell from that cell of its sources alone (``gm_local``), and every source answers
    ``request`` with a window of its own grid.  ``stencils``: how many Blocks that look around a cell instead
    (``gm_tiling_invariant``: Dilate, MovingMax, HillShade -- they ask their store for the box grown by their margin,
    so a tile's cells see exactly what the whole box's cells see) a path to the leaves may still hold: one, because their
    process arguments follow from the cell size of the request they are asked (the whole and its tiles agree on it,
    _whole_instead_of_tiles looks), not from that of a request another stencil has grown."""
    kind = type(block)
    if getattr(kind, "gm_local", False):
        aligned = getattr(block, "_aligned_window", None)
        if aligned is not None and aligned(request) is None:
            return False
        return all(_local_everywhere(a, request, stencils) for a in block.args if isinstance(a, Block))
    if getattr(kind, "gm_tiling_invariant", False) and stencils > 0 and _STENCILS_WHOLE and request.get("mode") == "vals":
        found = False
        for source, asked in block.get_sources_and_requests(**request):
            if isinstance(source, Block):
                if asked is None or not _local_everywhere(source, asked, stencils - 1):
                    return False
                found = True
        return found
    return False


# tiles of 2048^2 cells and more: the batched launch over the tiles (the test suites, whose rasters are small, set 0: their
# tiles go the way the bench's 4096^2 tiles go)
_BATCHED_TILE_CELLS = int(os.environ.get("GM_TILER_BATCH_CELLS", 1 << 22))


def _whole_instead_of_tiles(store, request, kwargs, tile_requests):
    """A mosaic of a pixel-local view over sources on the request's grid IS that view asked for the whole box: the reference
    computes it tile by tile to bound the memory of a task (parallelize.py:43-91) and pastes the tiles next to each other
    (:108-124), which changes no cell.  For a store with a kernel of its own -- a TemporalAggregate, a Cumulative: no program
    to batch over the tiles -- one request then replaces 16 x ~70 us of planning around 16 small launches (8 frames of 4096^2
    in 1024^2 tiles: 1.0 -> ~0.15 ms).  Only where nothing can differ: the tiles cover the box exactly (the reference's int()
    truncation of tile sizes can leave cells to the fill value), none of them is a single cell (NumPy sums the series of ONE
    cell pairwise), the box is of moderate size."""
    height, width = kwargs["shape_yx"]
    if height * width > (1 << 28) or len(tile_requests) < 2:
        return False
    # the tiler exists to bound the memory of a task (parallelize.py:43-91): the whole box at once must stay moderate in BYTES --
    # the frames a temporal store would hold in its window and the width of its cells count, not just the cells of one frame
    # (a series of N frames answered whole allocates N times the intermediates the tiles would, one after the other)
    try:
        frames = 1
        start, stop, delta = request.get("start"), request.get("stop"), getattr(store, "timedelta", None)
        if delta and start is not None and stop is not None:
            frames = max(1, int((stop - start) / delta) + 1)
        if height * width * frames * max(np.dtype(kwargs["dtype"]).itemsize, 4) > (4 << 30):
            return False
    except (TypeError, ValueError, OverflowError):
        return False
    if getattr(store, "gm_direct", False) and hasattr(type(store), "gm_batch") and not _batches_tiles(store) and \
            max(tile["width"] * tile["height"] for _, tile in tile_requests) >= _BATCHED_TILE_CELLS:
        # a stencil that takes tiles in one launch, over a COMPUTED store, in LARGE tiles: tile after tile is the faster order
        # (every intermediate tile is read back while the Infinity Cache still holds it: 1.04 against 1.64 ms for a 16-tile
        # HillShade(A * 2 + B) in 4096^2 tiles); it is the planning around small tiles that one request for the whole saves
        return False
    if getattr(type(store).process, "gm_op", None) is not None and \
            max(tile["width"] * tile["height"] for _, tile in tile_requests) >= _BATCHED_TILE_CELLS:
        # element-wise stores are batched over the tiles and written into the mosaic (TiledAssembly): one launch, at the kernel's
        # rate for tiles of this size; it is the planning around SMALL tiles (8 frames of 4096^2 in 16 tiles: 337 us against 164
        # for the whole) that one request saves
        return False
    covered = 0
    x1, y1, x2, y2 = request["bbox"]
    if x2 == x1 or y2 == y1:
        return False
    per_x, per_y, cell_x, cell_y = width / (x2 - x1), height / (y2 - y1), (x2 - x1) / width, (y2 - y1) / height
    for _, tile in tile_requests:
        if tile["width"] * tile["height"] < 2:
            return False
        covered += tile["width"] * tile["height"]
        # the tiles' cells are the box's cells to the last bit (what a HillShade's resolution is
        # computed from): cell sizes that are no binary fractions can differ in it from tile to tile
        tx1, ty1, tx2, ty2 = tile["bbox"]
        if tx2 == tx1 or ty2 == ty1 or tile["width"] / (tx2 - tx1) != per_x or tile["height"] / (ty2 - ty1) != per_y \
                or (tx2 - tx1) / tile["width"] != cell_x or (ty2 - ty1) / tile["height"] != cell_y:
            return False
    if covered != height * width:
        return False
    return _local_everywhere(store, request)


def _as_mosaic(response, kwargs):
    """The store's answer for the whole box as RasterTiler.process hands its mosaic on (parallelize.py:108-125): the
    declared dtype (values[...] = tile casts on assignment) and fill value."""
    if response is None:
        return None
    values = response["values"]
    height, width = kwargs["shape_yx"]
    if len(values.shape) != 3 or tuple(values.shape[1:]) != (height, width):
        return False
    dtype = np.dtype(kwargs["dtype"])
    if np.dtype(values.dtype) != dtype:
        from .raster.stencil import _cast_tile
        from .hip.array import as_device
        values = _cast_tile(as_device(values), dtype)
    return {"values": values, "no_data_value": kwargs["fillvalue"]}


def _run_tiler(tiler, request):
    """RasterTiler.get_data through the plans of its store: (True, response) or (False, None)."""
    from .fuse import BatchedTask, TiledAssembly
    pairs = list(tiler.get_sources_and_requests(**request))
    kwargs = pairs[0][0]
    if kwargs is None or len(pairs) < 2:
        return False, None  # non-tiled / point requests
    store = tiler.store
    if _whole_instead_of_tiles(store, request, kwargs, pairs[1:]):
        planned, response = _evaluate(store, dict(request))
        if planned and (response is None or (isinstance(response, dict) and "values" in response)):
            mosaic = _as_mosaic(response, kwargs)
            if mosaic is not False:
                return True, mosaic
    batches, members, layout = [], [], []   # per distinct plan: BatchedTask material
    index_of = {}
    from .fuse import FusedTask
    for _, tile_request in pairs[1:]:
        plan = view_plan(store, tile_request)
        if plan is None or not isinstance(plan.task, FusedTask):
            # no program to batch (a store with a kernel of its own, a single unfusable operation): tile by tile
            # through whatever plan the store has, assembled by the Block's own process -- unless the Block takes
            # the tiles of a mosaic in one launch and writes them in place (gm_batch: HillShade)
            # (only straight over a resident source, whose tiles are zero-copy windows: over a computed store the
            #  one-by-one order is the better one -- every intermediate tile is read back while the Infinity Cache still
            #  holds it, 1.04 ms against 1.64 for a 16-tile HillShade(A * 2 + B) mosaic)
            if _batches_tiles(store):
                done = _run_tiler_batch(tiler, store, kwargs, pairs[1:])
                if done is not None:
                    return True, done
            elif getattr(store, "gm_direct", False) and hasattr(type(store), "gm_batch"):
                # over a computed store: tile after tile (the order that keeps the intermediates in the cache), each
                # result still written straight into the mosaic
                from .fuse import tiles_in_place
                outer = getattr(_running, "memo", None)

                def arguments():
                    for _, tile_request in pairs[1:]:
                        _running.memo = {}
                        yield _process_arguments(store, tile_request)
                try:
                    return True, tiles_in_place(type(tiler).process, type(store), kwargs, arguments(), one_by_one=True)
                finally:
                    _running.memo = outer
            responses = []
            outer, outer_boxless = getattr(_running, "memo", None), getattr(_running, "boxless", None)
            _running.boxless = {}
            try:
                for _, request in pairs[1:]:
                    _running.memo = {}  # a memo per tile: the tiles share nothing, and theirs must not pile up
                    planned, response = _evaluate(store, request)
                    if not planned:
                        return False, None
                    responses.append(response)
            finally:
                _running.memo, _running.boxless = outer, outer_boxless
            return True, type(tiler).process(kwargs, *responses)
        b = index_of.get(id(plan))
        if b is None:
            b = index_of[id(plan)] = len(batches)
            batches.append(plan)
            members.append([])
        layout.append((b, len(members[b])))
        members[b].append(tile_request["bbox"])
    leaf_values, tasks = [], []
    outer = getattr(_running, "memo", None)
    try:
        for plan, boxes in zip(batches, members):
            leaf_values.extend(_leaf_values_of_tiles(plan, boxes))
            tasks.append(BatchedTask([plan.task] * len(boxes)))
    finally:
        _running.memo = outer
    assembly = TiledAssembly(type(tiler).process, layout, tasks)
    return True, assembly(kwargs, *leaf_values)


def _batches_tiles(block):
    """A Block that takes the tiles of a mosaic in one launch (gm_batch), straight over a resident source."""
    return (getattr(block, "gm_direct", False) and hasattr(type(block), "gm_batch") and
            hasattr(getattr(block, "store", None), "gm_plan_response"))


def _process_arguments(block, request):
    """What ``_direct`` hands to ``block.process`` for one request."""
    return tuple(_child(source, sub) if isinstance(source, Block) and sub is not None else source
                 for source, sub in block.get_sources_and_requests(**request))


def _leaf_values_of_tiles(plan, boxes):
    """ViewPlan.leaf_values for all tiles of a mosaic -> [leaf values of tile 0 ..., of tile 1 ...]: leaves that take the
    tiles of a mosaic in one launch (a HillShade under an element-wise view under a RasterTiler) are evaluated once for
    all tiles, the others tile by tile."""
    columns = []
    for block, request in plan.leaves:
        requests = [dict(request, bbox=bbox) for bbox in boxes]
        column = None
        if len(boxes) > 1 and _batches_tiles(block):
            calls = []
            for tile_request in requests:
                _running.memo = {}  # (a memo per tile: what the leaves of one tile evaluate underneath must not pile up)
                calls.append(_process_arguments(block, tile_request))
            column = type(block).gm_batch(calls, None)
        if column is None:
            short_cut = getattr(block, "gm_plan_response", None)
            column = []
            for tile_request in requests:
                _running.memo = {}
                response = short_cut(tile_request) if short_cut is not None else None
                column.append(response if response is not None else _direct(block, tile_request))
        columns.append(column)
    return [column[i] for i in range(len(boxes)) for column in columns]


def _run_tiler_batch(tiler, store, kwargs, tile_pairs):
    """The tiles of a RasterTiler over a Block with ``gm_batch``: the store's sources are evaluated tile by tile (their
    plans, zero-copy windows), the Block's kernel runs ONCE over all tiles and writes into the mosaic.  The mosaic is the
    result only when the tiles cover it (fuse.TiledAssembly's rule); otherwise, or when the Block declines, None / the
    reference's fill + paste."""
    calls = []
    outer = getattr(_running, "memo", None)
    try:
        for _, tile_request in tile_pairs:
            _running.memo = {}  # (a memo per tile)
            calls.append(_process_arguments(store, tile_request))
    finally:
        _running.memo = outer
    from .fuse import tiles_in_place
    return tiles_in_place(type(tiler).process, type(store), kwargs, calls)


def _direct(block, request):
    """``block.process`` on its sources' responses without a graph: what ``get_compute_graph`` + ``compute`` do for
    a Block whose sources are evaluated one by one (no key to share between them)."""
    args = []
    for source, sub in block.get_sources_and_requests(**request):
        args.append(_child(source, sub) if isinstance(source, Block) and sub is not None else source)
    return block.process(*args)


def _direct_kept(block, request):
    """``_direct`` for the view a request is asked of, with the ARGUMENTS of its ``process`` kept per request (round 6) when every
    answer of its sources is resident data as it stands (zero-copy windows: ``DeviceArray.resident``) or carries no raster at
    all (time stamps, metadata): asking immutable Blocks the same question again gives the same windows.  ``process`` itself --
    every launch -- runs each time.  Group(A, B) over two resident sources: 49 -> 33 us per request, kernel-bound."""
    try:
        key = tuple(sorted((k, tuple(v) if isinstance(v, list) else v) for k, v in request.items()))
        kept = block.__dict__.get("_gm_direct_args")
        args = kept.get(key) if kept is not None else None
    except TypeError:
        return _direct(block, request)
    if args is not None:
        return block.process(*args)
    args = []
    keep = True
    for source, sub in block.get_sources_and_requests(**request):
        if isinstance(source, Block) and sub is not None:
            answer = _child(source, sub)
            keep = keep and (_is_resident_answer(answer) or (isinstance(answer, dict) and "values" not in answer) or answer is None)
            args.append(answer)
        else:
            args.append(source)
    if keep:
        with _lock:
            kept = block.__dict__.setdefault("_gm_direct_args", collections.OrderedDict())
            kept[key] = args
            while len(kept) > MAX_KEPT_BOXES:
                kept.popitem(last=False)
    return block.process(*args)


_running = threading.local()  # .memo: {(id(block), request): response} of the evaluation in progress on this thread


def _child(block, request):
    """The response of a source Block for one request, not materialised: through its plan when it has one.  Within
    one evaluation the same (Block, request) is answered once -- what the shared keys of a graph achieve."""
    short_cut = getattr(block, "gm_plan_response", None)  # (a resident source answers an aligned window at once)
    if short_cut is not None and request.get("mode") == "vals":
        response = short_cut(request)
        if response is not None:
            return response
    memo = getattr(_running, "memo", None)
    if memo is not None and "bbox" not in request:
        # a question without a box (the time stamps / metadata of a window in time) has one answer for all tiles of a
        # mosaic: kept for the whole tiler request, not per tile
        shared = getattr(_running, "boxless", None)
        if shared is not None:
            memo = shared
    key = None
    if memo is not None:
        try:
            key = (id(block), tuple(sorted((k, tuple(v) if isinstance(v, list) else v) for k, v in request.items())))
            if key in memo:
                return memo[key]
        except TypeError:
            key = None
    if request.get("mode") in ("time", "meta"):
        # time stamps and metadata: no tiles, nothing to fuse -- the Blocks' own process() over their sources' answers,
        # without a graph, a token and a scheduler pass per question (a Group asks every source for its time stamps)
        response = _direct(block, request)
    elif not any(isinstance(a, Block) for a in block.args):
        # a source (nothing below it to plan or fuse): its own process(), without a graph per request -- the frames of a
        # series under a tiled temporal view are asked for tile by tile
        response = _direct(block, request)
    else:
        planned, response = evaluate(block, request)
        if not planned:
            from .core.graphs import compute
            from .template import fused_compute_graph
            graph, name = fused_compute_graph(block, request)
            response = compute(graph, name)
    if key is not None:
        memo[key] = response
    return response


def enabled():
    if not _config.get("geomodeling.fuse", True) or not _config.get("geomodeling.plans", True):
        return False
    if _config.get("scheduler_get") is not None:
        return False
    if _dask_get_scheduler is not None and _dask_get_scheduler() is not None:
        return False  # a scheduler configured in dask owns the evaluation
    return True


def evaluate(view, request):
    """(True, response) when the request was answered through a plan, else (False, None)."""
    if request.get("mode") != "vals" or "bbox" not in request or not enabled():
        return False, None
    if getattr(_running, "memo", None) is None:  # the outermost evaluation on this thread owns the memo
        _running.memo = {}
        try:
            return _evaluate(view, request)
        finally:
            _running.memo = None
    return _evaluate(view, request)


def _evaluate(view, request):
    if getattr(type(view).process, "gm_tiler", False) and hasattr(view, "store"):
        return _run_tiler(view, request)
    plan = view_plan(view, request)
    if plan is not None:
        return True, plan.run(request["bbox"])
    if getattr(view, "gm_direct", False):
        # a Block with a kernel of its own over ONE store (stencils, Smooth, resample): its store is evaluated
        # through the store's plan, the Block's own request planning is all that is redone per request
        return True, _direct_kept(view, request)
    return False, None


def clear(view):
    view.__dict__.pop("_gm_plans", None)
    view.__dict__.pop("_gm_direct_args", None)
