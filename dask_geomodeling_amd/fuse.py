"""Graph-level fusion: collapse sub-graphs of element-wise Blocks into one HIP launch.

The reference materialises one NumPy array per Block per tile (one dask task per Block,
core/graphs.py:179-189).  This pass rewrites the dask-style graph produced by
``Block.get_compute_graph``: every maximal set of tasks whose ``process`` function is one
of the element-wise bodies (recognised by their ``gm_op`` attribute: the
wrap_math_process_func family, Invert/IsData/IsNoData, Clip, Mask) and whose intermediate
results are consumed only inside the set becomes a single ``FusedTask``.  At run time the
FusedTask turns the member tasks into one expression and evaluates it as one program
(hip/program.py); each step keeps the reference's per-op no-data / non-finite rule, so the
result is bit-identical to the op-by-op path.

**Horizontal batching.**  A ``RasterTiler`` (raster/parallelize.py:82-89) fans a view out into
one identical sub-graph per tile.  After the vertical fusion above, fused tasks that have the
same structure (same ops, same constants, different leaves) and do not depend on one another
are merged into one ``BatchedTask``: all its tiles run the same compiled program in one launch
per 16 tiles (``gm_elemwise_run_batch``), which removes the kernel boundaries between tiles.
The per-tile keys stay in the graph as ``(getitem, batch_key, i)``.

``None`` / ``time`` / ``meta`` responses are not numerics: when any leaf of a fused group is
not a tile of values the member tasks are simply run one by one with their own ``process``
bodies, which carry the reference's control flow for those cases.
"""
import math

import numpy as np

from . import utils
from .hip import program as P
from .hip.array import DeviceArray
from .raster import devops, tables

__all__ = ["fuse_graph", "materialize", "FusedTask", "BatchedTask", "TiledAssembly"]

MAX_BATCH_TILES = 64  # tiles per BatchedTask: bounds how many tile results are alive at once


def _is_task(x):
    return isinstance(x, tuple) and len(x) > 0 and callable(x[0])


def _is_key(x, graph):
    if isinstance(x, (list, dict, np.ndarray)) or _is_task(x):
        return False
    try:
        return x in graph
    except TypeError:
        return False


def _fusable(task):
    if not _is_task(task):
        return False
    op = getattr(task[0], "gm_op", None)
    if op is None:
        return False
    # table lookups join a program only when they are small enough to run inside one; otherwise
    # they keep their own kernel (and must not drag the rest of a group to the op-by-op path)
    if op == "power" and len(task) > 3 and isinstance(task[1], dict):
        # integer ** raster: NumPy raises when any exponent is negative, which only the op's own body checks
        if np.dtype(task[1].get("dtype", "f4")).kind in "iu" and not isinstance(task[3], (int, float, np.number)):
            return False
    if op == "classify":
        bins = task[2] if len(task) > 2 else None
        return hasattr(bins, "__len__") and len(bins) <= P.GM_MAX_LINEAR_EDGES
    if op == "reclassify":
        kwargs = task[2] if len(task) > 2 else None
        if not isinstance(kwargs, dict) or not kwargs.get("data"):
            return False
        keys = sorted(int(pair[0]) for pair in kwargs["data"])
        best, lo = 0, 0
        for hi in range(len(keys)):
            while keys[hi] - keys[lo] >= P.LOOKUP_MAX_WINDOW:
                lo += 1
            best = max(best, hi - lo + 1)
        return len(keys) - best <= P.LOOKUP_MAX_EXCEPTIONS - 1  # one slot stays free for the no-data key
    return True


class FusedTask(object):
    """Callable graph node standing for a fused group of element-wise tasks."""

    def __init__(self, members, order, root, leaf_keys):
        self.members = members      # key -> task tuple (func, *args) of the original graph
        self.order = order          # member keys, producers first
        self.root = root
        self.leaf_keys = leaf_keys  # keys (outside the group) the call arguments correspond to
        self._plans = {}            # tile signature -> (compiled program, slot binding, shape)
        self._kept = {}             # addresses of resident leaf tiles -> their launch, marshalled (see _kept_launch)
        self._leaf_set = set(leaf_keys)
        self.__name__ = "fused_%d_ops" % len(order)

    def __repr__(self):
        return "<FusedTask %s: %s>" % (self.root, " -> ".join(self.members[k][0].gm_op for k in self.order))

    def __getstate__(self):
        # a FusedTask kept by a graph template travels to worker processes (dask's multiprocessing scheduler)
        # without its compiled programs: those hold device handles of this process; the worker compiles its own
        state = dict(self.__dict__)
        state["_plans"] = {}
        state["_kept"] = {}
        return state

    # -- slow path: exact reference control flow, one launch per op ---------------------
    def _run_unfused(self, env):
        for key in self.order:
            func, args = self.members[key][0], self.members[key][1:]
            env[key] = func(*[env[a] if self._ref(a) else a for a in args])
        return env[self.root]

    def _ref(self, a):
        if isinstance(a, (list, dict, np.ndarray)):
            return False
        try:
            return a in self.members or a in self._leaf_set
        except TypeError:
            return False

    @staticmethod
    def _ndv_key(ndv):
        # a NaN no-data value must hit the cache too (NaN != NaN would miss for ever)
        if isinstance(ndv, (float, np.floating)) and ndv != ndv:
            return "nan"
        return ndv

    def _signature(self, leaf_values):
        try:
            return tuple((v["values"].dtype.num, v["values"].shape, type(v.get("no_data_value")),
                          self._ndv_key(v.get("no_data_value"))) for v in leaf_values)
        except (TypeError, KeyError, AttributeError, IndexError):  # (IndexError: a leaf that answered with a bare ndarray)
            return None

    _UNFUSABLE = ("unfusable",)  # cached verdict: this kind of tiles runs op by op

    def plan_for(self, leaf_values):
        """-> (stages, shape) when this kind of tiles runs as a fixed sequence of programs -- one
        program in the common case, a few when no register class serves the whole group -- else
        None (control-flow responses, not fusable).  A stage is (compiled, binding); a binding entry
        is (0, call argument index), (1, constant array) or (2, index of an earlier stage)."""
        sig = self._signature(leaf_values)
        if sig is None:
            return None
        hit = self._plans.get(sig)
        if hit is not None:
            return None if hit is self._UNFUSABLE else hit
        if not all(isinstance(v, dict) and "values" in v for v in leaf_values):
            return None
        if len(self._plans) > 64:
            self._plans.clear()
        nodes, arg_of = {}, {}
        for i, (key, data) in enumerate(zip(self.leaf_keys, leaf_values)):
            leaf = P.Leaf(None, data["values"].dtype, data.get("no_data_value"), values=data["values"])
            nodes[key] = leaf
            arg_of[id(leaf)] = i
        try:
            for key in self.order:
                nodes[key] = self._node(key, nodes)
            staged, shape = P.plan_staged(nodes[self.root])
        except (_NotFusable, P.FusionLimit, NotImplementedError, TypeError, IndexError):
            # (IndexError: frames that NumPy's boolean indexing would refuse -- op by op the Block that refuses them raises,
            #  or hands back what the reference hands back before it gets there: Clip of an all-no-data store)
            self._plans[sig] = self._UNFUSABLE  # do not rebuild the node tree for every tile
            return None
        stages = []
        for compiled, leaves in staged:
            binding = []
            for l in leaves:
                if isinstance(l, P.StageRef):
                    binding.append((2, l.stage))
                elif id(l) in arg_of:
                    arrived = tuple(leaf_values[arg_of[id(l)]]["values"].shape)
                    if arrived != tuple(shape) and math.prod(arrived) != 1:
                        # a raster of 1 frame against rasters of n (NumPy broadcasts them in a ufunc): P.plan expanded THIS
                        # call's tile; a remembered binding would hand the next call's tile to the program as it arrives.
                        # Rare (TemporalSum / a single-frame source next to a series): such tiles run op by op.
                        self._plans[sig] = self._UNFUSABLE
                        return None
                    binding.append((0, arg_of[id(l)]))
                else:
                    binding.append((1, l.values))  # constant ndarray operand, uploaded once
            stages.append((compiled, binding))
        self._plans[sig] = (stages, shape)
        return self._plans[sig]

    def _kept_launch(self, leaf_values):
        """The launch over exactly these tiles with its arguments marshalled once (``gm_prepare``), when every leaf value is
        resident data as it stands (``DeviceArray.resident``: a source raster on the device, a zero-copy window of one -- the
        same bytes at the same address whenever the same request comes again): a graph made per request for a configured
        scheduler hands this task fresh window OBJECTS of the same windows.  None: not such tiles / first seen just now."""
        try:
            key = tuple((v["values"].ptr, v["values"].pitch, v["values"].shape) if v["values"].resident else None for v in leaf_values)
        except (AttributeError, KeyError, TypeError, IndexError):  # (IndexError: a leaf that answered with a bare ndarray)
            return None
        if None in key:
            return None
        kept = self._kept.get(key)
        if kept is None:  # first sighting: remembered, nothing prepared (a task made for one request never comes back)
            if len(self._kept) >= 256:
                self._kept.clear()
            self._kept[key] = 0
            return None
        if kept == 0:     # second sighting: prepared; this call still takes the general path
            self._kept[key] = self.gm_prepare(*leaf_values) or False
            return None
        return kept or None

    def __call__(self, *leaf_values):
        # fast path: a fixed sequence of compiled programs (cached per kind of tiles)
        kept = self._kept_launch(leaf_values)
        if kept is not None:
            return kept()
        hit = self.plan_for(leaf_values)
        if hit is not None:
            stages, shape = hit
            outs = []
            for compiled, binding in stages:
                tiles = [P.as_device_view(leaf_values[ref]["values"]) if kind == 0 else
                         (outs[ref] if kind == 2 else P.as_device(ref)) for kind, ref in binding]
                outs.append(P.launch(compiled, tiles, shape))
            return {"values": outs[-1], "no_data_value": stages[-1][0].out_ndv}
        self._leaf_set = set(self.leaf_keys)
        return self._run_unfused(dict(zip(self.leaf_keys, leaf_values)))

    def gm_prepare(self, *leaf_values):
        """``__call__`` for leaf values that are the same objects every time (resident windows: plan.ViewPlan keeps the launch
        of a bounding box, core/graphs.py PreparedGraph the launches of a kept job) -> a callable without arguments that
        allocates the result tile and launches, or None where the general path is needed."""
        hit = self.plan_for(leaf_values)
        if hit is None or len(hit[0]) != 1:
            return None
        (compiled, binding), = hit[0]
        tiles = [[P.as_device_view(leaf_values[ref]["values"]) if kind == 0 else P.as_device(ref) for kind, ref in binding]]
        prepared = P.prepare_batch(compiled, tiles, hit[1])
        if prepared is None:
            return None
        ndv = compiled.out_ndv

        def run():
            return {"values": prepared.run()[0], "no_data_value": ndv}

        return run

    def structure(self):
        """Hashable description of the group up to the identity of its leaves: two groups with
        equal structure evaluate the same expression on different tiles."""
        pos = {k: ("m", i) for i, k in enumerate(self.order)}
        pos.update({k: ("l", i) for i, k in enumerate(self.leaf_keys)})

        def const(a):
            if isinstance(a, dict):
                return ("d", tuple(sorted((str(k), const(v)) for k, v in a.items())))
            if isinstance(a, (list, tuple)):
                return ("t", tuple(const(x) for x in a))
            if isinstance(a, np.ndarray):
                return ("a", id(a))  # the same constant array object
            if isinstance(a, np.generic):
                return ("g", a.dtype.str, a.tobytes())
            if isinstance(a, float) and a != a:
                return ("nan",)
            try:
                hash(a)
                return ("c", type(a).__name__, a)
            except TypeError:
                return ("o", id(a))

        out = []
        for key in self.order:
            task = self.members[key]
            out.append((task[0].gm_op, tuple(pos[a] if self._ref(a) else const(a) for a in task[1:])))
        return (pos[self.root], tuple(out))

    def _node(self, key, nodes):
        task = self.members[key]
        op, args = task[0].gm_op, task[1:]

        def operand(a):
            if self._ref(a):
                return nodes[a]
            if isinstance(a, np.ndarray):
                a = devops.constant_operand(a)
                return P.Leaf(None, a.dtype, None, values=a)
            if isinstance(a, (bool, int, float, np.generic)):
                return P.Lit(a)
            raise _NotFusable()

        if op in P.MATH_OPS or op in P.COMPARE_OPS or op in P.LOGIC_OPS:
            if not isinstance(args[0], dict):
                raise _NotFusable()
            return devops.math_node(op, args[0], [operand(a) for a in args[1:]])
        if op == "reduce_max":
            if not isinstance(args[0], dict) or len(args) < 2:  # (no raster at all: the Block's own process answers)
                raise _NotFusable()
            return devops.reduce_max_node(args[0], [operand(a) for a in args[1:]])
        if op == "fillnodata":
            # FillNoData's loop (elemwise.py:749-755): every raster in turn pasted where it has data
            if not isinstance(args[0], dict) or len(args) < 2:
                raise _NotFusable()
            dtype = np.dtype(args[0]["dtype"])
            fill = utils.get_dtype_max(dtype)
            node = P.Lit(fill)
            for a in args[1:]:
                if not self._ref(a):
                    raise _NotFusable()
                node = devops.fillnodata_node(node, operand(a), dtype, fill)
            return node
        if op == "clip":
            return devops.clip_node(operand(args[0]), operand(args[1]))
        if op == "mask":
            return devops.mask_node(operand(args[0]), args[1])
        if op == "maskbelow":
            return devops.maskbelow_node(operand(args[0]), args[1])
        if op == "step":
            return devops.step_node(operand(args[0]), *args[1:5])
        if op == "invert":
            return devops.invert_node(operand(args[0]))
        if op == "isdata":
            return devops.isdata_node(operand(args[0]))
        if op == "isnodata":
            return devops.isnodata_node(operand(args[0]))
        try:
            if op == "classify":
                return tables.classify_node(operand(args[0]), args[1], args[2])
            if op == "reclassify":
                return tables.reclassify_node(operand(args[0]), args[1])
        except P.FusionLimit:
            raise _NotFusable()
        raise _NotFusable()


class _NotFusable(Exception):
    pass


class BatchedTask(object):
    """Several structurally identical fused groups (the tiles of one view) as one graph node:
    called with the leaf values of all its tiles, returns the list of per-tile results."""

    def __init__(self, tasks):
        self.tasks = tasks                      # FusedTasks, one per tile
        self.n_leaves = len(tasks[0].leaf_keys)
        self.__name__ = "batch_%d_x_%s" % (len(tasks), tasks[0].__name__)

    def __repr__(self):
        return "<BatchedTask %d tiles of %r>" % (len(self.tasks), self.tasks[0])

    def __call__(self, *all_leaf_values):
        return self.run(all_leaf_values)

    def run(self, all_leaf_values, dest=None):
        """``dest(i, shape, dtype)`` may hand out the DeviceArray (e.g. a window of a mosaic) that the
        result of tile i is to be written into; None -> a fresh tile."""
        k = self.n_leaves
        per_tile = [all_leaf_values[i * k:(i + 1) * k] for i in range(len(self.tasks))]
        # every tile is planned by the first group (same structure => same plan for the same kind of
        # tiles); tiles of a different kind (edge tiles of another shape, None responses) run alone
        rep = self.tasks[0]
        results = [None] * len(per_tile)
        groups = {}
        for i, values in enumerate(per_tile):
            hit = rep.plan_for(values)
            if hit is None:
                results[i] = self.tasks[i](*values)
            else:
                groups.setdefault(id(hit), (hit, []))[1].append(i)
        for hit, members in groups.values():
            stages, shape = hit
            outs = []  # per stage: one result per member tile
            for stage_no, (compiled, binding) in enumerate(stages):
                tiles = [[P.as_device_view(per_tile[i][ref]["values"]) if kind == 0 else
                          (outs[ref][m] if kind == 2 else P.as_device(ref)) for kind, ref in binding]
                         for m, i in enumerate(members)]
                targets = None
                if dest is not None and stage_no == len(stages) - 1:
                    targets = [dest(i, shape, compiled.out_dtype) for i in members]
                    targets = [t if t is not None else DeviceArray.empty(shape, compiled.out_dtype) for t in targets]
                outs.append(P.launch_batch(compiled, tiles, shape, outs=targets))
            ndv = stages[-1][0].out_ndv
            for i, out in zip(members, outs[-1]):
                results[i] = {"values": out, "no_data_value": ndv}
        return results


    def gm_prepare(self, *all_leaf_values):
        """What ``run`` does per call that does not depend on the call when every leaf value is the same object every time (the
        resident windows a kept job reads: core/graphs.py PreparedGraph) -> a callable without arguments that allocates the
        result tiles and launches, or None where ``run`` is needed (tiles that do not fuse, programs in several stages)."""
        k = self.n_leaves
        per_tile = [all_leaf_values[i * k:(i + 1) * k] for i in range(len(self.tasks))]
        rep = self.tasks[0]
        groups = {}
        for i, values in enumerate(per_tile):
            hit = rep.plan_for(values)
            if hit is None or len(hit[0]) != 1:
                return None
            groups.setdefault(id(hit), (hit, []))[1].append(i)
        launches = []
        for hit, members in groups.values():
            (compiled, binding), = hit[0]
            tiles = [[P.as_device_view(per_tile[i][ref]["values"]) if kind == 0 else P.as_device(ref) for kind, ref in binding]
                     for i in members]
            prepared = P.prepare_batch(compiled, tiles, hit[1])
            if prepared is None:
                return None
            launches.append((prepared, members, compiled.out_ndv))
        n = len(per_tile)

        def run():
            results = [None] * n
            for prepared, members, ndv in launches:
                for i, out in zip(members, prepared.run()):
                    results[i] = {"values": out, "no_data_value": ndv}
            return results

        return run


def _getitem(seq, index):
    return seq[index]


class TiledAssembly(object):
    """RasterTiler.process (raster/parallelize.py:93-125) fused with the batches that compute its
    tiles: the mosaic is allocated first and every batched tile program writes its result straight
    into its window of the mosaic (row pitch = mosaic width, y axis flipped as the reference
    pastes), so the fill + paste passes over the result disappear.  Tiles that cannot be written in
    place (another dtype, another number of bands than the first tile's, ``None`` responses) fall back to the reference's paste."""

    def __init__(self, tiler_process, layout, batches):
        self.tiler_process = tiler_process
        self.layout = layout      # per tile position: (batch index, member index)
        self.batches = batches    # BatchedTasks in call-argument order
        self.__name__ = "tiled_%d" % len(layout)

    def __call__(self, process_kwargs, *all_leaf_values):
        height, width = process_kwargs["shape_yx"]
        count_x, count_y = process_kwargs["count_xy"]
        tile_x, tile_y = process_kwargs["tilesize_xy"]
        dtype, fill = np.dtype(process_kwargs["dtype"]), process_kwargs["fillvalue"]
        position = {bm: t for t, bm in enumerate(self.layout)}
        state = {"mosaic": None, "covered": 0}

        def dest_for(b):
            def dest(member, shape, out_dtype):
                if len(shape) != 3 or shape[0] < 1 or np.dtype(out_dtype) != dtype:
                    return None
                if state["mosaic"] is None:  # (several bands: the frames of a series, every tile a window of as many planes)
                    state["mosaic"] = DeviceArray.empty((shape[0], height, width), dtype)
                if state["mosaic"].shape[0] != shape[0]:
                    return None
                ix, iy = divmod(position[(b, member)], count_y)
                th, tw = shape[1], shape[2]
                row0, col0 = height - (iy * tile_y + th), ix * tile_x  # low y -> high rows (parallelize.py:114-124)
                if row0 < 0 or col0 + tw > width:
                    return None
                state["covered"] += th * tw  # the slots of distinct tiles are disjoint
                return state["mosaic"].window(row0, col0, th, tw)
            return dest

        results, offset = {}, 0
        for b, batch in enumerate(self.batches):
            n = batch.n_leaves * len(batch.tasks)
            for member, res in enumerate(batch.run(all_leaf_values[offset:offset + n], dest_for(b))):
                results[(b, member)] = res
            offset += n
        tiles = [results[bm] for bm in self.layout]
        mosaic = state["mosaic"]
        in_place = mosaic is not None and all(
            isinstance(t, dict) and isinstance(t.get("values"), DeviceArray) and t["values"]._owner is mosaic
            for t in tiles)
        # the mosaic was never filled: it only IS the result when the tiles written into it cover every
        # pixel.  The reference's int() truncation of the tile sizes (parallelize.py:85-86) can leave a
        # tile one pixel short of its slot for cell sizes that are not exact in binary; those pixels
        # keep the fill value of np.full there, so such requests take the fill + paste below.
        if in_place and len(tiles) == count_x * count_y and state["covered"] == height * width:
            return {"values": mosaic, "no_data_value": fill}
        return self.tiler_process(process_kwargs, *tiles)  # the reference's fill + paste


def tiles_in_place(tiler_process, block_type, process_kwargs, calls, one_by_one=False):
    """The tiles of a RasterTiler over a Block with ``gm_batch`` (a stencil): ``calls`` = what the Block's ``process``
    would get tile by tile; its kernel writes every tile into its window of the mosaic (one launch for all tiles where
    the Block has one).  The mosaic is the result only when the tiles cover it (TiledAssembly's rule); otherwise the
    reference's fill + paste.  None when the Block declines these tiles.  ``one_by_one``: ``calls`` is consumed tile
    after tile, each tile processed before the next one's arguments are asked for (a stencil over a COMPUTED store: the
    intermediate tile is read back while the Infinity Cache still holds it)."""
    height, width = process_kwargs["shape_yx"]
    count_x, count_y = process_kwargs["count_xy"]
    tile_x, tile_y = process_kwargs["tilesize_xy"]
    dtype = np.dtype(process_kwargs["dtype"])
    state = {"mosaic": None, "covered": 0}

    def dest(i, shape, out_dtype):
        if len(shape) != 3 or shape[0] != 1 or np.dtype(out_dtype) != dtype:
            return None
        ix, iy = divmod(i, count_y)
        th, tw = shape[1], shape[2]
        row0, col0 = height - (iy * tile_y + th), ix * tile_x  # low y -> high rows (parallelize.py:114-124)
        if row0 < 0 or col0 + tw > width or th > tile_y or tw > tile_x:
            return None
        if state["mosaic"] is None:
            state["mosaic"] = DeviceArray.empty((1, height, width), dtype)
        state["covered"] += th * tw
        return state["mosaic"].window(row0, col0, th, tw)

    if one_by_one:
        responses = []
        for i, call in enumerate(calls):
            one = block_type.gm_batch([call], lambda _, shape, out_dtype, i=i: dest(i, shape, out_dtype))
            responses.append(one[0] if one is not None else block_type.process(*call))
    else:
        responses = block_type.gm_batch(calls, dest)
        if responses is None:
            return None
    mosaic = state["mosaic"]
    in_place = mosaic is not None and len(responses) == count_x * count_y and all(
        isinstance(r, dict) and isinstance(r.get("values"), DeviceArray) and r["values"]._owner is mosaic for r in responses)
    if in_place and state["covered"] == height * width:
        return {"values": mosaic, "no_data_value": process_kwargs["fillvalue"]}
    return tiler_process(process_kwargs, *responses)


class MosaicOfTiles(object):
    """Graph task of a RasterTiler over a Block with ``gm_batch``: called with the tiler's process kwargs, the number of
    ``process`` arguments of every tile and those arguments back to back (the tiles' source responses computed by the
    scheduler) -> the mosaic, written in place (tiles_in_place), or process + paste tile by tile."""

    def __init__(self, tiler_process, block_type):
        self.tiler_process, self.block_type = tiler_process, block_type
        self.__name__ = "mosaic_of_%s" % block_type.__name__.lower()

    def __call__(self, process_kwargs, counts, *flat):
        calls, at = [], 0
        for n in counts:
            calls.append(tuple(flat[at:at + n]))
            at += n
        done = tiles_in_place(self.tiler_process, self.block_type, process_kwargs, calls)
        if done is not None:
            return done
        return self.tiler_process(process_kwargs, *[self.block_type.process(*call) for call in calls])


def fuse_graph(graph, name):
    """Return a graph in which fusable element-wise sub-graphs are single FusedTasks and
    structurally identical, independent FusedTasks are BatchedTasks.  ``name`` is the key (or
    the list of keys) that will be asked of the graph."""
    names = set(name) if isinstance(name, (list, tuple, set)) else {name}
    deps_memo = {}

    def deps_of(key):  # (a tiled view has thousands of keys: every key's dependencies are parsed once)
        hit = deps_memo.get(key)
        if hit is None:
            hit = deps_memo[key] = _deps(graph[key], graph)
        return hit

    # consumers of every key, and a consumers-before-producers order starting at the outputs
    consumers = {}
    order, seen, stack = [], set(), [(k, False) for k in names]
    while stack:
        key, expanded = stack.pop()
        if expanded:
            order.append(key)
            continue
        if key in seen:
            continue
        seen.add(key)
        stack.append((key, True))
        for dep in deps_of(key):
            consumers.setdefault(dep, set()).add(key)
            if dep not in seen:
                stack.append((dep, False))
    order.reverse()  # consumers first

    group_of = {}
    for key in order:
        if not _fusable(graph[key]):
            continue
        cons = consumers.get(key, ())
        groups = {group_of.get(c) for c in cons}
        if key not in names and cons and None not in groups and len(groups) == 1:
            group_of[key] = groups.pop()  # every consumer sits in one fused group: join it
        else:
            group_of[key] = key           # output of its group (feeds non-fusable / several groups)

    groups = {}
    for key in reversed(order):  # producers first
        root = group_of.get(key)
        if root is not None:
            groups.setdefault(root, []).append(key)

    new_graph = {k: v for k, v in graph.items() if k not in group_of}
    fused = {}
    for root, members in groups.items():
        member_set = set(members)
        leaf_keys = []
        for key in members:
            for dep in deps_of(key):
                if dep not in member_set and dep not in leaf_keys:
                    leaf_keys.append(dep)
        fused[root] = FusedTask({k: graph[k] for k in members}, members, root, leaf_keys)

    # horizontal batching: same structure, same depth in the graph (=> mutually independent)
    batches = {}
    if len(fused) > 1:
        depth = {}
        for key in reversed(order):  # producers first
            depth[key] = 1 + max([depth[d] for d in deps_of(key)] or [0])
        for root, task in fused.items():
            if not task.leaf_keys:
                continue
            try:
                sig = (task.structure(), depth[root], len(task.leaf_keys))
                hash(sig)
            except TypeError:
                continue
            batches.setdefault(sig, []).append(root)
    batched = set()
    for sig, roots in batches.items():
        for start in range(0, len(roots), MAX_BATCH_TILES):
            chunk = roots[start:start + MAX_BATCH_TILES]
            if len(chunk) < 2:
                continue
            tasks = [fused[r] for r in chunk]
            batch_key = "batch_%s" % chunk[0]
            new_graph[batch_key] = (BatchedTask(tasks),) + tuple(k for t in tasks for k in t.leaf_keys)
            for i, r in enumerate(chunk):
                new_graph[r] = (_getitem, batch_key, i)
                batched.add(r)
    for root, task in fused.items():
        if root in batched:
            continue
        if len(task.order) == 1:
            new_graph[root] = graph[root]  # nothing to fuse
        else:
            new_graph[root] = (task,) + tuple(task.leaf_keys)

    # a RasterTiler whose tiles all come out of batches (and feed nothing else) assembles in place
    for key, task in list(new_graph.items()):
        if not (_is_task(task) and getattr(task[0], "gm_tiler", False) and len(task) > 2 and isinstance(task[1], dict)):
            continue
        tile_keys = task[2:]
        try:
            ok = all(k in batched and consumers.get(k) == {key} for k in tile_keys)
        except TypeError:
            ok = False
        if not ok or len(set(tile_keys)) != len(tile_keys):
            continue
        batch_keys, layout = [], []
        for k in tile_keys:
            _, bkey, member = new_graph[k]
            if bkey not in batch_keys:
                batch_keys.append(bkey)
            layout.append((batch_keys.index(bkey), member))
        batches = [new_graph[b][0] for b in batch_keys]
        if sum(len(b.tasks) for b in batches) != len(layout):
            continue  # a batch also serves tiles of another consumer
        leaf_args = tuple(a for b in batch_keys for a in new_graph[b][1:])
        new_graph[key] = (TiledAssembly(task[0], layout, batches), task[1]) + leaf_args
    return new_graph


def _deps(task, graph):
    out = []

    def walk(obj):
        if _is_task(obj):
            for a in obj[1:]:
                walk(a)
        elif isinstance(obj, list):
            for a in obj:
                walk(a)
        elif _is_key(obj, graph):
            out.append(obj)

    walk(task)
    return out


def materialize(result):
    """DeviceArray -> ndarray in a raster response (the end-point of ``get_data``)."""
    if isinstance(result, dict) and isinstance(result.get("values"), DeviceArray):
        result = dict(result)
        result["values"] = result["values"].to_numpy()
    elif isinstance(result, DeviceArray):
        result = result.to_numpy()
    return result
