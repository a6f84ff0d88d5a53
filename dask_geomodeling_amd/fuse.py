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

``None`` / ``time`` / ``meta`` responses are not numerics: when any leaf of a fused group is
not a tile of values the member tasks are simply run one by one with their own ``process``
bodies, which carry the reference's control flow for those cases.
"""
import numpy as np

from .hip import program as P
from .hip.array import DeviceArray
from .raster import devops

__all__ = ["fuse_graph", "materialize", "FusedTask"]


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
    return _is_task(task) and getattr(task[0], "gm_op", None) is not None


class FusedTask(object):
    """Callable graph node standing for a fused group of element-wise tasks."""

    def __init__(self, members, order, root, leaf_keys):
        self.members = members      # key -> task tuple (func, *args) of the original graph
        self.order = order          # member keys, producers first
        self.root = root
        self.leaf_keys = leaf_keys  # keys (outside the group) the call arguments correspond to
        self._plans = {}            # tile signature -> (compiled program, slot binding, shape)
        self._leaf_set = set(leaf_keys)
        self.__name__ = "fused_%d_ops" % len(order)

    def __repr__(self):
        return "<FusedTask %s: %s>" % (self.root, " -> ".join(self.members[k][0].gm_op for k in self.order))

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

    def __call__(self, *leaf_values):
        # fast path: the same kinds of tiles as last time -> the cached program, one launch
        try:
            sig = tuple((v["values"].dtype.num, v["values"].shape, type(v.get("no_data_value")),
                         v.get("no_data_value")) for v in leaf_values)
            hit = self._plans.get(sig)
        except (TypeError, KeyError, AttributeError):
            sig = hit = None
        if hit is not None:
            compiled, binding, shape = hit
            tiles = [leaf_values[ref]["values"] if kind == 0 else ref for kind, ref in binding]
            values = P.launch(compiled, [P.as_device(t) for t in tiles], shape)
            return {"values": values, "no_data_value": compiled.out_ndv}

        self._leaf_set = set(self.leaf_keys)
        env = dict(zip(self.leaf_keys, leaf_values))
        if not all(isinstance(v, dict) and "values" in v for v in leaf_values):
            return self._run_unfused(env)
        nodes, arg_of = {}, {}
        for i, (key, data) in enumerate(env.items()):
            leaf = P.Leaf(None, data["values"].dtype, data.get("no_data_value"), values=data["values"])
            nodes[key] = leaf
            arg_of[id(leaf)] = i
        try:
            for key in self.order:
                nodes[key] = self._node(key, nodes)
        except _NotFusable:
            return self._run_unfused(env)
        root = nodes[self.root]
        try:
            compiled, leaves, shape = P.plan(root)
        except P.FusionLimit:
            values, ndv = P.evaluate(root)  # several launches: sub-trees first
            return {"values": values, "no_data_value": ndv}
        if sig is not None:
            # which call argument (or constant ndarray operand, uploaded once) feeds each slot
            binding = [(0, arg_of[id(l)]) if id(l) in arg_of else (1, l.values) for l in leaves]
            if len(self._plans) > 64:
                self._plans.clear()
            self._plans[sig] = (compiled, binding, shape)
        values = P.launch(compiled, [leaf.values for leaf in leaves], shape)
        return {"values": values, "no_data_value": compiled.out_ndv}

    def _node(self, key, nodes):
        task = self.members[key]
        op, args = task[0].gm_op, task[1:]

        def operand(a):
            if self._ref(a):
                return nodes[a]
            if isinstance(a, np.ndarray):
                return P.Leaf(None, a.dtype, None, values=a)
            if isinstance(a, (bool, int, float, np.generic)):
                return P.Lit(a)
            raise _NotFusable()

        if op in P.MATH_OPS or op in P.COMPARE_OPS or op in P.LOGIC_OPS:
            if not isinstance(args[0], dict):
                raise _NotFusable()
            return devops.math_node(op, args[0], [operand(a) for a in args[1:]])
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
        raise _NotFusable()


class _NotFusable(Exception):
    pass


def fuse_graph(graph, name):
    """Return a graph in which fusable element-wise sub-graphs are single FusedTasks."""
    # consumers of every key, and a consumers-before-producers order starting at `name`
    consumers = {}
    order, seen, stack = [], set(), [(name, False)]
    while stack:
        key, expanded = stack.pop()
        if expanded:
            order.append(key)
            continue
        if key in seen:
            continue
        seen.add(key)
        stack.append((key, True))
        for dep in _deps(graph[key], graph):
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
        if key != name and cons and None not in groups and len(groups) == 1:
            group_of[key] = groups.pop()  # every consumer sits in one fused group: join it
        else:
            group_of[key] = key           # output of its group (feeds non-fusable / several groups)

    groups = {}
    for key in reversed(order):  # producers first
        root = group_of.get(key)
        if root is not None:
            groups.setdefault(root, []).append(key)

    new_graph = {k: v for k, v in graph.items() if k not in group_of}
    for root, members in groups.items():
        if len(members) == 1:
            new_graph[root] = graph[root]  # nothing to fuse
            continue
        member_set = set(members)
        leaf_keys = []
        for key in members:
            for dep in _deps(graph[key], graph):
                if dep not in member_set and dep not in leaf_keys:
                    leaf_keys.append(dep)
        task = FusedTask({k: graph[k] for k in members}, members, root, leaf_keys)
        new_graph[root] = (task,) + tuple(leaf_keys)
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
