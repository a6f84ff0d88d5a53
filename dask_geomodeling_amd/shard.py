"""One-worker-per-GPU sharding of independent tile tasks (no collective on the data path).

``RasterTiler.get_sources_and_requests`` emits its tiles x-major, low y first
(raster/parallelize.py:82); tile ``t`` of that order belongs to rank ``t % world``.  Each rank
evaluates only its own tile sub-graphs on its own GPU; nothing is exchanged between GPUs.
"""
from .core.graphs import get_sync

__all__ = ["tile_owner", "owned_tiles", "split_tiler_graph", "run_owned_tiles"]


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
