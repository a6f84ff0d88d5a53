"""Configuration (counterpart of the reference's config.py:4-12).

Uses ``dask.config`` when dask is importable, otherwise a dict-backed shim with the
same ``get / set / update_defaults`` surface.  Keys live under ``geomodeling.*``;
besides the reference's own keys this path adds:

  geomodeling.fuse         collapse element-wise sub-graphs into one launch (default True)
  geomodeling.plans        get_data() reuses the fused plan of a view for requests that differ in their
                           bbox only instead of rebuilding the graph (default True; plan.py)
  geomodeling.templates    the fused graph of a request is made from the view's template instead of built and fused
                           from scratch (default True; template.py)
  geomodeling.materialize  get_data() returns ndarrays (True) or DeviceArrays (False)
  geomodeling.devices      None: evaluate in this process, on its current device.  A list of HIP device indices (or "all"):
                           get_data() / compute() hand the work to one worker process per listed GPU (shard.py) -- the
                           tiles of a RasterTiler are dealt t % G over them, the host mosaic is assembled from their tiles
  geomodeling.worker-preload  "module:function" every such worker imports and calls before it binds its device (a
                           deployment's own Blocks)
  geomodeling.worker-affinity  every such worker pins itself to the host cores next to its GPU (the PCI device's
                           local_cpulist / NUMA node) before it binds the device (default True); workers that share a GPU's
                           cores split them; False leaves placement to the operating system
  geomodeling.resident-window-bytes  bound of the per-device window cache of resident SyntheticSources (LRU)
"""
import os
import threading

try:  # pragma: no cover - depends on the environment
    import dask.config as _dask_config
except ImportError:  # the GPU image ships no dask
    _dask_config = None

defaults = {
    "root": os.getcwd(),
    "strict-file-paths": False,
    "raster-limit": 12 * (1024 ** 2),
    "raster-limit-timesteps": 65536,
    "geometry-limit": 10000,
    "fuse": True,
    "plans": True,
    "templates": True,
    "materialize": True,
    "devices": None,
    "worker-preload": None,
    "worker-affinity": True,
    "resident-window-bytes": 192 << 30,
}


class _Shim:
    def __init__(self):
        self._data = {}
        self._flat = {}  # "a.b.c" -> value of the lookups made so far (get_data reads half a dozen keys per request)
        self._generation = 0  # bumped by every write: a lookup that raced a write is not remembered
        self._lock = threading.Lock()

    def update_defaults(self, new):
        with self._lock:
            for k, v in new.items():
                if isinstance(v, dict):
                    cur = self._data.setdefault(k, {})
                    for kk, vv in v.items():
                        cur.setdefault(kk, vv)
                else:
                    self._data.setdefault(k, v)
            self._generation += 1
            self._flat.clear()

    def get(self, key, default=KeyError):
        try:
            return self._flat[key]
        except KeyError:
            pass
        generation = self._generation
        value = self._lookup(key, default)
        if value is not default:
            with self._lock:
                if generation == self._generation:  # no set() since the lookup began: the value is current
                    self._flat[key] = value
        return value

    def _lookup(self, key, default):
        cur = self._data
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is KeyError:
                    raise KeyError(key)
                return default
            cur = cur[part]
        return cur

    class set:  # noqa: N801 - mirrors dask.config.set (usable as context manager)
        def __init__(self, arg=None, **kwargs):
            self._old = {}
            items = dict(arg or {})
            items.update({k.replace("__", "."): v for k, v in kwargs.items()})
            for key, value in items.items():
                self._old[key] = config.get(key, None)
                self._assign(key, value)

        @staticmethod
        def _assign(key, value):
            parts = key.split(".")
            with config._lock:
                cur = config._data
                for part in parts[:-1]:
                    cur = cur.setdefault(part, {})
                cur[parts[-1]] = value
                config._generation += 1
                config._flat.clear()

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            for key, value in self._old.items():
                self._assign(key, value)
            return False


config = _dask_config if _dask_config is not None else _Shim()
config.update_defaults({"geomodeling": defaults})


def get(key, default=None):
    try:
        return config.get(key)
    except KeyError:
        return default
