"""Import the *reference* package (read-only, /root/reference) in THIS container.

Test infrastructure only: used by ``make_golden.py`` to generate golden vectors.
It never travels to the GPU box (``/root/reference`` does not exist there) and is
never imported by the product package.

The reference needs GDAL / shapely / geopandas / pyproj / pyogrio / dask, none of
which are installed.  The hot-path ``process()`` bodies only need NumPy, so the
missing third-party modules are replaced by permissive stubs.
"""
import importlib.machinery
import hashlib
import pickle
import sys
import types

sys.dont_write_bytecode = True  # never write into /root/reference

REFERENCE_ROOT = "/root/reference"


class _Dummy:
    """Callable, attribute-able placeholder object."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def startswith(self, *a):
        return True

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


def _stub(name, **attrs):
    mod = _StubModule(name)
    mod.__path__ = []
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def _cls(name, base=object):
    return type(name, (base,), {"__init__": lambda self, *a, **k: None})


def install_stubs():
    if "osgeo" in sys.modules:
        return
    for name in ("osgeo", "osgeo.gdal", "osgeo.ogr", "osgeo.osr", "osgeo.gdal_array"):
        _stub(name)
    _stub("shapely", GEOSException=_cls("GEOSException", Exception),
          Point=_cls("Point"), from_wkt=lambda *a, **k: None)
    _stub("shapely.geometry", box=lambda *a, **k: None, Point=_cls("Point"),
          Polygon=_cls("Polygon"))
    _stub("shapely.geometry.base", BaseGeometry=_cls("BaseGeometry"))
    _stub("shapely.ops")
    _stub("shapely.errors", ShapelyError=_cls("ShapelyError", Exception))
    _stub("shapely.wkt")
    _stub("geopandas", GeoSeries=_cls("GeoSeries"), GeoDataFrame=_cls("GeoDataFrame"))
    _stub("pyproj", CRS=_cls("CRS"), Transformer=_cls("Transformer"))
    _stub("pyproj.exceptions", ProjError=_cls("ProjError", Exception))
    _stub("pyogrio")
    try:
        import dask  # noqa: F401
    except ImportError:
        _install_dask_stub()


def _install_dask_stub():
    cfg = {}

    def update_defaults(d):
        for k, v in d.items():
            cfg.setdefault(k, {}).update(v) if isinstance(v, dict) else cfg.setdefault(k, v)

    def get(key, default=None):
        cur = cfg
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                return default
            cur = cur[part]
        return cur

    class _Set:
        def __init__(self, d=None, **kw):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def tokenize(*args, **kwargs):
        try:
            payload = pickle.dumps((args, kwargs), protocol=4)
        except Exception:
            payload = repr((args, kwargs)).encode()
        return hashlib.md5(payload).hexdigest()

    class _Normalize:
        def register(self, typ):
            return lambda f: f

    def get_sync(dsk, keys, **kw):
        cache = {}

        def ev(x):
            if isinstance(x, str) and x in dsk:
                if x not in cache:
                    cache[x] = ev(dsk[x])
                return cache[x]
            if isinstance(x, tuple) and x and callable(x[0]):
                return x[0](*[ev(a) for a in x[1:]])
            if isinstance(x, list):
                return [ev(a) for a in x]
            return x

        if isinstance(keys, list):
            return [ev(k) for k in keys]
        return ev(keys)

    dask = _stub("dask")
    config = _stub("dask.config", update_defaults=update_defaults, get=get, set=_Set)
    base = _stub("dask.base", tokenize=tokenize, normalize_token=_Normalize(),
                 get_scheduler=lambda *a, **k: None)
    local = _stub("dask.local", get_sync=get_sync)
    dask.config, dask.base, dask.local = config, base, local


def import_reference():
    """Return the imported reference package ``dask_geomodeling``."""
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import dask_geomodeling  # noqa

    return dask_geomodeling
