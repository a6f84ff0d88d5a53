"""Tiny (de)serialiser for golden vectors: JSON manifest + one ``.npz`` of arrays.

Keeps the *exact python/NumPy type* of every scalar (``float`` vs ``np.float32``
matters for NumPy's promotion rules, e.g. in ``np.isclose``).
Test infrastructure only.
"""
import json
from datetime import datetime, timedelta

import numpy as np


class Writer:
    def __init__(self):
        self.arrays = {}
        self.cases = []
        self._seen = {}  # id(array) -> key: an array that many cases share (a probe source) is stored once

    def _enc(self, obj):
        if obj is None or isinstance(obj, (bool, str)):
            return obj
        if isinstance(obj, np.ndarray):
            key = self._seen.get(id(obj))
            if key is None or self.arrays[key] is not obj:
                key = "a%d" % len(self.arrays)
                self.arrays[key] = obj
                self._seen[id(obj)] = key
            return {"__nd__": key}
        if isinstance(obj, np.generic):
            if obj.dtype.kind == "f":
                return {"__np__": obj.dtype.str, "hex": float(obj).hex()}
            if obj.dtype.kind == "b":
                return {"__np__": obj.dtype.str, "v": bool(obj)}
            return {"__np__": obj.dtype.str, "v": int(obj)}
        if isinstance(obj, int):
            return {"__int__": str(obj)}
        if isinstance(obj, float):
            return {"__float__": obj.hex()}
        if isinstance(obj, np.dtype):
            return {"__dtype__": obj.str}
        if isinstance(obj, datetime):
            return {"__datetime__": obj.isoformat()}
        if isinstance(obj, timedelta):
            return {"__timedelta__": [obj.days, obj.seconds, obj.microseconds]}
        if isinstance(obj, (list, tuple)):
            return {"__seq__": type(obj).__name__, "v": [self._enc(x) for x in obj]}
        if isinstance(obj, dict):
            return {"__dict__": [[self._enc(k), self._enc(v)] for k, v in obj.items()]}
        raise TypeError("cannot encode %r" % type(obj))

    def add(self, **case):
        self.cases.append({k: self._enc(v) for k, v in case.items()})

    def save(self, stem):
        np.savez_compressed(stem + ".npz", **self.arrays)
        with open(stem + ".json", "w") as f:
            json.dump(self.cases, f, indent=0, separators=(",", ":"))


def _dec(obj, arrays):
    if obj is None or isinstance(obj, (bool, str)):
        return obj
    if isinstance(obj, dict):
        if "__nd__" in obj:
            return arrays[obj["__nd__"]]
        if "__np__" in obj:
            dt = np.dtype(obj["__np__"])
            if "hex" in obj:
                return dt.type(float.fromhex(obj["hex"]))
            return dt.type(obj["v"])
        if "__int__" in obj:
            return int(obj["__int__"])
        if "__float__" in obj:
            return float.fromhex(obj["__float__"])
        if "__dtype__" in obj:
            return np.dtype(obj["__dtype__"])
        if "__datetime__" in obj:
            return datetime.fromisoformat(obj["__datetime__"])
        if "__timedelta__" in obj:
            days, seconds, microseconds = obj["__timedelta__"]
            return timedelta(days=days, seconds=seconds, microseconds=microseconds)
        if "__seq__" in obj:
            seq = [_dec(x, arrays) for x in obj["v"]]
            return tuple(seq) if obj["__seq__"] == "tuple" else seq
        if "__dict__" in obj:
            return {_dec(k, arrays): _dec(v, arrays) for k, v in obj["__dict__"]}
    raise TypeError("cannot decode %r" % (obj,))


def load(stem):
    """Return the list of cases stored under ``stem`` (.json + .npz)."""
    with np.load(stem + ".npz") as z:
        arrays = {k: z[k] for k in z.files}
    with open(stem + ".json") as f:
        raw = json.load(f)
    return [{k: _dec(v, arrays) for k, v in case.items()} for case in raw]
