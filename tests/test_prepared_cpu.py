"""core/graphs.py PreparedGraph: get_sync for a graph that is run again and again (the kept jobs of the sharded scheduler's
workers).  No device here: the resident answers are stand-ins carrying the ``resident`` flag of a DeviceArray."""
import numpy as np

from dask_geomodeling_amd.core.graphs import PreparedGraph, get_sync


class Window(object):
    def __init__(self, data, resident):
        self.data, self.resident = data, resident


CALLS = []


def source(name, value, resident=True):
    CALLS.append(name)
    return {"values": Window(np.full((2, 2), value), resident), "no_data_value": -1}


def add(a, b):
    CALLS.append("add")
    return {"values": Window(a["values"].data + b["values"].data, False), "no_data_value": -1}


def total(parts):
    CALLS.append("total")
    return float(sum(p["values"].data.sum() for p in parts))


class Batch(object):
    """A function that offers gm_prepare (what fuse.BatchedTask does)."""
    prepared = 0

    def __call__(self, *values):
        CALLS.append("batch")
        return [v["values"].data.sum() for v in values]

    def gm_prepare(self, *values):
        Batch.prepared += 1
        sums = [v["values"].data.sum() for v in values]

        def run():
            CALLS.append("prepared-batch")
            return list(sums)
        return run


def _graph():
    batch = Batch()
    return {
        "a": (source, "src-a", 1.0), "b": (source, "src-b", 2.0), "c": (source, "src-c", 3.0, False),
        "ab": (add, "a", "b"), "abc": (add, "ab", "c"),
        "sum": (total, ["ab", "abc", "a"]),
        "batch": (batch, "a", "b"),
        "mixed": (batch, "a", "c"),
    }


def test_prepared_graph_equals_get_sync_and_keeps_resident_answers():
    graph = _graph()
    keys = ["abc", "sum", "batch", "mixed", "a"]
    want = get_sync(graph, keys)
    del CALLS[:]
    Batch.prepared = 0
    prepared = PreparedGraph(graph, keys)
    for turn in range(3):
        del CALLS[:]
        got = prepared.run()
        assert np.array_equal(got[0]["values"].data, want[0]["values"].data) and got[1] == want[1]
        assert got[2] == want[2] and got[3] == want[3] and got[4]["values"].data[0, 0] == 1.0
        if turn == 0:
            assert sorted(CALLS) == sorted(["src-a", "src-b", "src-c", "add", "add", "total", "batch", "batch"])
        else:
            # the resident sources are not asked again; the source whose answer is NOT resident is; every computation runs; the
            # batch over kept answers only runs prepared, the one with a live argument as it is
            assert sorted(CALLS) == sorted(["src-c", "add", "add", "total", "prepared-batch", "batch"])
    assert Batch.prepared == 1 and set(prepared.kept) == {"a", "b"} and prepared.runs == 3


def test_prepared_graph_single_key_and_release_order():
    graph = _graph()
    prepared = PreparedGraph(graph, "sum")
    assert prepared.run() == get_sync(graph, "sum") == prepared.run()
    # live results are dropped after their last live consumer: "ab" after "sum" (its last user in evaluation order)
    releases = {key: rel for key, _, rel in prepared.program}
    assert "ab" in releases["sum"] and "abc" in releases["sum"] and releases["ab"] == ()


def test_a_failed_first_run_is_retried_from_scratch():
    state = {"fail": True}

    def flaky():
        if state["fail"]:
            raise ValueError("not yet")
        return {"values": Window(np.ones((1, 1)), True), "no_data_value": 0}

    prepared = PreparedGraph({"x": (flaky,), "y": (add, "x", "x")}, "y")
    try:
        prepared.run()
        raise AssertionError("should have raised")
    except ValueError:
        pass
    state["fail"] = False
    assert prepared.run()["values"].data[0, 0] == 2.0 and prepared.run()["values"].data[0, 0] == 2.0
    assert set(prepared.kept) == {"x"}
