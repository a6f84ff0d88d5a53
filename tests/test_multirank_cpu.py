"""N > 1 path on the CPU: tile sharding logic and the bench timing harness under
torch.distributed (gloo, world_size 2)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from dask_geomodeling_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_ownership_is_a_partition():
    for n_tiles in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            seen = []
            for rank in range(world):
                mine = shard.owned_tiles(n_tiles, rank, world)
                assert all(shard.tile_owner(t, world) == rank for t in mine)
                seen += mine
            assert sorted(seen) == list(range(n_tiles))
            sizes = [len(shard.owned_tiles(n_tiles, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1  # balanced
    with pytest.raises(ValueError):
        shard.owned_tiles(4, 2, 2)


def test_run_owned_tiles_evaluates_only_own_subgraphs():
    calls = []

    def tile(i):
        calls.append(i)
        return {"values": i}

    graph = {"t%d" % i: (tile, i) for i in range(6)}
    graph["tiler"] = ((lambda kw, *tiles: tiles), {"count_xy": (3, 2)}) + tuple("t%d" % i for i in range(6))
    got = shard.run_owned_tiles(graph, "tiler", rank=1, world=2, sink=lambda t, r: r["values"] * 10)
    assert got == {1: 10, 3: 30, 5: 50} and calls == [1, 3, 5]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_bench_harness_world_size_2_gloo():
    """bench.py under torch.distributed.run with 2 ranks (gloo, no GPU: --dry-run): one JSON line
    from rank 0, n_gpus = 2, whole-job value = units of both ranks / max time."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--tiles", "4", "--side", "64", "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "Mpix/s" and out["vs_baseline"] is None and "cpu_baseline" not in out
    # the strong-scaling mosaic over the N devices, run by rank 0 after the ranks have dissolved their group, in the same line
    curve = out["secondary"]["mosaic"]
    assert curve["scaling"] == "strong" and curve["workers"] == 2 and len(curve["worker_devices"]) == 2
    assert len(curve["per_worker_device_ms_per_step"]) == 2 and curve["value"] > 0
    # 2 ranks x 4 tiles x 64^2 px per step
    assert abs(out["value"] - 2 * 4 * 64 * 64 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) / out["value"] < 0.01


@pytest.mark.timeout(300)
def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the documented CLI): the parent starts
    two fresh rank processes itself and relays rank 0's single JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tiles", "4", "--side", "64", "--dry-run"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 4 * 64 * 64 * 3 / (out["ms_per_step"] * 3e-3) / 1e6) / out["value"] < 0.01
    assert out["ranks_reported"] == 2 and [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert all(r["Mpix/s"] > 0 and r["ms_per_step"] <= out["ms_per_step"] * 1.001 for r in out["per_rank"])
    assert out["secondary"]["mosaic"]["scaling"] == "strong" and out["secondary"]["mosaic"]["workers"] == 2


@pytest.mark.timeout(300)
def test_bench_mosaic_curve_can_be_switched_off():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tiles", "4", "--side", "64", "--dry-run", "--no-mosaic-curve"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and "secondary" not in out


@pytest.mark.timeout(120)
def test_bench_fails_within_seconds_when_a_rank_dies():
    """A rank that dies (GM_BENCH_FAIL_RANK) must not leave the others waiting in a barrier until the process
    group's timeout -- the parent watches all its children, stops the survivors and reports which rank failed."""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tiles", "4", "--side", "64", "--dry-run"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GM_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=100, env=env, cwd=ROOT)
    took = time.time() - t0
    assert res.returncode != 0
    assert "rank(s) failed" in res.stderr and "(1, " in res.stderr, res.stderr[-1000:]
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")], "no result line from a failed run"
    assert took < 30, "took %.0f s" % took  # (seconds: interpreter + torch start-up, not a barrier timeout)


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_gpus_flag_two_ranks_on_the_gpu_box(gpu):
    """`python bench.py --gpus 2` on real hardware, no launcher: two self-spawned ranks (sharing the box's
    single MI355X), for the headline chain and for the sharded mosaic view (strong scaling, every rank
    holding only the leaf windows of the tiles it owns)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    res = subprocess.run(base + ["--steps", "10", "--warmup", "3", "--tiles", "4", "--side", "1024", "--secondary-steps", "2",
                                 "--no-e2e", "--leaves", "procedural"],
                         capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["roofline"]["achieved"] > 0
    # (leaves regenerated per request: the suite's other tests use the GPU's memory at the same time)
    # the driver's command carries the strong-scaling curve point of BASELINE configs[4]: the 65536^2 mosaic dealt over two
    # workers by the product's scheduler after the ranks have given their GPU back (here both on the box's one GPU)
    curve = out["secondary"]["mosaic"]
    assert "error" not in curve, curve
    assert curve["scaling"] == "strong" and curve["workers"] == 2 and curve["tiles_per_worker"] == 128
    assert abs(curve["value"] - 65536.0 ** 2 * 2 / (curve["ms_per_step"] * 2e-3) / 1e6) / curve["value"] < 0.01
    assert len(curve["worker_host_us_per_step"]) == 2 and curve["parent_us_per_step"] > 0
    # (leaves regenerated per request: the two workers share this box's one GPU with whatever else the suite is running)
    res = subprocess.run(base + ["--workload", "mosaic", "--side", "4096", "--steps", "2", "--warmup", "1", "--leaves", "procedural", "--no-e2e"],
                         capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["tiles_per_gpu"] == 128
    assert abs(out["value"] - 65536.0 ** 2 * 2 / (out["ms_per_step"] * 2e-3) / 1e6) / out["value"] < 0.01


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_world_size_2_on_the_gpu_box(gpu):
    """The N > 1 path on real hardware: two ranks under torch.distributed.run (gloo barrier +
    max-reduce), each with its own batch of tiles, here sharing the box's single MI355X; one JSON
    line from rank 0 whose value counts the pixels of both ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "10", "--warmup", "3", "--tiles", "4", "--side", "1024", "--no-mosaic-curve"]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 10 and out["data"] == "synthetic"
    assert abs(out["value"] - 2 * 4 * 1024 * 1024 * 10 / (out["ms_per_step"] * 10e-3) / 1e6) / out["value"] < 0.01
    assert out["roofline"]["achieved"] > 0 and "cpu_baseline" not in out
