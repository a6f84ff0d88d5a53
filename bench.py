#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of the fused 5-op element-wise chain on 4096x4096 f32 tiles.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU a batch of B device-resident tile
pairs (A_i, B_i), float32 4096x4096, no-data = float32 max on a [::97, ::89] lattice; one
"step" evaluates   V_i = Clip(Add(Subtract(Multiply(Add(A_i, B_i), 2.5), B_i), 1.5), A_i)
for every tile of the batch through the Block API: MemorySource leaves -> get_compute_graph of
every view, merged into one graph (the dask idiom for independent tiles) -> fusion pass: the five
Blocks of a tile become one program, the B identical tile programs one batched launch
(gm_elemwise_run_batch, 16 tiles per launch) -> the scheduler runs it.  The batch (B x 192 MiB) is
far larger than the 256 MiB Infinity Cache, so every launch streams from HBM.

Tiles are independent: with N GPUs every rank owns its own batch (weak scaling), there is no
collective on the data path; torch.distributed (gloo) is used for the timing barrier and the
max-over-ranks reduction only.

Rank 0 prints ONE JSON line.  roofline.achieved = algorithmic bytes per launch (12 B/px x
16.78 Mpix x tiles per launch) / average launch duration from HIP events (recorded on the
library's compute stream) over the timed region.  roofline.traffic = HBM bytes per launch from
the rocprofv3 PMC passes of the same command (profiles/<round>_pmc.json, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950), when that file is present.
cpu_baseline = the NumPy oracle (a restatement of the reference's process() bodies) on the
host cores of the same box, same chain, same synthetic tiles (bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
BYTES_PER_PX = 12              # read A (4) + read B (4) + write V (4)   
F32MAX = float(np.finfo("f4").max)
LATTICE = (97, 89)


MOSAIC_SIDE_DEFAULT = 65536


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="chain5", choices=["chain5", "reclass", "bilinear", "mosaic", "tiler"],
                   help="chain5 = the headline (BASELINE.json configs[1]); reclass = configs[2] "
                        "(Reclassify + Mask on int32 tiles); bilinear = configs[3] (2x upsample); "
                        "mosaic = configs[4] (10-Block view over a 65536^2 procedural mosaic, tiles sharded)")
    p.add_argument("--side", type=int, default=0, help="tile side in pixels (0 = the workload's own size)")
    p.add_argument("--tiles", type=int, default=0, help="tiles per GPU per step (0 = the workload's default)")
    p.add_argument("--cpu-tiles", type=int, default=-1, help="tiles of the CPU baseline sample (0 = skip)")
    p.add_argument("--no-check", action="store_true", help="skip the oracle check of one tile")
    p.add_argument("--no-e2e", action="store_true", help="skip the end_to_end block (get_data rates seen by a caller)")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the `secondary` block of the headline run (BASELINE configs[2], [3], [4] clocked in the same run)")
    p.add_argument("--secondary-steps", type=int, default=20)
    p.add_argument("--in-process", action="store_true",
                   help="mosaic only: run the tiles' fused programs in THIS process (what one worker of the scheduler does), "
                        "so that rocprofv3 can attach to the kernels; never the product's route")
    p.add_argument("--leaves", default="owned", choices=["owned", "whole", "procedural"],
                   help="mosaic leaves: owned = every rank keeps the leaf windows of the tiles it owns resident "
                        "(4 x 16 GiB / N per GPU, generated on the owning GPU during warm-up); whole = every rank "
                        "holds the four whole 16 GiB leaves as MemorySources (tiles = zero-copy windows); "
                        "procedural = every window is regenerated on the fly inside the timed region")
    p.add_argument("--procedural", action="store_true", help="same as --leaves procedural")
    p.add_argument("--mosaic-side", type=int, default=MOSAIC_SIDE_DEFAULT,
                   help="mosaic only: side of the whole mosaic (65536 = BASELINE configs[4]; small mosaics of small tiles leave "
                        "the GPU idle and time the HOST side of the scheduler, e.g. --mosaic-side 4096 --side 256)")
    p.add_argument("--no-affinity", action="store_true",
                   help="mosaic only: workers are not pinned to the cores next to their GPU (geomodeling.worker-affinity off)")
    p.add_argument("--no-mosaic-curve", action="store_true",
                   help="--gpus N > 1: skip the strong-scaling mosaic over the N GPUs that rank 0 runs after the weak-scaling "
                        "headline (secondary.mosaic)")
    p.add_argument("--dry-run", action="store_true",
                   help="exercise the multi-rank timing harness without a GPU (every tile is a 0.2 ms sleep); "
                        "used by the CPU gloo test, never a measurement")
    return p.parse_args()


def build_batch(raster, side, tiles, rank):
    """B independent views on device-resident MemorySources, merged into one prebuilt fused graph."""
    from dask_geomodeling_amd.fuse import fuse_graph
    from dask_geomodeling_amd.raster.sources import generate_window

    request = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side,
                   projection="EPSG:28992")
    ndv = np.float32(F32MAX)
    graph, names, seeds = {}, [], []
    for i in range(tiles):
        sa, sb = 1000 * rank + 2 * i, 1000 * rank + 2 * i + 1
        da = generate_window((side, side), "f4", sa, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
        db = generate_window((side, side), "f4", sb, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
        A = raster.MemorySource(da, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
        B = raster.MemorySource(db, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
        view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
        graph, name = view.get_compute_graph(cached_compute_graph=graph, **request)
        names.append(name)
        seeds.append((sa, sb))
    return [(fuse_graph(graph, names), names)], seeds, request


def build_reclass(raster, side, tiles, rank):
    """configs[2]: Mask(Reclassify(int32 tile, 128-entry table), 1): one int32 program (LOOKUP with the Mask
    folded into its table), int32 in, uint8 out = 5 B/px; the tiles of a step form one merged graph, i.e.
    one launch."""
    from dask_geomodeling_amd.fuse import fuse_graph
    from dask_geomodeling_amd.raster.sources import generate_window

    table = [[k, (7 * k) % 1000] for k in range(0, 256, 2)]
    request = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side,
                   projection="EPSG:28992")
    graph, names = {}, []
    for i in range(tiles):
        d = generate_window((side, side), "i4", 2000 + 100 * rank + i, 0.0, 256.0, 2147483647, (50, 50), 0, 0, side, side)
        src = raster.MemorySource(d, 2147483647, "EPSG:28992", 1.0, (0.0, float(side)))
        view = raster.Mask(raster.Reclassify(src, table), 1)
        graph, name = view.get_compute_graph(cached_compute_graph=graph, **request)
        names.append(name)
    return [(fuse_graph(graph, names), names)]


def build_bilinear(raster, side, tiles, rank):
    """configs[3]: bilinear 2x upsample of a float32 side^2 tile -> (2 side)^2: 5 B per output px."""
    from dask_geomodeling_amd.raster.sources import generate_window

    request = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=2 * side, height=2 * side,
                   projection="EPSG:28992")
    jobs = []
    for i in range(tiles):
        d = generate_window((side, side), "f4", 3000 + 100 * rank + i, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
        src = raster.MemorySource(d, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(side)))
        view = raster.SpatialResample(src, 2, "bilinear")
        graph, name = view.get_compute_graph(**request)
        jobs.append((graph, [name]))
    return jobs


def build_tiler(raster, side, tile, rank):
    """The canonical chain under a RasterTiler over two RESIDENT float32 MemorySources of side^2 pixels:
    every tile request is a zero-copy window of the sources, the tiles run as one batched launch per
    16 and are pasted into the (side, side) result -- the whole drop-in path of a tiled view."""
    from dask_geomodeling_amd.fuse import fuse_graph
    from dask_geomodeling_amd.raster.sources import generate_window

    ndv = np.float32(F32MAX)
    da = generate_window((side, side), "f4", 5000 + 2 * rank, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
    db = generate_window((side, side), "f4", 5001 + 2 * rank, 0.0, 100.0, F32MAX, LATTICE, 0, 0, side, side)
    A = raster.MemorySource(da, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
    B = raster.MemorySource(db, ndv, "EPSG:28992", 1.0, (0.0, float(side)))
    view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    tiler = raster.RasterTiler(view, tile)
    request = dict(mode="vals", bbox=(0.0, 0.0, float(side), float(side)), width=side, height=side,
                   projection="EPSG:28992")
    graph, name = tiler.get_compute_graph(**request)
    return [(fuse_graph(graph, name), [name])]


MOSAIC_SIDE = 65536


def mosaic_view(raster, leaves="owned", mosaic_side=MOSAIC_SIDE):
    """configs[4]: 10 Blocks (Add, Multiply, Subtract, Divide, Greater, Clip, Classify, Reclassify, Mask
    under a RasterTiler) over four float32 leaves of a 65536^2 mosaic.  leaves: "owned" = procedural
    sources that keep every window they served resident, so a rank holds the leaf windows of ITS tiles
    only (64 GiB / N per GPU); "whole" = MemorySources holding the whole mosaic in HBM (4 x 16 GiB on
    every GPU; tile requests are zero-copy windows); "procedural" = every window regenerated per request."""
    from dask_geomodeling_amd.raster.sources import generate_window

    def leaf(seed):
        if leaves == "whole":
            data = generate_window((mosaic_side, mosaic_side), "f4", seed, 0.0, 100.0, F32MAX, LATTICE, 0, 0,
                                   mosaic_side, mosaic_side)
            return raster.MemorySource(data, np.float32(F32MAX), "EPSG:28992", 1.0, (0.0, float(mosaic_side)))
        return raster.SyntheticSource((mosaic_side, mosaic_side), "f4", seed, 0.0, 100.0, F32MAX, LATTICE,
                                      "EPSG:28992", 1.0, (0.0, float(mosaic_side)), resident=(leaves == "owned"))
    A, B, C, D = leaf(11), leaf(12), leaf(13), leaf(14)
    x = raster.Divide(raster.Subtract(raster.Multiply(raster.Add(A, B), 0.5), C), 2.0)
    x = raster.Clip(x, raster.Greater(D, 20.0))
    x = raster.Classify(x, [-50.0, -10.0, 0.0, 10.0, 50.0])
    x = raster.Reclassify(x, [[0, 10], [1, 11], [3, 13], [5, 15]])
    return raster.Mask(x, 1)


def build_mosaic(raster, tile, rank, world, leaves="owned"):
    """The tile sub-graphs of RasterTiler(view, tile) over the whole mosaic that this rank owns."""
    from dask_geomodeling_amd import shard
    from dask_geomodeling_amd.fuse import fuse_graph

    tiler = raster.RasterTiler(mosaic_view(raster, leaves), tile)
    request = dict(mode="vals", bbox=(0.0, 0.0, float(MOSAIC_SIDE), float(MOSAIC_SIDE)), width=MOSAIC_SIDE,
                   height=MOSAIC_SIDE, projection="EPSG:28992")
    graph, name = tiler.get_compute_graph(**request)
    _, tile_keys = shard.split_tiler_graph(graph, name)
    mine = shard.owned_tiles(len(tile_keys), rank, world)
    keys = [tile_keys[t] for t in mine]
    return [(fuse_graph(graph, keys), keys)], len(tile_keys)


def oracle_chain(a, b):
    from oracle import ref_numpy as O
    ndv = np.float32(F32MAX)
    kw = {"dtype": "float32", "fillvalue": F32MAX}
    ta, tb = {"values": a, "no_data_value": ndv}, {"values": b, "no_data_value": ndv}
    x = O.math_process("add", kw, ta, tb)
    x = O.math_process("multiply", kw, x, 2.5)
    x = O.math_process("subtract", kw, x, tb)
    x = O.math_process("add", kw, x, 1.5)
    return O.clip_process(x, ta)


def host_tile(seed, side):
    from oracle import synth
    return synth.generate_window((side, side), "f4", seed, 0.0, 100.0, np.float32(F32MAX), LATTICE, 0, 0, side, side)


def cpu_baseline(side, n_tiles, budget_s=12.0):
    """The oracle chain on host cores (the restatement of the reference's process() bodies; the reference
    itself cannot travel to the GPU box): (i) the synchronous leg -- tiles one after the other on one thread,
    what dask's synchronous scheduler does -- and (ii) the threaded leg -- passes of n_tiles tiles over a
    thread pool (NumPy releases the GIL; dask's threaded scheduler), until about budget_s seconds are spent.
    `value` is the threaded rate (the faster one)."""
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    workers = min(cores, n_tiles)
    distinct = [(host_tile(2 * i, side), host_tile(2 * i + 1, side)) for i in range(min(n_tiles, 2))]
    tiles = [distinct[i % len(distinct)] for i in range(n_tiles)]  # timing does not depend on the values
    oracle_chain(*[t[:, :256, :256] for t in tiles[0]])  # warm
    t0, sync_tiles = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s / 4 and sync_tiles < 8:
        oracle_chain(*tiles[sync_tiles % n_tiles])
        sync_tiles += 1
    sync_rate = sync_tiles * side * side / (time.perf_counter() - t0) / 1e6
    passes, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(workers) as pool:
        while True:
            list(pool.map(lambda ab: oracle_chain(*ab)["values"][0, 0, 0], tiles))
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s * 3 / 4 or passes >= 64:
                break
    return {"value": round(passes * n_tiles * side * side / dt / 1e6, 2), "unit": "Mpix/s", "cores": workers,
            "kind": "port", "synchronous": {"value": round(sync_rate, 2), "unit": "Mpix/s", "cores": 1},
            "sample": "oracle/ref_numpy.py chain (NumPy %s): threaded leg %d passes over %d tiles of %dx%d f32 on a "
                      "%d-thread pool, %.1f s; synchronous leg %d tiles on 1 thread"
                      % (np.__version__, passes, n_tiles, side, side, workers, dt, sync_tiles)}


def kernel_source_hash():
    """Identity of the device code a measurement was taken with (kernel sources + the program compiler)."""
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "dask_geomodeling_amd")
    files = sorted(os.path.join(base, "csrc", f) for f in os.listdir(os.path.join(base, "csrc"))
                   if f.endswith((".hip", ".h")))
    for path in files + [os.path.join(base, "hip", "program.py"), os.path.join(base, "fuse.py")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, tiles_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this very
    command AND this very device code (profiles/*_pmc*.json, written by tools/collect_profiles.py with the
    hash above): a stale file -- another kernel version, another workload -- yields (None, None)."""
    folder = os.path.join(ROOT, "profiles")
    if not os.path.isdir(folder):
        return None, None
    want = kernel_source_hash()
    for name in sorted(os.listdir(folder), reverse=True):
        if "_pmc" not in name or not name.endswith(".json"):
            continue
        try:
            with open(os.path.join(folder, name)) as f:
                rec = json.load(f)
            if (rec.get("workload") == workload and rec.get("tiles_per_launch") == tiles_per_launch and
                    rec.get("kernel_source_hash") == want):
                return rec["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def end_to_end(raster, runtime, side, budget_s=6.0):
    """What a caller of the Block API sees (rank 0, N = 1 only; bounded): Mpix/s and host microseconds per
    request of ``view.get_data(**request)`` -- graph build / plan lookup + source windows + launch + (with
    materialize) the device-to-host copy -- for single 4096^2 tile requests at ever-changing bounding boxes of a
    resident 16384^2 source pair, and for ``RasterTiler(view, 4096).get_data`` over the whole 16384^2."""
    from dask_geomodeling_amd import config as gmconfig
    from dask_geomodeling_amd.raster.sources import generate_window
    big = 4 * side
    ndv = np.float32(F32MAX)
    da = generate_window((big, big), "f4", 7001, 0.0, 100.0, F32MAX, LATTICE, 0, 0, big, big)
    db = generate_window((big, big), "f4", 7002, 0.0, 100.0, F32MAX, LATTICE, 0, 0, big, big)
    A = raster.MemorySource(da, ndv, "EPSG:28992", 1.0, (0.0, float(big)))
    B = raster.MemorySource(db, ndv, "EPSG:28992", 1.0, (0.0, float(big)))
    view = raster.Clip(raster.Add(raster.Subtract(raster.Multiply(raster.Add(A, B), 2.5), B), 1.5), A)
    boxes = [(float(i * side), float(j * side), float((i + 1) * side), float((j + 1) * side))
             for i in range(4) for j in range(4)]
    req = dict(mode="vals", width=side, height=side, projection="EPSG:28992")

    def rate(fn, px, min_calls, share):
        fn()
        runtime.sync()
        calls, t0 = 0, time.perf_counter()
        while calls < min_calls or time.perf_counter() - t0 < budget_s * share:
            fn()
            calls += 1
            if calls >= 2000:
                break
        runtime.sync()
        dt = time.perf_counter() - t0
        return round(px * calls / dt / 1e6, 1), round(dt / calls * 1e6, 1), calls

    out = {"config": "view = the canonical chain over two resident %dx%d f32 MemorySources; requests = %dx%d tiles "
                     "at 16 different bounding boxes in turn" % (big, big, side, side)}
    state = {"k": 0}

    def one_tile():
        state["k"] += 1
        return view.get_data(bbox=boxes[state["k"] % 16], **req)

    # plans: no graph at all (plan.py; steps aside when a scheduler is configured in dask); graph_per_request: a real
    # graph for the configured scheduler, made from the view's template (template.py); graph_from_scratch: the
    # reference's way, get_compute_graph + the fusion pass for every request
    modes = (("plans", True, True, 0.2), ("graph_per_request", False, True, 0.2), ("graph_from_scratch", False, False, 0.1))
    for label, plans, templates, share in modes:
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": plans,
                                  "geomodeling.templates": templates}):
            v, us, n = rate(one_tile, side * side, 32, share)
        out["get_data_tile_device_" + label] = {"Mpix/s": v, "us_per_request": us, "requests": n}
    with gmconfig.config.set({"geomodeling.materialize": True}):
        v, us, n = rate(one_tile, side * side, 4, 0.15)
    out["get_data_tile_to_host"] = {"Mpix/s": v, "us_per_request": us, "requests": n,
                                    "note": "includes the 64 MiB device-to-host copy of every result (PCIe)"}
    # a stencil Block over the resident source (row N3): the window is read in place, the margin planned per request
    shade = raster.HillShade(A)
    inner = [(b[0] + 64.0, b[1] + 64.0, b[2] - 64.0, b[3] - 64.0) for b in boxes]
    shade_req = dict(req, width=side - 128, height=side - 128)

    def one_shade():
        state["k"] += 1
        return shade.get_data(bbox=inner[state["k"] % 16], **shade_req)

    for label, plans in (("plans", True), ("graph_per_request", False)):
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": plans}):
            v, us, n = rate(one_shade, (side - 128) ** 2, 32, 0.08)
        out["hillshade_get_data_tile_device_" + label] = {"Mpix/s": v, "us_per_request": us, "requests": n}
    tiler = raster.RasterTiler(view, side)
    whole = dict(req, bbox=(0.0, 0.0, float(big), float(big)), width=big, height=big)
    for label, plans, templates, share in modes:
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": plans,
                                  "geomodeling.templates": templates}):
            v, us, n = rate(lambda: tiler.get_data(**whole), big * big, 4, share * 0.75)
        out["tiler_get_data_device_" + label] = {"Mpix/s": v, "us_per_request": us, "requests": n, "tiles": 16}
    # mixed views (round 4; tools/e2e_views.py is the long form): microseconds per request of the inner boxes on the plan
    # path next to the bytes the view's kernels move / 6 TB/s -- a view far above that has a slow seam (MEASUREMENTS 6.7)
    from datetime import datetime, timedelta
    frames = 8
    dt = generate_window((frames * side, side), "f4", 7003, 0.0, 100.0, F32MAX, LATTICE, 0, 0, frames * side, side)
    T = raster.MemorySource(dt.reshape((frames, side, side)), ndv, "EPSG:28992", 1.0, (0.0, float(side)),
                            time_first=datetime(2020, 1, 1), time_delta=timedelta(hours=6))
    days = dict(req, bbox=(0.0, 0.0, float(side), float(side)), start=datetime(2020, 1, 1), stop=datetime(2020, 1, 3))
    px = (side - 128) ** 2
    mixed = [("HillShade(A + B) * 0.5", raster.HillShade(A + B) * 0.5, None, (12 + 5 + 5) * px),
             ("FillNoData(A, B)", raster.FillNoData(A, B), None, 12 * px),
             ("Group(A, B)", raster.Group(A, B), None, 12 * px),
             ("Max(A, B, A * 2)", raster.Max(A, B, A * 2), None, 12 * px),
             ("TemporalAggregate(T, 'D', 'mean'), 8 frames", raster.TemporalAggregate(T, "D", "mean"), days, (frames + 2) * 4 * side * side),
             ("Resample(T, '3h', 'nearest'), 8 -> 17 frames", raster.Resample(T, "3h", "nearest"), days, 2 * 17 * 4 * side * side),
             ("RasterTiler(T * 2, 1024), 8 frames in 16 tiles", raster.RasterTiler(T * 2, 1024), days, frames * 8 * side * side),
             ("T * 2 at a point, 8 frames", T * 2, dict(days, bbox=(2000.5, 2000.5, 2000.5, 2000.5), width=1, height=1), 64)]
    views = {}
    from dask_geomodeling_amd.hip import program as P
    for label, mixed_view, fixed, nbytes in mixed:
        def one_mixed():
            state["k"] += 1
            if fixed is not None:
                return mixed_view.get_data(**fixed)
            return mixed_view.get_data(bbox=inner[state["k"] % 16], **shade_req)
        with gmconfig.config.set({"geomodeling.materialize": False, "geomodeling.plans": True}):
            for _ in range(4):
                one_mixed()
            if P.spec_pending():
                P.spec_wait()
            _, us, n = rate(one_mixed, px, 16, 0.04)
        views[label] = {"us_per_request": us, "kernels_at_6TBps_us": round(nbytes / 6e6, 1), "requests": n}
    out["views"] = views
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes of this very
    command (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one per GPU), relay rank 0's
    JSON line and return non-zero if any rank failed.  The parent never touches the GPU (no HIP call, no
    library load) and never exec()s: the ranks are ordinary children.  All children are watched: as soon
    as one of them exits non-zero the others are terminated (they would otherwise sit in a barrier until
    the process group's timeout) and the command fails within seconds, saying which rank died."""
    import subprocess
    import threading
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, cwd=os.getcwd()))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # (rank 0's pipe is drained while the children are polled)
    bad = []
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or all(c is not None for c in codes):
            break
        time.sleep(0.05)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if bad:
        sys.stderr.write("bench: rank(s) failed (rank, exit code): %s -- the other ranks were stopped\n" % bad)
        return 1
    return 0


# the dominant kernel of every workload: tiles per launch, algorithmic bytes per OUTPUT pixel (SURVEY 8d / DESIGN
# section 4), arithmetic type, and its name when the program is interpreted
MODEL = {"chain5": (16, 12, "f32", "gmvm::vm_kernel<float, 2, 2, 1, 0>"),
         "reclass": (16, 5, "i32", "lookup32_kernel_w4<1>"),
         "bilinear": (1, 5, "f64", "bilinear2x_kernel<16, 512, 256, false, true, false>"),
         "mosaic": (16, 17, "f32", "gmvm::vm_kernel<float, 2, 4, 2, 2>"),
         "tiler": (16, 12, "f32", "gmvm::vm_kernel<float, 2, 2, 1, 0>")}
DEFAULTS = {"chain5": (4096, 16), "reclass": (8192, 4), "bilinear": (16384, 1), "mosaic": (4096, 0), "tiler": (16384, 1)}


def warm_up(step, warmup, runtime, P, min_busy_s=0.2):
    """W untimed warm-up steps -- and, when those are shorter than that, keep stepping until the GPU has been busy for
    200 ms, so that the clocks have left their idle state before the timed region (a 20-step timed region is only ~10 ms
    long; hot programs also get their specialised kernels here: csrc/gm_spec.hip, ~1 s of compilation on the first step
    that finds a program hot) -- then wait for kernels still being compiled and step until they have taken over."""
    done, t_busy = 0, None
    while done < warmup or (min_busy_s and (t_busy is None or time.perf_counter() - t_busy < min_busy_s)):
        if t_busy is None and done >= 1:
            t_busy = time.perf_counter()  # the first step may have compiled: the busy period starts after it
        step()
        done += 1
        if done % 8 == 0:
            runtime.sync()
    if P is not None and P.spec_pending():
        P.spec_wait()
        for _ in range(3):
            step()


def secondary_line(name, raster, runtime, P, get_sync, steps, warmup):
    """One of BASELINE.json's other GPU configs, clocked in the same process as the headline, the same way: `steps` steps
    after a warm-up, HIP events on the compute stream around the timed region, no host sync in between."""
    side, tiles = DEFAULTS[name]
    jobs = build_reclass(raster, side, tiles, 0) if name == "reclass" else build_bilinear(raster, side, tiles, 0)
    runtime.sync()

    def step():
        for graph, names in jobs:
            get_sync(graph, names)

    warm_up(step, warmup, runtime, P)
    runtime.sync()
    spec_before = P.spec_stats()["launches"]
    e0, e1 = runtime.Event(), runtime.Event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    runtime.sync()
    elapsed = time.perf_counter() - t0
    spec_launches = P.spec_stats()["launches"] - spec_before
    out_side = 2 * side if name == "bilinear" else side
    per_launch, bytes_px, dtype, kernel = MODEL[name]
    tiles_per_launch = min(tiles, per_launch)
    launches = (tiles + tiles_per_launch - 1) // tiles_per_launch
    if spec_launches >= steps * launches:
        kernel = P.spec_last_kernel() + " (specialised per program by csrc/gm_spec.hip)"
    launch_us = e0.elapsed_ms(e1) / (steps * launches) * 1e3
    launch_bytes = bytes_px * out_side * out_side * tiles_per_launch
    traffic, traffic_src = pmc_traffic(name, tiles_per_launch)
    del jobs
    return {"workload": {"reclass": "BASELINE configs[2]: Mask(Reclassify(int32 tile, 128 pairs), 1) on %d tiles of %dx%d in one launch"
                                     % (tiles, side, side),
                         "bilinear": "BASELINE configs[3]: bilinear 2x upsample %dx%d -> %dx%d f32" % (side, side, out_side, out_side)}[name],
            "value": round(tiles * out_side * out_side * steps / elapsed / 1e6, 1), "unit": "Mpix/s", "steps": steps,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "avg_launch_us": round(launch_us, 2),
            "wall_us_per_launch": round(elapsed / (steps * launches) * 1e6, 2), "launches_per_step": launches,
            "dtype": dtype, "bytes_per_px": bytes_px, "algorithmic_bytes_per_launch": launch_bytes,
            "achieved_GBps": round(launch_bytes / launch_us / 1e3, 1), "frac": round(launch_bytes / launch_us / 1e3 / HBM_PEAK_GBS, 4),
            "kernel": kernel, "traffic": traffic, "traffic_source": traffic_src}


def mosaic_through_the_scheduler(devices, side, steps, warmup, leaves, e2e=True, mosaic_side=MOSAIC_SIDE):
    """BASELINE configs[4] through the PRODUCT's one-worker-per-GPU scheduler (dask_geomodeling_amd/shard.py): this process
    only deals -- RasterTiler(view, side).get_data over the whole 65536^2 box with geomodeling.devices = `devices`, results left
    on the GPUs that computed them --, every worker process fuses and runs the tiles t % G it owns and keeps the leaf windows
    of those tiles resident.  A step = one such request; it returns when every worker's stream is empty.  -> result fields.
    STRONG scaling: the mosaic is the same whatever G is.  Besides the rate the line says what the host side costs per step:
    `parent_us_per_step` (wall time of the parent's get_data call: two small messages per worker, sent and collected by one
    thread; includes waiting for the slowest worker's reply), `parent_cpu_us_per_step` (the parent's own CPU time in it),
    `worker_host_us_per_step` (per worker: wall time between a request's arrival and its reply -- leaf windows, arguments,
    launches queued) and `device_ms_per_step` per worker (HIP events on its compute stream): a step is device-bound as long
    as parent + slowest worker host time stay well under the slowest device time."""
    from dask_geomodeling_amd import raster, shard
    G = len(devices)
    tiler = raster.RasterTiler(mosaic_view(raster, leaves, mosaic_side), side)
    whole = dict(mode="vals", bbox=(0.0, 0.0, float(mosaic_side), float(mosaic_side)), width=mosaic_side,
                 height=mosaic_side, projection="EPSG:28992")
    pool = shard.pool_for(devices)

    def step():
        # (returns when every worker has queued its launches, like a get_data that leaves its result on the device in the
        #  single-process path; the handle is dropped: the workers free the tiles)
        shard.get_data(tiler, whole, devices, materialize=False)

    everybody = list(range(G))
    t_busy, done = None, 0
    while done < warmup or t_busy is None or time.perf_counter() - t_busy < 0.3:
        if t_busy is None and done >= 1:
            pool.sync()
            t_busy = time.perf_counter()
        step()
        done += 1
    pool.request({w: ("spec_wait",) for w in everybody}, what="waiting for specialised kernels")
    for _ in range(3):
        step()
    pool.sync()
    before = pool.request({w: ("spec_wait",) for w in everybody}, what="kernel statistics")
    pool.request({w: ("host_stats", True) for w in everybody}, what="host-time counters")
    pool.request({w: ("mark",) for w in everybody}, what="timing marks")   # a HIP event on every worker's compute stream
    in_step = in_step_cpu = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        t1, c1 = time.perf_counter(), time.thread_time()
        step()
        in_step += time.perf_counter() - t1
        in_step_cpu += time.thread_time() - c1
    pool.sync()
    elapsed = time.perf_counter() - t0
    per_worker_ms = pool.request({w: ("elapsed",) for w in everybody}, what="timing marks")
    device_ms = max(per_worker_ms.values())
    host = pool.request({w: ("host_stats",) for w in everybody}, what="host-time counters")
    spec = pool.request({w: ("spec_wait",) for w in everybody}, what="kernel statistics")
    n_tiles = (mosaic_side // side) ** 2
    per_worker = (n_tiles + G - 1) // G
    per_launch, bytes_px, dtype, kernel = MODEL["mosaic"]
    tiles_per_launch = min(per_worker, per_launch)
    launches = (per_worker + tiles_per_launch - 1) // tiles_per_launch   # of the busiest worker, per step
    if all(spec[w]["launches"] - before[w]["launches"] >= steps * launches for w in range(G)):
        kernel = spec[0]["kernel"] + " (specialised per program by csrc/gm_spec.hip)"
    launch_us = device_ms / (steps * launches) * 1e3
    launch_bytes = bytes_px * side * side * tiles_per_launch
    traffic, traffic_src = pmc_traffic("mosaic", tiles_per_launch)
    out = {"workload": "BASELINE configs[4]: 10-Block view over the %d^2 mosaic, %d tiles of %dx%d dealt t %% %d over %d worker "
                       "process(es) by dask_geomodeling_amd.shard (leaves '%s'), results left on the devices"
                       % (mosaic_side, n_tiles, side, side, G, G, leaves),
           "value": round(mosaic_side * mosaic_side * steps / elapsed / 1e6, 1), "unit": "Mpix/s", "steps": steps,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "device_ms_per_step": round(device_ms / steps, 4),
           "avg_launch_us": round(launch_us, 2), "launches_per_step_per_worker": launches, "tiles_per_worker": per_worker,
           "dtype": dtype, "bytes_per_px": bytes_px, "algorithmic_bytes_per_launch": launch_bytes,
           "achieved_GBps": round(launch_bytes / launch_us / 1e3, 1), "frac": round(launch_bytes / launch_us / 1e3 / HBM_PEAK_GBS, 4),
           "kernel": kernel, "traffic": traffic, "traffic_source": traffic_src, "workers": G,
           "worker_devices": ["%d:%s" % (d, n) for d, n in zip(devices, pool.device_names)],
           "scaling": "strong", "mosaic": [mosaic_side, mosaic_side], "tiles": n_tiles,
           # (workers that SHARE a device -- a one-GPU box asked for several -- take turns on it: their per-worker device time
           #  is the device's, `frac` then says nothing about the kernel)
           "devices_distinct": len(set(devices)) == len(devices),
           "per_worker_device_ms_per_step": [round(per_worker_ms[w] / steps, 4) for w in everybody],
           "parent_us_per_step": round(in_step / steps * 1e6, 1), "parent_cpu_us_per_step": round(in_step_cpu / steps * 1e6, 1),
           "worker_host_us_per_step": [round(host[w]["wall_s"] / max(host[w]["runs"], 1) * 1e6, 1) for w in everybody],
           "worker_host_cpu_us_per_step": [round(host[w]["cpu_s"] / max(host[w]["runs"], 1) * 1e6, 1) for w in everybody],
           "worker_cpus": [None if host[w]["cpus"] is None else len(host[w]["cpus"]) for w in everybody],
           "host_over_device": round((in_step / steps * 1e3) / max(device_ms / steps, 1e-9), 4)}
    if e2e:
        # what an ndarray-returning caller gets: the 4 GiB uint8 mosaic assembled on the host, every worker copying its tiles
        # device-to-host straight to their place of a shared-memory mosaic it has pinned (the first request pins)
        try:
            timings = []
            for _ in range(2):
                t1 = time.perf_counter()
                got = shard.get_data(tiler, whole, devices, materialize=True)
                timings.append(round((time.perf_counter() - t1) * 1e3, 1))
                shape = tuple(got["values"].shape)
                del got
            out["to_host_ms"] = {"first_request": timings[0], "second_request": timings[1], "shape": shape,
                                 "GBps_second": round(mosaic_side ** 2 / timings[1] / 1e6, 1)}
        except Exception as e:  # noqa: BLE001 -- e.g. a box whose /dev/shm cannot hold 4 GiB: reported, the timed line stands
            out["to_host_ms"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return out


def dry_mosaic_line(world, steps):
    """Harness test only (--dry-run): the shape of secondary.mosaic at N ranks without a GPU -- a step is a 2 ms sleep / N."""
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.002 / world)
    elapsed = time.perf_counter() - t0
    return {"workload": "DRY RUN (no GPU work, harness test only)", "value": round(MOSAIC_SIDE ** 2 * steps / elapsed / 1e6, 1),
            "unit": "Mpix/s", "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": "strong",
            "workers": world, "worker_devices": ["%d:dry" % g for g in range(world)],
            "per_worker_device_ms_per_step": [0.0] * world, "parent_us_per_step": 0.0,
            "worker_host_us_per_step": [0.0] * world}


def main_mosaic(args):
    """`--workload mosaic`: the product's scheduler is what runs (and what is timed): this process is the parent of one worker
    per GPU and never touches a GPU itself.  Under a launcher (WORLD_SIZE set: one rank per GPU already) rank 0 is that parent
    and the other ranks have nothing to do."""
    if int(os.environ.get("RANK", "0")) != 0:
        return 0
    if args.dry_run:
        raise SystemExit("bench: --workload mosaic runs the real scheduler; its CPU coverage is tests/test_shard_cpu.py")
    side = args.side or 4096
    leaves = "procedural" if args.procedural else args.leaves
    from dask_geomodeling_amd import shard
    from dask_geomodeling_amd.hip import runtime
    visible = runtime.device_count()  # (counts; this process never initialises a GPU)
    if visible <= 0:
        raise SystemExit("bench: no HIP device visible")
    devices = tuple(g % visible for g in range(max(args.gpus, 1)))  # (more workers than GPUs share them: the 1-GPU test box)
    if args.no_affinity:
        from dask_geomodeling_amd import config as gm_config
        gm_config.config.set({"geomodeling.worker-affinity": False})
    try:
        line = mosaic_through_the_scheduler(devices, side, args.steps, args.warmup, leaves, e2e=not args.no_e2e,
                                            mosaic_side=args.mosaic_side)
    finally:
        shard.shutdown()
    out = {"metric": "Mpix/s, workload mosaic", "value": line["value"], "unit": "Mpix/s", "n_gpus": len(devices),
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": line["ms_per_step"], "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": line["dtype"], "data": "synthetic",
           "config": {"workload": line["workload"], "name": "mosaic", "tile": [side, side],
                      "tiles_per_gpu": line["tiles_per_worker"], "tiles_per_launch": min(line["tiles_per_worker"], 16),
                      "bytes_per_px": line["bytes_per_px"],
                      "parallelism": "dask_geomodeling_amd.shard: tiles dealt t % G, one worker process per GPU, no collective"},
           "roofline": {"bound": "hbm", "achieved": line["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": line["frac"], "traffic": line["traffic"], "traffic_source": line["traffic_source"],
                        "kernel": line["kernel"], "algorithmic_bytes_per_launch": line["algorithmic_bytes_per_launch"],
                        "avg_launch_us": line["avg_launch_us"], "launches_per_step": line["launches_per_step_per_worker"],
                        "kernel_source_hash": kernel_source_hash()},
           "pct_of_hbm_roofline": round(100.0 * line["value"] * 1e6 * line["bytes_per_px"] / len(devices) / (HBM_PEAK_GBS * 1e9), 2),
           "device_ms_per_step": line["device_ms_per_step"], "worker_devices": line["worker_devices"],
           "host": {k: line[k] for k in ("parent_us_per_step", "parent_cpu_us_per_step", "worker_host_us_per_step",
                                         "worker_host_cpu_us_per_step", "worker_cpus", "per_worker_device_ms_per_step",
                                         "host_over_device")}}
    if "to_host_ms" in line:
        out["end_to_end"] = {"sharded_get_data_to_host": line["to_host_ms"]}
    print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse_args()
    if args.workload == "mosaic" and not args.in_process:
        raise SystemExit(main_mosaic(args))
    if args.workload == "mosaic" and args.gpus > 1:
        raise SystemExit("bench: --in-process is the one-GPU profiling variant of the mosaic workload")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))

    dist = None
    if world > 1:
        import torch.distributed as dist  # timing barrier / max-reduce only (no data-path collective)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its peers on stdout ("[Gloo] Rank 0 is connected to ..."): stdout belongs to the one
        # JSON line, so file descriptor 1 points at stderr while the process group is set up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            from datetime import timedelta
            # (a rank that dies leaves the others in a collective: two minutes, not gloo's default thirty)
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    if os.environ.get("GM_BENCH_FAIL_RANK") == str(rank):  # harness test: this rank dies before its first barrier
        raise SystemExit("bench: rank %d fails on purpose (GM_BENCH_FAIL_RANK)" % rank)

    side = args.side or DEFAULTS[args.workload][0]
    tiles = args.tiles or DEFAULTS[args.workload][1]
    if args.workload != "chain5":
        args.no_check = True
    P = None
    if args.dry_run:
        class _NoDevice:  # harness test only: no device, no kernels, no numbers worth reading
            class Event:
                def __init__(self):
                    self.t = 0.0

                def record(self):
                    self.t = time.perf_counter()

                def synchronize(self):
                    pass

                def elapsed_ms(self, later):
                    return (later.t - self.t) * 1e3

            @staticmethod
            def sync():
                pass

        runtime, jobs, seeds = _NoDevice, [None] * tiles, []
        args.no_check = True

        def step():
            for _ in jobs:
                time.sleep(0.0002)
    else:
        from dask_geomodeling_amd import raster
        from dask_geomodeling_amd.core.graphs import get_sync
        from dask_geomodeling_amd.hip import program as P
        from dask_geomodeling_amd.hip import runtime

        runtime.require_device()
        runtime.set_device(local_rank % runtime.device_count())
        seeds = []
        if args.workload == "chain5":
            jobs, seeds, _ = build_batch(raster, side, tiles, rank)
        elif args.workload == "reclass":
            jobs = build_reclass(raster, side, tiles, rank)
        elif args.workload == "bilinear":
            jobs = build_bilinear(raster, side, tiles, rank)
        elif args.workload == "tiler":
            jobs = build_tiler(raster, side, 4096, rank)
        else:  # mosaic --in-process
            jobs, _ = build_mosaic(raster, side, 0, 1, leaves="procedural" if args.procedural else args.leaves)
            tiles = len(jobs[0][1])
        runtime.sync()

        def step():
            for graph, names in jobs:
                get_sync(graph, names)  # result DeviceArrays are dropped -> buffers return to the pool

    def barrier():
        runtime.sync()
        if dist is not None:
            dist.barrier()
        runtime.sync()

    warm_up(step, args.warmup, runtime, P, min_busy_s=0.0 if args.dry_run else 0.2)
    barrier()
    spec_before = P.spec_stats()["launches"] if P is not None else 0
    e0, e1 = runtime.Event(), runtime.Event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    e1.synchronize()
    spec_launches = (P.spec_stats()["launches"] - spec_before) if P is not None else 0

    out_side = 2 * side if args.workload == "bilinear" else side
    if args.workload == "tiler":
        tile_px, n_tiles_step = 4096 * 4096, (side // 4096) ** 2
    else:
        tile_px, n_tiles_step = out_side * out_side, tiles
    model = MODEL[args.workload]
    tiles_per_launch = min(n_tiles_step, model[0])
    launches_per_step = (n_tiles_step + tiles_per_launch - 1) // tiles_per_launch
    bytes_px, dtype, kernel = model[1], model[2], model[3]
    if spec_launches >= args.steps * launches_per_step and P is not None:
        kernel = P.spec_last_kernel() + " (specialised per program by csrc/gm_spec.hip)"
    kernel_ms = e0.elapsed_ms(e1) / (args.steps * launches_per_step)  # average launch duration (HIP events, compute stream)
    # kernel-only duration, outside the timed region: the same steps once more with every specialised launch bracketed
    # by events that the kernel itself stamps at its begin and end (no submission gaps in the figure)
    kernel_only_us = None
    if P is not None and spec_launches >= args.steps * launches_per_step:
        P.spec_timing(True)
        for _ in range(min(args.steps, 10)):
            step()
        runtime.sync()
        P.spec_timing(False)
        total_us, timed = P.spec_timing_read()
        if timed:
            kernel_only_us = total_us / timed

    if not args.no_check and rank == 0:
        # parity of what was just timed: first and last tile of the batch against the oracle
        graph, names = jobs[0]
        results = get_sync(graph, names)
        for t in sorted({0, len(names) - 1}):
            got = np.asarray(results[t]["values"])
            want = oracle_chain(host_tile(seeds[t][0], side), host_tile(seeds[t][1], side))["values"]
            if not np.array_equal(got.view("u4"), want.view("u4")):
                raise SystemExit("bench: fused chain differs from the oracle on tile %d" % t)
        del results

    own_elapsed, per_rank = elapsed, None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own time, so that a straggler shows in the line instead of hiding in the maximum
        mine = torch.tensor([float(rank), own_elapsed], dtype=torch.float64)
        gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = sorted((int(g[0].item()), float(g[1].item())) for g in gathered)

    if rank == 0:
        px_per_step, scaling = world * n_tiles_step * tile_px, "weak"
        value = px_per_step * args.steps / elapsed / 1e6
        launch_bytes = bytes_px * tile_px * tiles_per_launch
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, tiles_per_launch)
        workloads = {
            "chain5": "fused chain Clip(Add(Subtract(Multiply(Add(A,B),2.5),B),1.5),A) on %dx%d f32 MemorySource tiles, "
                      "%d tile pairs per GPU per step, %d tiles per fused launch" % (side, side, tiles, tiles_per_launch),
            "reclass": "Mask(Reclassify(int32 tile, 128 pairs), 1) on %dx%d tiles (BASELINE configs[2]), %d tiles per "
                       "GPU per step in one launch" % (side, side, tiles),
            "bilinear": "bilinear 2x upsample %dx%d -> %dx%d f32 (BASELINE configs[3])" % (side, side, out_side, out_side),
            "tiler": "canonical chain under RasterTiler(4096) over two resident %dx%d sources: zero-copy windows in, "
                     "mosaic written in place" % (side, side),
            "mosaic": "10-Block view over the 65536^2 mosaic (BASELINE configs[4]), the %d tile programs of %dx%d run IN THIS PROCESS "
                      "(profiling variant: what the one worker of the scheduler does)" % (tiles, side, side)}
        out = {
            "metric": "Mpix/s on 4096x4096 f32 tiles, 5-op elemwise chain; % of HBM roofline" if args.workload == "chain5"
                      else "Mpix/s, workload %s" % args.workload,
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype,
            "data": "synthetic" if not args.dry_run else "DRY RUN (no GPU work, harness test only)",
            "config": {"workload": workloads[args.workload], "name": args.workload, "tile": [side, side],
                       "tiles_per_gpu": tiles, "tiles_per_launch": tiles_per_launch, "bytes_per_px": bytes_px,
                       "parallelism": "tiles sharded 1 rank per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "kernel": kernel,
                         "algorithmic_bytes_per_launch": launch_bytes, "avg_launch_us": round(kernel_ms * 1e3, 2),
                         "kernel_only_us": None if kernel_only_us is None else round(kernel_only_us, 2),
                         "wall_us_per_launch": round(elapsed / (args.steps * launches_per_step) * 1e6, 2),
                         "launches_per_step": launches_per_step, "kernel_source_hash": kernel_source_hash()},
            "pct_of_hbm_roofline": round(100.0 * value * 1e6 * bytes_px / world / (HBM_PEAK_GBS * 1e9), 2),
        }
        if per_rank is not None:
            rank_px = px_per_step / world
            out["ranks_reported"] = len(per_rank)
            out["per_rank"] = [{"rank": r, "ms_per_step": round(t / args.steps * 1e3, 4),
                                "Mpix/s": None if rank_px is None else round(rank_px * args.steps / t / 1e6, 1)}
                               for r, t in per_rank]
        cpu_tiles = args.cpu_tiles
        if cpu_tiles < 0:
            cpu_tiles = min(max(os.cpu_count() or 1, 4), 32) if world == 1 else 0
        if args.workload == "chain5" and world == 1 and not args.dry_run:
            if cpu_tiles > 0:
                out["cpu_baseline"] = cpu_baseline(side, cpu_tiles)
            if not args.no_e2e:
                del jobs
                out["end_to_end"] = end_to_end(raster, runtime, side)
        if args.workload == "chain5" and world == 1 and not args.dry_run and not args.no_secondary:
            # BASELINE.json's other GPU configs, clocked in this very run (20 steps each, the headline's way): configs[2]
            # and [3] in this process, configs[4] through the product's one-worker-per-GPU scheduler (one worker: this GPU)
            try:
                del jobs
            except NameError:
                pass
            runtime.sync()
            runtime.pool_trim()
            secondary = {}
            for name in ("reclass", "bilinear", "mosaic"):
                # (a secondary workload that fails -- no room in HBM or in /dev/shm on some box -- is reported as such: the
                #  headline line above was measured and stands)
                try:
                    if name == "mosaic":
                        from dask_geomodeling_amd import shard
                        try:
                            secondary[name] = mosaic_through_the_scheduler((0,), 4096, args.secondary_steps, 3, "owned",
                                                                           e2e=not args.no_e2e)
                        finally:
                            shard.shutdown()
                    else:
                        secondary[name] = secondary_line(name, raster, runtime, P, get_sync, args.secondary_steps, 3)
                except Exception as e:  # noqa: BLE001
                    secondary[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                runtime.sync()
                runtime.pool_trim()
            out["secondary"] = secondary
    if world > 1 and not args.no_mosaic_curve and args.workload == "chain5":
        # The curve north_star asks for (BASELINE configs[4]: "tile tasks sharded 1/2/4/8 GPUs"): the headline above is WEAK
        # scaling (every rank its own 16 tile pairs).  STRONG scaling of the one 65536^2 mosaic runs through the product's
        # scheduler, which wants to be the parent of one worker per GPU: every rank gives its GPU's memory back, the ranks
        # meet, the process group is dissolved (ranks other than 0 leave), and rank 0 -- now an ordinary parent -- deals the
        # mosaic over the N GPUs and merges the result into its line as secondary.mosaic.
        try:
            del jobs
        except NameError:
            pass
        if not args.dry_run:
            runtime.sync()
            runtime.pool_trim()
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        if rank == 0:
            try:
                if args.dry_run:
                    line = dry_mosaic_line(world, args.secondary_steps)
                else:
                    from dask_geomodeling_amd import shard
                    visible = runtime.device_count()
                    try:
                        line = mosaic_through_the_scheduler(tuple(g % visible for g in range(world)), 4096, args.secondary_steps,
                                                            3, "procedural" if args.procedural else args.leaves, e2e=not args.no_e2e)
                    finally:
                        shard.shutdown()
            except Exception as e:  # noqa: BLE001 -- the headline was measured and stands
                line = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            out["secondary"] = {"mosaic": line}

    if rank == 0:
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
