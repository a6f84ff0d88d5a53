"""Round 6: a soak of the one-worker-per-GPU scheduler (GPU box) -- thousands of requests through G workers, results left on
the devices and handed back as ndarrays in turn, with the memory of the parent and of every worker (resident set, the device
pool's bytes, the shared-memory segments under /dev/shm) read at a few points on the way: whatever grows with the number of
requests leaks.  usage: shard_soak.py [G=4] [requests=3000] [mosaic side=4096] [tile side=256]"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from dask_geomodeling_amd import raster, shard  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
MOSAIC = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
SIDE = int(sys.argv[4]) if len(sys.argv) > 4 else 256


def rss_kb(pid):
    try:
        with open("/proc/%d/status" % pid) as f:
            for line in f:
                if line.startswith("VmRSS:"):
                    return int(line.split()[1])
    except OSError:
        pass
    return -1


def main():
    devices = tuple([0] * G)
    tiler = raster.RasterTiler(bench.mosaic_view(raster, "procedural", MOSAIC), SIDE)
    whole = dict(mode="vals", bbox=(0.0, 0.0, float(MOSAIC), float(MOSAIC)), width=MOSAIC, height=MOSAIC, projection="EPSG:28992")
    inner = dict(whole, bbox=(float(SIDE), float(SIDE), float(MOSAIC - SIDE), float(MOSAIC - SIDE)), width=MOSAIC - 2 * SIDE,
                 height=MOSAIC - 2 * SIDE)
    pool = shard.pool_for(devices)
    pids = [pool.request({w: ("host_stats",) for w in range(G)}, what="host-time counters")[w]["pid"] for w in range(G)]
    first = None

    def snapshot(label, k):
        pool.sync()
        stats = pool.request({w: ("host_stats",) for w in range(G)}, what="host-time counters")
        shm = [p for p in glob.glob("/dev/shm/gm-shard-*")]
        line = {"after": k, "parent_rss_mb": round(rss_kb(os.getpid()) / 1024, 1),
                "worker_rss_mb": [round(rss_kb(p) / 1024, 1) for p in pids], "shm_segments": len(shm),
                "shm_mb": round(sum(os.path.getsize(p) for p in shm) / 2 ** 20, 1),
                "worker_runs": [int(stats[w].get("runs", -1)) for w in range(G)]}
        print(label, line, flush=True)
        return line

    t0 = time.perf_counter()
    checksum = None
    for k in range(N):
        request = whole if k % 3 else inner          # (two request shapes: two kept jobs per worker)
        if k % 10 == 9:                               # every tenth result as an ndarray (a shared-memory mosaic)
            got = shard.get_data(tiler, request, devices, materialize=True)
            digest = float(np.asarray(got["values"], dtype=np.float64).sum())
            if request is whole:
                assert checksum in (None, digest), (checksum, digest)
                checksum = digest
            del got
        else:
            shard.get_data(tiler, request, devices, materialize=False)
        if k in (N // 10, N // 2, N - 1):
            line = snapshot("snapshot", k + 1)
            if first is None:
                first = line
    pool.sync()
    elapsed = time.perf_counter() - t0
    grew = {"parent_rss_mb": round(line["parent_rss_mb"] - first["parent_rss_mb"], 1),
            "worker_rss_mb": [round(b - a, 1) for a, b in zip(first["worker_rss_mb"], line["worker_rss_mb"])],
            "shm_segments": line["shm_segments"] - first["shm_segments"], "shm_mb": round(line["shm_mb"] - first["shm_mb"], 1)}
    print("done: %d requests through %d workers in %.1f s; growth between request %d and %d: %s" % (N, G, elapsed, first["after"], N, grew))
    shard.shutdown()
    left = glob.glob("/dev/shm/gm-shard-*")
    print("segments left after shutdown:", len(left))


if __name__ == "__main__":
    main()
