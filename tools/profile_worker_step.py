"""cProfile of what ONE worker of the sharded scheduler does per request of the configs[4] mosaic (shard_worker._evaluate:
get_sync over the fused graph of its tiles), on a small mosaic (4096^2 in 256 tiles of 256^2) so that the GPU is idle and the
figure is the host side alone."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dask_geomodeling_amd import raster
from dask_geomodeling_amd.core.graphs import get_sync
from dask_geomodeling_amd.hip import runtime
from dask_geomodeling_amd.fuse import fuse_graph
runtime.set_device(0)
side, mosaic = 256, 4096
tiler = raster.RasterTiler(bench.mosaic_view(raster, "owned", mosaic), side)
request = dict(mode="vals", bbox=(0.0, 0.0, float(mosaic), float(mosaic)), width=mosaic, height=mosaic, projection="EPSG:28992")
pairs = list(tiler.get_sources_and_requests(**request))
graph, keys = {}, []
for t in range(len(pairs) - 1):
    store, tile_request = pairs[1 + t]
    graph, key = store.get_compute_graph(cached_compute_graph=graph, **tile_request)
    keys.append(key)
graph = fuse_graph(graph, keys)
print("tiles", len(keys), "graph tasks", len(graph))
for _ in range(20):
    get_sync(graph, keys)
runtime.sync()
t0 = time.perf_counter()
for _ in range(50):
    get_sync(graph, keys)
host = time.perf_counter() - t0
runtime.sync()
print("host per step %.1f us (%.2f us per tile), with drain %.1f us" % (host / 50 * 1e6, host / 50 / len(keys) * 1e6, (time.perf_counter() - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    get_sync(graph, keys)
pr.disable(); runtime.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
