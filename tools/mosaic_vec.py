import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dask_geomodeling_amd.hip import program as P
import bench
for vec in (4, 8):
    P.set_tuning(vec=vec)
    sys.argv = ["bench.py", "--workload", "mosaic", "--steps", "5", "--warmup", "2"]
    print("vec", vec, flush=True)
    bench.main()
