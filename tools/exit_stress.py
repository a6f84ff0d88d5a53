"""Processes that exit while the library's worker thread is still compiling a hot program (round 3: `tools/kbench.py mask`
crashed or hung at exit).  Runs N children, each with an empty private kernel cache: the child makes a program hot (which
queues its compilation), optionally sleeps a few milliseconds, and exits without waiting.  Every child must exit with code 0
within the time limit.  python tools/exit_stress.py [N]"""
import os
import subprocess
import sys
import tempfile

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from dask_geomodeling_amd.hip import program as P
rng = np.random.default_rng(int(sys.argv[1]))
a = rng.random((1, 512, 512)).astype("f4")
b = rng.random((1, 512, 512)).astype("f4")
k = float(rng.random())  # a different program (and kernel) per child
from dask_geomodeling_amd import raster
from datetime import datetime
src = [raster.MemorySource(np.asarray(x), -9999.0, "EPSG:28992", 1.0, (0.0, 512.0), datetime(2000, 1, 1)) for x in (a, b)]
view = (src[0] * k + src[1]) - k
req = dict(mode="vals", bbox=(0.0, 0.0, 512.0, 512.0), projection="EPSG:28992", width=512, height=512)
for _ in range(int(os.environ.get("GM_SPEC_HOT", "3")) + 2):
    view.get_data(**req)
time.sleep(float(sys.argv[2]))
print("pending at exit:", P.spec_pending(), flush=True)
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = 0
    for i in range(n):
        with tempfile.TemporaryDirectory() as cache:
            env = dict(os.environ, GM_SPEC_CACHE=cache, GM_SPEC_HOT="3")
            delay = [0.0, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.3, 0.6][i % 9]
            try:
                res = subprocess.run([sys.executable, "-c", CHILD % root, str(i), str(delay)], env=env, capture_output=True, text=True,
                                     timeout=int(os.environ.get("GM_EXIT_STRESS_LIMIT", "40")))
                rc, out = res.returncode, (res.stdout + res.stderr)[-300:]
            except subprocess.TimeoutExpired:
                rc, out = "TIMEOUT", ""
            if rc != 0:
                bad += 1
            print("child %2d delay %.3f s rc %s %s" % (i, delay, rc, out.strip().replace("\n", " | ")), flush=True)
    print("EXIT STRESS children %d failures %d" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
