import numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import load_family
from dask_geomodeling_amd.raster import devops
case = [c for c in load_family("views") if c["id"] == "view0011"][0]
A = case["leaves"]["A"]
kw = {"dtype": "float32", "fillvalue": case["fillvalue"]}
out = devops.math_process("power", kw, A, -1.0)
got = np.asarray(out["values"]); want = case["expect"]["values"]
bad = (got.view("u4") != want.view("u4")) & ~(np.isnan(got) & np.isnan(want))
print(bad.sum())
for idx in np.argwhere(bad)[:10]:
    t = tuple(idx); print(t, repr(A["values"][t]), repr(got[t]), repr(want[t]))
