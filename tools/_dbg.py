import numpy as np, sys
sys.path.insert(0, "/root/repo")
from dask_geomodeling_amd.raster import stencil
from oracle import ref_numpy as O
rng = np.random.default_rng(1)
for shape, vals in [((1, 20, 80), [1]), ((1, 20, 80), [1, 3]), ((1, 5, 64), [1]), ((2, 20, 80), [1])]:
    v = rng.integers(0, 6, size=shape).astype("u1")
    want = O.dilate_process({"values": v, "no_data_value": None}, vals)["values"]
    got = stencil.dilate(v, vals).to_numpy()
    bad = np.argwhere(got != want)
    print(shape, vals, len(bad), bad[:12].tolist())
    if len(bad):
        b = bad[0]
        print("got", got[b[0], b[1], max(b[2]-4,0):b[2]+5], "want", want[b[0], b[1], max(b[2]-4,0):b[2]+5])
        print(v[b[0], b[1]:b[1]+3, max(b[2]-3,0):b[2]+7])
