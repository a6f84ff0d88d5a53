import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from dask_geomodeling_amd.hip import runtime, program as P
from dask_geomodeling_amd.raster import devops
from dask_geomodeling_amd.raster.sources import generate_window
from tools.kbench import timeit
side = 8192; n = side * side
runtime.set_device(0)
i4 = [generate_window((side, side), "i4", 60 + i, 0.0, 1000.0, 2147483647, (50, 50), 0, 0, side, side) for i in range(4)]
i8 = [generate_window((side, side), "i8", 50 + i, 0.0, 1000.0, 9223372036854775807, (50, 50), 0, 0, side, side) for i in range(4)]
f4 = [generate_window((side, side), "f4", 50 + i, 0.0, 1000.0, 3e38, (50, 50), 0, 0, side, side) for i in range(4)]
for vec in (4, 8):
    P.set_tuning(vec=vec)
    timeit("vec=%d mask i32->u8 (5 B/px)" % vec, 5.0 * n, [lambda t=t: devops.mask_process({"values": t, "no_data_value": 2147483647}, 1) for t in i4])
    timeit("vec=%d mask i64->u8 (9 B/px)" % vec, 9.0 * n, [lambda t=t: devops.mask_process({"values": t, "no_data_value": 9223372036854775807}, 1) for t in i8])
    timeit("vec=%d mask f32->u8 (5 B/px)" % vec, 5.0 * n, [lambda t=t: devops.mask_process({"values": t, "no_data_value": np.float32(3e38)}, 1) for t in f4])
    kw = {"dtype": "float32", "fillvalue": 3.4028234663852886e+38}
    timeit("vec=%d mul f32*2.5 (8 B/px)" % vec, 8.0 * n, [lambda t=t: devops.math_process("multiply", kw, {"values": t, "no_data_value": np.float32(3e38)}, 2.5) for t in f4])
    kw = {"dtype": "float64", "fillvalue": 1.7976931348623157e+308}
    timeit("vec=%d div i64/2.0 -> f64 (16 B/px)" % vec, 16.0 * n, [lambda t=t: devops.math_process("divide", kw, {"values": t, "no_data_value": 9223372036854775807}, 2.0) for t in i8])
