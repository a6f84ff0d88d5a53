"""Which kernels issue their global loads one memory round trip at a time?  (round 3: a per-lane test, a select or simply the
use of a value right behind its load made the compiler put a full `s_waitcnt vmcnt(0)` behind every load of the nearest warp,
Dilate u8's and Smooth's staging, the register-sorted median and the stand-alone classify / reclassify kernels.)

Compiles one csrc file to gfx950 assembly and, per kernel, counts the loads, the loads that are waited for on their own (a
full wait with exactly one load outstanding) and the most loads ever in flight between two full waits:

    python tools/isa_load_scan.py dask_geomodeling_amd/csrc/gm_stencil.hip [substring of the kernel name]

A streaming kernel with one load per loop iteration shows up too (its parallelism is the occupancy); what to look for is an
UNROLLED batch -- many loads, nearly as many lone waits, "most in flight" of 1-3.  No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-sink-common-insts=false",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "build"), "--cuda-device-only", "-S"]


def scan(asm_path, want):
    stats, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            cur = m.group(1)
            stats[cur] = dict(loads=0, lone=0, pending=0, best=0)
            continue
        if cur is None:
            continue
        st = stats[cur]
        if "s_endpgm" in line:
            cur = None
        elif re.search(r"\b(global_load|buffer_load|flat_load)", line):
            st["loads"] += 1
            st["pending"] += 1
            st["best"] = max(st["best"], st["pending"])
        elif re.search(r"s_waitcnt.*vmcnt\(0\)", line):
            if st["pending"] == 1:
                st["lone"] += 1
            st["pending"] = 0
    for name, st in sorted(stats.items(), key=lambda kv: -kv[1]["lone"]):
        if want and want not in name:
            continue
        if "rocprim" in name or st["loads"] < 4:
            continue
        print("%-110s loads %4d  lone waits %4d  most in flight %3d" % (name[:110], st["loads"], st["lone"], st["best"]))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    src = os.path.abspath(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "out.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", asm, src], check=True, cwd=os.path.dirname(src))
        scan(asm, want)


if __name__ == "__main__":
    main()
