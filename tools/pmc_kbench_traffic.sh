#!/bin/bash
# HBM traffic per kernel of tools/kbench.py <section> (run from the repo root through gpurun): FETCH_SIZE and WRITE_SIZE in
# separate passes (gfx950), averaged per (kernel, grid), FETCH x 2 per the guide's correction for wide streaming reads.
#   tools/pmc_kbench_traffic.sh <tag> <section> [kbench args]   ->  gpurun_out/kbench_traffic_<tag>.txt
set -u
TAG=${1:?tag}; shift
R=$PWD; BASE=$R/gpurun_out/kbt_$TAG; rm -rf "$BASE"; mkdir -p "$BASE"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$BASE/$c" -- python3 "$R/tools/kbench.py" "$@" > "$BASE/$c.log" 2>&1
done
cd "$R"
python3 - "$BASE" "$TAG" <<'PY'
import collections, csv, glob, sys
base, tag = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(base + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
                agg[(name, r["Grid_Size"])][c].append(float(r["Counter_Value"]))
with open("gpurun_out/kbench_traffic_%s.txt" % tag, "w") as out:
    for (name, grid), c in sorted(agg.items()):
        f = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"])) * 1024 * 2
        w = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])) * 1024
        if f + w < 16e6:
            continue
        line = "%-72s grid %-10s launches %3d  read %8.1f MB  written %8.1f MB  total %8.1f MB" % (name, grid, len(c["FETCH_SIZE"]), f / 1e6, w / 1e6, (f + w) / 1e6)
        print(line)
        out.write(line + "\n")
PY
