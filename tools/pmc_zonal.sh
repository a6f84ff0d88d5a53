#!/bin/bash
# HBM traffic of one zonal statistic (run from the repo root through gpurun): tools/pmc_zonal.sh <tag> [n_labels] [statistic] [positive]
# FETCH_SIZE and WRITE_SIZE in separate passes (gfx950; never combined with trace domains other than --kernel-trace), summed
# over the kernels of ONE call, with the guide's correction (FETCH_SIZE x 2 for wide streaming reads) ->
# gpurun_out/zonal_pmc_<tag>.json.  Copy into profiles/ what is to be kept.
set -u
TAG=${1:?tag}; NL=${2:-1024}; ST=${3:-sum}; POS=${4:-}
REPO=$PWD; BASE=$REPO/gpurun_out/zonal_pmc_$TAG; rm -rf "$BASE"; mkdir -p "$BASE"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$BASE/$c" -- python3 "$REPO/tools/zonal_trace_driver.py" $NL $ST 4 $POS > "$BASE/$c.log" 2>&1
done
cd "$REPO"
python3 - "$BASE" "$TAG" "$NL" "$ST" "$POS" <<'PY'
import collections, csv, glob, json, sys
base, tag, nl, st, pos = sys.argv[1:6]
out = {"workload": "zonal %s, %s labels, 3072 x 4096 float32%s (tools/zonal_trace_driver.py)" % (st, nl, ", positive cells" if pos else ""),
       "cells": 3072 * 4096, "algorithmic_bytes": 8 * 3072 * 4096, "calls": 4, "kernels": {}}
total = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(base + "/" + c + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            per[r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]] += float(r["Counter_Value"])
    for k, v in per.items():
        out["kernels"].setdefault(k, {})[c + "_KiB_per_call"] = v / 4
    total[c] = sum(per.values()) / 4 * 1024  # (the counters are in KiB)
out["fetch_bytes_per_call_raw"] = total["FETCH_SIZE"]
out["write_bytes_per_call"] = total["WRITE_SIZE"]
out["hbm_bytes_per_call"] = 2 * total["FETCH_SIZE"] + total["WRITE_SIZE"]  # gfx950: FETCH_SIZE tallies 128-byte requests at 64
out["ratio_to_algorithmic"] = out["hbm_bytes_per_call"] / out["algorithmic_bytes"]
json.dump(out, open("gpurun_out/zonal_pmc_%s.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
PY
