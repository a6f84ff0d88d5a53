#!/bin/bash
# rocprofv3 kernel trace of one zonal statistic (run from the repo root through gpurun):
#   tools/zonal_trace.sh <tag> <n_labels> <statistic> [positive]
# -> gpurun_out/zt_<tag>/: kernel stats (all calls) in stats.csv, the kernels of the LAST call in order in last_call.txt
set -u
TAG=${1:?tag}; NL=${2:-1024}; ST=${3:-sum}; POS=${4:-}
REPO=$PWD; BASE=$REPO/gpurun_out/zt_$TAG; rm -rf "$BASE"; mkdir -p "$BASE"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$BASE/raw" -- python3 "$REPO/${ZT_DRIVER:-tools/zonal_trace_driver.py}" $NL $ST 5 $POS > "$BASE/log" 2>&1
cd "$REPO"
python3 - "$BASE" <<'PY'
import csv, glob, sys
base = sys.argv[1]
tr = glob.glob(base + "/raw/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(tr[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 5
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
with open(base + "/last_call.txt", "w") as f:
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%8.1f us  +gap %6.1f  dur %7.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
        prev_end = e
    f.write("total %.1f us over %d kernels\n" % ((prev_end - t0) / 1e3, len(last)))
print(open(base + "/last_call.txt").read())
PY
