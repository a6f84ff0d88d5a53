#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/k_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k_trace -o k -- python3 $R/tools/stencil_driver.py zonalsum > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/k_trace/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:12]:
        print("%-90s %5s avg us %9.1f" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
