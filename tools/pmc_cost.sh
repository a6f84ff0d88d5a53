#!/bin/bash
# VALU instructions per pixel of N-step chains (fixed vs marginal cost of a micro-op): tools/pmc_cost.sh "<step name>"
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_cost
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cost -- python3 $R/tools/vm_cost.py "$@" > $R/gpurun_out/pmc_cost.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_cost/**/*counter_collection.csv", recursive=True)[-1]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "vm_kernel" not in r["Kernel_Name"]: continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][17:45]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
px = 16 * 4096 * 4096
for i in range(0, len(ids), 5):
    d = rows[ids[i + 2]]
    print("dispatch group %d %s VALU/px %.1f SALU/wave %.0f wait_any %.0f%% valu-active %.0f%%" % (
        i // 5, d["k"], d["SQ_INSTS_VALU"] * 64 / px, d["SQ_INSTS_SALU"] / d["SQ_WAVES"],
        100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]))
PY
