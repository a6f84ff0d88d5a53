#!/bin/bash
# SQ counters of the VM on the mosaic (configs[4]) program pieces: tools/pmc_mosaic.sh
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mosaic
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mosaic -- python3 $R/tools/vm_cost.py mosaic > $R/gpurun_out/pmc_mosaic.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_mosaic/**/*counter_collection.csv", recursive=True)[-1]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "vm_kernel" not in r["Kernel_Name"]: continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][17:45]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
labels = ["arith", "clipped", "+classify5", "full", "clipped->mask", "clipped->classify2", "arith->classify5", "(A+B)->classify5"]
ids = sorted(rows)
px = 16 * 4096 * 4096
for i in range(0, len(ids), 5):
    d = rows[ids[i + 2]]
    print("%-20s %s VALU/px %.1f SALU/wave %.0f  wait_any %.0f%% wait_inst %.0f%% active %.0f%% (valu %.0f%%)" % (
        labels[i // 5], d["k"], d["SQ_INSTS_VALU"] * 64 / px, d["SQ_INSTS_SALU"] / d["SQ_WAVES"],
        100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
        100 * d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]))
PY
