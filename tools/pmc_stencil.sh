#!/bin/bash
# tools/pmc_stencil.sh <which>  : kernel durations + SQ counters of tools/stencil_driver.py <which>
W=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/k_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k_trace -o k -- python3 $R/tools/stencil_driver.py $W > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/k_pmc -o k -- python3 $R/tools/stencil_driver.py $W > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/k_pmc2 -o k -- python3 $R/tools/stencil_driver.py $W > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/k_trace/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:4]:
        print(row["Name"][:60], row["Calls"], "avg ns", row["AverageNs"])
for d in ("k_pmc", "k_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:50]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(k, row["Counter_Name"])] += 1
    for k, cs in acc.items():
        if "generate" in k: continue
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-24s %.4g per launch" % (c, v / n[(k, c)]))
PY
