#!/bin/bash
# SQ counters of the dominant kernel of a bench workload: tools/pmc_workload.sh <workload> <pixels per launch>
R=$PWD
W=${1:-reclass}
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_wl
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wl -- python3 $R/bench.py --workload $W --steps 10 --warmup 3 --no-check --cpu-tiles 0 > $R/gpurun_out/pmc_wl.log 2>&1
cd $R
python - "$2" <<'PY'
import csv, glob, collections, sys
px = float(sys.argv[1])
f = glob.glob("gpurun_out/pmc_wl/**/*counter_collection.csv", recursive=True)[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m["SQ_WAVES"] < 1000: continue
    print("%s\n   dispatches %d VALU/px %.1f SALU/wave %.0f wait_any %.0f%% wait_inst %.0f%% active %.0f%% (valu %.0f%%) waves %.0f" % (
        k, len(c["SQ_WAVES"]), m["SQ_INSTS_VALU"] * 64 / px, m["SQ_INSTS_SALU"] / m["SQ_WAVES"], 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        100 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAVES"]))
PY
