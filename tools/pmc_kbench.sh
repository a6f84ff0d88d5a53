#!/bin/bash
# SQ counters per kernel of tools/kbench.py <section>: tools/pmc_kbench.sh stencil
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_kb
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_kb -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/pmc_kb.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_kb/**/*counter_collection.csv", recursive=True)[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"][:78], r["Grid_Size"], r["LDS_Block_Size"], r.get("VGPR_Count", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m["SQ_WAVES"] < 2000: continue
    px = float(k[1])
    print("%s grid %s lds %s vgpr %s\n   VALU/thread %.1f LDS/thread %.1f wait_any %.0f%% wait_inst %.0f%% active %.0f%% (valu %.0f%%) waves %.0f" % (
        k[0], k[1], k[2], k[3], m["SQ_INSTS_VALU"] * 64 / px, m["SQ_INSTS_LDS"] * 64 / px, 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        100 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAVES"]))
PY
