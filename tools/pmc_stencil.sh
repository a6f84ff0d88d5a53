#!/bin/bash
# tools/pmc_stencil.sh <which>  : kernel durations + SQ counters of tools/stencil_driver.py <which>
# (run from the repo root on the GPU box; three rocprofv3 passes: kernel trace, then two counter passes of 8 SQ
#  counters each -- the number gfx950 schedules in one pass; the program itself follows `--`)
set -u
W=${1:?usage: pmc_stencil.sh <hillshade|mm<size>|smooth<sigma>|dilate8|bilinear|zonal<stat>>}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/tools/stencil_driver.py" ] || { echo "pmc_stencil.sh: $R is not the repo root" >&2; exit 2; }
OUT="$R/gpurun_out/pmc_stencil_$W"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k_trace" -o k -- python3 "$R/tools/stencil_driver.py" "$W" > "$OUT/trace.log" 2>&1 \
  || { echo "pmc_stencil.sh: kernel-trace pass failed, see $OUT/trace.log" >&2; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/k_pmc" -o k -- python3 "$R/tools/stencil_driver.py" "$W" > "$OUT/pmc1.log" 2>&1 \
  || { echo "pmc_stencil.sh: first counter pass failed, see $OUT/pmc1.log" >&2; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM \
  --output-format csv -d "$OUT/k_pmc2" -o k -- python3 "$R/tools/stencil_driver.py" "$W" > "$OUT/pmc2.log" 2>&1 \
  || { echo "pmc_stencil.sh: second counter pass failed, see $OUT/pmc2.log" >&2; exit 1; }
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in glob.glob(out + "/k_trace/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f)))[:4]:
        print(row["Name"][:60], row["Calls"], "avg ns", row["AverageNs"])
for d in ("k_pmc", "k_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
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
