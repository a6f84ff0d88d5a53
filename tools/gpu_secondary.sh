#!/bin/bash
# Secondary workloads (BASELINE.json configs[2..4]) with their rocprofv3 kernel stats: tools/gpu_secondary.sh <tag>
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for w in reclass bilinear mosaic tiler; do
  rm -rf "$OUT/prof_$w"
  # (the cheap workloads need more steps than the 20 ms mosaic for a timed region of some milliseconds)
  STEPS=30; [ $w = mosaic ] && STEPS=5
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$w" -- python3 "$REPO/bench.py" --workload $w --steps $STEPS --warmup 3 > "$OUT/bench_$w.log" 2> "$OUT/bench_$w.err"
  grep '^{' "$OUT/bench_$w.log" | cut -c1-330
  tail -2 "$OUT/bench_$w.err"
done
cd "$REPO"
python - <<'PY'
import csv, glob, os
for w in ["reclass", "bilinear", "mosaic", "tiler"]:
    fs = sorted(glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % w, recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    print("==", w)
    for r in list(csv.DictReader(open(fs[-1])))[:6]:
        print("  %-64s calls=%-5s avg=%9.1f us total=%8.1f ms %s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
