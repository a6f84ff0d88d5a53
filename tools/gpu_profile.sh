#!/bin/bash
# Measurement of one bench workload on the GPU box (run from the repo root through gpurun):
#   tools/gpu_profile.sh <tag> <workload> [bench args...]
# the bench line, the rocprofv3 kernel stats of the same command and the two PMC passes (FETCH_SIZE /
# WRITE_SIZE need separate passes on gfx950; never combined with trace domains other than --kernel-trace).
# Everything lands under gpurun_out/prof_<tag>_<workload>/; tools/collect_profiles.py copies the summaries
# into profiles/.
set -u
TAG=${1:?usage: gpu_profile.sh <tag> <workload> [bench args]}; WL=${2:?usage: gpu_profile.sh <tag> <workload> [bench args]}; shift 2
case "$TAG$WL" in *[!A-Za-z0-9_,.-]*) echo "gpu_profile.sh: tag / workload may only hold [A-Za-z0-9_,.-]" >&2; exit 2;; esac
REPO=$PWD
[ -f "$REPO/bench.py" ] || { echo "gpu_profile.sh: run from the repo root" >&2; exit 2; }
BASE=$REPO/gpurun_out/prof_${TAG}_${WL}
rm -rf "$BASE"; mkdir -p "$BASE"
EXTRA="--no-secondary"; [ "$WL" = mosaic ] && EXTRA="--in-process"   # (nothing under the profiler starts worker processes)
python bench.py --workload $WL $EXTRA "$@" > "$BASE/bench.log" 2> "$BASE/bench.err"; cut -c1-300 "$BASE/bench.log"; tail -2 "$BASE/bench.err"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$BASE/stats" -- python3 "$REPO/bench.py" --workload $WL $EXTRA "$@" --cpu-tiles 0 --no-e2e > "$BASE/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$BASE/pmc_$c" -- python3 "$REPO/bench.py" --workload $WL $EXTRA "$@" --steps 3 --warmup 2 --cpu-tiles 0 --no-check --no-e2e > "$BASE/pmc_$c.log" 2>&1
done
cd "$REPO"
python tools/collect_profiles.py "$TAG" "$WL" --dry
