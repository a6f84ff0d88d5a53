#!/bin/bash
# Round-end measurement on the GPU box (run from the repo root through gpurun):
#   tools/gpu_round.sh <tag> [quick]
# GPU tests, smoke, the bench line, the rocprofv3 kernel stats of the same command and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950).  Everything lands under
# gpurun_out/; tools/collect_profiles.py copies the summaries into profiles/.
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
if [ "$2" != "quick" ]; then
  python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
  python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
fi
python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; cat "$OUT/bench.log"; tail -3 "$OUT/bench.err"
cd /tmp; export TMPDIR=/tmp
rm -rf "$OUT/prof_$TAG" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -- python3 "$REPO/bench.py" > "$OUT/prof_bench.log" 2>&1
grep '^{' "$OUT/prof_bench.log" | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --cpu-tiles 0 --no-check > "$OUT/pmc_$c.log" 2>&1
done
cd "$REPO"
python tools/collect_profiles.py "$TAG" --dry
