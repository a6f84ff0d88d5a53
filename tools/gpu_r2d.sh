#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_spec.py -x -q -k "literals or strided or hot" 2>&1 | tail -5
for w in 0 2 3 4 6; do for g in 4 8; do
  echo "mosaic waves=$w vec=$g"; GM_SPEC_WAVES=$w GM_VM_VEC=$g python bench.py --workload mosaic --steps 5 --warmup 2 2>/dev/null | grep -o '"value": [0-9.]*'
done; done
for w in 0 4 8; do for g in 4 8; do
  echo "tiler waves=$w vec=$g"; GM_SPEC_WAVES=$w GM_VM_VEC=$g python bench.py --workload tiler --steps 30 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*'
done; done
for i in 1 2 3; do for cfg in "0 0" "4 8" "8 4"; do set -- $cfg
  echo "chain waves=$1 vec=$2"; GM_SPEC_WAVES=$1 GM_VM_VEC=$2 python bench.py --cpu-tiles 0 --no-check --steps 100 2>/dev/null | grep -o '"value": [0-9.]*'
done; done
