#!/bin/bash
# spec kernels: new tests, mosaic with digimap, chain5 tuning sweep (G, waves)
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_spec.py -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py -x -q -k "one_pixel_short" 2>&1 | tail -3
GM_SPEC_VERBOSE=1 python bench.py --workload mosaic --steps 5 --warmup 2 2>&1 | cut -c1-220
for w in 0 4 6 8; do for g in 0 4 8; do
  echo "waves=$w vec=$g"; GM_SPEC_WAVES=$w GM_VM_VEC=$g python bench.py --cpu-tiles 0 --no-check --steps 60 2>/dev/null | grep -o '"value": [0-9.]*'
done; done
echo interp; GM_SPEC=0 python bench.py --cpu-tiles 0 --no-check --steps 60 2>/dev/null | grep -o '"value": [0-9.]*'
echo interp; GM_SPEC=0 python bench.py --cpu-tiles 0 --no-check --steps 60 2>/dev/null | grep -o '"value": [0-9.]*'
