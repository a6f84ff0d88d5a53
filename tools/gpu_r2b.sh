#!/bin/bash
# specialised kernels vs interpreter: headline chain and mosaic, A/B on one box
OUT=$PWD/gpurun_out; mkdir -p $OUT
export GM_SPEC_VERBOSE=1
for m in 0 1; do
  GM_SPEC=$m python bench.py --cpu-tiles 0 > $OUT/ab_chain_$m.log 2> $OUT/ab_chain_$m.err; cut -c1-200 $OUT/ab_chain_$m.log; grep -o '"roofline".*' $OUT/ab_chain_$m.log | cut -c1-330; tail -2 $OUT/ab_chain_$m.err
  GM_SPEC=$m python bench.py --workload mosaic --steps 5 --warmup 2 > $OUT/ab_mosaic_$m.log 2> $OUT/ab_mosaic_$m.err; cut -c1-200 $OUT/ab_mosaic_$m.log; tail -2 $OUT/ab_mosaic_$m.err
  GM_SPEC=$m python bench.py --workload tiler --steps 20 --warmup 3 > $OUT/ab_tiler_$m.log 2> $OUT/ab_tiler_$m.err; cut -c1-200 $OUT/ab_tiler_$m.log; tail -2 $OUT/ab_tiler_$m.err
  GM_SPEC=$m python bench.py --workload reclass --steps 20 --warmup 3 > $OUT/ab_reclass_$m.log 2> $OUT/ab_reclass_$m.err; cut -c1-200 $OUT/ab_reclass_$m.log; tail -2 $OUT/ab_reclass_$m.err
done
GM_SPEC=2 python -m pytest tests/test_gpu_fuzz.py -x -q -k "on_device and (0 or 1)" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
