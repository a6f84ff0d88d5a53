#!/bin/bash
# round-2 first GPU pass: full GPU tests, headline bench, mosaic with per-rank leaves (1 and 2 self-spawned ranks)
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.log 2> $OUT/bench.err; cat $OUT/bench.log | cut -c1-600
python bench.py --workload mosaic --steps 5 --warmup 2 > $OUT/mosaic1.log 2> $OUT/mosaic1.err; cat $OUT/mosaic1.log; tail -2 $OUT/mosaic1.err
python bench.py --workload mosaic --steps 5 --warmup 2 --leaves whole > $OUT/mosaic1w.log 2> $OUT/mosaic1w.err; cat $OUT/mosaic1w.log; tail -2 $OUT/mosaic1w.err
python bench.py --workload mosaic --steps 5 --warmup 2 --gpus 2 > $OUT/mosaic2.log 2> $OUT/mosaic2.err; cat $OUT/mosaic2.log; tail -2 $OUT/mosaic2.err
python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench2.log 2> $OUT/bench2.err; cut -c1-300 $OUT/bench2.log; tail -2 $OUT/bench2.err
