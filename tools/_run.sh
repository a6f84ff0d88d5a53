#!/bin/bash
# round-2 final measurements: GPU tests, smoke, profiles of every bench workload, stand-alone kernels
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r02.log 2>&1; tail -3 gpurun_out/pytest_gpu_r02.log
python __graft_entry__.py smoke 2>&1 | tail -1
tools/gpu_profile.sh r02 chain5 > gpurun_out/prof_chain5.out 2>&1
tools/gpu_profile.sh r02 mosaic --steps 5 --warmup 2 > gpurun_out/prof_mosaic.out 2>&1
tools/gpu_profile.sh r02 reclass --steps 20 --warmup 3 > gpurun_out/prof_reclass.out 2>&1
tools/gpu_profile.sh r02 bilinear --steps 20 --warmup 3 > gpurun_out/prof_bilinear.out 2>&1
tools/gpu_profile.sh r02 tiler --steps 20 --warmup 3 > gpurun_out/prof_tiler.out 2>&1
for w in chain5 mosaic reclass bilinear tiler; do tail -n 3 gpurun_out/prof_$w.out | cut -c1-400; done
python tools/kbench.py > gpurun_out/kbench_r02.log 2>&1; cat gpurun_out/kbench_r02.log
GM_SPEC=0 python bench.py --cpu-tiles 0 --no-e2e 2>/dev/null | cut -c1-160
GM_SPEC=0 python bench.py --workload mosaic --steps 5 --warmup 2 2>/dev/null | cut -c1-160
python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | cut -c1-200
python bench.py --gpus 2 --workload mosaic --steps 5 --warmup 2 2>/dev/null | cut -c1-200
