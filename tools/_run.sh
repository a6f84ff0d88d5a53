#!/bin/bash
tools/gpu_profile.sh r02 chain5 2>&1 | tail -1 | cut -c1-200
tools/gpu_profile.sh r02 reclass --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-200
tools/gpu_profile.sh r02 bilinear --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-200
tools/gpu_profile.sh r02 tiler --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-200
tools/gpu_profile.sh r02 mosaic --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-200
python tools/kbench.py > gpurun_out/kbench_r02.log 2>&1; grep zonal gpurun_out/kbench_r02.log
