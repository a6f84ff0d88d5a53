#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -s -k "hillshade or stencil" 2>&1 | grep -E "hillshade|passed|failed|Error|error" | tail -15
python tools/kbench.py stencil 2>&1 | head -3
python bench.py --cpu-tiles 0 2>/dev/null | cut -c1-150
python bench.py --workload mosaic --steps 5 --warmup 2 2>/dev/null | cut -c1-150
python bench.py --workload tiler --steps 30 --warmup 5 2>/dev/null | cut -c1-150
