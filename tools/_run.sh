#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "zonal" 2>&1 | tail -2
python tools/zonal_shapes.py | grep "sum"
python tools/kbench.py zonal cpu=0 2>&1 | grep -E "sum|mean"
