#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "long_bins or many_rasters or smooth or integer_power or reduce or math_golden" 2>&1 | grep -E "^E  |passed|failed" | head -12
