#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "temporal or reduce or cumulative or group" 2>&1 | grep -E "^E  |passed|failed" | head
python tools/kbench.py temporal 2>&1 | grep -E "median|mean"
