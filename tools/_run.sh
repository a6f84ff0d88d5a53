#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_keys" 2>&1 | grep -E "^E  |passed|failed" | head
