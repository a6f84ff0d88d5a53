#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -s -k "hillshade or stencil" 2>&1 | grep -E "^E|passed|failed|hillshade" | head -20
python tools/kbench.py stencil 2>&1 | head -1
