#!/bin/bash
export GM_HS_STREAM=1
python -m pytest tests/test_gpu_parity.py -x -q -k "hillshade" 2>&1 | tail -2
python tools/kbench.py stencil 2>&1 | grep -i "hillshade"
