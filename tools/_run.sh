#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "stencil or window or dilate or hillshade or movingmax or smooth or gauss" 2>&1 | tail -3
python -m pytest tests/test_gpu_plans.py -x -q 2>&1 | tail -2
python tests/stencil_stress.py 300 2>&1 | tail -1
python tools/e2e_stencil.py 2>&1 | tail -8
