#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "hillshade or stencil" 2>&1 | tail -2
python tests/stencil_stress.py 2>&1 | tail -1
python tools/kbench.py stencil 2>&1 | grep -i "hillshade"
