#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "dilate or stencil" 2>&1 | tail -3
python tests/stencil_stress.py 2>&1 | tail -2
python tools/kbench.py stencil 2>&1 | grep -i dilate
