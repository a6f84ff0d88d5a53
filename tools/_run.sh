#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "gauss or smooth or stencil" 2>&1 | tail -3
python tests/stencil_stress.py 2>&1 | tail -1
