#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "movingmax or stencil or moving" 2>&1 | tail -2
python tests/stencil_stress.py 2>&1 | tail -1
