#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_plans.py tests/test_gpu_fuzz.py -x -q -k "tiler or tiled or plans or paste or fuzz" 2>&1 | tail -2
python tools/e2e_tiler_stencil.py 2>&1 | tail -6
