#!/bin/bash
python -m pytest tests/test_gpu_plans.py -x -q 2>&1 | tail -2
python tools/e2e_stencil.py 2>&1 | grep plans
