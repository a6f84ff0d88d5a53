#!/bin/bash
python -m pytest tests/test_gpu_plans.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
