#!/bin/bash
python -m pytest tests/test_gpu_plans.py -x -q 2>&1 | tail -2
