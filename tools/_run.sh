#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "zonal_edge" 2>&1 | tail -15
