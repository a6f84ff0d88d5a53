#!/bin/bash
python tests/fuzz_more.py 3000 2>&1 | tail -2
python tests/stencil_stress.py 1500 2>&1 | tail -1
python tests/zonal_stress.py 3000 2>&1 | tail -1
