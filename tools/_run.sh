#!/bin/bash
python tests/zonal_stress.py 4000 2>&1 | tail -8
