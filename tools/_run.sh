#!/bin/bash
python tools/kbench.py zonal cpu=0 2>&1 | grep -E "i32"
