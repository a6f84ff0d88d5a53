#!/bin/bash
python tests/fuzz_more.py 4000 2>&1 | tail -4
GM_SPEC=2 python tests/fuzz_more.py 500 2>&1 | tail -2
python tests/stencil_stress.py 1200 2>&1 | tail -2
