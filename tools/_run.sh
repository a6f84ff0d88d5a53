#!/bin/bash
python -m pytest tests/test_gpu_plans.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
python tools/e2e_stencil.py 2>&1 | grep plans
python bench.py --steps 20 --warmup 3 --cpu-tiles 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print({k:v for k,v in e.items() if k!='config'})"
