#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "bilinear or resample or canonical or nearest" 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
python bench.py --workload bilinear --steps 20 --warmup 3 --cpu-tiles 0 --no-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"
