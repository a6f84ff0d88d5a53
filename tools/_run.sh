#!/bin/bash
python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['traffic']); print(json.dumps(d['end_to_end'], indent=1)[:1800])"
