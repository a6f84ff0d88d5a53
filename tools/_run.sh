#!/bin/bash
python bench.py --steps 20 --warmup 3 --cpu-tiles 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print(d['value'], d['roofline']['traffic'] is not None); print({k:(v['Mpix/s'], v.get('us_per_request')) for k,v in e.items() if isinstance(v,dict)})"
python bench.py --workload mosaic --steps 3 --warmup 1 --cpu-tiles 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('end_to_end'))"
