#!/bin/bash
mkdir -p gpurun_out/prof_r02_mosaic
python bench.py --workload mosaic --steps 5 --warmup 2 2>gpurun_out/mosaic.err | tee gpurun_out/prof_r02_mosaic/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], json.dumps(d.get('end_to_end')))"
tail -2 gpurun_out/mosaic.err
