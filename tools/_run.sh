#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
tools/gpu_profile.sh r02 chain5 2>&1 | tail -1 | cut -c1-160
tools/gpu_profile.sh r02 reclass --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-120
tools/gpu_profile.sh r02 bilinear --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-120
tools/gpu_profile.sh r02 tiler --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-120
tools/gpu_profile.sh r02 mosaic --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-120
