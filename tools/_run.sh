#!/bin/bash
tools/gpu_profile.sh r02 tiler --steps 20 --warmup 3 2>&1 | head -1 | cut -c1-200
tools/gpu_profile.sh r02 mosaic --steps 5 --warmup 2 2>&1 | head -1 | cut -c1-200
