#!/bin/bash
tools/gpu_profile.sh r02 mosaic --steps 5 --warmup 2 2>&1 | tail -4
