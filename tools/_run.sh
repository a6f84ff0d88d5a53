#!/bin/bash
python tools/zonal_shapes.py > gpurun_out/zonal_shapes_r02.log 2>&1; cat gpurun_out/zonal_shapes_r02.log | grep "sum f32"
