#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kb_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kb_trace -o kb -- python3 $R/tools/kbench.py cpu=0 > $R/gpurun_out/kb_trace.log 2>&1
cd $R; ls gpurun_out/kb_trace; head -5 gpurun_out/kb_trace/kb_kernel_stats.csv | cut -c1-200
