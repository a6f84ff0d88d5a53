set -u
cd $GRAFT_REPO_ROOT
for L in blocks32x64; do echo "== $L f32"; ZT_LAYOUT=$L bash tools/zonal_trace.sh ${L}_p 0 sum 2>&1 | cut -c1-150 | tail -16; grep routes gpurun_out/zt_${L}_p/log; done
