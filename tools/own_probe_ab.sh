#!/bin/bash
# Run ON THE GPU BOX: tools/own_probe.py under rocprofv3 for each library build in $VARIANTS; prints median kernel durations
L=graspbalance_amd/libgraspbal_hip
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-A B}; do
  cp ${L}_$v.so $L.so
  OUT=gpurun_out/own_prof_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python3 tools/own_probe.py > $OUT/out.txt 2> $OUT/err.txt
  cp $(find $OUT/kt -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv; rm -rf $OUT/kt
done
