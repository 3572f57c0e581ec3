#!/bin/bash
# Run ON THE GPU BOX: tools/fewrow_breakdown.py for each library build in $VARIANTS (graspbalance_amd/libgraspbal_hip_<V>.so)
L=graspbalance_amd/libgraspbal_hip
for v in ${VARIANTS:-A B}; do
  cp ${L}_$v.so $L.so
  echo "== $v"; python tools/fewrow_breakdown.py "$@" 2>/dev/null | grep -v amdgpu
done
