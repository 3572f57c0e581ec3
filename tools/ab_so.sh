#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library on one box.  usage: tools/ab_so.sh [bench args...]
# expects graspbalance_amd/libgraspbal_hip_<V>.so for every V in $VARIANTS (default "A B A B"); prints ms_per_step of each run
# and leaves the LAST variant installed - copy the build you want afterwards.
L=graspbalance_amd/libgraspbal_hip
for v in ${VARIANTS:-A B A B}; do
  cp ${L}_$v.so $L.so
  echo -n "$v: "; python bench.py --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python tools/ab_line.py
done
cp ${L}_${KEEP:-B}.so $L.so
