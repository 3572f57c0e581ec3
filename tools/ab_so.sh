#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library on one box.  usage: tools/ab_so.sh [bench args...]
# expects graspbalance_amd/libgraspbal_hip_<V>.so for every V in $VARIANTS (default "A B A B"); prints ms_per_step of each run
# and leaves the LAST variant installed - copy the build you want afterwards.
L=graspbalance_amd/libgraspbal_hip
for v in ${VARIANTS:-A B A B}; do
  cp ${L}_$v.so $L.so
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['ms_per_step'], 'cl', d['roofline']['ms_per_step'] if 'cl' in d['roofline']['kernel'] else d['roofline_gemm2']['ms_per_step'], 'rs', d['roofline_gemm2']['ms_per_step'] if 'rs' in d['roofline_gemm2']['kernel'] else d['roofline']['ms_per_step'])"
done
