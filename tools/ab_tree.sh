#!/bin/bash
# Run ON THE GPU BOX: A/B of two source trees (root vs _old) on one box.
R=$(pwd)
for v in old new old new; do
  if [ $v = old ]; then cd $R/_old; else cd $R; fi
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=[d['roofline'],d['roofline_gemm2']]
cl=[x for x in r if 'gemm_cl' in x['kernel']][0]; rs=[x for x in r if 'gemm_rs' in x['kernel']][0]
print('$v', d['ms_per_step'], 'cl', cl['ms_per_step'], cl['launches'], 'rs', rs['ms_per_step'], rs['launches'])"
done
