#!/bin/bash
# A/B on ONE box: tools/ab_bench.sh "ENV1=a ENV2=b" "ENV1=c" ...  -> ms/step of each setting, interleaved twice
for rep in 1 2; do
  for setting in "$@"; do
    out=$(env $setting python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$setting  rep $rep: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "fps", d["roofline_fps"]["launch_ms"], "cl", d["roofline"]["frac"], "rs", d["roofline_gemm2"]["frac"])')"
  done
done
