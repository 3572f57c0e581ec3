#!/bin/bash
# GPU box: kernel trace of the default bench with / without the paired dgrad+wgrad launch; ring kernels per step.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pair_ab
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for mode in 0 1; do
  GB_PAIR=$mode GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$mode -o bench -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-configs > $OUT/line$mode.json 2> $OUT/kt$mode.log
  (cd tools && python3 prof_steady.py $(find $OUT/kt$mode -name "*kernel_trace.csv" | head -1) 3 80) > $OUT/steady$mode.txt
  rm -rf $OUT/kt$mode
done
grep -i "ring\|steady\|gemm" $OUT/steady0.txt > $OUT/ring0.txt || true
grep -i "ring\|steady\|gemm" $OUT/steady1.txt > $OUT/ring1.txt || true
