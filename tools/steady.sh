#!/bin/bash
# GPU box: steady-state per-kernel breakdown of the default bench step (kernel trace of the timed replays).  usage: steady.sh [top]
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/steady
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o bench -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra-configs > $OUT/line.json 2> $OUT/kt.log
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 3 ${1:-45}) > $OUT/steady.txt
rm -rf $OUT/kt
cat $OUT/steady.txt
