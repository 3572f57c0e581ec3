#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the default bench, steady-state rows matching $1 (grep pattern).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_tmp; rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --no-cpu-baseline > $OUT/line.json 2> $OUT/kt.log
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 3 60) > $OUT/steady.txt
head -2 $OUT/steady.txt; grep -E "$1" $OUT/steady.txt
rm -rf $OUT/kt
