#!/bin/bash
# kernel-trace stats of the configs[4] bench line (bf16 MLP mode): run ON THE GPU BOX from the repo root
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_stress; rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --config stress --steps 5 --warmup 3 > $OUT/line.json 2> $OUT/kt.log
tail -1 $OUT/line.json > $OUT/l.tmp && mv $OUT/l.tmp $OUT/line.json
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/kernel_stats.csv 10 40 > $OUT/kernel_stats_summary.txt
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 2 45) > $OUT/steady_state.txt
rm -rf $OUT/kt
