set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_stress; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python3 bench.py --config stress --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $OUT/line.json 2> $OUT/kt.log
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 3 45) > $OUT/steady.txt
rm -rf $OUT/kt
