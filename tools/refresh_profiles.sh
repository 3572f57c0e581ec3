#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: regenerates the evidence bench.py's roofline numbers rest on.
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command  -> kernel stats + steady-state breakdown
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run -> per-kernel HBM traffic (json + text)
# Outputs land in gpurun_out/profiles_new/ ; copy what is to be judged into profiles/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_new
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --no-cpu-baseline --no-extra-configs > $OUT/bench_line_under_rocprof.json 2> $OUT/kt.log
tail -1 $OUT/bench_line_under_rocprof.json > $OUT/line.tmp && mv $OUT/line.tmp $OUT/bench_line_under_rocprof.json
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/kernel_stats.csv 8 30 > $OUT/kernel_stats_summary.txt
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 3 60) > $OUT/steady_state.txt
rm -rf $OUT/kt
echo "kernel trace done"
GB_BENCH_TIMED_ONLY=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2> $OUT/pmc_f.log
echo "pmc fetch done"
GB_BENCH_TIMED_ONLY=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2> $OUT/pmc_w.log
echo "pmc write done"
python3 tools/pmc_summary.py $OUT/pmc_f $OUT/pmc_w gb:: > $OUT/pmc_per_kernel.txt
python3 tools/pmc_summary.py $OUT/pmc_f $OUT/pmc_w gb:: --json > $OUT/pmc_traffic.json
rm -rf $OUT/pmc_f $OUT/pmc_w
ls -la $OUT
