# kernel-trace part of refresh_profiles.sh only (fast): steady state of the timed graph replays
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/profiles_new; rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/kt.log
tail -1 $OUT/bench_line_under_rocprof.json > $OUT/line.tmp && mv $OUT/line.tmp $OUT/bench_line_under_rocprof.json
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/kernel_stats.csv 8 30 > $OUT/kernel_stats_summary.txt
(cd tools && python3 prof_steady.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 3 70) > $OUT/steady_state.txt
rm -rf $OUT/kt
cat $OUT/bench_line_under_rocprof.json; cat $OUT/steady_state.txt
