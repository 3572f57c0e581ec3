set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ring_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o b -- python3 tools/ring_prof.py run > $OUT/out.txt 2> $OUT/err.txt
cp $(find $OUT/kt -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/kt
python3 tools/ring_prof.py parse $OUT/kernel_trace.csv | tee $OUT/table.txt
