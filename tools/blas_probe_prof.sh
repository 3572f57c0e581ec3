set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/blas_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python3 tools/blas_probe.py > $OUT/out.txt 2> $OUT/err.txt
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/kt
