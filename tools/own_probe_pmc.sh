set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/own_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc -o p -- python3 tools/own_probe.py > $OUT/out.txt 2> $OUT/err.txt
cp $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $OUT/counters.csv
rm -rf $OUT/pmc
