#!/bin/bash
# GPU box: every launch of the kernels matching $1 in the last timed step of the default bench: grid, duration.  usage: kernel_calls.sh <substring>
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/kcalls
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs > $OUT/line.json 2> $OUT/kt.log
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kcalls/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_rows_kernel" in r["Kernel_Name"]]
lo, hi = marks[-2], marks[-1]
for r in rows[lo:hi]:
    if sys.argv[1] in r["Kernel_Name"]:
        print("%8.1f us  grid %-8s wg %-5s lds %-7s %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")),
              r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r["Kernel_Name"][:70]))
PY
rm -rf $OUT/kt
