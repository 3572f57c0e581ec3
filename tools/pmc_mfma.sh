#!/bin/bash
# GPU box: where the GEMM kernels of the default bench step spend their cycles (SQ counters + GRBM_GUI_ACTIVE per dispatch).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_mfma
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
GB_BENCH_TIMED_ONLY=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > /dev/null 2> $OUT/p.log
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/pmc_mfma/p/**/*counter_collection.csv", recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "wg": r.get("Workgroup_Size", "")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for i, d in disp.items():
    k = d["k"]
    if not re.search(r"gemm_|wgrad_direct", k):
        continue
    gui = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui <= 0:
        continue
    rows.append((re.sub(r"\(.*", "", k.replace("void gb::", ""))[:44], d["grid"], gui, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024),
                 4 * d.get("SQ_WAVE_CYCLES", 0) / (gui * 1024), d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
                 d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), d.get("SQ_ACTIVE_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1)))
# the last third of the dispatches = the last timed step, roughly; aggregate by (kernel, grid)
agg = collections.OrderedDict()
for r in rows[len(rows) * 2 // 3:]:
    a = agg.setdefault((r[0], r[1]), [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1
    for j in range(6):
        a[j + 1] += r[j + 2]
print("%-46s %9s %4s %9s %6s %6s %6s %6s %6s" % ("kernel", "grid", "n", "kcyc/call", "mfma", "waves", "wait", "stall", "active"))
for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    n = a[0]
    print("%-46s %9s %4d %9.1f %6.2f %6.2f %6.2f %6.2f %6.2f" % (k, g, n, a[1] / n / 1e3, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n))
PY
