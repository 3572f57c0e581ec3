#!/bin/bash
# GPU box: PMC counters of the register-direct wgrad on the probe shapes (one pass per counter group).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/wg_pmc
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p1 -o p -- python3 tools/wg_probe.py graspbalance_amd/libgraspbal_hip.so 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -o p -- python3 tools/wg_probe.py graspbalance_amd/libgraspbal_hip.so 2 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o p -- python3 tools/wg_probe.py graspbalance_amd/libgraspbal_hip.so 2 > $OUT/kt.log 2>&1
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/wg_pmc"
for d in ("p1", "p2"):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no counter file"); continue
    rows = list(csv.DictReader(open(f[0])))
    by = collections.OrderedDict()
    for r in rows:
        if "wgrad_direct" not in r["Kernel_Name"]:
            continue
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["_k"] = r["Kernel_Name"][:60]
    for k, v in by.items():
        print(d, k, " ".join("%s=%.4g" % (a, b) for a, b in v.items() if a != "_k"), v["_k"])
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "wgrad_direct" in r["Kernel_Name"]:
        print("kt", r["Dispatch_Id"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us", r["Kernel_Name"][:60])
PY
