#!/bin/bash
# usage: tools/kernel_resources.sh graspbalance_amd/csrc/<file>.hip  -> table of VGPR/AGPR/spill/occupancy per kernel
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math $EXTRA \
  -I$(dirname $f) -c $f -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)",line)
    if m and cur:
        d[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith("LDS Size"):
            print("%-70s VGPR %3s AGPR %3s spill %3s occ %s scratch %s" % (cur[:70], d.get("VGPRs"), d.get("AGPRs"), d.get("VGPRs Spill"), d.get("Occupancy [waves/SIMD]"), d.get("ScratchSize [bytes/lane]")))
'
