#!/bin/bash
# CPU-only (the flags of csrc/Makefile): compile each given csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and list the kernels that
# use scratch memory or spill registers (name, VGPRs, scratch bytes per lane, spills, occupancy).
cd "$(dirname "$0")/../graspbalance_amd/csrc"
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -I../../include -c $f.hip -o /tmp/$f.spills.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}
    for k,pat in (('vgpr','VGPRs'),('scratch',r'ScratchSize \[bytes/lane\]'),('occ',r'Occupancy \[waves/SIMD\]'),('vspill','VGPRs Spill'),('sspill','SGPRs Spill')):
        m=re.search(r'remark:\s+'+pat+r': (\d+)',l)
        if m and cur: d[cur][k]=int(m.group(1))
for k,v in d.items():
    if v.get('scratch',0)>0 or v.get('vspill',0)>0:
        n=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:120]
        print('$f',n,v)
"
done
