"""Registers / spills / LDS / occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py graspbalance_amd/csrc/gemm_cl.hip [name filter]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
       "-munsafe-fp-atomics", "-fno-fast-math", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if filt not in d:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    i = d.find("<")
    print("%-70s VGPR %3s AGPR %3s spill %3s LDS %6s occ %s" % (
        d[:d.find("(")][-70:] if i < 0 else d[d.find("::") + 2:d.find(">", i) + 1][-70:], g(" VGPRs"), g("AGPRs"),
        g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
