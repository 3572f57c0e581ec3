"""Print the interesting numbers of bench.py JSON lines read from stdin (one per line, other lines ignored)."""
import json
import sys

for l in sys.stdin:
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    parts = ["ms_per_step %.3f" % d["ms_per_step"]]
    for k in ("roofline", "roofline_gemm2", "roofline_gemm3", "roofline_gemm4", "roofline_gemm5"):
        r = d.get(k)
        if r:
            parts.append("%s %.3f ms (%d launches, frac %.3f)" % (r["kernel"].split()[0], r["ms_per_step"], r["launches"], r["frac"]))
    print(" | ".join(parts))
