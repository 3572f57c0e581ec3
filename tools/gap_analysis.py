"""Idle time of the GPU inside a train step, from a rocprofv3 kernel trace of bench.py: gaps between consecutive
kernels (all streams merged), split at the host synchronisation of the step (the crop row counts: cyl_unique_kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"]]
lo, hi = marks[-4], marks[-1]
steps = 3
busy_end = None
gaps_pre = gaps_post = 0.0
big = []
phase = "pre"
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    if "fps_pruned_kernel" in name:
        phase = "pre"
    if busy_end is not None and s > busy_end:
        g = (s - busy_end) / 1e3
        if phase == "pre":
            gaps_pre += g
        else:
            gaps_post += g
        if g > 30:
            big.append((round(g, 1), phase, name[:60]))
    if "cyl_rows_kernel" in name:
        phase = "post"
    busy_end = e if busy_end is None else max(busy_end, e)
wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / steps
print("wall %.2f ms/step; idle before the row-count sync %.2f ms/step, after it %.2f ms/step" % (wall, gaps_pre / 1e3 / steps, gaps_post / 1e3 / steps))
print("gaps > 30 us:", sorted(big, reverse=True)[:25])
