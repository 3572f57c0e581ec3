"""Steady-state per-step kernel breakdown from a rocprofv3 kernel_trace.csv of bench.py.

The first-level FPS kernel (fps_pruned_kernel<1024, 20> / fps_reg_kernel<1024, 20>) is launched exactly once per train step, so
its launches mark step boundaries; only the last `nsteps` whole steps are summarised (drops the
one-time MIOpen find / lazy-init kernels of the first steps).
"""
import csv
import re
import sys
from collections import defaultdict

from prof_summary import category, short


def main(path, nsteps=3, top=30):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "fps_reg_kernel<1024, 20>" in r["Kernel_Name"] or "fps_rows_kernel" in r["Kernel_Name"]
             or "fps_pruned_kernel<1024, 20>" in r["Kernel_Name"] or "fps_pruned_big_kernel" in r["Kernel_Name"]
             or "fps_multi_kernel<1024, 20>" in r["Kernel_Name"]]
    if len(marks) < nsteps + 1:
        raise SystemExit("not enough steps in trace (%d markers)" % len(marks))
    lo, hi = marks[-nsteps - 1], marks[-1]
    sel = rows[lo:hi]
    wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / nsteps
    tot = defaultdict(float)
    cnt = defaultdict(int)
    cats = defaultdict(float)
    for r in sel:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        k = short(r["Kernel_Name"])
        tot[k] += d
        cnt[k] += 1
        cats[category(r["Kernel_Name"])] += d
    total = sum(tot.values())
    print("steady state over the last %d steps: wall %.2f ms/step, kernel time %.2f ms/step, %d launches/step"
          % (nsteps, wall, total / nsteps, len(sel) // nsteps))
    for c, t in sorted(cats.items(), key=lambda kv: -kv[1]):
        print("  %-32s %8.2f ms/step  %5.1f%%" % (c, t / nsteps, 100 * t / total))
    print("%-62s %8s %10s %9s %6s" % ("kernel", "calls/st", "ms/step", "avg us", "%"))
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
        print("%-62s %8.1f %10.3f %9.1f %6.2f" % (k, cnt[k] / nsteps, t / nsteps, 1e3 * t / cnt[k], 100 * t / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 30)
