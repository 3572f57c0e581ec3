"""Compact view of a rocprofv3 --stats kernel_stats.csv: top kernels with short names + category totals."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name)
    name = name.replace("void ", "").replace("at::native::", "")
    return name[:60]


def category(name):
    if name.startswith("void gb::") or name.startswith("gb::"):
        return "own HIP kernels (gb::)"
    if name.startswith("Cijk") or "igemm" in name or "Conv" in name:
        return "GEMM/conv (rocBLAS/MIOpen)"
    if "BatchNorm" in name:
        return "MIOpen batchnorm"
    if "transpose" in name:
        return "MIOpen transposes"
    if "max_pool" in name:
        return "torch max_pool"
    if "nccl" in name.lower() or "rccl" in name.lower():
        return "RCCL"
    return "other torch kernels"


def main(path, steps=1.0, top=25):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    cats = {}
    for r in rows:
        cats[category(r["Name"])] = cats.get(category(r["Name"]), 0.0) + float(r["TotalDurationNs"])
    print("WHOLE PROCESS, warm-up included (MIOpen find-mode trial kernels, lazy initialisation, the CPU-baseline legs' GPU "
          "cross-checks): NOT the per-step picture - that is the steady-state file next to this one.")
    print("total kernel time %.2f ms  (%.2f ms per timed step if it were spread over the %g timed steps)"
          % (total / 1e6, total / 1e6 / steps, steps))
    for c, t in sorted(cats.items(), key=lambda kv: -kv[1]):
        print("  %-32s %8.2f ms/step  %5.1f%%" % (c, t / 1e6 / steps, 100 * t / total))
    print("%-62s %6s %10s %10s %6s" % ("kernel", "calls", "ms/step", "avg us", "%"))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print("%-62s %6s %10.3f %10.1f %6.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps,
                                                 float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 25)
