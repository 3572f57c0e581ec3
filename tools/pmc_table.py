"""Per-kernel means of the counters in a rocprofv3 --pmc csv directory."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}, "n=%d" % len(next(iter(d.values()))))
