"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), gfx950 corrections
of MI355X_MICROARCH.md §HBM: both counters are in KiB; FETCH_SIZE tallies 128-B requests at 64 B for
wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics."""
import csv
import glob
import sys
from collections import defaultdict


def load(dirpath, counter):
    files = glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True)
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
            tot[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main(fetch_dir, write_dir, pattern):
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    for name in sorted(ft, key=lambda k: -ft[k]):
        if pattern not in name:
            continue
        n = fc[name]
        fetch = ft[name] * 1024 * 2 / n
        write = wt.get(name, 0.0) * 1024 / max(wc.get(name, 1), 1)
        print("%-70s launches %5d  fetch %10.3f MB (x2 corrected)  write %10.3f MB  total %10.3f MB per launch"
              % (name, n, fetch / 1e6, write / 1e6, (fetch + write) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "gb::")
