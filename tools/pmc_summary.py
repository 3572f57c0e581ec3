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


def main(fetch_dir, write_dir, pattern, as_json=False):
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    out = {}
    for name in sorted(ft, key=lambda k: -ft[k]):
        if pattern not in name:
            continue
        n = fc[name]
        fetch = ft[name] * 1024 * 2 / n
        write = wt.get(name, 0.0) * 1024 / max(wc.get(name, 1), 1)
        if as_json:
            # key = the kernel's short name as bench.py's pmc_traffic() looks it up
            key = name.replace("gb::", "")
            key = key if key.startswith("fps_reg_kernel<1024, 20>") else key.split("<")[0]
            ent = out.setdefault(key, {"launches_profiled": 0, "fetch": 0.0, "write": 0.0})
            ent["launches_profiled"] += n
            ent["fetch"] += fetch * n
            ent["write"] += write * n
        else:
            print("%-70s launches %5d  fetch %10.3f MB (x2 corrected)  write %10.3f MB  total %10.3f MB per launch"
                  % (name, n, fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
    if as_json:
        import json
        res = {}
        for k, e in out.items():
            n = e["launches_profiled"]
            res[k] = {"launches_profiled": n, "fetch_bytes_per_launch_x2_corrected": e["fetch"] / n,
                      "write_bytes_per_launch": e["write"] / n, "hbm_bytes_per_launch": (e["fetch"] + e["write"]) / n}
            # the git blob hashes of the kernel's source files AS PROFILED: bench.py reports the traffic only while they
            # still match (a figure from a file is not a measurement of a changed kernel)
            try:
                import os
                sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                import bench
                src = bench.kernel_source_hashes(k)
                if src:
                    res[k]["sources"] = src
            except Exception:
                pass
        res["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                       "--warmup 1 --no-cpu-baseline; counters in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                       "(128-B requests tallied at 64 B); mean over all launches of the kernel (all template variants)")
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--json"]
    main(args[0], args[1], args[2] if len(args) > 2 else "gb::", "--json" in sys.argv)
