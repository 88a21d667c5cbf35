#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: tools/pmc_table.py <dir with *counter_collection.csv (searched recursively)> [kernel substring ...]
Prints one row per (kernel, counter): launches, mean value per launch.  Used for the SQ wait / busy breakdowns in DESIGN.md section 6."""
import collections, csv, glob, os, sys
root = sys.argv[1]
subs = sys.argv[2:]
agg = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if subs and not any(s in k for s in subs):
                continue
            a = agg[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
kernels = sorted({k for k, _ in agg})
for k in kernels:
    print(k[:150])
    for (kk, c), (s, n) in sorted(agg.items()):
        if kk == k:
            print(f"    {c:34s} launches {n:5d}  mean {s / n:16.1f}")
