#!/usr/bin/env python3
"""top N rows of a rocprofv3 kernel_stats.csv: tools/top_kernels.py <kernel_stats.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{float(r['Percentage']):5.2f}% calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
