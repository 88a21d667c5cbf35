#!/usr/bin/env python3
"""Per-kernel averages of FETCH_SIZE / WRITE_SIZE from rocprofv3 --pmc runs (one counter per run).
usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_csv> <out_json> <probe kernel substring> [algorithmic bytes per launch]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads -> x2;
WRITE_SIZE exact; both in KB."""
import csv, json, sys, collections
fetch, write, out_csv, out_json, probe = sys.argv[1:6]
rows = []
agg = collections.defaultdict(lambda: [0.0, 0])
for path in (fetch, write):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = (r["Counter_Name"], r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "launches", "avg_value_KB"])
    for (c, k), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([c, k[:120], n, round(s / n, 1)])
fk = [(k, v) for (c, k), v in agg.items() if c == "FETCH_SIZE" and probe in k]
wk = [(k, v) for (c, k), v in agg.items() if c == "WRITE_SIZE" and probe in k]
assert len(fk) == 1 and len(wk) == 1, (fk, wk)
f_avg, w_avg = fk[0][1][0] / fk[0][1][1], wk[0][1][0] / wk[0][1][1]
M, N, K = 8192, 4608, 1152
alg = float(sys.argv[6]) if len(sys.argv) > 6 else 2 * (M * K + N * K + 2 * M * N)
json.dump({"kernel": fk[0][0], "FETCH_SIZE_KB_avg": f_avg, "WRITE_SIZE_KB_avg": w_avg,
           "traffic_bytes_per_launch": (2 * f_avg + w_avg) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on `python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline`; gfx950 correction: FETCH_SIZE counts 1/2 of a wide coalesced read stream (x2), WRITE_SIZE exact; units KB",
           "algorithmic_bytes_per_launch": alg}, open(out_json, "w"), indent=1)
print(open(out_json).read())
