#!/bin/bash
# One GPU call: per-kernel average durations of bench.py's step for several library builds (rocprofv3 --kernel-trace --stats), side by side.
#   tools/ab_prof.sh <outdir> <lib suffix|product> ...
ROOT=$(pwd); out=$ROOT/gpurun_out/$1; shift; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then unset SFRON_LIB_NAME; else export SFRON_LIB_NAME=libsfron_$v.so; fi
  rm -rf $out/st_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$v -- python3 $ROOT/tools/bench_ab.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $out/bench_$v.json 2> $out/bench_$v.err
  cp $(ls $out/st_$v/*/*kernel_stats.csv | head -1) $out/stats_$v.csv; rm -rf $out/st_$v
done
python3 - "$out" "$@" <<'PY'
import csv, sys
out, vs = sys.argv[1], sys.argv[2:]
tab = {}
for v in vs:
    for r in csv.DictReader(open(f"{out}/stats_{v}.csv")):
        tab.setdefault(r["Name"], {})[v] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
rows = sorted(tab.items(), key=lambda kv: -max(x[2] for x in kv[1].values()))[:28]
for name, d in rows:
    print(f"{name[:70]:70s} " + "  ".join(f"{v}: {d[v][0]:5d} x {d[v][1]:7.1f} us = {d[v][2]:7.2f} ms" if v in d else f"{v}: -" for v in vs))
PY
