#!/usr/bin/env python3
"""HBM-side traffic per launch of the GroupNorm kernels, shape by shape, from two rocprofv3 --pmc passes over tools/bench_gn.py (FETCH_SIZE and
WRITE_SIZE separately; gfx950: FETCH_SIZE counts half of a wide coalesced read stream -> x2, WRITE_SIZE exact, KB -- MI355X_MICROARCH.md).
  on the GPU box, from the repo root:  python3 tools/pmc_gn_traffic.py gpurun_out/pmc_gn > gpurun_out/gn_traffic.txt
A row per (kernel, workgroups): launches, mean MB read / written per launch."""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", os.path.join(out, c), "--", "python3", os.path.join(ROOT, "tools", "bench_gn.py"),
                    "--reps", "3"], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"]
        if "k_gn" not in k:
            continue
        wg = int(r["Workgroup_Size"]) if r.get("Workgroup_Size") else 0
        grid = int(r["Grid_Size"]) // wg if wg else 0
        name = [n for n in ("k_gn3_fwd", "k_gn3_bwd", "k_gn2_stats", "k_gn2_apply", "k_gn2_bwd_stats", "k_gn2_bwd_apply", "k_gn_fwd", "k_gn_bwd") if n in k]
        a = agg[(name[0] if name else k[:40], grid)][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"{'kernel':18s} {'workgroups':>10s} {'launches':>8s} {'read MB':>9s} {'written MB':>10s}")
for (k, grid), v in sorted(agg.items()):
    f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
    if not f[1] or not w[1]:
        continue
    print(f"{k:18s} {grid:10d} {f[1]:8d} {2 * f[0] / f[1] / 1024:9.1f} {w[0] / w[1] / 1024:10.1f}")
