#!/usr/bin/env python3
"""HBM-side traffic per launch of the convolution tiles, shape by shape, from two rocprofv3 --pmc passes over tools/bench_conv.py
(FETCH_SIZE and WRITE_SIZE separately; gfx950: FETCH_SIZE counts half of a wide coalesced read stream -> x2, WRITE_SIZE exact, KB).
  on the GPU box, from the repo root:  python3 tools/pmc_conv_traffic.py gpurun_out/pmc_conv > gpurun_out/conv_traffic.txt
A row per (kernel, grid): launches, mean bytes read / written per launch.  Shapes are told apart by their grid (workgroups x splits)."""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", os.path.join(out, c), "--", "python3", os.path.join(ROOT, "tools", "bench_conv.py"),
                    "--reps", "3"], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        k = r["Kernel_Name"]
        if "k_cgemm" not in k:
            continue
        wg = int(r["Workgroup_Size"]) if r.get("Workgroup_Size") else 0
        grid = int(r["Grid_Size"]) // wg if wg else 0
        a = agg[(k.split("(")[0][:48], grid)][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print(f"{'kernel':50s} {'workgroups':>10s} {'launches':>8s} {'read MB':>9s} {'written MB':>10s}")
for (k, grid), v in sorted(agg.items()):
    f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
    if not f[1] or not w[1]:
        continue
    print(f"{k:50s} {grid:10d} {f[1]:8d} {2 * f[0] / f[1] / 1024:9.1f} {w[0] / w[1] / 1024:10.1f}")
