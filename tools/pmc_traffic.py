#!/usr/bin/env python3
"""HBM-side traffic per launch of the three kernels bench.py reports a roofline for, from two rocprofv3 --pmc passes over
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (FETCH_SIZE and WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes;
gfx950: FETCH_SIZE counts half of a wide coalesced read stream -> x2, WRITE_SIZE exact, both in KB).
  on the GPU box, from the repo root:  python3 tools/pmc_traffic.py gpurun_out/pmc_r03 rNN   -> gpurun_out/pmc_r03/rNN_{wgrad,fc1,sweep}_traffic.json
Each JSON carries `csrc_sha` (tools/csrc_sha.py) and the kernel name: bench.py emits the figure only for the same sources."""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from csrc_sha import csrc_sha
out, tag = os.path.abspath(sys.argv[1]), sys.argv[2]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", os.path.join(out, c), "--", "python3", os.path.join(ROOT, "bench.py"),
                    "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs"], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
agg = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        a = agg[(r["Counter_Name"], r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
M, D, F = 8192, 1152, 4608
nt = 674_834_720
probes = {"wgrad": ("k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2, 2", 0.25 * sum(2 * M * (n + k) + 4 * n * k for n, k in ((3 * D, D), (D, D), (F, D), (D, F)))),
          # round 6: fc1 + GELU writes its second output as ONE byte per element (SFRON_EPI_GELU_Q): X, W1, h (bf16) + codes (u8)
          "fc1": ("k_gemm_pipe<4, 2, 4, 6, false, false, 7, 1, 2", 2 * (M * D + F * D + M * F) + M * F),
          "sweep": ("k_masked_clip_adam", None)}
sha = csrc_sha()
# the parameter sweeps of a whole step (norm pre-pass + forget-stage AdamW + remain-stage AdamW / EMA, flat and rank-R kernels): totals
# over the run / 3 steps (1 warm-up + 2)
n_ada = (6 * 28 + 2) * D * D
n_gemm = 28 * 12 * D * D      # the block GEMM weights: since round 5 their share of the clip norm is formed in the weight-gradient GEMMs' epilogues
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for (c, k), (sm, n) in agg.items():
    if any(x in k for x in ("k_masked_clip_adam", "k_adam_lowrank", "k_sumsq_masked", "k_sumsq_lowrank", "k_sumsq_ranges")):
        tot[c] += sm
# steps in the profiled run: NOT the 3 of the command line -- when the timed steps split their remain-stage sweep across the step boundary,
# bench.py appends two more steps to time that sweep whole (round 3 divided by 3 and reported 5/3 of the traffic: 78.7 GB against 47.2 GB
# algorithmic, which would have been more than the HBM peak).  k_adam_lowrank runs exactly twice per step (forget + remain stage).
n_lowrank = max([n for (c, k), (sm, n) in agg.items() if c == "FETCH_SIZE" and "k_adam_lowrank" in k] or [0])
n_steps = n_lowrank / 2.0 if n_lowrank else 3.0
json.dump({"kernel": "k_sumsq_ranges + k_sumsq_masked + k_sumsq_lowrank + k_masked_clip_adam + k_adam_lowrank, all launches of one SFR-on step",
           "steps_in_run": n_steps,
           "fetch_raw_bytes_per_step": tot["FETCH_SIZE"] * 1024.0 / n_steps, "write_bytes_per_step": tot["WRITE_SIZE"] * 1024.0 / n_steps,
           "traffic_bytes_per_step": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / n_steps,
           "algorithmic_bytes_per_step": (31 + 38) * nt - (4 + 4) * n_ada + 5 * (nt - n_gemm - n_ada), "csrc_sha": sha,
           "note": "same passes; totals over the run / its step count (launches of k_adam_lowrank / 2); FETCH_SIZE x2 (the kernels' reads are "
                   "16-B-per-lane streams but for the 1-B-per-parameter mask), WRITE_SIZE exact.  Algorithmic: forget AdamW 31 B/param, remain AdamW + EMA "
                   "38, minus the gradient reads of the adaLN matrix (formed inside the sweep from its factors), plus the norm pre-pass (g, mask: 5 B) "
                   "over what is neither a block GEMM weight (summed in the weight-gradient GEMMs' epilogues) nor the adaLN matrix (summed on the "
                   "matrix core from its factors by a GEMM launch that is not in this kernel list)"},
          open(os.path.join(out, f"{tag}_sweep_traffic.json"), "w"), indent=1)
probes.pop("sweep")
with open(os.path.join(out, f"{tag}_pmc_hbm_traffic_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["counter", "kernel", "launches", "avg_value_KB"])
    for (c, k), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        w.writerow([c, k[:140], n, round(s / n, 1)])
for name, (sub, alg) in probes.items():
    fk = [(k, v) for (c, k), v in agg.items() if c == "FETCH_SIZE" and sub in k]
    wk = [(k, v) for (c, k), v in agg.items() if c == "WRITE_SIZE" and sub in k]
    if len(fk) != 1 or len(wk) != 1:
        print("skip", name, [k for k, _ in fk]); continue
    fa, wa = fk[0][1][0] / fk[0][1][1], wk[0][1][0] / wk[0][1][1]
    json.dump({"kernel": fk[0][0], "launches": fk[0][1][1], "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa,
               "traffic_bytes_per_launch": (2 * fa + wa) * 1024.0, "algorithmic_bytes_per_launch": alg, "csrc_sha": sha,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on `python3 bench.py --steps 2 --warmup 1 "
                       "--no-cpu-baseline`; gfx950 correction: FETCH_SIZE x2, WRITE_SIZE exact; KB; mean over all launches of the kernel"},
              open(os.path.join(out, f"{tag}_{name}_traffic.json"), "w"), indent=1)
    print(name, fk[0][0][:60], "traffic MB", (2 * fa + wa) / 1024, "algorithmic MB", (alg or 0) / 1e6)
