#!/usr/bin/env python3
"""Per-kernel-class roofline table from a rocprofv3 `--kernel-trace --stats` summary of `python3 bench.py` (DiT-XL/2, batch 32).

    tools/kernel_table.py profiles/rNN_bench_kernel_stats.csv [--steps N] [--md out.md] [--json out.json]

Every `frac` is reproducible from the csv: frac = algorithmic FLOP (or bytes) per launch / AverageNs / peak, with the per-launch
work below (DiT-XL/2 256 px, B = 32: M = 8192 token rows, D = 1152, F = 4608, T = 256, 16 heads of 72).
A kernel name shared by several shapes is priced at the mean of the shapes the engine (csrc/dit_engine.hip) sends to it.
Peaks: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E (MI355X_MICROARCH.md).
"""
import argparse
import csv
import json
import re

M, D, F, T, H, HD, B, L = 8192, 1152, 4608, 256, 16, 72, 32, 28
NPAR = 674_834_720 + 0          # trainable parameters of DiT-XL/2 (arena padding ignored)
NADA = (6 * L + 2) * D * D      # the adaLN_modulation weights of all blocks + final layer, one [NM][D] matrix
G = lambda m, n, k: 2.0 * m * n * k
QKV, PROJ, FC1, FC2 = G(M, 3 * D, D), G(M, D, D), G(M, F, D), G(M, D, F)
ATT_F = 4.0 * T * T * HD * H * B            # QK^T + PV
MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12

# (regex on the kernel name, class label, bound, work per launch [FLOP or bytes], note)
RULES = [
    (r"k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2, 2, true>", "wgrad qkv/fc1 + bias row sums (192x192, 3 slots)", "mfma", (QKV + FC1) / 2, "mean of qkv 65.2 / fc1 87.0 GFLOP"),
    (r"k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2, 2(, false)?(, \d)?>", "wgrad qkv/proj/fc1/fc2 (192x192, 3 slots, loader waves)", "mfma", None, "mean of 65.2 / 21.7 / 87.0 / 87.0 GFLOP"),
    (r"k_gemm_pipe<8, 1, 2, 9, false, true, 0, 2, 3", "dgrad qkv/proj/fc1 -> 1152 wide (256x144)", "mfma", (QKV + PROJ + FC1) / 3, "mean of 65.2 / 21.7 / 87.0"),
    (r"k_gemm_pipe<4, 2, 4, 6, false, true, [48], 1, 2", "dgrad fc2 + GELU' (256x192; round 6: GELU' read as one byte per element)", "mfma", FC2, ""),
    (r"k_gemm_pipe<4, 2, 4, 6, false, false, [27], 1, 2", "fwd fc1 + GELU (256x192; round 6: second output = GELU' as one byte per element)", "mfma", FC1, ""),
    (r"k_gemm_pipe<8, 1, 2, 9, false, false, 0, 2, 3", "fwd qkv (256x144)", "mfma", QKV, ""),
    (r"k_gemm_pipe<8, 1, 2, 9, false, false, 3, 2, 3", "fwd proj/fc2 + gated residual (256x144)", "mfma", (PROJ + FC2) / 2, "mean of 21.7 / 87.0"),
    (r"k_attn_fwd", "attention forward", "mfma", ATT_F, "4 T^2 hd per (batch, head)"),
    (r"k_attn_bwd_fused", "attention backward (fused dQ, dK, dV)", "mfma", 2.5 * ATT_F, "5 products"),
    (r"k_attn_bwd_dq", "attention backward dQ (recomputes S, dP)", "mfma", 1.5 * ATT_F, "3 products of the 5 algorithmic ones"),
    (r"k_attn_bwd_dkv", "attention backward dK, dV (recomputes S, dP)", "mfma", 1.0 * ATT_F, "remaining 2 of the 5 algorithmic products"),
    (r"k_masked_clip_adam", "mask -> clip -> AdamW (+EMA, +bf16 shadow), flat ranges (round 3: many launches of different length)", "hbm", None,
     "31 B/param forget stage, 38 B/param remain stage; bytes per launch vary: see bench.py's others.hbm for the timed remain-stage sweep"),
    (r"k_adam_lowrank", "AdamW of the adaLN matrix, gradient formed from its two factors", "hbm", 30.5 * NADA, "27 B/param forget stage, 34 remain stage"),
    (r"k_sumsq_lowrank", "masked sum of squares of the rank-(batch) adaLN gradient", "hbm", 1.0 * NADA, "mask byte only; 32 FMA per element"),
    (r"k_sumsq_masked", "masked sum of squares (clip norm), flat ranges", "hbm", None, "g fp32 + mask byte"),
    (r"k_gemm8", "fp8 (e4m3) forward GEMM (config 5)", "mfma", None, ""),
    (r"k_row_bwd<true, true(, \w+)?>", "LN backward + gate backward (fused)", "hbm", 170e6, "DESIGN.md section 4"),
    (r"k_row_bwd<true, false(, \w+)?>", "LN backward", "hbm", 113e6, ""),
    (r"k_row_bwd<false, true(, \w+)?>", "gate backward", "hbm", 94e6, ""),
    (r"k_ln_mod_fwd", "LayerNorm + modulate forward", "hbm", 56.6e6, "x fp32 in, bf16 out"),
    (r"k_colsum_partial", "column sums (bias gradients)", "hbm", None, "round 1: 75 MB (fc1) / 57 MB (qkv) per launch"),
    (r"k_ema", "EMA of frozen parameters", "hbm", None, ""),
]


def classify(name, wgrad_plain_flops):
    for rx, label, bound, work, note in RULES:
        if re.search(rx, name):
            if label.startswith("wgrad qkv/proj/fc1/fc2"):
                work = wgrad_plain_flops
            return label, bound, work, note
    return None, None, None, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--md")
    ap.add_argument("--json")
    ap.add_argument("--bsum-build", action="store_true",
                    help="debug-knob build with the bias row sums inside the qkv / fc1 weight gradients: the plain kernel then runs proj / fc2 only")
    a = ap.parse_args()
    plain = (PROJ + FC2) / 2 if a.bsum_build else (QKV + PROJ + FC1 + FC2) / 4
    rows = []
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            label, bound, work, note = classify(r["Name"], plain)
            avg_ns = float(r["AverageNs"])
            frac = ach = None
            if work:
                ach = work / (avg_ns * 1e-9)
                frac = ach / (MFMA_PEAK if bound == "mfma" else HBM_PEAK)
            rows.append(dict(kernel=r["Name"], cls=label or "(other)", calls=int(r["Calls"]), avg_us=avg_ns / 1e3,
                             pct=float(r["Percentage"]), bound=bound, work_per_launch=work, achieved=ach, frac=frac, note=note))
    rows.sort(key=lambda x: -x["pct"])
    lines = ["| % GPU time | launches | avg us | class | bound | work / launch | achieved | frac of peak |", "|---|---|---|---|---|---|---|---|"]
    for r in rows[:24]:
        if r["work_per_launch"]:
            w = f"{r['work_per_launch'] / 1e9:.1f} GFLOP" if r["bound"] == "mfma" else f"{r['work_per_launch'] / 1e6:.0f} MB"
            ach = f"{r['achieved'] / 1e12:.0f} TFLOP/s" if r["bound"] == "mfma" else f"{r['achieved'] / 1e12:.2f} TB/s"
            fr = f"{r['frac']:.3f}"
        else:
            w = ach = fr = "-"
        cls = r["cls"] if r["cls"] != "(other)" else r["kernel"][:60]
        lines.append(f"| {r['pct']:.2f} | {r['calls']} | {r['avg_us']:.1f} | {cls} | {r['bound'] or '-'} | {w} | {ach} | {fr} |")
    out = "\n".join(lines)
    print(out)
    dom = rows[0]
    print(f"\ndominant by GPU time: {dom['kernel']}  ({dom['pct']:.2f} %), frac = {dom['frac']}")
    if a.md:
        open(a.md, "w").write(f"Per-kernel-class table of `{a.csv}` (tools/kernel_table.py; top 24 by GPU time)\n\n" + out + "\n")
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
