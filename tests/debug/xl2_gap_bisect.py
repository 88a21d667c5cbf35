#!/usr/bin/env python3
"""Where does the held-out eps-MSE gap between the HIP path and the oracle at DiT-XL/2 come from, and what is it after 50 steps?
(test infrastructure: imports oracle/, lives under tests/.)

Part 1 (bisect, before any step): the oracle restated with ONE class of bf16 rounding points at a time (torch hooks on the oracle's own
modules, everything else fp32), run on the GPU in fp32 next to the HIP forward pass, on several held-out batches:
   W     block / embedder / adaLN / final Linear weights rounded to bf16 (the weight shadow the GEMMs read)
   A     the inputs of the block Linears rounded to bf16 (xmod1, o, xmod2, h: the GEMM A operands)
   QKV   the qkv Linear's output rounded to bf16 (what the attention kernel reads)
   P     softmax probabilities rounded to bf16 before P.V (the flash kernel's P operand)
   H     fc1's pre-activation rounded to bf16 before GELU is NOT applied (the kernel applies GELU to the fp32 accumulator): h only
   E     the GEMM inputs OUTSIDE the blocks: patch-embedding input, timestep frequencies, silu(h1), silu(c) into every adaLN Linear, final
         Linear input (the HIP path feeds all of them to its GEMMs as bf16)
   ALL   all of the above together = the model of the HIP path's rounding points
For each: held-out mse minus the fp32 oracle's, per batch (mean, std over batches) -- a BIAS shows as a mean that does not average out.

Part 2: 50 SFR-on iterations at batch 4 (hyper-parameters of tests/test_gpu_baseline_shapes.py::test_xl2_ten_sfron_iterations_vs_oracle),
oracle on the GPU in fp32; held-out gap at 0 / 10 / 20 / 50 steps.

  python3 tests/debug/xl2_gap_bisect.py [--steps 50] [--batches 4] [--skip-bisect]
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda:0"


from oracle.bf16_ref import OperandRounding as Rounding, bf, sfron_step_bf16_operands  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--skip-bisect", action="store_true")
    ap.add_argument("--sensitivity", action="store_true",
                    help="Part 3: how far do two fp32-ORACLE trajectories drift apart over the same 50 steps when they differ by (a) one fp32 ulp "
                         "of every weight at step 0, (b) bf16 rounding of weights + GEMM inputs in the forward pass (fake-quant oracle)")
    args = ap.parse_args()
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    from test_gpu_baseline_shapes import _pair
    B = 4
    ref, model = _pair("DiT-XL/2", B, seed=51, std=0.02)
    tab = dref.DiffusionTables(1000)
    ref_cpu_out = None
    hbs = [data.synthetic_batch(24 + i, 0, "remain", global_batch=16, num_classes=1000, forget_class=207) for i in range(args.batches)]
    # the oracle on the CPU (the contract) against the same modules on the GPU in fp32: they must agree far below the gaps studied here
    ref.eval()
    with torch.no_grad():
        t_cpu = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), hbs[0]["x0"][:4], hbs[0]["t"][:4], dict(y=hbs[0]["y"][:4]), hbs[0]["noise"][:4])
    ref.to(DEV)
    with torch.no_grad():
        g = {k: v[:4].to(DEV) for k, v in hbs[0].items()}
        t_gpu = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])
    print(f"oracle fp32 CPU vs the same modules on the GPU (4 samples): mse {t_cpu['mse'].mean().item():.7f} vs {t_gpu['mse'].mean().item():.7f}, "
          f"max per-sample |diff| {(t_cpu['mse'] - t_gpu['mse'].cpu()).abs().max().item():.2e}", flush=True)
    d = diffusion.create_diffusion("")

    def held_ref(hb):
        ref.eval()
        with torch.no_grad():
            g = {k: v.to(DEV) for k, v in hb.items()}
            return dref.training_losses(tab, lambda x, t, y: ref(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])["mse"].mean().item()

    def held_hip(hb, m=None):
        m = m or model
        m.set_batch_size(16)
        m.eval()
        with torch.no_grad():
            g = {k: v.to(DEV) for k, v in hb.items()}
            out = m(d.q_sample(g["x0"], g["t"], g["noise"]), g["t"], g["y"])
        mse, _, _ = d.loss_fwd_bwd(out.contiguous(), g["x0"], g["t"], g["noise"], 1.0)
        m.train()
        return mse.mean().item()

    if not args.skip_bisect:
        base = [held_ref(hb) for hb in hbs]
        hip = [held_hip(hb) for hb in hbs]
        rows = [("HIP path", hip)]
        for kinds in (["W"], ["A"], ["QKV"], ["P"], ["E"], ["W", "A"], ["W", "A", "QKV", "P", "E"]):
            with Rounding(ref, kinds):
                rows.append(("oracle + bf16 " + "+".join(kinds), [held_ref(hb) for hb in hbs]))
        print(f"\nheld-out eps-MSE of the fp32 oracle per batch: {['%.5f' % b for b in base]}")
        print("| variant | mse - fp32 oracle, per held-out batch | mean | std |")
        print("|---|---|---|---|")
        for name, vals in rows:
            dv = torch.tensor([v - b for v, b in zip(vals, base)], dtype=torch.float64)
            print(f"| {name} | {' '.join('%+.2e' % x for x in dv.tolist())} | {dv.mean().item():+.2e} | {dv.std().item() if len(dv) > 1 else 0:.1e} |", flush=True)

    if args.sensitivity:
        del model
        ref.cpu()
        del ref
        torch.cuda.empty_cache()
        sensitivity(args.steps, hbs)
        return
    if args.steps <= 0:
        return
    # ---- Part 2: 50 iterations, both paths from the same state
    model.set_batch_size(B)
    model.train()
    ref.train()
    gm = torch.Generator().manual_seed(52)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, unlearn_loss="ga", forget_class=207)
    orc = sfron_ref.DiTSfronOracle(ref, tab, mask={k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in mask.items()}, **hp)
    runner = step.DiTSFRon(model, d, mask=mask, **hp)
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)

    def report(it):
        hs, rs = [held_hip(hb) for hb in hbs], [held_ref(hb) for hb in hbs]
        ref.train()
        model.set_batch_size(B)
        gaps = [h - r for h, r in zip(hs, rs)]
        print(f"after {it:3d} steps: held-out mse oracle {sum(rs) / len(rs):.5f}  HIP {sum(hs) / len(hs):.5f}  gap per batch "
              f"{' '.join('%+.2e' % x for x in gaps)}  |mean gap| {abs(sum(gaps) / len(gaps)):.2e}", flush=True)
    report(0)
    for it in range(args.steps):
        f, r = data.synthetic_batch(23, it, "forget", **kw), data.synthetic_batch(23, it, "remain", **kw)
        fd, rd = {k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()}
        orc.step({k: v.long() if k == "drop" else v for k, v in fd.items()}, {k: v.long() if k == "drop" else v for k, v in rd.items()})
        runner.step(fd, rd)
        if it + 1 in (1, 5, 10, 20, 30, 50, args.steps):
            report(it + 1)


def sensitivity(steps, hbs):
    """Three oracles on the GPU from the same initial weights over the same batches: fp32; fp32 with every weight moved by one ulp at
    step 0; fp32 masters with bf16-rounded weights and GEMM inputs in every forward pass (W + A + E: what ANY bf16-operand implementation
    of this model does).  Prints each one's held-out gap against the first."""
    import copy
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data
    from test_gpu_baseline_shapes import _pair
    B = 4
    ref, model = _pair("DiT-XL/2", B, seed=51, std=0.02)
    del model
    tab = dref.DiffusionTables(1000)
    gm = torch.Generator().manual_seed(52)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5).to(DEV) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, unlearn_loss="ga", forget_class=207)
    refs = {"fp32": ref.to(DEV)}
    refs["fp32 + 1 ulp"] = copy.deepcopy(ref)
    with torch.no_grad():
        for p in refs["fp32 + 1 ulp"].parameters():
            if p.requires_grad:
                p.copy_(torch.nextafter(p, torch.full_like(p, float("inf"))))
    refs["bf16 W+A+E forward"] = copy.deepcopy(ref)
    refs["bf16 W+A+E forward + 1 ulp"] = copy.deepcopy(refs["fp32 + 1 ulp"])
    orcs = {k: sfron_ref.DiTSfronOracle(m, tab, mask=mask, **hp) for k, m in refs.items()}
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)

    def held(m, fake):
        m.eval()
        vals = []
        with torch.no_grad():
            for hb in hbs:
                g = {k: v.to(DEV) for k, v in hb.items()}
                if fake:
                    with Rounding(m, ["W", "A", "E"]):
                        vals.append(dref.training_losses(tab, lambda x, t, y: m(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])["mse"].mean().item())
                else:
                    vals.append(dref.training_losses(tab, lambda x, t, y: m(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])["mse"].mean().item())
        m.train()
        return sum(vals) / len(vals)

    def report(it):
        base = held(refs["fp32"], False)
        line = f"after {it:3d} steps: fp32 oracle {base:.5f}"
        for k in list(refs)[1:]:
            line += f" | {k}: gap {held(refs[k], k.startswith('bf16')) - base:+.2e}"
        print(line, flush=True)
    print("\nPart 3: sensitivity of the 50-step trajectory itself (all three are the ORACLE, fp32 arithmetic on the GPU)")
    report(0)
    for it in range(steps):
        f, r = data.synthetic_batch(23, it, "forget", **kw), data.synthetic_batch(23, it, "remain", **kw)
        fd = {k: (v.long() if k == "drop" else v).to(DEV) for k, v in f.items()}
        rd = {k: (v.long() if k == "drop" else v).to(DEV) for k, v in r.items()}
        for k, o in orcs.items():
            if k.startswith("bf16"):
                sfron_step_bf16_operands(o, fd, rd)
            else:
                o.step(fd, rd)
        if it + 1 in (1, 5, 10, 20, 30, 50, steps):
            report(it + 1)


if __name__ == "__main__":
    main()
