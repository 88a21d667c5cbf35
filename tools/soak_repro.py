#!/usr/bin/env python3
"""GPU: the headline runner (DiT-XL/2, batch 32, every schedule switch as bench.py sets it) for N steps, TWICE from the same weights and batches:
the parameter / moment / EMA arenas must come out bit-identical -- every overlap of the step (sweeps beside the forward pass and across the
step boundary, weight gradients and the clip norm's adaLN share on their own streams) is ordered by events, and a missing one shows here as
a difference.  tools/soak_repro.py [steps=60]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import data, diffusion, dit, step

dev = torch.device("cuda:0")


def run(N):
    torch.manual_seed(0)
    model = dit.DiT_models["DiT-XL/2"](input_size=32, num_classes=1000, batch_size=32, device=dev)
    torch.manual_seed(1234)
    model.initialize_weights()
    dit.randomize_zero_init(model, std=0.02, seed=1)
    model.train()
    eng = model.engine
    gm = torch.Generator().manual_seed(0)
    mask_arena = (torch.rand(eng.n_trainable, generator=gm) < 0.5).to(torch.uint8).to(dev)
    runner = step.DiTSFRon(model, diffusion.create_diffusion("", device=dev), lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999,
                           mask=None, unlearn_loss="ga", forget_class=207)
    runner.mask_arena = mask_arena
    runner.opt.mask = mask_arena
    runner.sweep_across_steps = True
    kw = dict(global_batch=32, num_classes=1000, forget_class=207)
    batches = [tuple({k: v.to(dev) for k, v in data.synthetic_batch(5, i, s, **kw).items()} for s in ("forget", "remain")) for i in range(4)]
    for i in range(N):
        out = runner.step(*batches[i % 4])
    runner.sync_sweep()
    torch.cuda.synchronize()
    runner.guard.poll(block=True)
    nt = eng.n_trainable
    sums = [t.view(torch.int32).to(torch.int64).sum().item() for t in (eng.params[:nt], runner.opt.m[:nt], runner.opt.v[:nt], runner.ema[:nt])]
    loss = (out["forget_mse"].mean().item(), out["remain_mse"].mean().item())
    eng.close()
    return sums, loss


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    a = run(N)
    b = run(N)
    print(f"{N} steps twice: checksums (params, m, v, ema) {a[0]} vs {b[0]}; last losses {a[1]} vs {b[1]}")
    print("BIT-IDENTICAL" if a == b else "DIFFERENT")
    sys.exit(0 if a == b else 1)
