#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels at DiT-XL/2 B=32 shapes. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, sweep
DEV = "cuda:0"
B, T, D, F = 32, 256, 1152, 4608
M = B * T
g = torch.Generator(device=DEV).manual_seed(0)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
x = torch.randn(M, D, device=DEV, generator=g)
mod = torch.randn(B, 6 * D, device=DEV, generator=g) * 0.1
dxm = torch.randn(M, D, device=DEV, generator=g).to(torch.bfloat16)
dx = torch.randn(M, D, device=DEV, generator=g)
br = torch.randn(M, D, device=DEV, generator=g).to(torch.bfloat16)
dh = torch.randn(M, F, device=DEV, generator=g).to(torch.bfloat16)
out, mean, rstd = ops.ln_modulate_fwd(x, mod[:, 3 * D:], mod[:, 4 * D:], 6 * D, T)
def rep(name, ms, mbytes): print(f"{name:22s} {ms*1e3:8.1f} us  {mbytes/ms/1e3:6.2f} TB/s ({mbytes:.0f} MB)")
rep("ln_modulate_fwd", timeit(lambda: ops.ln_modulate_fwd(x, mod[:, 3 * D:], mod[:, 4 * D:], 6 * D, T)), M * D * 6 / 1e6)
rep("ln_modulate_bwd", timeit(lambda: ops.ln_modulate_bwd(dxm, x, mean, rstd, mod[:, 4 * D:], 6 * D, T, dx, True)), M * D * 14 / 1e6)
rep("gate_bwd", timeit(lambda: ops.gate_bwd(dx, br, mod[:, 2 * D:], 6 * D, T)), M * D * 8 / 1e6)
rep("ln+gate bwd fused", timeit(lambda: ops.ln_gate_bwd(dxm, x, mean, rstd, mod[:, 4 * D:], 6 * D, T, dx, True, br, mod[:, 2 * D:], 6 * D)), M * D * 18 / 1e6)
rep("colsum bf16 [M,4D]", timeit(lambda: ops.colsum(dh)), M * F * 2 / 1e6)
n = 675_000_000
p = torch.randn(n, device=DEV); gr = torch.randn(n, device=DEV) * 1e-3
mask = (torch.rand(n, device=DEV) < 0.5).to(torch.uint8)
wbf = torch.empty(n, dtype=torch.bfloat16, device=DEV); ema = p.clone()
opt = sweep.FlatAdam(p, gr, lr=1e-4, mask=mask, w_bf16=wbf)
rep("forget stage sweep", timeit(lambda: opt.step(max_norm=1.0, use_mask=True), 5), n * 36 / 1e6)
rep("remain stage sweep+ema", timeit(lambda: opt.step(max_norm=None, ema=ema, ema_decay=0.9999, ema_mode=1), 5), n * 38 / 1e6)
