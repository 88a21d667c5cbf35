#!/usr/bin/env python3
"""The parameter sweep alone at DiT-XL/2 size (674.8 M parameters): forget-stage form (mask + clip, 31 B/param) and remain-stage form
(+ EMA, 38 B/param), at the HBM roofline (5.7 TB/s measured: 72 % of the 8 TB/s spec, 90 % of the 6.3 TB/s a float4 copy reaches; nontemporal and 2x-unrolled variants measured the same within 1 %).   python tools/bench_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sfron
from sfron import _lib, sweep

n = 674_834_720
dev = "cuda"
p = torch.randn(n, device=dev) * 0.02; g = torch.randn(n, device=dev) * 1e-3
mask = (torch.rand(n, device=dev) < 0.5).to(torch.uint8); w16 = torch.empty(n, dtype=torch.bfloat16, device=dev); ema = p.clone()
opt = sweep.FlatAdam(p, g, lr=1e-4, mask=mask, w_bf16=w16)
def run(kind, it=6):
    for i in range(2 + it):
        if i == 2: torch.cuda.synchronize(); t0 = time.time()
        if kind == "forget": opt.step(max_norm=1.0, use_mask=True)
        else: opt.step(max_norm=None, use_mask=False, ema=ema, ema_decay=0.9999, ema_mode=1)
    torch.cuda.synchronize(); return (time.time() - t0) / it
for kind, bpp in (("forget", 36.0), ("remain", 38.0)):
    dt = run(kind)
    print(f"{kind}: {dt * 1e3:.2f} ms  ({bpp * n / dt / 1e12:.2f} TB/s algorithmic incl. the norm pre-pass)")
