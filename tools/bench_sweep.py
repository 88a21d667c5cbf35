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

# workgroup-count sweep of the update kernel alone (sfron_masked_clip_adam_wg): fewer, longer-lived workgroups
import ctypes
from sfron._lib import check, ptr, stream_ptr
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
L = _lib.lib()
def one(W, with_ema):
    check(L.sfron_masked_clip_adam_wg(ptr(p), ptr(g), None, ptr(m), ptr(v), None if with_ema else ptr(mask), None, n, 0.9, 0.999, 1e-8,
                                      1e-4, 1.0, 1.0, ptr(w16), ptr(ema) if with_ema else None, 0.9999, 1 if with_ema else 0, W, stream_ptr()), "adam_wg")
for with_ema, bpp in ((False, 31.0), (True, 38.0)):
    for W in (0, 1536, 1024, 768, 512, 384, 320, 256, 192):
        for _ in range(2): one(W, with_ema)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(6): one(W, with_ema)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 6
        print(f"  update only, {'remain' if with_ema else 'forget'} form, W={W:5d}: {dt * 1e3:.2f} ms  {bpp * n / dt / 1e12:.2f} TB/s")
