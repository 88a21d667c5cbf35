#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels at the DiT-XL/2 B=32 shape (random data). GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops
DEV = "cuda:0"
B, T, H, hd = 32, 256, 16, 72
if len(sys.argv) > 1: B, T, H, hd = map(int, sys.argv[1:5])
D = H * hd
g = torch.Generator(device=DEV).manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
d_o = torch.randn(B * T, D, device=DEV, generator=g).to(torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
from sfron import _lib
o, lse = ops.attn_fwd(qkv, B, T, H, hd)
fl = 4.0 * B * H * T * T * hd
for form in (4, 8, 16, 2):          # 2 = both query blocks of a head in one workgroup (round 6)
    _lib.lib().sfron_attn_fwd_form(form)
    ms = timeit(lambda: ops.attn_fwd(qkv, B, T, H, hd))
    print(f"attn fwd  B{B} T{T} H{H} hd{hd}, form {form}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
_lib.lib().sfron_attn_fwd_form(0)
ms = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hd))
print(f"attn bwd  (delta+dq+dkv)      : {ms*1e3:8.1f} us  {2.5*fl/ms/1e9:7.1f} TFLOP/s (2.5x fwd flops)")
