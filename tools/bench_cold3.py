#!/usr/bin/env python3
"""Forward-layout GEMMs at the DiT-XL/2 shapes, hot vs cold operands: 256x192 two-slot (hint 42) vs 256x144 three-slot (62). GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; L = 28; M = 8192
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
def run(name, N, K, hint, rotate):
    n = L if rotate else 1
    As = [rnd(M, K) for _ in range(n)]; Ws = [rnd(N, K) for _ in range(n)]
    Cs = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(n)]
    def one(i): ops.gemm(As[i % n], Ws[i % n], M, N, K, c_bf16=Cs[i % n], tile_hint=hint)
    for i in range(L): one(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(4 * L): one(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / (4 * L) * 1e3
    print(f"{name:40s} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for nm, N, K in (("proj", 1152, 1152), ("qkv", 3456, 1152), ("fc1", 4608, 1152), ("fc2", 1152, 4608)):
    for rot in (False, True):
        for hint, lab in ((42, "256x192 2-slot"), (62, "256x144 3-slot")):
            run(f"{nm:4s} {'cold' if rot else 'hot '} {lab}", N, K, hint, rot)
