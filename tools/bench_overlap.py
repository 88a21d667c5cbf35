#!/usr/bin/env python3
"""Feasibility of overlapping the optimizer sweep (HBM-bound) with forward GEMMs (cold operands): serial vs concurrent. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, sweep, _lib
DEV = "cuda:0"; M, D, F, L = 8192, 1152, 4608, 28
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
n = 450_000_000                                      # the 28 blocks' share of the arena
p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV) * 1e-3
mask = (torch.rand(n, device=DEV) < 0.5).to(torch.uint8); wbf = torch.empty(n, dtype=torch.bfloat16, device=DEV)
opt = sweep.FlatAdam(p, g, lr=1e-4, mask=mask, w_bf16=wbf)
As = [rnd(M, D) for _ in range(L)]; Ws = [rnd(F, D) for _ in range(L)]
Cs = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(L)]; Xs = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(L)]
def gemms():
    for i in range(L): ops.gemm(As[i], Ws[i], M, F, D, epilogue=_lib.EPI_GELU, c_bf16=Cs[i], aux=Xs[i])
side = torch.cuda.Stream()
def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
def serial(): opt.step(max_norm=None, use_mask=True); gemms()
def concurrent():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): opt.step(max_norm=None, use_mask=True)
    gemms()
    torch.cuda.current_stream().wait_stream(side)
print(f"28 cold fc1+GELU GEMMs alone : {timed(gemms):7.2f} ms")
print(f"sweep of 450M params alone   : {timed(lambda: opt.step(max_norm=None, use_mask=True)):7.2f} ms")
print(f"serial                       : {timed(serial):7.2f} ms")
print(f"concurrent (side stream)     : {timed(concurrent):7.2f} ms")
