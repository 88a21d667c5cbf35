#!/usr/bin/env python3
"""wgrad fc1 time vs the number of distinct operand sets cycled through (1 = hot ... 28 = the step's footprint). GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; M, D, F = 8192, 1152, 4608
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
N, K = F, D
sets = [(rnd(M, N), rnd(M, K), torch.empty(N, K, dtype=torch.float32, device=DEV)) for _ in range(28)]
for ns in (1, 2, 3, 4, 8, 16, 28):
    def one(i):
        dY, X, C = sets[i % ns]
        ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C)
    for i in range(28): one(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(112): one(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 112 * 1e3
    print(f"wgrad fc1, {ns:2d} operand sets ({ns*94.4:6.0f} MB): {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
