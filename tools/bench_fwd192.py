#!/usr/bin/env python3
"""Experiment: the forward layout on the 192 x 192 tile (4 x 2 waves of 48 x 96, three slots) with and without loader waves, against the
256 x 144 tile (8 x 1 waves of 32 x 144) the forward GEMMs use -- M = 8064 = 42 x 192 rows for the 192-row tile (252 / 756 tiles: one / three rounds, as 256 / 768 at M = 8192).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; D, F = 1152, 4608
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
L = _lib.lib()
for name, N, K in (("qkv-like", 3 * D, D), ("proj-like", D, D), ("fc2-like", D, F)):
    for M, hint, lw, label in ((8192, 62, 0, "256x144 8x1 shared waves"), (8064, 55, 0, "192x192 4x2 shared waves"), (8064, 55, 7, "192x192 4x2 loader waves")):
        A, B = rnd(M, K), rnd(N, K); C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        L.sfron_gemm_loader_waves(lw)
        t = timeit(lambda: ops.gemm(A, B, M, N, K, c_bf16=C, tile_hint=hint))
        print(f"{name:10s} M {M} {label:26s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s", flush=True)
L.sfron_gemm_loader_waves(4)
