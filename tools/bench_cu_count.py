#!/usr/bin/env python3
"""Is a K-step of the dgrad GEMM slower than the weight gradient's because of the kernel or because of the NUMBER OF ACTIVE CUs?
The same dgrad kernel (256 x 144 tile, loader waves, K = 4608: 72 K-steps) on 144 tiles (M = 4608) and on 256 tiles (M = 8192), and the
weight gradient (192 x 192 tile, 144 tiles, 128 K-steps) -- time per K-step.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; D, F = 1152, 4608
g = torch.Generator(device=DEV).manual_seed(0)
ZERO = len(sys.argv) > 1 and sys.argv[1] == "zeros"      # zero-filled operands: same instructions, far fewer toggling bits (power)
rnd = lambda *s: torch.zeros(*s, device=DEV, dtype=torch.bfloat16) if ZERO else torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for M in (2048, 4608, 8192, 16384):
    dY, W = rnd(M, F), rnd(F, D); C = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: ops.gemm(dY, W, M, D, F, b_t=True, c_bf16=C))
    tiles = (M // 256) * (D // 144)
    print(f"dgrad fc1-like M {M:5d}: {tiles:4d} tiles  {t:7.1f} us  {t / 72 / -(-tiles // 256):6.3f} us per K-step and round  {2.0 * M * D * F / t / 1e6:7.1f} TFLOP/s", flush=True)
M = 8192
dY, X = rnd(M, F), rnd(M, D); Cw = torch.empty(F, D, dtype=torch.float32, device=DEV)
t = timeit(lambda: ops.gemm(dY, X, F, D, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cw))
print(f"wgrad fc1 (144 tiles of 192 x 192, 128 K-steps): {t:7.1f} us  {t / 128:6.3f} us per K-step  {2.0 * M * D * F / t / 1e6:7.1f} TFLOP/s")
