#!/usr/bin/env python3
"""GEMM time with cache-hot operands (same buffers every call) vs the real step's access pattern (each layer its own
weights and its own saved-activation output, touched once per pass). GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
HINT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
MODE = sys.argv[2] if len(sys.argv) > 2 else "all"      # all | fwd | wgrad
M, D, F, L = 8192, 1152, 4608, 28
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
def run(name, N, K, epi=_lib.EPI_BF16, rotate_w=True, rotate_c=True, rotate_a=True):
    As = [rnd(M, K) for _ in range(L if rotate_a else 1)]
    Ws = [rnd(N, K) for _ in range(L if rotate_w else 1)]
    Cs = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(L if rotate_c else 1)]
    Xs = [torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(L if rotate_c else 1)] if epi == _lib.EPI_GELU else None
    def one(i):
        kw = dict(aux=Xs[i % len(Xs)]) if Xs else {}
        ops.gemm(As[i % len(As)], Ws[i % len(Ws)], M, N, K, epilogue=epi, c_bf16=Cs[i % len(Cs)], tile_hint=HINT, **kw)
    for i in range(L): one(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(4):
        for i in range(L): one(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / (4 * L) * 1e3
    print(f"{name:34s} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for nm, N, K, epi in () if MODE == "wgrad" else (("qkv", 3 * D, D, _lib.EPI_BF16), ("fc1+gelu", F, D, _lib.EPI_GELU), ("fc2", D, F, _lib.EPI_BF16)):
    run(f"{nm} hot", N, K, epi, False, False, False)
    run(f"{nm} rotate W", N, K, epi, True, False, False)
    run(f"{nm} rotate all", N, K, epi, True, True, True)


# ---- weight gradient dW[N,K] = dY[M,N]^T X[M,K]: X is a saved activation (cold), dY was just written
def run_w(name, N, K, rotate):
    dYs = [rnd(M, N) for _ in range(L if rotate else 1)]; Xs = [rnd(M, K) for _ in range(L if rotate else 1)]
    Cs = [torch.empty(N, K, dtype=torch.float32, device=DEV) for _ in range(L if rotate else 1)]
    def one(i):
        ops.gemm(dYs[i % len(dYs)], Xs[i % len(Xs)], N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cs[i % len(Cs)], tile_hint=HINT)
    for i in range(L): one(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(4):
        for i in range(L): one(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / (4 * L) * 1e3
    print(f"{name:34s} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for nm, N, K in () if MODE == "fwd" else (("wgrad qkv", 3 * D, D), ("wgrad fc1", F, D)):
    run_w(f"{nm} hot", N, K, False); run_w(f"{nm} rotate all", N, K, True)
