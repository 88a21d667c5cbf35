#!/usr/bin/env python3
"""Block GEMMs with cache-hot operands against the real step's access pattern (operands cycled through more than the 256 MB
Infinity Cache).  GPU only.  One script, three experiments (DESIGN.md section 6.2.1 quotes all of them):
    tools/bench_cold.py layers [tile_hint] [all|fwd|wgrad]   every block GEMM, same buffers each call vs one set per layer
    tools/bench_cold.py footprint                            wgrad fc1 time vs the number of distinct operand sets (1 ... 28)
    tools/bench_cold.py tiles                                forward layouts: 256x192 two-slot (hint 42) vs 256x144 three-slot (62), hot / cold"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib

EXP = sys.argv[1] if len(sys.argv) > 1 else "layers"


def exp_layers(argv):
    sys_argv = [sys.argv[0]] + argv
    DEV = "cuda:0"
    HINT = int(sys_argv[1]) if len(sys_argv) > 1 else 0
    MODE = sys_argv[2] if len(sys_argv) > 2 else "all"      # all | fwd | wgrad
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


def exp_footprint(argv):
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


def exp_tiles(argv):
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


if __name__ == "__main__":
    {"layers": exp_layers, "footprint": exp_footprint, "tiles": exp_tiles}[EXP](sys.argv[2:])
