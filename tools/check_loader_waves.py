#!/usr/bin/env python3
"""The loader-wave form of the three-slot GEMM tiles (sfron_gemm_loader_waves(4)) against the default form: same bits, and the time
of each block GEMM alone in both forms.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; M, D, F = 8192, 1152, 4608
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
cases = []
def fwd(name, N, K, epi=_lib.EPI_BF16):
    A, B = rnd(M, K), rnd(N, K); bias = torch.randn(N, device=DEV, generator=g)
    def run():
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); kw = {}
        if epi == _lib.EPI_GELU: kw = dict(aux=torch.empty(M, N, dtype=torch.bfloat16, device=DEV))
        ops.gemm(A, B, M, N, K, epilogue=epi, c_bf16=C, bias=bias, **kw); return (C,) + tuple(kw.values())
    cases.append((name, run))
def gate_res(name, N, K):
    A, B = rnd(M, K), rnd(N, K); x0 = torch.randn(M, N, device=DEV, generator=g)
    gate = torch.randn(32, 6 * N, device=DEV, generator=g); bias = torch.randn(N, device=DEV, generator=g)
    def run():
        x1 = torch.empty_like(x0); aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_GATE_RES, bias=bias, c_f32=x1, resid=x0, aux=aux, gate=gate[:, 2 * N:], ldgate=6 * N, tokens=256)
        return x1, aux
    cases.append((name, run))
def dgrad(name, N, K):
    dY, W = rnd(M, N), rnd(N, K)
    def run():
        C = torch.empty(M, K, dtype=torch.bfloat16, device=DEV); ops.gemm(dY, W, M, K, N, b_t=True, c_bf16=C); return (C,)
    cases.append((name, run))
def wgrad(name, N, K):
    dY, X = rnd(M, N), rnd(M, K)
    def run():
        C = torch.empty(N, K, dtype=torch.float32, device=DEV); ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C); return (C,)
    cases.append((name, run))
fwd("fwd qkv", 3 * D, D); gate_res("fwd proj + gate-res", D, D); gate_res("fwd fc2 + gate-res", D, F)
dgrad("dgrad qkv", 3 * D, D); dgrad("dgrad proj", D, D); dgrad("dgrad fc1", F, D)
wgrad("wgrad qkv", 3 * D, D); wgrad("wgrad proj", D, D); wgrad("wgrad fc1", F, D); wgrad("wgrad fc2", D, F)
L = _lib.lib()
bad = 0
for name, run in cases:
    L.sfron_gemm_loader_waves(0); ref = run(); t0 = timeit(run)
    L.sfron_gemm_loader_waves(4); out = run(); t1 = timeit(run)
    same = all(torch.equal(a, b) for a, b in zip(ref, out))
    bad += 0 if same else 1
    print(f"{name:24s} default {t0:7.1f} us   loader waves {t1:7.1f} us   {'bit-identical' if same else 'DIFFERENT'}", flush=True)
L.sfron_gemm_loader_waves(0)
sys.exit(1 if bad else 0)
