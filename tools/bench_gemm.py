#!/usr/bin/env python3
"""Micro-benchmark of sfron_gemm_bf16 on the DiT-XL/2 B=32 shapes (random bf16 data). GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib
_lib = ab_lib.select()
if os.environ.get("SFRON_DBG_LIB"):       # the debug-knob build (SFRON_GEMM_SAME_TILE ...): `make -C .../csrc dbg`
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsfron_dbg.so")
from sfron import ops
if os.environ.get("SFRON_LOADER_WAVES"): _lib.lib().sfron_gemm_loader_waves(int(os.environ["SFRON_LOADER_WAVES"]))

DEV = "cuda:0"
HINT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M, D, F = 8192, 1152, 4608
g = torch.Generator(device=DEV).manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=DEV, generator=g)).to(torch.bfloat16)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

cases = []
def fwd(name, N, K, epi=_lib.EPI_BF16):
    A, B = rnd(M, K), rnd(N, K)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    kw = {}
    if epi == _lib.EPI_GELU: kw = dict(aux=torch.empty(M, N, dtype=torch.bfloat16, device=DEV))
    cases.append((name, 2.0 * M * N * K, lambda: ops.gemm(A, B, M, N, K, epilogue=epi, c_bf16=C, tile_hint=HINT, **kw)))
def dgrad(name, N, K):   # dX[M,K] = dY[M,N] W[N,K]
    dY, W = rnd(M, N), rnd(N, K)
    C = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    cases.append((name, 2.0 * M * N * K, lambda: ops.gemm(dY, W, M, K, N, b_t=True, c_bf16=C, tile_hint=HINT)))
def wgrad(name, N, K):   # dW[N,K] = dY[M,N]^T X[M,K]
    dY, X = rnd(M, N), rnd(M, K)
    C = torch.empty(N, K, dtype=torch.float32, device=DEV)
    cases.append((name, 2.0 * M * N * K, lambda: ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C, tile_hint=HINT)))

fwd("fwd qkv  N3456 K1152", 3 * D, D); fwd("fwd proj N1152 K1152", D, D)
fwd("fwd fc1  N4608 K1152 gelu", F, D, _lib.EPI_GELU); fwd("fwd fc2  N1152 K4608", D, F)
dgrad("dgrad qkv  N3456->1152", 3 * D, D); dgrad("dgrad proj N1152->1152", D, D)
dgrad("dgrad fc1  N4608->1152", F, D); dgrad("dgrad fc2  N1152->4608", D, F)
wgrad("wgrad qkv  3456x1152", 3 * D, D); wgrad("wgrad proj 1152x1152", D, D)
wgrad("wgrad fc1  4608x1152", F, D); wgrad("wgrad fc2  1152x4608", D, F)
# epilogue variants used by the engine
def dgelu(name, N, K):
    dY, W = rnd(M, N), rnd(N, K)
    C = torch.empty(M, K, dtype=torch.bfloat16, device=DEV); aux = rnd(M, K)
    cases.append((name, 2.0 * M * N * K, lambda: ops.gemm(dY, W, M, K, N, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=C, aux=aux, tile_hint=HINT)))
def gate_res(name, N, K):
    A, B = rnd(M, K), rnd(N, K)
    x0 = torch.randn(M, N, device=DEV); x1 = torch.empty_like(x0); aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    gate = torch.randn(32, 6 * N, device=DEV); bias = torch.randn(N, device=DEV)
    cases.append((name, 2.0 * M * N * K, lambda: ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_GATE_RES, bias=bias, c_f32=x1, resid=x0, aux=aux,
                                                         gate=gate[:, 2 * N:], ldgate=6 * N, tokens=256, tile_hint=HINT)))
dgelu("dgrad fc2 + GELU'  ->4608", D, F)
gate_res("fwd proj + gate-res N1152", D, D); gate_res("fwd fc2 + gate-res K4608", D, F)
# phases of a workgroup's life (diagnostic build + SFRON_GEMM_CLK=1: SFRON_DBG_LIB=1 SFRON_GEMM_CLK=1 python3 tools/bench_gemm.py):
# stamps of the constant 100 MHz counter at entry / K-loop start / K-loop end / last store issued (csrc/gemm.hip sfron_dbg_gemm_phases)
PH = bool(os.environ.get("SFRON_DBG_LIB") and os.environ.get("SFRON_GEMM_CLK"))
def phases(name):
    import ctypes, numpy as np
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.sfron_dbg_gemm_phases.argtypes = [ctypes.c_int, ctypes.c_void_p]
    cls = 2 if name.startswith("wgrad") else 1 if name.startswith("dgrad") else 0
    a = np.zeros((1024, 6), dtype=np.int64)
    assert raw.sfron_dbg_gemm_phases(cls, a.ctypes.data) == 0
    a = a[(a[:, 1] > 0) & (a[:, 5] > 0)]
    if not len(a): return ""
    t = a[:, 2:6].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    late = (t[:, 0] - t0) > 1.0
    med = lambda x: float(np.median(x))
    return (f"   | {len(a)} wg ({int(late.sum())} start > 1 us late), life {med(t[:, 3] - t[:, 0]):5.1f} us = prologue {med(t[:, 1] - t[:, 0]):4.1f}"
            f" + K-loop {med(t[:, 2] - t[:, 1]):5.1f} + epilogue {med(t[:, 3] - t[:, 2]):4.1f}; last exit {(t[:, 3] - t0).max():5.1f} us"
            f"; clock {med(100.0 * a[:, 0] / a[:, 1]):4.0f} MHz")
tot_f = tot_t = 0
for name, fl, fn in cases:
    ms = timeit(fn)
    tot_f += fl; tot_t += ms
    print(f"{name:28s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s" + (phases(name) if PH else ""))
print(f"hint {HINT} {'block total':21s} {tot_t*1e3:8.1f} us  {tot_f/tot_t/1e9:7.1f} TFLOP/s")
