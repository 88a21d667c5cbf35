#!/usr/bin/env python3
"""GPU: the few-tile products of BASELINE config 2 (DiT-B/4 batch 32: [2048 x 768] outputs, K = 768 / 2304 / 3072), one launch on each tile
against the contraction split over S workgroups per tile + its finish kernel.  us per (product [+ finish]), back-to-back launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib
_lib = ab_lib.select()
import torch
from sfron import ops
from sfron._lib import check, ptr, stream_ptr

DEV = "cuda:0"
M, N, T = 2048, 768, 64
L = _lib.lib()
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for K in (768, 2304, 3072):
    X = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    Wt = (torch.randn(K, N, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=DEV) * 0.1
    gate = torch.randn(M // T, N, generator=g, device=DEV)
    resid = torch.randn(M, N, generator=g, device=DEV)
    x1, a1 = torch.empty(M, N, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    slabs = torch.empty(8, M, N, dtype=torch.float32, device=DEV)
    row = [f"K {K:4d}"]
    for hint in (0, 1):
        try:
            t = timeit(lambda: ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_GATE_RES, bias=bias, c_f32=x1, resid=resid, aux=a1, gate=gate, ldgate=N, tokens=T, tile_hint=hint))
            row.append(f"fwd gate_res hint {hint}: {t:5.1f}")
        except Exception as e:
            row.append(f"fwd hint {hint}: n/a")
        try:
            t = timeit(lambda: ops.gemm(X, Wt, M, N, K, b_t=True, c_bf16=out, tile_hint=hint))
            row.append(f"dgrad hint {hint}: {t:5.1f}")
        except Exception as e:
            row.append(f"dgrad hint {hint}: n/a")
    print(" | ".join(row))
    for S in (2, 3, 4, 6, 8):
        kt = K // 64
        if kt % S or (kt // S) % 2:
            continue

        def f():
            ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_F32, c_f32=slabs, ldc_f32=N, split_k=S, split_stride=M * N)
            check(L.sfron_split_gate_res(ptr(slabs), S, M * N, ptr(bias), ptr(gate), N, T, ptr(resid), ptr(x1), ptr(a1), M, N, stream_ptr()), "fin")

        def d():
            ops.gemm(X, Wt, M, N, K, b_t=True, epilogue=_lib.EPI_F32, c_f32=slabs, ldc_f32=N, split_k=S, split_stride=M * N)
            check(L.sfron_split_sum_bf16(ptr(slabs), S, M * N, M * N, ptr(out), stream_ptr()), "fin")
        print(f"        split {S}: fwd + gate_res finish {timeit(f):5.1f} | dgrad + bf16 finish {timeit(d):5.1f}")
