#!/usr/bin/env python3
"""GPU diagnosis (round 6): are the one-byte-GELU' epilogues repeatable bit for bit, alone and beside another stream's traffic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib
ab_lib.select()
import torch
from sfron import _lib, ops
DEV = "cuda:0"
M, F, D = 8192, 4608, 1152
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(DEV)
X, W1, b1 = r(M, D), r(F, D, sc=0.03), (torch.randn(F, generator=g) * 0.1).to(DEV)
dY, W2 = r(M, D, sc=0.1), r(D, F, sc=0.03)
side = torch.cuda.Stream()
junk = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
ref = None
for it in range(8):
    H, codes = torch.empty(M, F, dtype=torch.bfloat16, device=DEV), torch.empty(M, F, dtype=torch.uint8, device=DEV)
    dq, part = torch.empty(M, F, dtype=torch.bfloat16, device=DEV), torch.zeros(M // 256, F, dtype=torch.float32, device=DEV)
    if it >= 4:
        with torch.cuda.stream(side):
            junk.add_(1.0)                      # HBM traffic beside the kernels
    ops.gemm(X, W1, M, F, D, epilogue=_lib.EPI_GELU_Q, bias=b1, c_bf16=H, aux=codes)
    ops.gemm(dY, W2, M, F, D, b_t=True, epilogue=_lib.EPI_DGELU_Q, c_bf16=dq, aux=codes, col_partials=part)
    torch.cuda.synchronize()
    cur = (H.clone(), codes.clone(), dq.clone(), part.clone())
    if ref is None:
        ref = cur
    else:
        print(f"run {it}: H {torch.equal(cur[0], ref[0])} codes {torch.equal(cur[1], ref[1])} dX {torch.equal(cur[2], ref[2])} colpart {torch.equal(cur[3], ref[3])}"
              + ("" if torch.equal(cur[1], ref[1]) else f"  ({int((cur[1] != ref[1]).sum())} codes differ)")
              + ("" if torch.equal(cur[2], ref[2]) else f"  ({int((cur[2] != ref[2]).sum())} dX elements differ)"), flush=True)
