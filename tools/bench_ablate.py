#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV="cuda:0"; M,D,F=8192,1152,4608
g=torch.Generator(device=DEV).manual_seed(0)
def rnd(*s): return torch.randn(*s,device=DEV,generator=g).to(torch.bfloat16)
def timeit(fn,iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(iters)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/iters
for name,N,K in (("fc2 N1152 K4608",D,F),("qkv N3456 K1152",3*D,D),("fc1 N4608 K1152",F,D)):
    A,B=rnd(M,K),rnd(N,K); C=torch.empty(M,N,dtype=torch.bfloat16,device=DEV)
    for h,lab in ((2,"full"),(21,"no-mfma"),(22,"no-dma")):
        ms=timeit(lambda: ops.gemm(A,B,M,N,K,c_bf16=C,tile_hint=h))
        print(f"{name} {lab:8s} {ms*1e3:8.1f} us")

# wgrad fc1: dW[4608,1152] = dY[8192,4608]^T X[8192,1152]
dY, X = rnd(M, F), rnd(M, D); C = torch.empty(F, D, dtype=torch.float32, device=DEV)
for h, lab in ((2, "full"), (21, "no-mfma"), (22, "no-dma"), (-1, "generic"), (32, "pipe")):
    ms = timeit(lambda: ops.gemm(dY, X, F, D, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C, tile_hint=h))
    print(f"wgrad fc1 (108 tiles of 256x192, K=8192) {lab:8s} {ms*1e3:8.1f} us")
