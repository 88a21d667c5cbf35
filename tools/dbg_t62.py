import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((256, 144, 192), (512, 288, 384), (512, 1152, 576), (8192, 1152, 1152), (1024, 3456, 4608)):
    ints = lambda *s: torch.randint(-3, 4, s, generator=g).to(torch.bfloat16).to(DEV)
    A, B = ints(M, K), ints(N, K)
    C1 = torch.zeros(M, N, device=DEV); C2 = torch.zeros(M, N, device=DEV)
    ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C1, tile_hint=62)
    ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C2, tile_hint=-1)
    d = (C1 - C2).abs()
    print("fwd 256x144 3-slot", (M, N, K), "max diff", d.max().item(), "n bad", int((d > 0).sum()))
