import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for K in (128, 320, 512, 896, 8192):
    M, N = 384, 576
    ints = lambda *s: torch.randint(-3, 4, s, generator=g).to(torch.bfloat16).to(DEV)
    A, B, kw = ints(K, M), ints(K, N), dict(a_t=True, b_t=True)
    C1 = torch.zeros(M, N, device=DEV); C2 = torch.zeros(M, N, device=DEV)
    ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C1, tile_hint=55, **kw)
    ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C2, tile_hint=-1, **kw)
    print("wgrad 3-slot K", K, "max diff", (C1 - C2).abs().max().item())
