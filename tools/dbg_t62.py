import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for layout in ("fwd", "dgrad"):
    for (M, N, K) in ((256, 144, 192), (512, 288, 384), (512, 1152, 576), (8192, 1152, 1152), (1024, 1152, 3456)):
        ints = lambda *s: torch.randint(-3, 4, s, generator=g).to(torch.bfloat16).to(DEV)
        if layout == "fwd": A, B, kw = ints(M, K), ints(N, K), {}
        else: A, B, kw = ints(M, K), ints(K, N), dict(b_t=True)
        C1 = torch.zeros(M, N, device=DEV); C2 = torch.zeros(M, N, device=DEV)
        ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C1, tile_hint=62, **kw)
        ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C2, tile_hint=-1, **kw)
        d = (C1 - C2).abs()
        print(layout, "256x144 3-slot", (M, N, K), "max diff", d.max().item(), "n bad", int((d > 0).sum()))
