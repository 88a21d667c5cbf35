import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for layout in ("fwd", "dgrad", "wgrad"):
    for K in (128, 256, 384, 1024):
        M, N = 512, 768
        ints = lambda *s: torch.randint(-3, 4, s, generator=g).to(torch.bfloat16).to(DEV)
        if layout == "fwd": A, B, kw = ints(M, K), ints(N, K), {}
        elif layout == "dgrad": A, B, kw = ints(M, K), ints(K, N), dict(b_t=True)
        else: A, B, kw = ints(K, M), ints(K, N), dict(a_t=True, b_t=True)
        C1 = torch.zeros(M, N, device=DEV); C2 = torch.zeros(M, N, device=DEV)
        ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C1, tile_hint=42, **kw)
        ops.gemm(A, B, M, N, K, epilogue=_lib.EPI_F32, c_f32=C2, tile_hint=-1, **kw)
        d = (C1 - C2).abs()
        bad = (d > 0).nonzero()
        print(layout, K, "max diff", d.max().item(), "n bad", len(bad), "first", bad[:3].tolist(),
              "bad rows%16", sorted(set((bad[:, 0] % 16).tolist()))[:16], "cols%16", sorted(set((bad[:, 1] % 16).tolist()))[:16],
              "row blocks", sorted(set((bad[:, 0] // 64).tolist()))[:8], "col blocks", sorted(set((bad[:, 1] // 96).tolist()))[:8])
