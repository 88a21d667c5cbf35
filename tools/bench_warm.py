#!/usr/bin/env python3
"""Would warming the NEXT GEMM's weights into the Infinity Cache (side stream, under the preceding elementwise kernel)
pay?  Per iteration: flush the cache (1 GB fill), produce A (stand-in for LN+modulate), GEMM with a weight not touched
since ~28 layers ago.  Times produce+GEMM with and without the warm-up read.  GPU only; torch ops are stand-ins."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"; M, D, F, L = 8192, 1152, 4608, 8
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
side = torch.cuda.Stream()
flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)     # 1 GB
for nm, N, K in (("qkv", 3 * D, D), ("fc1", F, D), ("fc2", D, F)):
    A0, A = rnd(M, K), rnd(M, K)
    Ws = [rnd(N, K) for _ in range(L)]
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for warm in (False, True, False, True):
        tot = 0.0
        for i in range(2 * L):
            flush.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if warm:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    Ws[i % L].view(torch.int32).sum()
            A.copy_(A0)                                            # "previous kernel": writes the GEMM's A operand
            if warm: torch.cuda.current_stream().wait_stream(side)
            ops.gemm(A, Ws[i % L], M, N, K, c_bf16=C)
            b.record(); torch.cuda.synchronize()
            if i >= 2: tot += a.elapsed_time(b)
        print(f"{nm}: produce A + GEMM, W cold, warm-up read {'on ' if warm else 'off'}: {tot/(2*L-2)*1e3:7.1f} us", flush=True)
