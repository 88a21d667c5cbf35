#!/usr/bin/env python3
"""Yardstick only (never on the product path): torch.matmul (hipBLASLt/rocBLAS) bf16 on the DiT-XL/2 B=32 GEMM shapes,
to see what the vendor library attains on this box next to tools/bench_gemm.py. GPU only."""
import torch
DEV = "cuda:0"
M, D, F = 8192, 1152, 4608
def rnd(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
tot_f = tot_t = 0
for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", F, D), ("fc2", D, F)):
    X, W, dY = rnd(M, K), rnd(N, K), rnd(M, N)
    out1 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); out2 = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    out3 = torch.empty(N, K, device=DEV, dtype=torch.bfloat16)
    for kind, fn in (("fwd  ", lambda: torch.matmul(X, W.t(), out=out1)), ("dgrad", lambda: torch.matmul(dY, W, out=out2)),
                     ("wgrad", lambda: torch.matmul(dY.t(), X, out=out3))):
        ms = timeit(fn); fl = 2.0 * M * N * K
        tot_f += fl; tot_t += ms
        print(f"lib {kind} {name:5s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
print(f"lib block total {tot_t*1e3:8.1f} us  {tot_f/tot_t/1e9:7.1f} TFLOP/s")
# big square for reference
for n in (4096, 8192):
    A, B = rnd(n, n), rnd(n, n); C = torch.empty(n, n, device=DEV, dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(A, B.t(), out=C), 10)
    print(f"lib square {n}: {ms*1e3:8.1f} us {2.0*n**3/ms/1e9:7.1f} TFLOP/s")
