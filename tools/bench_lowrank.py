#!/usr/bin/env python3
"""GPU: the rank-(batch) AdamW sweep of the adaLN_modulation matrix alone, at DiT-XL/2's size (R = 32, 28 x 6912 rows, 1152 columns):
matrix-core form against the vector form (SFRON_LOWRANK_VECTOR=1, read once per process).  tools/bench_lowrank.py [max_workgroups]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib  # noqa: E402
ab_lib.select()
import torch  # noqa: E402
from sfron import _lib  # noqa: E402
from sfron._lib import check, ptr, stream_ptr  # noqa: E402

L = _lib.lib()
dev = "cuda:0"
R, NM, D = 32, 28 * 6912, 1152
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = NM * D
g = torch.Generator().manual_seed(0)
dmod = (torch.randn(R, NM, generator=g) * 0.1).to(torch.bfloat16).to(dev)
sc = torch.randn(R, D, generator=g).to(torch.bfloat16).to(dev)
p = (torch.randn(n, generator=g) * 0.05).to(dev)
m, v, ema = torch.zeros(n, device=dev), torch.zeros(n, device=dev), p.clone()
mask = (torch.rand(n, generator=g) < 0.5).to(torch.uint8).to(dev)
w16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
stats = torch.tensor([0.0, 1.0, 0.0, 0.0], device=dev)
args = (0.9, 0.999, 1e-8, 1e-4, 1.0, 1.0)
for name, mk, em, mode in (("forget stage (mask, no EMA)", mask, None, 0), ("remain stage (EMA, no mask)", None, ema, 1)):
    def run():
        check(L.sfron_adam_lowrank_rows(ptr(p), ptr(m), ptr(v), ptr(mk), ptr(stats), ptr(dmod), ptr(sc), R, NM, 0, NM, D, *args, ptr(w16),
                                        ptr(em), 0.9999, mode, cap, stream_ptr()), "adam_lowrank_rows")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    byts = n * (24 + 2 + (1 if mk is not None else 0) + (8 if mode else 0))
    print(f"{'vector' if os.environ.get('SFRON_LOWRANK_VECTOR') == '1' else 'matrix-core'} form, cap {cap}: {name}: {dt * 1e3:.3f} ms = {byts / dt / 1e12:.2f} TB/s")
