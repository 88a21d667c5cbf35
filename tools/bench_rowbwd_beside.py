#!/usr/bin/env python3
"""k_row_bwd (fused LN + gate backward, 170 MB per launch) alone and BESIDE a stream of 144-workgroup weight-gradient GEMMs, as in
the backward pass of the step: how much of its in-step slow-down (45 us alone, 105-119 us in the step) is the co-resident GEMM?
Operands rotated through > 256 MB (cold in the Infinity Cache).  GPU only.
    python tools/bench_rowbwd_beside.py [dbg]        # dbg: the debug-knob build (SFRON_ROW_RPW ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import _lib
if len(sys.argv) > 1 and sys.argv[1] == "dbg":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsfron_dbg.so")
from sfron import ops
DEV = "cuda:0"; B, T, D, F = 32, 256, 1152, 4608; M = B * T
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
NSET = 4
rows = [dict(x=rnd(M, D), dxm=rnd(M, D).bfloat16(), dx=rnd(M, D), br=rnd(M, D).bfloat16()) for _ in range(NSET)]
mod = rnd(B, 6 * D) * 0.1
mean, rstd = rnd(M), rnd(M).abs() + 0.5
wsets = [(rnd(M, F).bfloat16(), rnd(M, D).bfloat16()) for _ in range(NSET)]
dW = torch.empty(F, D, dtype=torch.float32, device=DEV)
side = torch.cuda.Stream()
def row(i):
    r = rows[i % NSET]
    ops.ln_gate_bwd(r["dxm"], r["x"], mean, rstd, mod[:, 4 * D:], 6 * D, T, r["dx"], True, r["br"], mod[:, 2 * D:], 6 * D)
def wg(i):
    dY, X = wsets[i % NSET]
    ops.gemm(dY, X, F, D, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=dW)
def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record()
    return a, b
for i in range(3): row(i); wg(i)
torch.cuda.synchronize()
a, b = timed(row, 24); torch.cuda.synchronize(); t_row = a.elapsed_time(b) / 24 * 1e3
a, b = timed(wg, 24); torch.cuda.synchronize(); t_wg = a.elapsed_time(b) / 24 * 1e3
print(f"alone: row_bwd {t_row:.1f} us, wgrad fc1 (144 workgroups) {t_wg:.1f} us")
# beside: the side stream runs 40 weight gradients back to back; the main stream 24 row kernels in the middle of that
cur = torch.cuda.current_stream()
side.wait_stream(cur)
with torch.cuda.stream(side):
    sa, sb = timed(wg, 40)
for i in range(4): row(i)            # (let the side stream get going)
a, b = timed(row, 24)
torch.cuda.synchronize()
print(f"beside: row_bwd {a.elapsed_time(b) / 24 * 1e3:.1f} us, wgrad {sa.elapsed_time(sb) / 40 * 1e3:.1f} us (average over 40, 28 of them beside row kernels)")
