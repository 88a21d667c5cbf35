#!/usr/bin/env python3
"""GroupNorm(32) + swish forward / backward alone at the DDPM batch-64 shapes (and two LDM shapes): the one-launch form (k_gn3_*) where the rule
takes it, microseconds per launch by HIP events, algorithmic bytes (forward: x fp32 read once + y bf16 written; backward: x, dy read once + dx
written + pg / pb) and the rate they imply.   python tools/bench_gn.py [--reps 20]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import ab_lib; ab_lib.select()          # SFRON_LIB_NAME=libsfron_<variant>.so: another build of the library (tools/build_variant.sh)
from sfron import _lib
from sfron._lib import check, ptr, stream_ptr
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20); a = ap.parse_args()
L = _lib.lib()
DEV = "cuda"
SHAPES = [(64, 1024, 128), (64, 1024, 256), (64, 1024, 384), (64, 256, 256), (64, 256, 512), (64, 64, 256), (64, 64, 512), (64, 16, 256), (64, 16, 512),
          (8, 256, 1280), (8, 4096, 320)]
print(f"{'B x HW x C':>18s} {'form':>10s} {'fwd us':>8s} {'fwd GB/s':>9s} {'bwd us':>8s} {'bwd GB/s':>9s}")
for B, HW, C in SHAPES:
    rows = B * HW
    x = torch.randn(rows, C, device=DEV); dy = torch.randn(rows, C, device=DEV) * 0.1
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV); mean = torch.empty(B * 32, device=DEV); rstd = torch.empty_like(mean)
    dx = torch.empty(rows, C, device=DEV); pg = torch.empty(B, C, device=DEV); pb = torch.empty_like(pg)
    ws = torch.empty(L.sfron_groupnorm_scratch_bytes(B, HW, C, 32) // 8 + 2, dtype=torch.float64, device=DEV)
    junk = torch.empty(192 << 20, dtype=torch.uint8, device=DEV)          # pushed through the caches between repetitions: operands come from HBM
    def fwd():
        check(L.sfron_groupnorm_fwd(ptr(x), C, ptr(gamma), ptr(beta), B, HW, C, 32, 1e-6, 1, None, 1.0, ptr(y), ptr(mean), ptr(rstd), ptr(ws), stream_ptr()), "fwd")
    def bwd():
        check(L.sfron_groupnorm_bwd(ptr(dy), ptr(x), C, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), B, HW, C, 32, 1, None, 1.0, ptr(dx), C, 0, ptr(pg), ptr(pb),
                                    ptr(ws), stream_ptr()), "bwd")
    def timed(fn):
        fn(); torch.cuda.synchronize(); tot = 0.0
        for _ in range(a.reps):
            junk.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
        return tot / a.reps * 1e3
    tf, tb = timed(fwd), timed(bwd)
    bf, bb = rows * C * (4 + 2), rows * C * (4 + 4 + 4)
    form = "one launch" if L.sfron_groupnorm_one_launch(B, HW, C, 32) else "pair"
    print(f"{B:4d} x {HW:4d} x {C:4d} {form:>10s} {tf:8.1f} {bf / tf / 1e3:9.0f} {tb:8.1f} {bb / tb / 1e3:9.0f}")
