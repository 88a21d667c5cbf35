#!/usr/bin/env python3
"""Where the time of the fused attention kernels goes (diagnostic build only: `make -C .../csrc dbg`): per-workgroup phase stamps of
the constant 100 MHz counter (csrc/attn.hip ATTN_STAMP), DiT-XL/2 shape B 32, T 256, H 16, hd 72, random data.  GPU only.
    python3 tools/attn_probe.py
Prints, for the forward and the fused backward: the launch time (events), when workgroups START relative to the first one (rounds on
a CU), and the median duration of each phase.  The stamps go to a buffer of their own; no output value depends on them."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sfron  # noqa: E402,F401
from sfron import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("SFRON_PROBE_LIB", "libsfron_dbg.so"))
torch.zeros(1, device="cuda:0")
L = _lib.lib()
from sfron import ops  # noqa: E402
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sfron_dbg_attn_clock.restype = ctypes.c_int
raw.sfron_dbg_attn_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
DEV = "cuda:0"
B, T, H, hd = 32, 256, 16, 72
D = H * hd
g = torch.Generator(device=DEV).manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device=DEV, generator=g).to(torch.bfloat16)
d_o = torch.randn(B * T, D, device=DEV, generator=g).to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, B, T, H, hd)


def stamps(n):
    a = np.zeros((n, 8), dtype=np.int64)
    rc = raw.sfron_dbg_attn_clock(a.ctypes.data, n)
    assert rc == 0, rc
    return a


def launch_us(fn, iters=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def report(name, fn, n_wg, phases):
    us = launch_us(fn)
    assert raw.sfron_dbg_attn_clock(None, 0) == 0      # arm
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = stamps(n_wg)
    t = a[:, :7].astype(np.float64) / 100.0      # us
    t0 = t[:, 0].min()
    start = np.sort(t[:, 0] - t0)
    end = t[:, 6] - t0
    print(f"{name}: {us:.1f} us per launch (events); {n_wg} workgroups; stamped span {end.max():.1f} us")
    print("  workgroup start times (us after the first), deciles:", " ".join(f"{start[int(q * (n_wg - 1) / 10)]:.1f}" for q in range(11)))
    print(f"  workgroup duration: median {np.median(t[:, 6] - t[:, 0]):.2f}  min {np.min(t[:, 6] - t[:, 0]):.2f}  max {np.max(t[:, 6] - t[:, 0]):.2f} us")
    first = (t[:, 0] - t0) < 1.0
    for label, i, j in phases:
        d = t[:, j] - t[:, i]
        print(f"  {label:44s} median {np.median(d):6.2f} us   first-round {np.median(d[first]):6.2f}   later {np.median(d[~first]) if (~first).any() else float('nan'):6.2f}")
    hw = a[:, 7]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = (hw >> 32) & 0xf
    slot = xcc * 10000 + se * 100 + sh * 50 + cu
    print(f"  distinct (XCC, SE, SH, CU): {len(np.unique(slot))}; workgroups per CU: min {np.bincount(np.unique(slot, return_inverse=True)[1]).min()} max {np.bincount(np.unique(slot, return_inverse=True)[1]).max()}")


if os.environ.get("SFRON_PROBE_SET") == "2":      # a library built with -DSFRON_ATTN_PROBE=2: the stamps sit inside the prologue
    P = [("entry -> operand loads issued", 0, 1), ("LDS-DMA of the first chunks issued", 1, 2), ("pads zeroed (+ delta, lse in the backward)", 2, 3),
         ("wait for chunk 0 + barrier", 3, 4), ("rest of the workgroup", 4, 6)]
    L.sfron_attn_fwd_form(4)
    report("attention forward, prologue", lambda: ops.attn_fwd(qkv, B, T, H, hd), T // 128 * B * H, P)
    report("attention backward (fused), prologue", lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hd), B * H, P)
    sys.exit(0)
for form in (4,):
    L.sfron_attn_fwd_form(form)
    report(f"attention forward ({form}-wave workgroups)", lambda: ops.attn_fwd(qkv, B, T, H, hd), T // 128 * B * H,
           [("entry -> first chunk landed (barrier 0)", 0, 1), ("chunk 0", 1, 2), ("chunk 1", 2, 3), ("chunk 2", 3, 4),
            ("chunk 3", 4, 5), ("epilogue (normalise, stores)", 5, 6)])
report("attention backward (fused)", lambda: ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hd), B * H,
       [("entry -> first chunk landed (barrier 0)", 0, 1), ("chunk 0: S / dP / softmax / dS^T", 1, 2), ("chunk 0: dV, dK", 2, 3),
        ("chunk 0: dQ + stores", 3, 4), ("chunks 1..3", 4, 5), ("epilogue (dK / dV stores)", 5, 6)])
