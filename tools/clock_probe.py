#!/usr/bin/env python3
"""In-kernel clock of the block GEMMs' K-loops (diagnostic build only: `make -C .../csrc dbg`, SFRON_GEMM_CLK=1): delta s_memtime / delta
s_memrealtime x 100 MHz, median over the workgroups of the last launch per layout class (MI355X guide, DVFS give-back item 6).
    SFRON_GEMM_CLK=1 python3 tools/clock_probe.py alone      # each class back to back for ~1.5 s, random and zero-filled operands
    SFRON_GEMM_CLK=1 python3 tools/clock_probe.py step       # inside bench.py's steps (--steps 12): the last step's launches
The stamps go to a buffer of their own; no output value depends on them."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SFRON_GEMM_CLK", "1")
import torch  # noqa: E402
import sfron  # noqa: E402,F401
from sfron import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsfron_dbg.so")
torch.zeros(1, device="cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.sfron_dbg_gemm_clock.restype = ctypes.c_int
raw.sfron_dbg_gemm_clock.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
NAMES = ("forward", "dgrad", "weight gradient")


def read():
    mhz, n = (ctypes.c_double * 3)(), (ctypes.c_int * 3)()
    rc = raw.sfron_dbg_gemm_clock(mhz, n)
    assert rc == 0, rc
    return {NAMES[i]: (round(mhz[i]), n[i]) for i in range(3) if n[i]}


mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
if mode == "alone":
    from sfron import ops
    DEV, M, D, F = "cuda:0", 8192, 1152, 4608
    g = torch.Generator(device=DEV).manual_seed(0)
    for fill in ("random", "zeros"):
        mk = (lambda *s: torch.randn(*s, device=DEV, generator=g).to(torch.bfloat16)) if fill == "random" else (lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV))
        A, W = mk(M, D), mk(3 * D, D)
        C = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
        dY, Cd = mk(M, 3 * D), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        dYf, X, Cw = mk(M, F), mk(M, D), torch.empty(F, D, dtype=torch.float32, device=DEV)
        cases = {"forward": lambda: ops.gemm(A, W, M, 3 * D, D, c_bf16=C),
                 "dgrad": lambda: ops.gemm(dY, W, M, D, 3 * D, b_t=True, c_bf16=Cd),
                 "weight gradient": lambda: ops.gemm(dYf, X, F, D, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cw)}
        for name, fn in cases.items():
            t0 = time.time(); it = 0
            while time.time() - t0 < 1.5:
                for _ in range(200): fn()
                torch.cuda.synchronize(); it += 200
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): fn()
            b.record(); torch.cuda.synchronize()
            clk = read().get(name)
            print(f"alone, {fill:6s} operands, {name:16s}: {a.elapsed_time(b) / 50 * 1e3:7.1f} us per launch, in-kernel clock {clk[0]} MHz ({clk[1]} workgroups)", flush=True)
else:
    sys.argv = [sys.argv[0], "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-configs"]
    import bench  # noqa: E402
    bench.main()
    print("in the step (last launches of each class):", read(), flush=True)
