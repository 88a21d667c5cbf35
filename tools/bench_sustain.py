#!/usr/bin/env python3
"""Does a sustained MFMA load slow down (power / clock)?  Times one GEMM back-to-back in chunks. GPU only."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import ops, _lib
DEV = "cuda:0"
M, N, K = 8192, 4608, 1152
A = torch.randn(M, K, device=DEV).to(torch.bfloat16); B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
fn = lambda: ops.gemm(A, B, M, N, K, c_bf16=C)
for _ in range(3): fn()
torch.cuda.synchronize()
smi = []
def poll():
    for _ in range(12):
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            smi.append([l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l or "power" in l][:3])
        except Exception as e:
            smi.append([repr(e)])
        time.sleep(0.25)
th = threading.Thread(target=poll); th.start()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for c in range(40):
    for _ in range(400): fn()
    evs[c + 1].record()
torch.cuda.synchronize()
th.join()
fl = 2.0 * M * N * K
for c in range(0, 40, 3):
    ms = evs[c].elapsed_time(evs[c + 1]) / 400
    print(f"chunk {c:2d} (t={evs[0].elapsed_time(evs[c]):7.1f} ms): {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF")
for s in smi[:8]: print(s)
