#!/usr/bin/env python3
"""Does a HIP-graph replay of a dependent chain of C-launched kernels beat the stream launches?  The DiT forward pass (one C call = ~200 launches
on one stream at DiT-XL/2, ~90 at DiT-B/4) eager against captured + replayed.   python tools/bench_fwd_graph.py [xl2|b4] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sfron import dit
name = sys.argv[1] if len(sys.argv) > 1 else "xl2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
model = dit.DiT_models["DiT-XL/2" if name == "xl2" else "DiT-B/4"](input_size=32, num_classes=1000).cuda()
model.set_batch_size(B)
eng = model.engine
with torch.no_grad():
    eng.params[:eng.n_trainable].normal_(0, 0.02); eng.sync_bf16()
x = torch.randn(B, 4, 32, 32, device="cuda"); t = torch.randint(0, 1000, (B,), device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
out = torch.empty(eng.out_shape, device="cuda")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
eager = timed(lambda: eng.forward(x, t, y, None, out=out))
ref = out.clone()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    eng.forward(x, t, y, None, out=out)
out.zero_(); g.replay(); torch.cuda.synchronize()
assert torch.equal(out, ref)
rep = timed(g.replay)
print(f"{name} batch {B}: forward pass eager (stream launches from one C call) {eager:.3f} ms, HIP-graph replay {rep:.3f} ms")
