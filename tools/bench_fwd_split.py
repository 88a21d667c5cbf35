#!/usr/bin/env python3
"""Does the forward pass gain from running as TWO independent half-batch chains on two streams (memory-bound phases of one chain -- epilogue
stores, LayerNorm, attention -- under the K-loops of the other)?  DiT-XL/2, batch 32 as one chain against 2 x 16 (and 4 x 8) on their own streams,
same weights (sibling engines).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfron import data, dit
DEV = "cuda:0"
model = dit.DiT_models["DiT-XL/2"](input_size=32, num_classes=1000, batch_size=32)
dit.randomize_zero_init(model, std=0.02, seed=1)
eng = model.engine
b = data.synthetic_batch(0, 0, "remain", 32, device=DEV)


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters


print(f"one chain, batch 32: {timeit(lambda: eng.forward(b['x0'], b['t'], b['y'], b['drop'])):.2f} ms per forward pass", flush=True)
for n in (2, 4):
    per = 32 // n
    engs = [eng.sibling(per) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    parts = [{k: v[i * per:(i + 1) * per].contiguous() for k, v in b.items()} for i in range(n)]

    def run():
        cur = torch.cuda.current_stream()
        for e, s, p in zip(engs, streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.forward(p["x0"], p["t"], p["y"], p["drop"])
        for s in streams:
            cur.wait_stream(s)
    print(f"{n} chains of batch {per} on {n} streams: {timeit(run):.2f} ms per forward pass", flush=True)
    one = timeit(lambda: engs[0].forward(parts[0]["x0"], parts[0]["t"], parts[0]["y"], parts[0]["drop"]))
    print(f"   (one chain of batch {per} alone: {one:.2f} ms)", flush=True)
    for e in engs: e.close()
