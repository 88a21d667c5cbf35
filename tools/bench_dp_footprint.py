#!/usr/bin/env python3
"""GPU (one device): what does the FOOTPRINT of the data-parallel gradient exchange cost the step before a byte crosses a link?  

The N > 1 step overlaps the exchange of each gradient stream with its backward pass: RCCL's ring kernels occupy a few
CUs (one workgroup of 512 threads per channel) and move the gradient arena through HBM about twice (reduce-scatter + all-gather) while the
backward pass -- which DESIGN section 6 calls 90 % packed -- runs.  Here the headline runner (DiT-XL/2, batch 32, bench.py's schedule) runs with a
stand-in for that footprint (tools/probes/dp_footprint.hip: k workgroups streaming `payload` bytes src -> dst twice, summing on the second pass) launched
on a side stream at the start of EVERY backward pass; the optimizer step waits for it, as it would for the exchange.  Reported: ms / step
against k (0 = no stand-in) for the fp32 (1.8 GB) and bf16 (0.9 GB) transports, and the stand-in's own duration.  No links, no ranks: a
prediction for the first 8-GPU run to be checked against, not a measurement of it.
    python3 tools/bench_dp_footprint.py [steps=12]"""
import ctypes
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sfron import data, diffusion, dit, step

dev = torch.device("cuda:0")
_so = os.path.join(ROOT, "tools", "probes", "libdpfoot.so")
if not os.path.exists(_so):          # (git-ignored; hipcc cross-compiles it anywhere)
    import subprocess
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", _so, os.path.join(ROOT, "tools", "probes", "dp_footprint.hip")], check=True)
foot = ctypes.CDLL(_so)
foot.dp_footprint.restype = ctypes.c_int
foot.dp_footprint.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def build():
    model = dit.DiT_models["DiT-XL/2"](input_size=32, num_classes=1000, batch_size=32, device=dev)
    torch.manual_seed(1234)
    model.initialize_weights()
    dit.randomize_zero_init(model, std=0.02, seed=1)
    model.train()
    eng = model.engine
    gm = torch.Generator().manual_seed(0)
    mask_arena = (torch.rand(eng.n_trainable, generator=gm) < 0.5).to(torch.uint8).to(dev)
    runner = step.DiTSFRon(model, diffusion.create_diffusion("", device=dev), lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999,
                           mask=None, unlearn_loss="ga", forget_class=207)
    runner.mask_arena = runner.opt.mask = mask_arena
    runner.sweep_across_steps = True
    return model, runner


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    model, runner = build()
    eng = model.engine
    kw = dict(global_batch=32, num_classes=1000, forget_class=207)
    bt = [tuple({k: v.to(dev) for k, v in data.synthetic_batch(5, i, s, **kw).items()} for s in ("forget", "remain")) for i in range(4)]
    from sfron import streams
    # the stream the data-parallel runner puts its collectives on (step.py): probed to run beside the caller's and the weight-gradient streams
    side = streams.get("comm", beside=[torch.cuda.current_stream()] + eng.side_streams())
    nbytes = eng.n_trainable * 4 // 16 * 16
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).zero_()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev).zero_()
    cfg = {"k": 0, "bytes": nbytes}
    timed = []
    orig_bwd, orig_step = eng.backward_factored_ada, runner.opt.step

    def bwd(*a, **kws):
        if cfg["k"] > 0:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            assert foot.dp_footprint(src.data_ptr(), dst.data_ptr(), cfg["bytes"], 2, cfg["k"], ctypes.c_void_p(side.cuda_stream)) == 0
            e1.record(side)
            timed.append((e0, e1))
        return orig_bwd(*a, **kws)

    def opt_step(*a, **kws):
        if cfg["k"] > 0:
            torch.cuda.current_stream().wait_stream(side)        # the sweep needs the exchanged gradient
        return orig_step(*a, **kws)
    eng.backward_factored_ada, runner.opt.step = bwd, opt_step

    rows = []
    for label, nb in (("fp32 transport, 1.8 GB per stream", nbytes), ("bf16 transport, 0.9 GB per stream", nbytes // 2 // 16 * 16)):
        for k in (0, 8, 16, 32, 64):
            if k == 0 and rows and nb != nbytes:
                continue
            cfg["k"], cfg["bytes"] = k, nb
            for i in range(4):
                runner.step(*bt[i % 4])
            runner.sync_sweep()
            torch.cuda.synchronize()
            timed.clear()
            t0 = time.perf_counter()
            for i in range(steps):
                runner.step(*bt[i % 4])
            runner.sync_sweep()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            foot_ms = sum(a.elapsed_time(b) for a, b in timed) / max(1, len(timed))
            rows.append({"transport": label if k else "none", "workgroups": k, "ms_per_step": round(ms, 2), "stand_in_ms_per_pass": round(foot_ms, 2)})
            print(json.dumps(rows[-1]), flush=True)
    runner.guard.poll(block=True)
    base = rows[0]["ms_per_step"]
    print(f"\nbaseline (no stand-in) {base:.2f} ms / step")
    for r in rows[1:]:
        print(f"  {r['transport']:36s} {r['workgroups']:3d} workgroups: {r['ms_per_step']:.2f} ms / step ({r['ms_per_step'] - base:+.2f}), stand-in busy {r['stand_in_ms_per_pass']:.2f} ms per backward pass")


if __name__ == "__main__":
    main()
