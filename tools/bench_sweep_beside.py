#!/usr/bin/env python3
"""Does a THROTTLED optimizer sweep hide beside the real DiT-XL/2 forward / backward pass?  GPU only.

tools/bench_overlap.py found no gain for a full-grid sweep beside cold fc1 GEMMs (the HBM queue saturates and the GEMMs'
operand loads wait behind it).  Here the sweep runs on at most W workgroups (sfron_masked_clip_adam_wg), i.e. at a bounded
share of the HBM rate, on a second stream beside the real pass of the engine, over the 28 blocks' share of the arena.
Prints: pass alone, sweep alone (per W), both concurrently, and the serial sum."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sfron import _lib, data, diffusion, dit  # noqa: E402
from sfron._lib import check, ptr  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "32"))
model = dit.DiT_models["DiT-XL/2"](input_size=32, num_classes=1000, batch_size=B, device=dev)
torch.manual_seed(0)
model.initialize_weights()
dit.randomize_zero_init(model, std=0.02, seed=1)
model.train()
eng = model.engine
diff = diffusion.create_diffusion("", device=dev)
lay = eng.layout
lo, hi = lay["blocks"], lay["blocks"] + eng.cfg.depth * lay["blk_stride"]
n = hi - lo
# a SECOND set of arenas for the sweep (so that the pass's weights do not change under it): same sizes, same traffic
p = torch.randn(n, device=dev) * 0.02
g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev)
v = torch.zeros(n, device=dev)
ema = p.clone()
mask = (torch.rand(n, device=dev) < 0.5).to(torch.uint8)
wbf = torch.empty(n, dtype=torch.bfloat16, device=dev)
L = _lib.lib()
side = torch.cuda.Stream()
b = data.synthetic_batch(0, 0, "forget", B, 0, 1, input_size=32, device=dev)
x_t = diff.q_sample(b["x0"], b["t"], b["noise"])


def sweep(W, stream, with_ema, pieces=1):
    per = (n // pieces) // 8 * 8
    for i in range(pieces):
        a, e = i * per, (n if i == pieces - 1 else (i + 1) * per)
        check(L.sfron_masked_clip_adam_wg(ptr(p[a:e]), ptr(g[a:e]), None, ptr(m[a:e]), ptr(v[a:e]), None if with_ema else ptr(mask[a:e]), None,
                                          e - a, 0.9, 0.999, 1e-8, 1e-4, 1.0, 1.0, ptr(wbf[a:e]), ptr(ema[a:e]) if with_ema else None, 0.9999,
                                          1 if with_ema else 0, W, ctypes.c_void_p(stream.cuda_stream)), "adam_wg")


def fwd():
    return eng.forward(x_t, b["t"], b["y"], b["drop"])


out = fwd()
_, _, d_out = diff.loss_fwd_bwd(out, b["x0"], b["t"], b["noise"], 1.0 / B)


def bwd():
    eng.backward(d_out, b["y"], b["drop"])


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e))
    return best


def both(pass_fn, W, with_ema, pieces):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        sweep(W, side, with_ema, pieces)
        pass_fn()
        cur.wait_stream(side)
    return f


def alone(W, with_ema, pieces):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        sweep(W, side, with_ema, pieces)
        cur.wait_stream(side)
    return f


t_f, t_b = timed(fwd), timed(bwd)
print(f"forward alone {t_f:.2f} ms, backward alone {t_b:.2f} ms; sweep range {n / 1e6:.0f} M params", flush=True)
for with_ema, name, pass_fn, t_pass in ((False, "forget-stage Adam (31 B/param) beside the FORWARD pass", fwd, t_f),
                                         (True, "remain-stage Adam+EMA (38 B/param) beside the BACKWARD pass", bwd, t_b)):
    print(name)
    for W in (0, 1024, 512, 256, 128, 64, 32):
        for pieces in (1, 28):
            t_s = timed(alone(W, with_ema, pieces))
            t_c = timed(both(pass_fn, W, with_ema, pieces))
            print(f"  W={W:5d} pieces={pieces:2d}: sweep alone {t_s:6.2f} ms ({n * (38 if with_ema else 31) / t_s / 1e9:5.2f} TB/s)  "
                  f"concurrent {t_c:6.2f} ms  serial sum {t_s + t_pass:6.2f} ms  hidden {t_s + t_pass - t_c:5.2f} ms", flush=True)
