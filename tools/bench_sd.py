#!/usr/bin/env python3
"""BASELINE config 4 on the HIP path: SD v1 UNet nsfw_removal SFR-on iterations/s (batch 2, 64x64 latents, 77-token context).
    python tools/bench_sd.py [--steps 5] [--batch 2] [--method full|xattn]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--method", default="full")
ap.add_argument("--eager", action="store_true", help="no HIP-graph replay of the stages")
a = ap.parse_args()
from sfron import sd, sd_unet
if os.environ.get("SFRON_FUSE_SPLIT_FINISH"):       # A-B knob: 0 = every split convolution finishes its own output (before round 6, late)
    from sfron import unet as _u2; _u2._TapeNet.FUSE_SPLIT_FINISH = os.environ["SFRON_FUSE_SPLIT_FINISH"] != "0"
if os.environ.get("SFRON_BATCH_REDUCTIONS"):        # A-B knob: 0 = every parameter-gradient finish as its own launch (before round 6)
    from sfron import unet as _u; _u._TapeNet.BATCH_REDUCTIONS = os.environ["SFRON_BATCH_REDUCTIONS"] != "0"
if os.environ.get("SFRON_LOADER_WAVES"):            # A-B knob: 10 = convolution tiles in the shared-wave form
    from sfron import _lib as _L; _L.lib().sfron_gemm_loader_waves(int(os.environ["SFRON_LOADER_WAVES"]))
DEV = "cuda"
torch.manual_seed(0)
model = sd_unet.UNetModel()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for p in model.parameters():
        if not bool(p.any()):
            p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
model.sync_bf16()
run = sd.SDSFRon(model, lr=1e-5, train_method=a.method, use_graphs=not a.eager)
B = a.batch
gd = torch.Generator(device=DEV).manual_seed(2)
rn = lambda *s: torch.randn(*s, device=DEV, generator=gd)
c_f, c_p = rn(1, 77, 768).expand(B, -1, -1).contiguous(), rn(1, 77, 768).expand(B, -1, -1).contiguous()
def batch():
    xf = rn(B, 4, 64, 64)
    return (dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.randint(0, 1000, (B,), device=DEV, generator=gd), noise=rn(B, 4, 64, 64)),
            dict(x=rn(B, 4, 64, 64), c=c_p, t=torch.randint(0, 1000, (B,), device=DEV, generator=gd), noise=rn(B, 4, 64, 64)))
bts = [batch() for _ in range(2)]
for i in range(2): run.step(*bts[i % 2])
torch.cuda.synchronize(); t0 = time.time()
for i in range(a.steps): run.step(*bts[i % 2])
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
t1 = time.time()
for i in range(2): run.step(*bts[i % 2])
host = (time.time() - t1) / 2
print(f"SD v1 UNet SFR-on iteration, batch {B}, train_method {a.method}, {'eager' if a.eager else 'graph replay'}: {dt * 1e3:.0f} ms = {1 / dt:.2f} it/s (host enqueue {host * 1e3:.0f} ms)")
