#!/usr/bin/env python3
"""BASELINE config 1 on the HIP path: DDPM CIFAR-10 SFR-on steps/s at batch 64 (cifar10_sfron.yml model), synthetic inputs.
    python tools/bench_ddpm.py [--steps 20] [--batch 64]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
sys.path.insert(0, os.path.join(ROOT, "tools")); import ab_lib; ab_lib.select()     # SFRON_LIB_NAME: another build of the library
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--eager", action="store_true", help="no HIP-graph replay of the stages")
a = ap.parse_args()
from sfron import ddpm, unet
if os.environ.get("SFRON_FUSE_SPLIT_FINISH"):       # A-B knob: 0 = every split convolution finishes its own output (before round 6, late)
    unet._TapeNet.FUSE_SPLIT_FINISH = os.environ["SFRON_FUSE_SPLIT_FINISH"] != "0"
if os.environ.get("SFRON_BATCH_REDUCTIONS"):        # A-B knob: 0 = every parameter-gradient finish as its own launch (before round 6)
    unet._TapeNet.BATCH_REDUCTIONS = os.environ["SFRON_BATCH_REDUCTIONS"] != "0"
if os.environ.get("SFRON_LOADER_WAVES"):            # A-B knob: 10 = convolution tiles in the shared-wave form
    from sfron import _lib as _L; _L.lib().sfron_gemm_loader_waves(int(os.environ["SFRON_LOADER_WAVES"]))
import test_gpu_unet as T
torch.manual_seed(1234)
model = unet.Conditional_Model(unet.config_namespace())
run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, mask=None, unlearn_loss="adaga", lambd=0.5, n_iters=50, use_graphs=not a.eager)
g = torch.Generator().manual_seed(1)
bt = [({k: v.cuda() for k, v in T._synthetic(i, "forget", a.batch, g).items()}, {k: v.cuda() for k, v in T._synthetic(i, "remain", a.batch, g).items()}) for i in range(2)]
for i in range(3): run.step(i, *bt[i % 2])
torch.cuda.synchronize(); t0 = time.time()
for i in range(a.steps): run.step(i, *bt[i % 2])
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
t1 = time.time()
for i in range(3): run.step(i, *bt[i % 2])
host = (time.time() - t1) / 3
print(f"DDPM SFR-on step, batch {a.batch}, {'eager' if a.eager else 'graph replay'}: {dt * 1e3:.1f} ms/step = {1 / dt:.2f} steps/s (host enqueue time per step {host * 1e3:.1f} ms)")
