"""Run the DDPM U-Net forward + backward twice on identical inputs and list the gradient tensors that differ bitwise.
   python tools/dbg_determinism.py [--dbg-lib]   (with --dbg-lib the debug-knob library is loaded: SFRON_NO_CGEMM=<mask> applies)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sfron import _lib
if "--dbg-lib" in sys.argv:
    _lib.LIB_PATH = os.path.join(ROOT, "unified-unlearning-w-remain-geometry_amd", "libsfron_dbg.so")
from sfron import unet
DEV = "cuda:0"
torch.manual_seed(40)
model = unet.Conditional_Model(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.0, resolution=16, n_classes=10)
with torch.no_grad():                         # zero-initialised tensors would hide whole branches of the backward pass
    for p in model.parameters():
        if not bool(p.any()):
            p.copy_(torch.randn(p.shape, device=p.device) * 0.05)
model.sync_bf16()
model.train()
g = torch.Generator().manual_seed(1)
B = 8
x = torch.randn(B, 3, 16, 16, generator=g).to(DEV); t = torch.randint(0, 1000, (B,), generator=g).float().to(DEV)
c = torch.randint(0, 10, (B,), generator=g).to(DEV); keep = torch.ones(B, dtype=torch.uint8, device=DEV)
w = torch.randn(B, 3, 16, 16, generator=g).to(DEV)
outs, grads = [], []
for rep in range(3):
    junk = torch.randn(1 << 24, device=DEV) * (rep + 1)      # different garbage in freed memory between repetitions
    del junk
    out, bwd = model._run(x, t, c, keep, None, need_grad=True)
    bwd(w.clone())
    outs.append(out.clone()); grads.append(model.grads.clone())
print("outputs equal:", [torch.equal(outs[0], o) for o in outs[1:]])
for r in (1, 2):
    bad = [n for n in model.index if not torch.equal(model.view(grads[0], n), model.view(grads[r], n))]
    print(f"rep {r}: {len(bad)} differing gradient tensors", bad[:12])
