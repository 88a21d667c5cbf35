import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_unet as T
from sfron import ddpm
DEV = "cuda:0"
loss = sys.argv[1] if len(sys.argv) > 1 else "adaga"
cfg = dict(T.SMALL, dropout=0.0)
B, n_it = 8, 4
g = torch.Generator().manual_seed(41)
batches = []
for it in range(n_it):
    pair = []
    for stream in ("forget", "remain"):
        b = T._synthetic(it, stream, B, g)
        b["x0"], b["e"] = b["x0"][:, :, :16, :16].contiguous(), b["e"][:, :, :16, :16].contiguous()
        b["keep_mask"] = (torch.rand(B, generator=g) >= 0.1).to(torch.uint8)
        pair.append({k: v.to(DEV) for k, v in b.items()})
    batches.append(pair)
for use in (False, False, True):
    _, model = T._pair(cfg, seed=40)
    run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, unlearn_loss=loss, n_iters=n_it, use_graphs=use)
    out = []
    for it in range(n_it):
        l = run.step(it, *batches[it])
        out.append((l["forget_loss"].item(), l["remain_loss"].item(), run.flat.g.double().norm().item(), run.flat.p.double().norm().item()))
    print(use, out)
