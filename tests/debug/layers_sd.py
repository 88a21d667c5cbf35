import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/debug/ -> repo root (debugging aid; uses the oracle, hence under tests/)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_sd as T
cfg, B, S, Lc = T.SMALL, 3, 8, 5
ref, model = T._pair(cfg, seed=B)
ref.train(); model.train()
g = torch.Generator().manual_seed(9)
x = torch.randn(B, 4, S, S, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
ctx = torch.randn(B, Lc, cfg["context_dim"], generator=g); w = torch.randn(B, 4, S, S, generator=g) * 0.1
out_ref = ref(x, timesteps=t, context=ctx); (out_ref * w).sum().backward()
out = model(x.to("cuda"), timesteps=t.to("cuda"), context=ctx.to("cuda")); (out * w.to("cuda")).sum().backward()
rows = []
for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
    ga, gb = p.grad.detach().cpu().flatten(), q.grad.flatten()
    rows.append((((ga - gb).norm() / (gb.norm() + 1e-30)).item(), n, gb.norm().item(), ga.norm().item()))
rows.sort(key=lambda r: -r[0])
for r in rows[:25]:
    print("%.3e %-70s ref %.3e ours %.3e" % r)
