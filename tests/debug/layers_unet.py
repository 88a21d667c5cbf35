import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/debug/ -> repo root (debugging aid; uses the oracle, hence under tests/)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_unet as T
cfg, B, p = T.SMALL, 4, 0.0
ref, model = T._pair(cfg, seed=B)
model.dropout_p = p
first = []
def hook(i, step):
    torch.cuda.synchronize()
    if not first and bool(torch.isnan(model.grads).any()):
        import inspect
        first.append(i)
        cv = inspect.getclosurevars(step)
        print("first NaN in the gradient arena after backward step", i, step.__name__, {k: (v if isinstance(v, (str, int)) else type(v).__name__) for k, v in cv.nonlocals.items()})
model._tape_hook = hook
model.BATCH_REDUCTIONS = False          # the hook looks at the gradient arena after every step: every finish launched where it is asked for
out, out_ref = T._fwd_bwd_both(ref, model, cfg, B, seed=11, p_drop=p)
rows = []
for (n, pp), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
    ga, gb = pp.grad.detach().cpu().flatten(), q.grad.flatten()
    rows.append((((ga - gb).norm() / (gb.norm() + 1e-30)).item(), n, gb.norm().item(), ga.norm().item(), bool(torch.isnan(ga).any())))
rows.sort(key=lambda r: -r[0] if r[0] == r[0] else -1e30)
print("ok tensors:", [r[1] for r in rows if not r[4]][:60])
for r in rows[:6]:
    print("%.3e %-45s ref %.3e ours %.3e nan=%s" % r)
print("nan tensors:", [r[1] for r in rows if r[4] or r[0] != r[0]][:30])
