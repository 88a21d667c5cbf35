"""Oracle (test infrastructure): the parameter-side sweep of one SFR-on stage.

Restates, with torch itself as the arithmetic reference (torch.optim.Adam/AdamW and
torch.nn.utils.clip_grad_norm_ are importable here, so they ARE the pinned reference):
  * mask application ... /root/reference/DiT/forget.py:289-292,
                         /root/reference/DDPM/runners/diffusion.py:1126-1129
  * global-norm clip ... DiT/forget.py:293-298, DDPM/runners/diffusion.py:1131-1136,1169-1174
  * optimizer .......... DiT/forget.py:199 (AdamW, wd=0), DDPM/functions/__init__.py:9-18 (Adam)
  * EMA ................ DiT/forget.py:52-62 (ema = d*ema + (1-d)*p over ALL named params),
                         DDPM/models/ema.py:17-24 (shadow = (1-mu)*p + mu*shadow, trainable only)
  * mask from Fisher ... DiT/generate_mask.py:31-36, DDPM/generate_fisher_mask.py:39-46
  * Fisher accumulate .. DiT/generate_fisher.py:236-239

All functions work on lists of tensors or flat fp32 tensors on CPU.
"""
import torch


def apply_mask_(grads, masks):
    """grad *= mask (bool -> float multiply).  ``masks[i]`` may be None (no entry) or the
    python int 0 the reference stores for never-grad params."""
    for g, m in zip(grads, masks):
        if g is None or m is None:
            continue
        if isinstance(m, int):
            g.mul_(m)
        else:
            g.mul_(m.to(g.dtype))


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics: total = ||(||g_i||)||_2,
    coef = max_norm / (total + 1e-6) clamped to 1.0, g *= coef.  Returns total."""
    params = []
    for g in grads:
        p = torch.nn.Parameter(torch.empty_like(g))
        p.grad = g
        params.append(p)
    # call the real thing on stand-in Parameters whose .grad are the given tensors
    return torch.nn.utils.clip_grad_norm_(params, max_norm)


class AdamRef:
    """Thin holder around torch.optim.AdamW/Adam on flat CPU tensors so tests can drive
    'two .step() per iteration on one shared state' (DiT/forget.py:199,299,320)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=True):
        self.params = [torch.nn.Parameter(p.clone()) for p in params]
        cls = torch.optim.AdamW if adamw else torch.optim.Adam
        self.opt = cls(self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def step(self, grads):
        for p, g in zip(self.params, grads):
            p.grad = g.clone()
        self.opt.step()

    def state(self, i):
        st = self.opt.state[self.params[i]]
        return st["exp_avg"], st["exp_avg_sq"], int(st["step"])


def ema_update_dit_(ema, params, decay=0.9999):
    # DiT/forget.py:52-62 : ema.mul_(decay).add_(p, alpha=1-decay)
    for e, p in zip(ema, params):
        e.mul_(decay).add_(p, alpha=1 - decay)


def ema_update_ddpm_(shadow, params, mu):
    # DDPM/models/ema.py:17-24 : shadow = (1-mu)*p + mu*shadow
    for i, p in enumerate(params):
        shadow[i] = (1.0 - mu) * p + mu * shadow[i]


def mask_from_fisher(forget_fisher, remain_fisher, th):
    """DiT/generate_mask.py:34-35: ((F_f + 1e-15) / (F_r + 1e-15)) >= th, fp32 tensors,
    python-float threshold compared against an fp32 tensor."""
    return ((forget_fisher + 1e-15) / (remain_fisher + 1e-15)) >= th


def fisher_accumulate_(fisher, grad, n_iters):
    # DiT/generate_fisher.py:239 : F += grad**2 / n_iters
    fisher += (grad ** 2) / n_iters
