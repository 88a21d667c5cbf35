"""Oracle (test infrastructure): one SFR-on iteration, restated.

DiT policy   -- /root/reference/DiT/forget.py:256-322 (method "ron"; losses "ga" / "rl"):
   forget fwd -> alpha * (-/+) loss.mean() -> zero_grad, backward -> grad *= mask ->
   clip_grad_norm_(1.0) -> AdamW.step -> remain fwd -> zero_grad, backward -> AdamW.step
   (no mask, no clip) -> update_ema(decay 0.9999, all named params).
DDPM policy  -- /root/reference/DDPM/runners/diffusion.py:1075-1180: same with cosine-decayed
   alpha (:1076-1079), losses "ga"/"adaga" (functions/losses.py:22-69), clip in BOTH stages
   (:1131-1136,1169-1174) and EMAHelper.update (models/ema.py:17-24).

Every random draw (t, noise, label/CFG drop) is an explicit input so that the HIP
path and this oracle consume identical values.
"""
import math

import torch

from . import diffusion_ref as dref


def cosine_alpha(base, step, n_iters):
    # DDPM/functions/losses.py:71-72
    return base * (1 + math.cos(math.pi * step / n_iters)) / 2


def _named_trainable(model):
    return [(n, p) for n, p in model.named_parameters() if p.requires_grad]


class DiTSfronOracle:
    """State = (model, AdamW on model.parameters(), ema list over all named params)."""

    def __init__(self, model, tables, lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999,
                 mask=None, unlearn_loss="ga", forget_class=0, num_classes=1000, method="ron"):
        if method not in ("ron", "joint"):
            raise ValueError(method)
        self.method = method        # "joint": forget.py:314-316 -- one step on remain + alpha * forget, no mask, no clip
        self.model = model
        self.tab = tables
        self.opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=0)   # forget.py:199
        # forget.py:190,230 : ema = deepcopy(model); update_ema(decay=0) == copy
        self.ema = {n: p.detach().clone() for n, p in model.named_parameters()}
        self.forget_alpha = forget_alpha
        self.grad_clip = grad_clip
        self.ema_decay = ema_decay
        self.mask = mask            # dict name -> bool tensor / int 0, keys with or without "module."
        self.unlearn_loss = unlearn_loss
        self.forget_class = forget_class
        self.num_classes = num_classes
        model.train()

    def _mask_for(self, name):
        if self.mask is None:
            return None
        if name in self.mask:
            return self.mask[name]
        return self.mask.get("module." + name)

    def _losses(self, x0, y, t, noise, drop):
        fn = lambda x_t, ts, y: self.model(x_t, ts, y, force_drop_ids=drop)
        return dref.training_losses(self.tab, fn, x0, t, dict(y=y), noise)

    def step(self, forget, remain):
        """forget / remain: dicts with x0 [N,4,H,W], y [N] int64, t [N] int64, noise, drop [N] (0/1).
        Returns dict of python floats + the pre-clip forget grad norm."""
        m = self.model
        # ---- forget stage (forget.py:258-299)
        if self.unlearn_loss == "ga":
            terms_f = self._losses(forget["x0"], forget["y"], forget["t"], forget["noise"], forget["drop"])
            ori_forget = -terms_f["loss"].mean()
        elif self.unlearn_loss == "rl":
            pseudo = torch.full_like(forget["y"], (self.forget_class + 100) % 1000)
            terms_f = self._losses(forget["x0"], pseudo, forget["t"], forget["noise"], forget["drop"])
            ori_forget = terms_f["loss"].mean()
        else:
            raise ValueError(f"unsupported unlearn_loss {self.unlearn_loss!r} (SURVEY.md section 9 Q4)")
        gnorm = float("nan")
        if self.method == "ron":                                   # forget.py:283-299
            forget_loss = self.forget_alpha * ori_forget
            self.opt.zero_grad()
            forget_loss.backward()
            if self.mask is not None:
                for name, p in m.named_parameters():
                    if p.grad is not None:
                        mk = self._mask_for(name)
                        if mk is not None:
                            p.grad *= mk
            gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), self.grad_clip)
            self.opt.step()
        # ---- remain stage (forget.py:301-320)
        terms_r = self._losses(remain["x0"], remain["y"], remain["t"], remain["noise"], remain["drop"])
        ori_remain = terms_r["loss"].mean()
        self.opt.zero_grad()
        loss = ori_remain + self.forget_alpha * ori_forget if self.method == "joint" else ori_remain       # forget.py:314-318
        loss.backward()
        self.opt.step()
        # ---- EMA (forget.py:322)
        with torch.no_grad():
            for name, p in m.named_parameters():
                self.ema[name].mul_(self.ema_decay).add_(p.data, alpha=1 - self.ema_decay)
        return {"forget_loss": ori_forget.item(), "remain_loss": ori_remain.item(),
                "forget_mse": terms_f["mse"].mean().item(), "remain_mse": terms_r["mse"].mean().item(),
                "forget_gnorm": float(gnorm)}


# ----------------------------------------------------------------------------- DDPM side
def ddpm_alphas_cumprod_fp32(betas_fp32):
    # DDPM/functions/losses.py:32 : fp32 cumprod on device every call
    return (1 - betas_fp32).cumprod(dim=0)


def ddpm_get_betas(num_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    # DDPM/runners/diffusion.py:36-66,83 : fp64 linspace -> fp32 tensor
    import numpy as np
    return torch.from_numpy(np.linspace(beta_start, beta_end, num_timesteps, dtype=np.float64)).float()


def ddpm_loss_per_sample(model_fn, x0, t, e, b):
    """DDPM/functions/losses.py:22-38 with keepdim=True: sum over C,H,W of (e - eps_hat)^2."""
    a = ddpm_alphas_cumprod_fp32(b).index_select(0, t).view(-1, 1, 1, 1)
    x = x0 * a.sqrt() + e * (1.0 - a).sqrt()
    out = model_fn(x, t.float())
    return (e - out).square().sum(dim=(1, 2, 3))


def ddpm_adaptive_loss(per_sample, lambd):
    # DDPM/functions/losses.py:49-69
    size = per_sample.shape[0]
    coef = 1 / (torch.pow(per_sample.detach().clone(), lambd) + 1e-8)
    return ((coef / coef.sum()) * per_sample * size).mean(dim=0)


class DDPMSfronOracle:
    """DDPM/runners/diffusion.py:1075-1180 over an arbitrary eps-model ``model(x, t_float, c, drop)``."""

    def __init__(self, model, betas, lr=1e-4, forget_alpha=10.0, remain_alpha=1.0, grad_clip=1.0,
                 ema_mu=1e-4, mask=None, unlearn_loss="adaga", lambd=0.5, n_iters=50, decay_forget_alpha=True, label_to_forget=0,
                 n_classes=10, method="ron"):
        self.label_to_forget, self.n_classes = label_to_forget, n_classes
        self.method = method        # "joint": runners/diffusion.py:1160-1167 -- ONE clipped Adam step on remain_loss + forget_loss
        self.model = model
        self.b = betas
        # DDPM/functions/__init__.py:9-18 with cifar10_sfron.yml:48-56
        self.opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0, betas=(0.9, 0.999),
                                    amsgrad=False, eps=1e-8)
        self.shadow = {n: p.data.clone() for n, p in model.named_parameters() if p.requires_grad}
        self.forget_alpha, self.remain_alpha = forget_alpha, remain_alpha
        self.grad_clip, self.mu, self.mask = grad_clip, ema_mu, mask
        self.unlearn_loss, self.lambd = unlearn_loss, lambd
        self.n_iters, self.decay = n_iters, decay_forget_alpha
        model.train()

    def step(self, step_idx, forget, remain):
        m = self.model
        alpha = cosine_alpha(self.forget_alpha, step_idx, self.n_iters) if self.decay else self.forget_alpha
        fn_f = lambda x, tf: m(x, tf, forget["c"], forget["drop"])
        if self.unlearn_loss == "rl":
            # DDPM/runners/diffusion.py:1101-1113: MSE between the prediction under the forget label and the (detached)
            # prediction under the pseudo label (label_to_forget + 1) % 10, on the same x_t
            a = ddpm_alphas_cumprod_fp32(self.b).index_select(0, forget["t"]).view(-1, 1, 1, 1)
            x = forget["x0"] * a.sqrt() + forget["e"] * (1.0 - a).sqrt()
            output = m(x, forget["t"].float(), forget["c"], forget["drop"])
            pseudo_c = torch.full(forget["c"].shape, (self.label_to_forget + 1) % self.n_classes)
            pseudo = m(x, forget["t"].float(), pseudo_c, forget["drop"]).detach()
            ori_forget = torch.nn.functional.mse_loss(pseudo, output)
        else:
            per = ddpm_loss_per_sample(fn_f, forget["x0"], forget["t"], forget["e"], self.b)
            if self.unlearn_loss == "ga":
                ori_forget = -per.mean(dim=0)
            elif self.unlearn_loss == "adaga":
                ori_forget = -ddpm_adaptive_loss(per, self.lambd)
            else:
                raise ValueError(self.unlearn_loss)
        forget_loss = alpha * ori_forget
        if self.method == "ron":                                      # :1122-1139
            self.opt.zero_grad()
            forget_loss.backward()
            if self.mask is not None:
                for name, p in m.named_parameters():
                    if p.grad is not None and name in self.mask:
                        p.grad *= self.mask[name]
            torch.nn.utils.clip_grad_norm_(m.parameters(), self.grad_clip)
            self.opt.step()
        fn_r = lambda x, tf: m(x, tf, remain["c"], remain["drop"])
        ori_remain = ddpm_loss_per_sample(fn_r, remain["x0"], remain["t"], remain["e"], self.b).mean(dim=0)
        if self.method == "ron":
            self.opt.zero_grad()
            (self.remain_alpha * ori_remain).backward()
        else:
            # :1160-1167 as written: the mask loop runs on the STALE gradients of the previous step and zero_grad() follows it, so
            # the mask has no effect on the update; both graphs are back-propagated from the same (not yet updated) weights
            loss = self.remain_alpha * ori_remain + forget_loss
            if self.mask is not None:
                for name, p in m.named_parameters():
                    if p.grad is not None and name in self.mask:
                        p.grad *= self.mask[name]
            self.opt.zero_grad()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), self.grad_clip)
        self.opt.step()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.requires_grad:
                    self.shadow[n] = (1.0 - self.mu) * p.data + self.mu * self.shadow[n]
        return {"forget_loss": ori_forget.item(), "remain_loss": ori_remain.item(), "alpha": alpha}


def ddpm_fisher(model, batches, b, cond_scale=2.0, grad_clip=1.0):
    """DDPM/runners/diffusion.py:1244-1299 (one of the two identical loops): eval mode, guided mode="test" forward WITH
    gradients through both branches, Sigma_chw (e - output)^2 averaged over the batch, clip_grad_norm_ BEFORE squaring,
    F[name] += grad^2 / len(loader).  ``model(x, t_float, c, mode="test", cond_scale=...)`` is oracle.ddpm_ref.ConditionalUNet."""
    model.eval()
    F = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    for bt in batches:
        a = ddpm_alphas_cumprod_fp32(b).index_select(0, bt["t"]).view(-1, 1, 1, 1)
        x = bt["x0"] * a.sqrt() + bt["e"] * (1.0 - a).sqrt()
        output = model(x, bt["t"].float(), bt["c"], cond_scale=cond_scale, mode="test")
        loss = (bt["e"] - output).square().sum(dim=(1, 2, 3)).mean(dim=0)
        model.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.grad is not None:
                    F[n] += p.grad.data ** 2 / len(batches)
    return F


# ----------------------------------------------------------------------------- DDPM sampler (snapshots)
def ddpm_compute_alpha(beta, t):
    # DDPM/functions/denoising.py:4-7
    beta = torch.cat([torch.zeros(1), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def ddpm_generalized_steps_conditional(x, c, seq, model, b, cond_scale=3.0, eta=0.0, step_noise=None):
    """DDPM/functions/denoising.py:72-95 (DDIM-style update with eta); ``step_noise[k]`` replaces the k-th randn_like draw."""
    with torch.no_grad():
        n = x.size(0)
        seq_next = [-1] + list(seq[:-1])
        xs, x0_preds = [x], []
        for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
            t = torch.ones(n) * i
            next_t = torch.ones(n) * j
            at = ddpm_compute_alpha(b, t.long())
            at_next = ddpm_compute_alpha(b, next_t.long())
            xt = xs[-1]
            et = model(xt, t, c, cond_scale=cond_scale, mode="test")
            x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
            x0_preds.append(x0_t)
            c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            c2 = ((1 - at_next) - c1 ** 2).sqrt()
            noise = torch.randn_like(x) if step_noise is None else step_noise[k]
            xs.append(at_next.sqrt() * x0_t + c1 * noise + c2 * et)
    return xs, x0_preds
