"""Host side of the DDPM (CIFAR-10) epsilon loss (csrc/loss.hip, "DDPM" section).

API mirror of /root/reference/DDPM/functions/losses.py:
  ``noise_estimation_loss_conditional(model, x0, t, c, e, b, cond_drop_prob=0.1, keepdim=False)`` (:22-38),
  ``loss_registry_conditional["simple"]`` (:45-47), ``adaptive_loss(loss_fn, model, x0, t, c, e, b, lambd)`` (:49-69),
  ``cosine_lr_scheduler(alpha, step, n_steps)`` (:71-72), and of ``get_beta_schedule`` for the schedule cifar10_sfron.yml
  uses (DDPM/runners/diffusion.py:36-66: fp64 linspace -> fp32 tensor).
The denoiser is any ``model(x_t, t.float(), c, mode="train", cond_drop_prob=...)`` callable whose output takes part in
autograd; everything around it (alphas_cumprod, q-sample, per-sample sums, adaptive weights, the gradient handed to the
denoiser's backward) runs in the HIP library.  With ``dp_group`` the adaptive weights are normalised over the GLOBAL
batch (one all-reduce of a scalar).
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def get_beta_schedule(beta_schedule="linear", *, beta_start=1e-4, beta_end=2e-2, num_diffusion_timesteps=1000, device="cuda"):
    if beta_schedule != "linear":
        raise NotImplementedError("cifar10_sfron.yml uses the linear schedule")
    return torch.from_numpy(np.linspace(beta_start, beta_end, num_diffusion_timesteps, dtype=np.float64)).float().to(device)


def alphas_cumprod(b):
    out = torch.empty_like(b)
    check(_lib.lib().sfron_ddpm_alphas_cumprod(ptr(b), b.numel(), ptr(out), stream_ptr()), "ddpm_alphas_cumprod")
    return out


def q_sample(x0, e, t, abar):
    x0, e = x0.contiguous(), e.contiguous()
    xt = torch.empty_like(x0)
    check(_lib.lib().sfron_ddpm_q_sample(ptr(x0), ptr(e), ptr(t), ptr(abar), x0.shape[0], x0[0].numel(), ptr(xt), stream_ptr()),
          "ddpm_q_sample")
    return xt


class _SampleLoss(torch.autograd.Function):
    """per_sample[i] = sum_chw (e - out)^2 ; backward: d out_i = g_i * 2 (out_i - e_i) through sfron_ddpm_loss_bwd."""

    @staticmethod
    def forward(ctx, out, e):
        out, e = out.contiguous(), e.contiguous()
        n, chw = out.shape[0], out[0].numel()
        per = torch.empty(n, dtype=torch.float32, device=out.device)
        check(_lib.lib().sfron_ddpm_sample_loss(ptr(e), ptr(out), n, chw, ptr(per), stream_ptr()), "ddpm_sample_loss")
        ctx.save_for_backward(out, e)
        return per

    @staticmethod
    def backward(ctx, g):
        out, e = ctx.saved_tensors
        coef = (2.0 * g).contiguous().float()
        d = torch.empty_like(out)
        check(_lib.lib().sfron_ddpm_loss_bwd(ptr(e), ptr(out), ptr(coef), out.shape[0], out[0].numel(), ptr(d), stream_ptr()),
              "ddpm_loss_bwd")
        return d, None


class _Reduce(torch.autograd.Function):
    """loss = mean(per) ("simple") or sum(w*per)/sum(w) with detached adaptive weights ("adaga"); the backward hands
    per-sample coefficients to _SampleLoss (coef_i / 2 = d loss / d per_i)."""

    @staticmethod
    def forward(ctx, per, mode, lambd, dp_group):
        n = per.shape[0]
        world = torch.distributed.get_world_size(dp_group) if dp_group is not None else 1
        coef = torch.empty_like(per)
        loss = torch.empty((), dtype=torch.float32, device=per.device)
        wsum = torch.zeros((), dtype=torch.float32, device=per.device)
        L = _lib.lib()
        check(L.sfron_ddpm_loss_coef(ptr(per), n, mode, float(lambd), 1.0, n * world, ptr(wsum), 0, ptr(coef), ptr(loss),
                                     stream_ptr()), "ddpm_loss_coef")
        if mode == 1 and dp_group is not None:
            torch.distributed.all_reduce(wsum, group=dp_group)
            check(L.sfron_ddpm_loss_coef(ptr(per), n, mode, float(lambd), 1.0, n * world, ptr(wsum), 1, ptr(coef), ptr(loss),
                                         stream_ptr()), "ddpm_loss_coef")
        ctx.save_for_backward(coef)
        return loss

    @staticmethod
    def backward(ctx, g):
        (coef,) = ctx.saved_tensors
        return coef * (0.5 * g), None, None, None


def noise_estimation_loss_conditional(model, x0, t, c, e, b, cond_drop_prob=0.1, keepdim=False, dp_group=None):
    abar = alphas_cumprod(b)
    x = q_sample(x0, e, t, abar)
    output = model(x, t.float(), c, cond_drop_prob=cond_drop_prob, mode="train")
    per = _SampleLoss.apply(output, e)
    if keepdim:
        return per
    return _Reduce.apply(per, 0, 0.0, dp_group)


loss_registry_conditional = {"simple": noise_estimation_loss_conditional}


def adaptive_loss(loss_fn, model, x0, t, c, e, b, lambd=0.5, dp_group=None):
    per = loss_fn(model, x0, t, c, e, b, keepdim=True)
    return _Reduce.apply(per, 1, lambd, dp_group)


def cosine_lr_scheduler(alpha, step, n_steps):
    return alpha * (1 + math.cos(math.pi * step / n_steps)) / 2
