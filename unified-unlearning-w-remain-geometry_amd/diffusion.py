"""Host side of the DiT diffusion loss (csrc/loss.hip).

API mirror of /root/reference/DiT/diffusion: ``create_diffusion("")`` returns an object with
``num_timesteps``, ``q_sample(x_start, t, noise)`` and
``training_losses(model, x_start, t, model_kwargs=None, noise=None) -> {"loss","mse","vb"}``
(gaussian_diffusion.py:215-230,715-787; __init__.py:10-46).  Tables are built host-side in fp64
with the same operation order as the reference and rounded once to fp32 for the kernels.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _tables_fp64(num_timesteps=1000):
    # gaussian_diffusion.py:98-115 (linear), respace.py:70-87 (betas re-derived from the cumprod),
    # gaussian_diffusion.py:163-201
    scale = 1000 / num_timesteps
    base = np.linspace(scale * 0.0001, scale * 0.02, num_timesteps, dtype=np.float64)
    base_ac = np.cumprod(1.0 - base, axis=0)
    last, nb = 1.0, []
    for ac in base_ac:
        nb.append(1 - ac / last)
        last = ac
    betas = np.array(nb, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(
        betas=betas, alphas_cumprod=ac,
        sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
        sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac), sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
        posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        posterior_log_variance_clipped=np.log(np.append(post_var[1], post_var[1:])),
        log_betas=np.log(betas))


_TAB_ORDER = ["sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped", "log_betas"]


class GaussianDiffusion:
    """epsilon-prediction, LEARNED_RANGE variance, MSE loss: what create_diffusion("") builds."""

    def __init__(self, num_timesteps=1000, device="cuda"):
        self.tables = _tables_fp64(num_timesteps)
        self.num_timesteps = num_timesteps
        packed = np.stack([self.tables[k] for k in _TAB_ORDER], axis=1).astype(np.float32)
        self.tab = torch.from_numpy(packed).to(device).contiguous()

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        x_start, noise = x_start.contiguous(), noise.contiguous()
        out = torch.empty_like(x_start)
        n = x_start.shape[0]
        check(_lib.lib().sfron_q_sample(ptr(x_start), ptr(noise), ptr(t.contiguous()), ptr(self.tab), n,
                                        x_start.numel() // n, ptr(out), stream_ptr()), "q_sample")
        return out

    def loss_fwd_bwd(self, model_output, x_start, t, noise, grad_scale):
        """Fused loss forward + d(grad_scale * sum_i loss_i)/d model_output.  Returns (mse[N], vb[N], d_out)."""
        n, c2 = model_output.shape[0], model_output.shape[1]
        c = c2 // 2
        hw = model_output.numel() // (n * c2)
        mse = torch.empty(n, dtype=torch.float32, device=model_output.device)
        vb = torch.empty_like(mse)
        d_out = torch.empty_like(model_output)
        check(_lib.lib().sfron_dit_loss_fwd_bwd(ptr(x_start), ptr(noise), ptr(model_output), ptr(t), ptr(self.tab),
                                                n, c, hw, float(grad_scale), ptr(mse), ptr(vb), ptr(d_out),
                                                stream_ptr()), "dit_loss_fwd_bwd")
        return mse, vb, d_out

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """Reference signature (gaussian_diffusion.py:715).  ``terms["loss"]`` carries an autograd edge into
        ``model``'s output, so ``terms["loss"].mean().backward()`` works as in DiT/forget.py:271-288."""
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = torch.randn_like(x_start)
        x_t = self.q_sample(x_start, t, noise)
        model_output = model(x_t, t, **model_kwargs)
        mse, vb, loss = _DitLoss.apply(model_output, self, x_start.contiguous(), t.contiguous(), noise.contiguous())
        return {"loss": loss, "mse": mse, "vb": vb}


class _DitLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_output, diff, x_start, t, noise):
        mse, vb, d_out = diff.loss_fwd_bwd(model_output.contiguous(), x_start, t, noise, 1.0)
        ctx.save_for_backward(d_out)
        ctx.c = model_output.shape[1] // 2
        loss = mse + vb
        ctx.mark_non_differentiable(mse, vb)
        return mse, vb, loss

    @staticmethod
    def backward(ctx, g_mse, g_vb, g_loss):
        (d_out,) = ctx.saved_tensors
        # per-sample upstream weights; d_out already holds d loss_i / d out for unit weights
        g = g_loss.view(-1, *([1] * (d_out.dim() - 1)))
        return d_out * g, None, None, None, None


def create_diffusion(timestep_respacing="", diffusion_steps=1000, device="cuda", **unused):
    if timestep_respacing not in ("", None):
        raise NotImplementedError("only the training configuration create_diffusion('') is on the hot path")
    return GaussianDiffusion(diffusion_steps, device=device)
