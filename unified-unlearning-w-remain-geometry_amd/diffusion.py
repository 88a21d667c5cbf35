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


def space_timesteps(num_timesteps, section_counts):
    """respace.py:12-62: which original timesteps a respaced process keeps (returned as a set, like the reference)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


def _tables_fp64(num_timesteps=1000, use_timesteps=None):
    # gaussian_diffusion.py:98-115 (linear), respace.py:65-87 (betas re-derived from the cumprod of the retained
    # steps; also when every step is kept), gaussian_diffusion.py:163-201
    scale = 1000 / num_timesteps
    base = np.linspace(scale * 0.0001, scale * 0.02, num_timesteps, dtype=np.float64)
    base_ac = np.cumprod(1.0 - base, axis=0)
    last, nb, tmap = 1.0, [], []
    for i, ac in enumerate(base_ac):
        if use_timesteps is None or i in use_timesteps:
            nb.append(1 - ac / last)
            last = ac
            tmap.append(i)
    betas = np.array(nb, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(
        timestep_map=tmap, betas=betas, alphas_cumprod=ac,
        sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
        sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac), sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
        posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        posterior_log_variance_clipped=np.log(np.append(post_var[1], post_var[1:])),
        log_betas=np.log(betas))


_TAB_ORDER = ["sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
              "posterior_log_variance_clipped", "log_betas"]


class _WrappedModel:
    """respace.py:117-129: the model sees the ORIGINAL timestep of each respaced index."""

    def __init__(self, model, map_tensor):
        self.model, self.map_tensor = model, map_tensor

    def __call__(self, x, ts, **kwargs):
        return self.model(x, self.map_tensor.to(ts.dtype)[ts], **kwargs)


class GaussianDiffusion:
    """epsilon-prediction, LEARNED_RANGE variance, MSE loss: what create_diffusion(timestep_respacing) builds
    (a SpacedDiffusion; "" keeps all steps)."""

    def __init__(self, num_timesteps=1000, device="cuda", use_timesteps=None):
        self.tables = _tables_fp64(num_timesteps, use_timesteps)
        self.original_num_steps = num_timesteps
        self.timestep_map = self.tables["timestep_map"]
        self.num_timesteps = len(self.timestep_map)
        self._identity_map = self.timestep_map == list(range(num_timesteps))
        self.map_tensor = torch.tensor(self.timestep_map, dtype=torch.int64, device=device)
        packed = np.stack([self.tables[k] for k in _TAB_ORDER], axis=1).astype(np.float32)
        self.tab = torch.from_numpy(packed).to(device).contiguous()

    def _wrap_model(self, model):
        if self._identity_map or isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.map_tensor)

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
        model_output = self._wrap_model(model)(x_t, t, **model_kwargs)
        mse, vb, loss = _DitLoss.apply(model_output, self, x_start.contiguous(), t.contiguous(), noise.contiguous())
        return {"loss": loss, "mse": mse, "vb": vb}


    # ------------------------------------------------------------------ sampling (gaussian_diffusion.py:376-511)
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, noise=None):
        """One ancestral step; ``noise`` (optional) replaces the reference's ``th.randn_like(x)`` draw."""
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are not used by the unlearning scripts")
        x = x.contiguous().float()
        t = t.contiguous()
        model_output = self._wrap_model(model)(x, t, **(model_kwargs or {}))
        if isinstance(model_output, tuple):
            model_output = model_output[0]
        n, c = x.shape[0], x.shape[1]
        assert model_output.shape == (n, 2 * c, *x.shape[2:])
        if noise is None:
            noise = torch.randn_like(x)
        sample, pred = torch.empty_like(x), torch.empty_like(x)
        mo, nz = model_output.contiguous().float(), noise.contiguous()       # named: temporaries must outlive the launch
        check(_lib.lib().sfron_p_sample(ptr(x), ptr(mo), ptr(t), ptr(self.tab), ptr(nz),
                                        n, c, x[0, 0].numel(), int(bool(clip_denoised)), ptr(sample), ptr(pred), stream_ptr()),
              "p_sample")
        return {"sample": sample, "pred_xstart": pred}

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, step_noise=None):
        device = device if device is not None else self.tab.device
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for k, i in enumerate(indices):
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                                    model_kwargs=model_kwargs, noise=None if step_noise is None else step_noise[k])
            yield out
            img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False, step_noise=None):
        """Reference signature (gaussian_diffusion.py:423); ``step_noise[k]`` optionally fixes the k-th per-step draw."""
        final = None
        for final in self.p_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
                                                    progress, step_noise):
            pass
        return final["sample"]


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
    """DiT/diffusion/__init__.py:10-46 for the defaults the unlearning scripts use (linear schedule, learn_sigma, MSE)."""
    use = None if timestep_respacing in ("", None) else space_timesteps(diffusion_steps, timestep_respacing)
    return GaussianDiffusion(diffusion_steps, device=device, use_timesteps=use)
