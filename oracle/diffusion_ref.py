"""Oracle (test infrastructure): DiT-side Gaussian diffusion loss, restated.

Follows /root/reference/DiT/diffusion:
  * tables ............ gaussian_diffusion.py:98-115,153-201 (+ the SpacedDiffusion
                        re-derivation of betas, respace.py:70-87, with
                        timestep_respacing="" i.e. all 1000 steps kept)
  * q_sample .......... gaussian_diffusion.py:215-230
  * _extract .......... gaussian_diffusion.py:861-873 (fp64 table -> .float() at gather)
  * vb term ........... gaussian_diffusion.py:232-252,285-293,334-339,682-713
  * normal_kl / cdf ... diffusion_utils.py:10-44,62-88
  * training_losses ... gaussian_diffusion.py:715-787 (MSE + LEARNED_RANGE branch)
  * sampling .......... respace.py:12-62 (space_timesteps), :65-129 (SpacedDiffusion tables, timestep map, wrapped
                        model), gaussian_diffusion.py:254-332 (p_mean_variance, LEARNED_RANGE / EPSILON),
                        :376-421 (p_sample), :423-511 (p_sample_loop)

Plain PyTorch fp32 on CPU; no reference code is imported here.
"""
import math

import numpy as np
import torch as th


def space_timesteps(num_timesteps, section_counts):
    """respace.py:12-62: the retained original timesteps, as a sorted list."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return sorted(range(0, num_timesteps, stride))
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
    return sorted(set(steps))


class DiffusionTables:
    """fp64 numpy tables exactly as create_diffusion(timestep_respacing) builds them ("" = all steps kept)."""

    def __init__(self, num_timesteps=1000, timestep_respacing=""):
        # gaussian_diffusion.py:98-115 : linear schedule, scale = 1000/T
        scale = 1000 / num_timesteps
        base_betas = np.linspace(scale * 0.0001, scale * 0.02, num_timesteps, dtype=np.float64)
        # respace.py:70-87 : SpacedDiffusion recomputes betas from the cumprod even
        # when every step is kept (not bit-identical to base_betas in fp64).
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        use = set(range(num_timesteps)) if timestep_respacing in ("", None) else set(space_timesteps(num_timesteps, timestep_respacing))
        last = 1.0
        new_betas, self.timestep_map = [], []
        for i, ac in enumerate(base_ac):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        self.set_betas(betas)

    def set_betas(self, betas):
        # gaussian_diffusion.py:163-201
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)

    def packed_fp32(self):
        """[T, 8] fp32 table handed to the HIP loss kernel, in the column order
        include/sfron.h documents (SFRON_TAB_*).  Each entry is the fp64 value
        rounded once to fp32, which is what _extract_into_tensor does."""
        cols = [self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod,
                self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod,
                self.posterior_mean_coef1, self.posterior_mean_coef2,
                self.posterior_log_variance_clipped, self.log_betas]
        return np.stack(cols, axis=1).astype(np.float32)


def _extract(arr, t, shape):
    # gaussian_diffusion.py:861-873
    res = th.from_numpy(arr).to(device=t.device)[t].float()      # (:870 moves the table to the timesteps' device)
    while len(res.shape) < len(shape):
        res = res[..., None]
    return res + th.zeros(shape, device=t.device)


def mean_flat(x):
    return x.mean(dim=list(range(1, len(x.shape))))


def normal_kl(mean1, logvar1, mean2, logvar2):
    # diffusion_utils.py:10-36
    return 0.5 * (-1.0 + logvar2 - logvar1 + th.exp(logvar1 - logvar2)
                  + ((mean1 - mean2) ** 2) * th.exp(-logvar2))


def approx_standard_normal_cdf(x):
    # diffusion_utils.py:39-44
    return 0.5 * (1.0 + th.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * th.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, means, log_scales):
    # diffusion_utils.py:62-88
    centered_x = x - means
    inv_stdv = th.exp(-log_scales)
    plus_in = inv_stdv * (centered_x + 1.0 / 255.0)
    cdf_plus = approx_standard_normal_cdf(plus_in)
    min_in = inv_stdv * (centered_x - 1.0 / 255.0)
    cdf_min = approx_standard_normal_cdf(min_in)
    log_cdf_plus = th.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = th.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    return th.where(x < -0.999, log_cdf_plus,
                    th.where(x > 0.999, log_one_minus_cdf_min, th.log(cdf_delta.clamp(min=1e-12))))


def q_sample(tab, x_start, t, noise):
    # gaussian_diffusion.py:215-230
    return (_extract(tab.sqrt_alphas_cumprod, t, x_start.shape) * x_start
            + _extract(tab.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)


def vb_terms_bpd(tab, eps_frozen, var_values, x_start, x_t, t):
    """gaussian_diffusion.py:682-713 with the LEARNED_RANGE / EPSILON branches of
    p_mean_variance (:285-293, :317-332) and clip_denoised=False."""
    shape = x_t.shape
    true_mean = (_extract(tab.posterior_mean_coef1, t, shape) * x_start
                 + _extract(tab.posterior_mean_coef2, t, shape) * x_t)
    true_logvar = _extract(tab.posterior_log_variance_clipped, t, shape)
    min_log = _extract(tab.posterior_log_variance_clipped, t, shape)
    max_log = _extract(tab.log_betas, t, shape)
    frac = (var_values + 1) / 2
    model_logvar = frac * max_log + (1 - frac) * min_log
    pred_xstart = (_extract(tab.sqrt_recip_alphas_cumprod, t, shape) * x_t
                   - _extract(tab.sqrt_recipm1_alphas_cumprod, t, shape) * eps_frozen)
    model_mean = (_extract(tab.posterior_mean_coef1, t, shape) * pred_xstart
                  + _extract(tab.posterior_mean_coef2, t, shape) * x_t)
    kl = mean_flat(normal_kl(true_mean, true_logvar, model_mean, model_logvar)) / np.log(2.0)
    nll = -discretized_gaussian_log_likelihood(x_start, means=model_mean, log_scales=0.5 * model_logvar)
    nll = mean_flat(nll) / np.log(2.0)
    return th.where(t == 0, nll, kl)


def training_losses_from_output(tab, model_output, x_start, x_t, t, noise):
    """gaussian_diffusion.py:746-783 given the model output [N, 2C, H, W]."""
    C = x_t.shape[1]
    eps_hat, var_values = th.split(model_output, C, dim=1)
    vb = vb_terms_bpd(tab, eps_hat.detach(), var_values, x_start, x_t, t)
    mse = mean_flat((noise - eps_hat) ** 2)
    return {"loss": mse + vb, "mse": mse, "vb": vb}


def training_losses(tab, model, x_start, t, model_kwargs=None, noise=None):
    """gaussian_diffusion.py:715-787 (same signature as the reference, with the
    tables object in place of ``self``)."""
    if model_kwargs is None:
        model_kwargs = {}
    if noise is None:
        noise = th.randn_like(x_start)
    x_t = q_sample(tab, x_start, t, noise)
    model_output = model(x_t, t, **model_kwargs)
    return training_losses_from_output(tab, model_output, x_start, x_t, t, noise)


# ----------------------------------------------------------------------------- sampling
def p_mean_variance(tab, model, x, t, clip_denoised=True, model_kwargs=None):
    """gaussian_diffusion.py:254-332 for LEARNED_RANGE / EPSILON; `model` sees ORIGINAL timesteps (respace.py:117-129)."""
    B, C = x.shape[:2]
    ts = th.tensor(tab.timestep_map, dtype=t.dtype)[t]
    model_output = model(x, ts, **(model_kwargs or {}))
    assert model_output.shape == (B, C * 2, *x.shape[2:])
    eps, var_values = th.split(model_output, C, dim=1)
    min_log = _extract(tab.posterior_log_variance_clipped, t, x.shape)
    max_log = _extract(tab.log_betas, t, x.shape)
    frac = (var_values + 1) / 2
    log_variance = frac * max_log + (1 - frac) * min_log
    pred_xstart = _extract(tab.sqrt_recip_alphas_cumprod, t, x.shape) * x - _extract(tab.sqrt_recipm1_alphas_cumprod, t, x.shape) * eps
    if clip_denoised:
        pred_xstart = pred_xstart.clamp(-1, 1)
    mean = _extract(tab.posterior_mean_coef1, t, x.shape) * pred_xstart + _extract(tab.posterior_mean_coef2, t, x.shape) * x
    return {"mean": mean, "log_variance": log_variance, "pred_xstart": pred_xstart}


def p_sample(tab, model, x, t, clip_denoised=True, model_kwargs=None, noise=None):
    """gaussian_diffusion.py:376-421 (cond_fn None); noise defaults to th.randn_like(x), the reference's draw."""
    out = p_mean_variance(tab, model, x, t, clip_denoised, model_kwargs)
    if noise is None:
        noise = th.randn_like(x)
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return {"sample": out["mean"] + nonzero * th.exp(0.5 * out["log_variance"]) * noise, "pred_xstart": out["pred_xstart"]}


def p_sample_loop(tab, model, shape, noise=None, clip_denoised=True, model_kwargs=None, step_noise=None):
    """gaussian_diffusion.py:423-511.  step_noise[k] (optional) replaces the k-th randn_like draw (k = 0 at t = T-1)."""
    img = noise if noise is not None else th.randn(*shape)
    with th.no_grad():
        for k, i in enumerate(reversed(range(tab.num_timesteps))):
            t = th.tensor([i] * shape[0])
            img = p_sample(tab, model, img, t, clip_denoised, model_kwargs, None if step_noise is None else step_noise[k])["sample"]
    return img
