"""The SFR-on concept-erasure iteration for Stable Diffusion (BASELINE config 4) on the native UNet.

One call of ``SDSFRon.step`` = one iteration of /root/reference/SD/train-scripts/nsfw_removal.py:108-173:
  forget: x_f / x_p noised with the SAME t and noise (:134-141) -> eps(x_f, c_forget) and stop-gradient eps(x_p, c_pseudo)
          -> forget_alpha * MSE (:143-147) -> backward -> [mask] -> Adam.step
  remain: LDM eps loss of shared_step (:164-166; ldm/models/diffusion/ddpm.py:1286-1319 with logvar = 0) -> backward -> Adam.step
Same Adam state for both steps (:81), no gradient clipping, no EMA.  ``train_method`` "full" / "xattn" (:66-77: parameters whose
name contains "attn2").  The reference's mask application (:157-160) tests a parameter NAME against a list of Parameters and is
therefore never true: ``mask_mode="as_written"`` reproduces that (no masking), ``"intended"`` multiplies
the forget-stage gradients by the saliency mask as the sibling scripts do.
Latents and prompt embeddings arrive resident on the device (the VAE / CLIP front-end of model.get_input is outside the path).
"""
import numpy as np
import torch

from . import _lib, graphs, sweep
from ._lib import check, ptr, stream_ptr


class LDMSchedule:
    """register_schedule (ldm/models/diffusion/ddpm.py:153-240) for v1-inference.yaml: make_beta_schedule "linear" (a linspace of
    sqrt(beta), squared; util.py:21-30), fp64 numpy tables rounded to fp32 buffers.  Packed as the [T][8] table of sfron_q_sample."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.012, device="cuda"):
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = timesteps
        tab = np.zeros((timesteps, _lib.TAB_COLS if hasattr(_lib, "TAB_COLS") else 8), dtype=np.float32)
        tab[:, 0] = np.sqrt(ac).astype(np.float32)
        tab[:, 1] = np.sqrt(1.0 - ac).astype(np.float32)
        self.tab = torch.from_numpy(tab).to(device).contiguous()

    def q_sample(self, x_start, t, noise):
        x_start, noise = x_start.contiguous(), noise.contiguous()
        out = torch.empty_like(x_start)
        n = x_start.shape[0]
        check(_lib.lib().sfron_q_sample(ptr(x_start), ptr(noise), ptr(t.contiguous()), ptr(self.tab), n, x_start.numel() // n, ptr(out),
                                        stream_ptr()), "q_sample")
        return out


class LatentDiffusion:
    """The part of ldm.models.diffusion.ddpm.LatentDiffusion the unlearning scripts call on the denoiser side, over the native UNet:
    ``model.diffusion_model`` (nsfw_removal.py:66, generate_fisher.py:30), ``num_timesteps``, ``q_sample`` (ddpm.py:424-445),
    ``apply_model(x_noisy, t, cond)`` (:1121-1131, crossattn conditioning: cond = the prompt embedding [B, 77, 768] or
    {"c_crossattn": [embedding]}), ``p_losses(x_start, cond, t, noise)`` -> (loss, loss_dict) (:1286-1319 with the v1-inference.yaml
    settings: eps-parameterisation, l2, logvar = 0 and not learned, l_simple_weight 1, original_elbo_weight 0).  The outputs take part
    in torch autograd (gradients land in the UNet's flat arena).  The first stage (VAE) and the text encoder (get_input,
    encode_first_stage, get_learned_conditioning, shared_step) are outside the path: latents and embeddings arrive resident."""

    parameterization, first_stage_key, cond_stage_key = "eps", "jpg", "txt"

    def __init__(self, unet, schedule=None):
        import types
        self.model = types.SimpleNamespace(diffusion_model=unet, conditioning_key="crossattn")
        self.schedule = schedule or LDMSchedule(device=unet.device_)
        self.num_timesteps = self.schedule.num_timesteps
        self.training = True

    def train(self, mode=True):
        self.training = bool(mode)
        self.model.diffusion_model.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def q_sample(self, x_start, t, noise=None):
        return self.schedule.q_sample(x_start, t, torch.randn_like(x_start) if noise is None else noise)

    def apply_model(self, x_noisy, t, cond):
        if isinstance(cond, dict):
            cond = torch.cat(cond["c_crossattn"], 1)
        return self.model.diffusion_model(x_noisy, t, context=cond)

    def p_losses(self, x_start, cond, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        out = self.apply_model(self.q_sample(x_start, t, noise), t, cond)
        loss_simple = ((out - noise) ** 2).mean([1, 2, 3])
        prefix = "train" if self.training else "val"
        loss = loss_simple.mean()                       # / exp(logvar) + logvar with logvar = 0; elbo weight 0
        return loss, {f"{prefix}/loss_simple": loss_simple.mean().detach(), f"{prefix}/loss": loss.detach()}

    def _outside(self, *a, **k):
        raise NotImplementedError("the VAE / CLIP front-end of LatentDiffusion is outside the unlearning hot path: hand latents and prompt "
                                  "embeddings in (sfron.latents for cached VAE moments)")
    get_input = shared_step = encode_first_stage = decode_first_stage = get_learned_conditioning = _outside


class SDSFRon:
    def __init__(self, unet, schedule=None, lr=1e-5, forget_alpha=1.0, remain_alpha=1.0, train_method="full", mask=None,
                 mask_mode="as_written", process_group=None, use_graphs=False):
        from . import dp
        self.use_graphs, self._graphs, self._pool = bool(use_graphs), {}, (graphs.shared_pool() if use_graphs else None)
        if train_method not in ("full", "xattn"):
            raise ValueError("train_method must be 'full' or 'xattn' (nsfw_removal.py:66-77)")
        if mask_mode not in ("as_written", "intended"):
            raise ValueError("mask_mode must be 'as_written' or 'intended'")
        self.unet, self.s = unet, schedule or LDMSchedule(device=unet.device_)
        self.fa, self.ra, self.train_method = forget_alpha, remain_alpha, train_method
        self.pg, self._dp, self.world = process_group, dp, dp.world_size(process_group)
        p, g, w16, index = unet.flat_arena()
        # which coordinates the optimizer owns: Adam leaves a coordinate whose gradient is always zero where it is
        train = torch.zeros(p.numel(), dtype=torch.uint8, device=p.device)
        for name, (off, shape) in index.items():
            if train_method == "full" or "attn2" in name:
                n = 1
                for d in shape:
                    n *= d
                train[off:off + n] = 1
        self.train_mask = train
        self.forget_mask = train
        if mask is not None and mask_mode == "intended":
            fm = train.clone()
            for name, (off, shape) in index.items():
                m = mask.get(name, mask.get("model.diffusion_model." + name))
                if m is None:
                    raise KeyError(f"saliency mask has no entry for {name}")
                n = 1
                for d in shape:
                    n *= d
                fm[off:off + n] &= m.reshape(-1).to(device=p.device, dtype=torch.uint8)
            self.forget_mask = fm
        # contiguous arena ranges that hold the trainable tensors (offsets are multiples of 8): with "xattn" the sweep streams 5 % of the arena
        ranges = None
        if train_method != "full":
            spans = sorted((off, off + (int(np.prod(shape)) + 7) // 8 * 8) for name, (off, shape) in index.items() if "attn2" in name)
            ranges = []
            for lo, hi in spans:
                if ranges and lo <= ranges[-1][1]:
                    ranges[-1][1] = max(ranges[-1][1], hi)
                else:
                    ranges.append([lo, hi])
            ranges = [(lo, min(hi, p.numel())) for lo, hi in ranges]
        self.opt = sweep.FlatAdam(p, g, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=True, mask=self.train_mask, w_bf16=w16,
                                  ranges=ranges)
        unet.auto_prep = False                      # this loop tells the model when its weights changed
        unet.wgrad_filter = (lambda n: "attn2" in n) if train_method == "xattn" else None

    def _d_loss(self, out, target, scale):
        """d(scale * mean((out - target)^2)) / d out through the HIP loss kernel (this rank's share of the global batch mean)."""
        n, chw = out.shape[0], out[0].numel()
        coef = torch.full((n,), 2.0 * scale / (n * chw * self.world), dtype=torch.float32, device=out.device)
        d = torch.empty_like(out)
        tg, oc = target.contiguous(), out.contiguous()
        check(_lib.lib().sfron_ddpm_loss_bwd(ptr(tg), ptr(oc), ptr(coef), n, chw, ptr(d), stream_ptr()), "loss_bwd")
        return d

    def _mse(self, a, b):
        n, chw = a.shape[0], a[0].numel()
        per = torch.empty(n, dtype=torch.float32, device=a.device)
        bc, ac = b.contiguous(), a.detach().contiguous()
        check(_lib.lib().sfron_ddpm_sample_loss(ptr(bc), ptr(ac), n, chw, ptr(per), stream_ptr()), "sample_loss")
        return per.sum() / (n * chw)

    # the two stages: forward pass(es), loss, backward pass -- stream-ordered device work only, so each replays as one HIP graph
    def _forget_pass(self, x_f, x_p, c_f, c_p, t, noise):
        u, s = self.unet, self.s
        f_noisy, p_noisy = s.q_sample(x_f, t, noise), s.q_sample(x_p, t, noise)          # the SAME t and noise (:134-141)
        p_out, _ = u._run(p_noisy, t, c_p, need_grad=False)                               # stop-gradient branch (:145)
        f_out, bwd = u._run(f_noisy, t, c_f, need_grad=True)
        loss = self._mse(f_out, p_out)
        bwd(self._d_loss(f_out, p_out, self.fa))
        return loss

    def _remain_pass(self, x, c, t, noise):
        r_out, bwd = self.unet._run(self.s.q_sample(x, t, noise), t, c, need_grad=True)
        loss = self._mse(r_out, noise)
        bwd(self._d_loss(r_out, noise, self.ra))
        return loss

    def _stage(self, name, fn, **inputs):
        if not self.use_graphs:
            return fn(**inputs)
        if name not in self._graphs:
            self._graphs[name] = graphs.StageGraph(fn, warmup=1, pool=self._pool)
        return self._graphs[name](**inputs)

    def _exchange(self):
        """SUM all-reduce of this stage's gradients: the whole arena ("full"), or only the ranges the optimizer owns ("xattn" --
        the frozen layers' weight gradients are not even formed, their arena entries are stale)."""
        if self.opt.ranges is not None:
            self._dp.allreduce_ranges_(self.opt.g, self.opt.ranges, 64 << 20, self.pg)
        else:
            self._dp.allreduce_flat_(self.opt.g, 64 << 20, self.pg)

    def step(self, forget, remain):
        """forget: dict(x_f, x_p, c_f, c_p, t, noise); remain: dict(x, c, t, noise) -- device tensors, this rank's shard."""
        u = self.unet
        u.train()
        ori_forget = self._stage("forget", self._forget_pass, **{k: forget[k] for k in ("x_f", "x_p", "c_f", "c_p", "t", "noise")})
        if self.world > 1:
            self._exchange()
        self.opt.mask = self.forget_mask
        self.opt.step(max_norm=None, use_mask=True)                       # nsfw_removal.py:162 (no clipping)
        u.weights_updated(convs=self.train_method == "full")
        ori_remain = self._stage("remain", self._remain_pass, **{k: remain[k] for k in ("x", "c", "t", "noise")})
        if self.world > 1:
            self._exchange()
        self.opt.mask = self.train_mask
        self.opt.step(max_norm=None, use_mask=True)                       # :170
        u.weights_updated(convs=self.train_method == "full")
        return {"forget_loss": ori_forget, "remain_loss": ori_remain}
