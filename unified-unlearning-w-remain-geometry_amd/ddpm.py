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
    """(1 - b).cumprod(0) in fp32 (functions/losses.py:25, recomputed there on every call).  A schedule that a runner has pinned with
    ``pin_alphas_cumprod`` answers from that copy while the tensor is unchanged: the one-workgroup sequential product costs 70 us per call,
    twice per SFR-on step."""
    memo = getattr(b, "_sfron_abar", None)
    if memo is not None and memo[0] == b._version:
        return memo[1]
    out = torch.empty_like(b)
    check(_lib.lib().sfron_ddpm_alphas_cumprod(ptr(b), b.numel(), ptr(out), stream_ptr()), "ddpm_alphas_cumprod")
    return out


def pin_alphas_cumprod(b):
    """Compute alphas_cumprod(b) once, eagerly (outside any stage graph: the copy must outlive every capture), and attach it to ``b``."""
    b._sfron_abar = None
    b._sfron_abar = (b._version, alphas_cumprod(b))
    return b


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


def compute_alpha(beta, t):
    """DDPM/functions/denoising.py:4-7 (host-side, tiny): alpha_bar of timestep t with alpha_bar(-1) = 1."""
    beta = torch.cat([torch.zeros(1, device=beta.device), beta], dim=0)
    return alphas_cumprod(beta).index_select(0, t + 1)


def generalized_steps_conditional(x, c, seq, model, b, cond_scale=3.0, step_noise=None, **kwargs):
    """DDPM/functions/denoising.py:72-95: the classifier-free-guided generalized (DDIM, ``eta``) sampler the runner's snapshot
    path uses.  Returns (xs, x0_preds) like the reference (tensors stay on the device).  The denoiser is the caller's
    ``model(x, t, c, cond_scale=..., mode="test")``; the update runs in sfron_ddim_step.  ``step_noise[k]`` optionally fixes
    the k-th ``randn_like`` draw."""
    eta = float(kwargs.get("eta", 0))
    abar = torch.cat([torch.ones(1, device=b.device), alphas_cumprod(b)]).cpu()          # index t+1; alpha_bar(-1) = 1
    with torch.no_grad():
        n = x.size(0)
        seq = list(seq)
        seq_next = [-1] + seq[:-1]
        xs, x0_preds = [x], []
        for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
            t = (torch.ones(n) * i).to(x.device)
            at, at_next = abar[i + 1], abar[j + 1]                                         # fp32 0-dim tensors: torch's scalar math
            et = model(xs[-1], t, c, cond_scale=cond_scale, mode="test").contiguous().float()
            c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            c2 = ((1 - at_next) - c1 ** 2).sqrt()
            noise = None
            if eta != 0.0:
                noise = (torch.randn_like(x) if step_noise is None else step_noise[k]).contiguous()
            xt = xs[-1].contiguous().float()
            x_next, x0 = torch.empty_like(xt), torch.empty_like(xt)
            check(_lib.lib().sfron_ddim_step(ptr(xt), ptr(et), ptr(noise), xt.numel(), float((1 - at).sqrt()), float(at.sqrt()),
                                             float(at_next.sqrt()), float(c1), float(c2), ptr(x_next), ptr(x0), stream_ptr()),
                  "ddim_step")
            x0_preds.append(x0)
            xs.append(x_next)
    return xs, x0_preds


# ----------------------------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes the trainable parameters of ANY autograd module in one flat fp32 arena (parameters and their .grad
    become views), so that the mask -> clip -> Adam -> EMA sweep of csrc/sweep.hip runs over it in two launches per
    stage instead of the reference's per-tensor loops (DDPM/runners/diffusion.py:1126-1138,1169-1180)."""

    def __init__(self, model):
        self.names, self.params = [], []
        for n, p in model.named_parameters():
            if p.requires_grad:
                self.names.append(n)
                self.params.append(p)
        self.w_bf16 = None
        if hasattr(model, "flat_arena"):
            # sfron.unet.Conditional_Model: the parameters already live in one arena (with a gradient arena and a bf16 weight
            # shadow at the same offsets) -- adopt it
            self.p, self.g, self.w_bf16, index = model.flat_arena()
            self.offsets = [index[n][0] for n in self.names]
            self.n = self.p.numel()
            return
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8                       # 16-byte aligned tensors
        self.n = off
        self.p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.g = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            k = p.numel()
            self.p[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + k].view(p.shape)
            p.grad = self.g[o:o + k].view(p.shape)                # autograd accumulates in place into the arena

    def mask_arena(self, mask):
        """name -> bool tensor / python int (DDPM/generate_fisher_mask.py:39-46; keys with or without 'module.')."""
        arena = torch.zeros(self.n, dtype=torch.uint8, device=self.p.device)     # arena padding stays masked out
        for n, p, o in zip(self.names, self.params, self.offsets):
            m = mask.get(n, mask.get("module." + n))
            if m is None:
                raise KeyError(f"saliency mask has no entry for {n}")
            if isinstance(m, int):
                arena[o:o + p.numel()] = 1 if m else 0
            else:
                arena[o:o + p.numel()] = m.reshape(-1).to(device=self.p.device, dtype=torch.uint8)
        return arena

    def named_views(self, flat):
        return {n: flat[o:o + p.numel()].view(p.shape) for n, p, o in zip(self.names, self.params, self.offsets)}


class DDPMSFRon:
    """The SFR-on iteration of DDPM/runners/diffusion.py:1075-1180 (method "ron"): cosine-decayed alpha, forget loss
    "ga" / "adaga" -> mask -> clip -> Adam, remain loss -> clip -> Adam, EMAHelper update.  The denoiser is the caller's
    autograd module (``model(x, t_float, c, mode="train", cond_drop_prob=...)``); loss, gradient hand-off and the whole
    parameter sweep run in the HIP library.  ``step(forget, remain)`` takes dicts with x0 (already data_transform-ed),
    c, t, e."""

    def __init__(self, model, betas=None, lr=1e-4, forget_alpha=10.0, remain_alpha=1.0, grad_clip=1.0, ema_rate=None, mask=None,
                 unlearn_loss="adaga", lambd=0.5, n_iters=50, decay_forget_alpha=True, cond_drop_prob=0.1, process_group=None,
                 label_to_forget=0, n_classes=10, use_graphs=False, method="ron"):
        from . import dp, sweep
        self.use_graphs, self._graphs, self._pool = bool(use_graphs), {}, None
        # method "joint" (runners/diffusion.py:1160-1167): ONE clipped Adam step per iteration on remain_alpha * remain_loss +
        # alpha * forget_loss, both back-propagated from the same weights; the reference's mask loop there runs on the stale gradients
        # of the previous step, just before zero_grad(), so the mask does not touch the update
        if method not in ("ron", "joint"):
            raise ValueError(f"unsupported method {method!r} (DDPM/runners/diffusion.py:1121,1154-1167 defines 'ron' and 'joint')")
        self.method, self._g_forget = method, None
        if unlearn_loss not in ("ga", "adaga", "rl"):
            raise ValueError(f"unsupported unlearn_loss {unlearn_loss!r} (DDPM/runners/diffusion.py:1095-1120 defines ga, rl, adaga)")
        self.label_to_forget, self.n_classes = label_to_forget, n_classes
        self.model, self.flat = model, FlatParams(model)
        dev = self.flat.p.device
        self.b = pin_alphas_cumprod(betas if betas is not None else get_beta_schedule(device=dev))
        self.forget_alpha, self.remain_alpha, self.grad_clip = forget_alpha, remain_alpha, grad_clip
        self.unlearn_loss, self.lambd, self.n_iters, self.decay = unlearn_loss, lambd, n_iters, decay_forget_alpha
        self.cond_drop_prob, self.pg, self._dp = cond_drop_prob, process_group, dp
        self.world = dp.world_size(process_group)
        marena = self.flat.mask_arena(mask) if mask is not None else None
        # DDPM/functions/__init__.py:9-18 with cifar10_sfron.yml:48-56: Adam, wd 0, betas (0.9, 0.999), eps 1e-8
        self.opt = sweep.FlatAdam(self.flat.p, self.flat.g, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=True, mask=marena,
                                  w_bf16=self.flat.w_bf16)
        self.mu = ema_rate
        self.shadow = self.flat.p.clone() if ema_rate is not None else None       # EMAHelper.register (models/ema.py:11-15)

    def _loss(self, batch, kind):
        fn = loss_registry_conditional["simple"]
        extra = {}
        if batch.get("keep_mask") is not None:      # explicit classifier-free keep mask (tests; the reference draws it inside the model)
            extra["keep_mask"] = batch["keep_mask"]
        wrapped = lambda x, tf, c, cond_drop_prob=0.1, mode="train": self.model(x, tf, c, cond_drop_prob=self.cond_drop_prob, mode=mode, **extra)
        if kind == "rl":
            # runners/diffusion.py:1101-1113: criteria(pseudo, output) = mean over ALL elements of (pseudo - output)^2, the pseudo
            # branch detached; both branches see the same x_t.  The per-sample sums and their gradient are the HIP loss kernels.
            x = q_sample(batch["x0"], batch["e"], batch["t"], alphas_cumprod(self.b))
            tf = batch["t"].float()
            output = wrapped(x, tf, batch["c"])
            pseudo_c = torch.full_like(batch["c"], (self.label_to_forget + 1) % self.n_classes)
            with torch.no_grad():
                pseudo = wrapped(x, tf, pseudo_c)
            per = _SampleLoss.apply(output, pseudo)
            return _Reduce.apply(per, 0, 0.0, self.pg if self.world > 1 else None) / float(output[0].numel())
        args = (wrapped, batch["x0"], batch["t"], batch["c"], batch["e"], self.b)
        pg = self.pg if self.world > 1 else None
        if kind == "adaga":
            return adaptive_loss(fn, *args, lambd=self.lambd, dp_group=pg)
        return fn(*args, dp_group=pg)

    def _weights_updated(self):
        if hasattr(self.model, "weights_updated"):                # native U-Net: re-lay the bf16 conv operands once per Adam step
            self.model.auto_prep = False
            self.model.weights_updated()

    def _backward(self, loss):
        self.flat.g.zero_()                                       # optimizer.zero_grad()
        loss.backward()

    # the two stages (forward pass(es), loss, backward pass): stream-ordered device work only, so each can replay as one HIP graph
    # (sfron.graphs); the decayed forget alpha arrives as a device scalar
    def _forget_pass(self, alpha, **batch):
        if self.unlearn_loss == "rl":
            ori_forget = self._loss(batch, "rl")
        else:
            ori_forget = -self._loss(batch, "adaga" if self.unlearn_loss == "adaga" else "simple")
        self._backward(alpha * ori_forget)
        return ori_forget.detach()

    def _remain_pass(self, **batch):
        ori_remain = self._loss(batch, "simple")
        self._backward(self.remain_alpha * ori_remain)
        return ori_remain.detach()

    def _stage(self, name, fn, **inputs):
        # the adaga normaliser of a data-parallel run is a collective in the middle of the stage: those runs stay eager
        if not self.use_graphs or (self.world > 1 and self.unlearn_loss == "adaga" and name == "forget"):
            return fn(**inputs)
        if name not in self._graphs:
            from . import graphs
            if self._pool is None:
                self._pool = graphs.shared_pool()
            self._graphs[name] = graphs.StageGraph(fn, warmup=1, pool=self._pool)
        return self._graphs[name](**inputs)

    def step(self, step_idx, forget, remain):
        alpha = cosine_lr_scheduler(self.forget_alpha, step_idx, self.n_iters) if self.decay else self.forget_alpha
        self.model.train()
        keys = ("x0", "c", "t", "e", "keep_mask")
        a_dev = torch.full((), float(alpha), dtype=torch.float32, device=self.flat.p.device) if self.use_graphs else alpha
        ori_forget = self._stage("forget", self._forget_pass, alpha=a_dev, **{k: forget.get(k) for k in keys})
        if self.method == "joint":
            # the forget stage's gradient is parked (the remain stage zeroes the arena) and handed to the sweep as its second gradient
            # arena: the norm pre-pass and the Adam kernel add the two (csrc/sweep.hip g2)
            if self._g_forget is None:
                self._g_forget = torch.empty_like(self.flat.g)
            self._g_forget.copy_(self.flat.g)
        else:
            if self.world > 1:
                self._dp.allreduce_flat_(self.flat.g, 64 << 20, self.pg)
            self.opt.step(max_norm=self.grad_clip, use_mask=True)
            self._weights_updated()
        ori_remain = self._stage("remain", self._remain_pass, **{k: remain.get(k) for k in keys})
        if self.method == "joint":
            if self.world > 1:
                self.flat.g.add_(self._g_forget)
            else:
                self.opt.g2 = self._g_forget
        if self.world > 1:
            self._dp.allreduce_flat_(self.flat.g, 64 << 20, self.pg)
        self.opt.step(max_norm=self.grad_clip, use_mask=False, ema=self.shadow, ema_decay=self.mu if self.mu is not None else 0.0, ema_mode=2)
        self.opt.g2 = None
        self._weights_updated()
        return {"forget_loss": ori_forget, "remain_loss": ori_remain, "alpha": alpha}

    def ema_state_dict(self):
        return {n: v.clone() for n, v in self.flat.named_views(self.shadow).items()}

    def checkpoint(self, step):
        """[model.state_dict(), optimizer.state_dict(), step, ema_helper.state_dict()] as DDPM/runners/diffusion.py:160-171 saves
        it; the optimizer entry is in torch.optim.Adam's layout over ``model.parameters()``."""
        names = [n for n, _ in self.model.named_parameters()]
        m, v = self.flat.named_views(self.opt.m), self.flat.named_views(self.opt.v)
        state = {i: {"step": torch.tensor(float(self.opt.step_count)), "exp_avg": m[n].clone(), "exp_avg_sq": v[n].clone()}
                 for i, n in enumerate(names) if n in m and self.opt.step_count > 0}
        group = {"lr": self.opt.lr, "betas": tuple(self.opt.betas), "eps": self.opt.eps, "weight_decay": 0.0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(names)))}
        states = [{k: t.clone() for k, t in self.model.state_dict().items()}, {"state": state, "param_groups": [group]}, step]
        if self.shadow is not None:
            states.append(self.ema_state_dict())
        return states
