"""Fisher-diagonal accumulation and saliency-mask generation on the device (SURVEY.md section 8f "next #1").

Mirrors /root/reference/DiT/generate_fisher.py:216-291 (two loops: forget, then remain, each
``F[name] += grad**2 / n_iters`` of ``training_losses(...)["loss"].mean()``) and DiT/generate_mask.py:27-46
(``mask = (F_f + 1e-15) / (F_r + 1e-15) >= th``), with the reference's on-disk dict formats
(``module.``-prefixed keys, python int 0 for parameters that never receive a gradient).
The reference squares gradients on the CPU per tensor per step; here the accumulate is one sweep kernel over the
flat gradient arena.  Data parallel: per-rank batches are distinct Fisher samples, so ranks accumulate
g**2 locally and the caller all-reduce-SUMs F at the end (averaging gradients first would NOT be equivalent).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib, sweep
from ._lib import check, ptr, stream_ptr


class FisherAccumulator:
    def __init__(self, model, diffusion, n_iters):
        self.model, self.diffusion, self.n_iters = model, diffusion, n_iters
        eng = model.engine
        self.fisher = torch.zeros(eng.n_trainable, dtype=torch.float32, device=eng.device)

    def accumulate(self, batch):
        """One iteration: grads of loss.mean() on this batch, then F += g^2 / n_iters (generate_fisher.py:225-239)."""
        eng, diff = self.model.engine, self.diffusion
        n = batch["x0"].shape[0]
        self.model.set_batch_size(n)
        eng = self.model.engine
        x_t = diff.q_sample(batch["x0"], batch["t"], batch["noise"])
        out = eng.forward(x_t, batch["t"], batch["y"], batch.get("drop"))
        _, _, d_out = diff.loss_fwd_bwd(out, batch["x0"], batch["t"], batch["noise"], 1.0 / n)
        eng.backward(d_out, batch["y"], batch.get("drop"))
        sweep.fisher_accum(self.fisher, eng.grads[:eng.n_trainable], self.n_iters)

    def all_reduce(self, group=None):
        """Data parallel: every rank has accumulated g^2 / n_iters of ITS batches with n_iters = the GLOBAL number of batches, so the
        SUM over ranks is the reference's estimator over all batches (all-reducing gradients before
        squaring would not be -- per-rank batches are distinct Fisher samples).  Call once, after the last accumulate()."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.fisher, op=dist.ReduceOp.SUM, group=group)
        return self

    def state_dict(self, prefix="module."):
        """name -> fp32 CPU tensor, python int 0 for never-grad params (generate_fisher.py:218-239 format)."""
        eng = self.model.engine
        out = {}
        for name, (off, shape, trainable) in eng.index.items():
            if not trainable:
                out[prefix + name] = 0
            else:
                out[prefix + name] = eng.view(self.fisher, name).detach().cpu().clone()
        return out


def masks_from_fisher(forget_fisher, remain_fisher, th, device="cuda"):
    """Reference file format in, reference file format out (generate_mask.py:27-46): name -> bool tensor / int 0."""
    out = {}
    for name, ff in forget_fisher.items():
        rf = remain_fisher[name]
        if isinstance(ff, int) or isinstance(rf, int):
            out[name] = 0                       # the reference's try/except leaves the initial 0 (generate_mask.py:32,42-43)
            continue
        out[name] = sweep.mask_from_fisher(ff.to(device), rf.to(device), th).cpu()
    return out


class DDPMFisherAccumulator:
    """Fisher diagonal of the DDPM runner (DDPM/runners/diffusion.py:1210-1364) on the native U-Net (sfron.unet.Conditional_Model):
    per batch, x_t = q_sample(x0, t, e); output = model(x_t, t, c, cond_scale=s, mode="test") = (1 + s) f(cond) - s f(null) WITH
    gradients through both branches (:1260-1262); loss = sum_chw (e - output)^2 averaged over the batch (:1265);
    clip_grad_norm_(grad_clip) (:1271-1276) and only then F += grad^2 / len(loader) (:1278-1282).  The model is in eval mode
    (:1227: no dropout).  Two backward passes (one per branch, with d_out scaled by (1 + s) and -s) fill two gradient arenas;
    the clip norm and the squared, clipped sum are sweep kernels over both.  n_batches = len(loader) over ALL ranks; finish a
    data-parallel run with all_reduce()."""

    def __init__(self, model, betas, n_batches, cond_scale=2.0, grad_clip=1.0):
        from . import ddpm
        self._ddpm = ddpm
        self.model, self.b, self.n, self.s, self.clip = model, betas, n_batches, float(cond_scale), grad_clip
        p, g, _, _ = model.flat_arena()
        self.fisher = torch.zeros_like(p)
        self.g_first = torch.zeros_like(p)
        self._partials = torch.empty(_lib.lib().sfron_sweep_partials_len(), dtype=torch.float64, device=p.device)
        self._stats = torch.zeros(4, dtype=torch.float32, device=p.device)

    def accumulate(self, batch):
        """batch: x0 (data_transform-ed), c, t (int64), e."""
        L, m, ddpm = _lib.lib(), self.model, self._ddpm
        was_training = m.training
        m.eval()
        dev = m.device_
        x0, e, t, c = batch["x0"], batch["e"].contiguous(), batch["t"], batch["c"]
        B = x0.shape[0]
        x_t = ddpm.q_sample(x0, e, t, ddpm.alphas_cumprod(self.b))
        tf = t.float()
        out_c, bwd_c = m._run(x_t, tf, c, torch.ones(B, dtype=torch.uint8, device=dev), None, need_grad=True)
        out_n, bwd_n = m._run(x_t, tf, c, torch.zeros(B, dtype=torch.uint8, device=dev), None, need_grad=True)
        out = torch.empty_like(out_c)
        check(L.sfron_axpby(ptr(out_c), ptr(out_n), 1.0 + self.s, -self.s, out.numel(), ptr(out), stream_ptr()), "axpby")
        # d loss / d output_i = 2 (output_i - e_i) / B ; the branches receive it scaled by (1 + s) and -s
        chw = out[0].numel()
        per = torch.empty(B, dtype=torch.float32, device=dev)
        check(L.sfron_ddpm_sample_loss(ptr(e), ptr(out), B, chw, ptr(per), stream_ptr()), "ddpm_sample_loss")
        d_out = torch.empty_like(out)
        for scale, bwd, keep in (((1.0 + self.s), bwd_c, True), (-self.s, bwd_n, False)):
            coef = torch.full((B,), 2.0 * scale / B, dtype=torch.float32, device=dev)
            check(L.sfron_ddpm_loss_bwd(ptr(e), ptr(out), ptr(coef), B, chw, ptr(d_out), stream_ptr()), "ddpm_loss_bwd")
            bwd(d_out)
            if keep:
                self.g_first.copy_(m.grads)
        n = m.grads.numel()
        stats = None
        if self.clip is not None:
            nblk = ctypes.c_int(0)
            check(L.sfron_sumsq_masked(ptr(m.grads), ptr(self.g_first), None, n, ptr(self._partials), ctypes.byref(nblk), stream_ptr()), "sumsq")
            check(L.sfron_clip_coef(ptr(self._partials), nblk.value, float(self.clip), ptr(self._stats), stream_ptr()), "clip_coef")
            stats = self._stats
        check(L.sfron_fisher_accum_clipped(ptr(self.fisher), ptr(m.grads), ptr(self.g_first), ptr(stats), n, float(self.n), stream_ptr()),
              "fisher_accum_clipped")
        m.train(was_training)
        return per.mean()

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.fisher, op=dist.ReduceOp.SUM, group=group)
        return self

    def state_dict(self, prefix="module."):
        """name -> fp32 CPU tensor (the runner saves the DataParallel-prefixed names, :1284)."""
        return {prefix + n: self.model.view(self.fisher, n).detach().cpu().clone() for n in self.model.index}


class SDFisherAccumulator:
    """Fisher diagonal of the concept-erasure script (SD/train-scripts/generate_fisher.py:36-79 forget loop, :87-128 remain loop) on
    the native UNet (sfron.sd_unet.UNetModel): per batch one shared t and noise, x_t = q_sample(latents, t, noise) (:57-62);
    preds = (1 + c_guidance) eps(x_t, c_prompt) - c_guidance eps(x_t, c_null) with gradients through BOTH branches (:64-67);
    loss = -MSELoss(noise, preds) (:70, mean over every element); F[name] += grad^2 / len(loader) (:73-77), no clipping; the model
    is in eval mode (:25).  Two backward passes (d_preds scaled by (1 + c) and -c) fill two gradient arenas, one sweep squares
    their sum.  Latents and the two prompt embeddings arrive resident (the VAE / CLIP front-end of get_input is outside the path).
    n_batches = len(loader) over ALL ranks; finish a data-parallel run with all_reduce()."""

    def __init__(self, unet, schedule, n_batches, c_guidance=7.5):
        self.unet, self.s, self.n, self.c = unet, schedule, n_batches, float(c_guidance)
        p, g, _, _ = unet.flat_arena()
        self.fisher = torch.zeros_like(p)
        self.g_first = torch.zeros_like(p)

    def accumulate(self, batch):
        """batch: x (latents [B,4,h,w]), c (prompt embedding [B,L,ctx]), c_null (embedding of ""), t (int64), noise."""
        L, u = _lib.lib(), self.unet
        was_training = u.training
        u.eval()
        x, noise, t = batch["x"], batch["noise"].contiguous(), batch["t"]
        B, chw = x.shape[0], x[0].numel()
        x_t = self.s.q_sample(x, t, noise)
        keep, u.wgrad_filter = u.wgrad_filter, None           # every parameter's gradient (an xattn SDSFRon may have narrowed it)
        try:
            out_c, bwd_c = u._run(x_t, t, batch["c"], need_grad=True)
            out_n, bwd_n = u._run(x_t, t, batch["c_null"], need_grad=True)
        finally:
            u.wgrad_filter = keep
        preds = torch.empty_like(out_c)
        check(L.sfron_axpby(ptr(out_c), ptr(out_n), 1.0 + self.c, -self.c, preds.numel(), ptr(preds), stream_ptr()), "axpby")
        per = torch.empty(B, dtype=torch.float32, device=x.device)
        check(L.sfron_ddpm_sample_loss(ptr(noise), ptr(preds), B, chw, ptr(per), stream_ptr()), "sample_loss")
        # d(-mean((noise - preds)^2)) / d preds = -2 (preds - noise) / (B chw); the branches receive it scaled by (1 + c) and -c
        d_out = torch.empty_like(preds)
        for scale, bwd, keep in (((1.0 + self.c), bwd_c, True), (-self.c, bwd_n, False)):
            coef = torch.full((B,), -2.0 * scale / (B * chw), dtype=torch.float32, device=x.device)
            check(L.sfron_ddpm_loss_bwd(ptr(noise), ptr(preds), ptr(coef), B, chw, ptr(d_out), stream_ptr()), "loss_bwd")
            bwd(d_out)
            if keep:
                self.g_first.copy_(u.grads)
        check(L.sfron_fisher_accum_clipped(ptr(self.fisher), ptr(u.grads), ptr(self.g_first), None, u.grads.numel(), float(self.n),
                                           stream_ptr()), "fisher_accum")
        u.train(was_training)
        return -per.sum() / (B * chw)

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.fisher, op=dist.ReduceOp.SUM, group=group)
        return self

    def state_dict(self):
        """name -> fp32 CPU tensor keyed by the UNet-relative parameter names (generate_fisher.py:31-32,79: nude_forget.pt / nude_remain.pt)."""
        return {n: self.unet.view(self.fisher, n).detach().cpu().clone() for n in self.unet.index}
