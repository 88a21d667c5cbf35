"""Fisher-diagonal accumulation and saliency-mask generation on the device (SURVEY.md section 8f "next #1").

Mirrors /root/reference/DiT/generate_fisher.py:216-291 (two loops: forget, then remain, each
``F[name] += grad**2 / n_iters`` of ``training_losses(...)["loss"].mean()``) and DiT/generate_mask.py:27-46
(``mask = (F_f + 1e-15) / (F_r + 1e-15) >= th``), with the reference's on-disk dict formats
(``module.``-prefixed keys, python int 0 for parameters that never receive a gradient).
The reference squares gradients on the CPU per tensor per step; here the accumulate is one sweep kernel over the
flat gradient arena.  Data parallel: per-rank batches are distinct Fisher samples, so ranks accumulate
g**2 locally and the caller all-reduce-SUMs F at the end (averaging gradients first would NOT be equivalent).
"""
import torch

from . import sweep


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
