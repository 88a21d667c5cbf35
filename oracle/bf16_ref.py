"""Oracle (test infrastructure): the DiT oracle with the rounding points of bf16 GEMM OPERANDS, everything else fp32.

BASELINE.json's north_star prescribes bf16 MFMA for the attention and MLP GEMMs of the denoiser; ANY implementation that
feeds the matrix core bf16 operands rounds (a) the weights of every Linear / the patch-embedding convolution (the bf16
weight shadow) and (b) the activation operand of every product.  This module restates that -- and nothing else -- on top of
oracle.dit_ref (DiT/models.py:101-248) with torch hooks: arithmetic, accumulation, LayerNorm, softmax, GELU, the residual
stream, the loss (gaussian_diffusion.py:715-787) and the optimizer (forget.py:199,289-322) stay the fp32 oracle's.

What it is for: tests/test_gpu_baseline_shapes.py::test_xl2_fifty_sfron_iterations_vs_oracles and
tests/debug/xl2_gap_bisect.py separate "the HIP path differs from the reference" into (1) what the prescribed operand type
costs (this oracle against the fp32 oracle: 1.3e-4 of held-out eps-MSE at random-init DiT-XL/2 before any step, almost all
of it the WEIGHT rounding) and (2) what the implementation adds on top (the HIP path against this oracle).

Rounding classes (names used by the bisect script):
  W    weights of every matrix-shaped trainable tensor (Linear weights, patch-embedding kernel, label table)
  A    inputs of the four block Linears (qkv, proj, fc1, fc2)
  E    inputs of the products outside the blocks: patch embedding, timestep MLP, every adaLN Linear, final Linear
  QKV  the qkv Linear's output (what an attention kernel reads)
  P    softmax probabilities in front of P.V
"""
import torch

from . import dit_ref


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class OperandRounding:
    """Context manager: installs the requested rounding classes on an oracle DiT (any device), restores everything on exit."""

    def __init__(self, model, kinds=("W", "A", "E")):
        self.m, self.kinds, self.handles, self.saved = model, set(kinds), [], {}

    def __enter__(self):
        m, K = self.m, self.kinds
        if "W" in K:
            for n, p in m.named_parameters():
                if p.dim() >= 2 and p.requires_grad:
                    self.saved[n] = p.data.clone()
                    p.data.copy_(bf(p.data))
        pre = lambda mod, a: (bf(a[0]),)
        for blk in m.blocks:
            if "A" in K:
                for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                    self.handles.append(lin.register_forward_pre_hook(pre))
            if "QKV" in K:
                self.handles.append(blk.attn.qkv.register_forward_hook(lambda mod, a, o: bf(o)))
        if "E" in K:
            mods = [m.x_embedder.proj, m.t_embedder.mlp[0], m.t_embedder.mlp[2], m.final_layer.linear, m.final_layer.adaLN_modulation[1]]
            mods += [blk.adaLN_modulation[1] for blk in m.blocks]
            for mod in mods:
                self.handles.append(mod.register_forward_pre_hook(pre))
        if "P" in K:
            self._attn_fwd = dit_ref.Attention.forward

            def fwd(self_, x):
                B, N, C = x.shape
                qkv = self_.qkv(x).reshape(B, N, 3, self_.num_heads, self_.head_dim).permute(2, 0, 3, 1, 4)
                q, k, v = qkv.unbind(0)
                attn = ((q * self_.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
                x = bf(attn) @ v
                return self_.proj(x.transpose(1, 2).reshape(B, N, C))
            dit_ref.Attention.forward = fwd
        return self

    def __exit__(self, *a):
        for h in self.handles:
            h.remove()
        self.handles = []
        for n, p in self.m.named_parameters():
            if n in self.saved:
                p.data.copy_(self.saved[n])
        self.saved = {}
        if "P" in self.kinds:
            dit_ref.Attention.forward = self._attn_fwd


def sfron_step_bf16_operands(orc, forget, remain):
    """One oracle.sfron_ref.DiTSfronOracle.step (DiT/forget.py:256-322) whose forward / backward passes see bf16-rounded weights and
    bf16-rounded product inputs (classes W + A + E) while the optimizer keeps updating the fp32 MASTERS: the optimizer's step() is
    wrapped to restore the masters first and to round the updated ones afterwards -- the host-side picture of an fp32 arena with a
    bf16 shadow that every optimizer sweep rewrites."""
    m = orc.model
    masters = {}

    def round_in():
        for n, p in m.named_parameters():
            if p.dim() >= 2 and p.requires_grad:
                masters[n] = p.data.clone()
                p.data.copy_(bf(p.data))

    def restore():
        for n, p in m.named_parameters():
            if n in masters:
                p.data.copy_(masters[n])
        masters.clear()
    opt_step = orc.opt.step

    def wrapped(*a, **k):
        restore()
        r = opt_step(*a, **k)
        round_in()
        return r
    orc.opt.step = wrapped
    round_in()
    try:
        with OperandRounding(m, ("A", "E")):
            return orc.step(forget, remain)
    finally:
        orc.opt.step = opt_step
        restore()
