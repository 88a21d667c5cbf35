"""Oracle (test infrastructure): fake-quantised restatement of BASELINE config 5 -- "DiT-XL/2 fp8 weights + bf16 activations on the
CDNA4 fp8 MFMA".  The reference repository has no fp8 path: this file restates what csrc/fp8.hip computes so that the HIP path can be
checked against plain PyTorch, and the tolerance of config 5 against config 3 (the bf16 path / the fp32 reference) can be stated.

  e4m3        OCP e4m3fn (torch.float8_e4m3fn), round to nearest even, saturating at +-448
  weights     one power-of-two scale per tensor: 2^floor(log2(224 / amax))
  activations static power-of-two scales per producer: LayerNorm+modulate output 8, attention output 32, gelu(fc1) 16
  where       the four token Linears of every DiTBlock (attn.qkv, attn.proj, mlp.fc1, mlp.fc2 -- DiT/models.py:108-121), forward only;
              the backward pass sees the un-quantised operands (straight-through estimator), like the HIP path whose backward GEMMs
              keep their bf16 operands.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

E4M3_MAX = 448.0
ACT_SCALES = dict(xmod=8.0, o=32.0, h=16.0)


def q_e4m3(x, scale):
    """fake quantisation: e4m3(x * scale) / scale (the value the fp8 MFMA multiplies, after the epilogue's de-scaling)"""
    return (x * scale).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32) / scale


def e4m3_bytes(x, scale):
    """the e4m3 codes themselves (uint8), for bit-exact comparison with the HIP quantisers"""
    return (x * scale).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).view(torch.uint8)


def weight_scale(w):
    amax = float(w.detach().abs().max())
    if amax <= 0.0 or not math.isfinite(amax):
        return 1.0
    return 2.0 ** math.floor(math.log2(224.0 / amax))


def ste(x, xq):
    """value of xq, gradient of x"""
    return x + (xq - x).detach()


class FakeQuantLinear(nn.Linear):
    """nn.Linear whose forward multiplies e4m3-rounded input and weight (fp32 accumulation), straight-through backward."""

    act_scale = 1.0
    w_scale_override = None        # delayed scaling tests set the scale the HIP path used

    def forward(self, x):
        ws = self.w_scale_override if self.w_scale_override is not None else weight_scale(self.weight)
        return F.linear(ste(x, q_e4m3(x, self.act_scale)), ste(self.weight, q_e4m3(self.weight, ws)), self.bias)


def apply_fake_quant(dit_model, act_scales=None):
    """Swap the four token Linears of every block of an oracle.dit_ref.DiT for FakeQuantLinear (weights shared, in place)."""
    a = dict(ACT_SCALES, **(act_scales or {}))
    for blk in dit_model.blocks:
        for mod, name, s in ((blk.attn, "qkv", a["xmod"]), (blk.attn, "proj", a["o"]), (blk.mlp, "fc1", a["xmod"]), (blk.mlp, "fc2", a["h"])):
            lin = getattr(mod, name)
            fq = FakeQuantLinear(lin.in_features, lin.out_features, bias=lin.bias is not None)
            fq.weight, fq.bias = lin.weight, lin.bias
            fq.act_scale = s
            setattr(mod, name, fq)
    return dit_model
