"""DiT module with the reference's API surface over the HIP engine.

Mirrors /root/reference/DiT/models.py: ``DiT_models[name](input_size=..., num_classes=...)``,
``forward(x, t, y)``, ``forward_with_cfg``, identical ``state_dict()`` keys (so ``find_model`` checkpoints,
DiT/download.py:18-29, load with ``load_state_dict``), ``train()/eval()`` controlling label dropout
(models.py:78-94).  Every nn.Parameter is a VIEW into the engine's flat arena.

Autograd: ``forward`` returns a tensor with a grad edge; its backward runs the HIP backward pass and
publishes ``param.grad`` as views of the engine's gradient arena (overwrite semantics -- call
``zero_grad()`` before each backward, as the reference does at DiT/forget.py:287,312).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr
from .engine import DitEngine


class _Holder(nn.Module):
    """Name-space node so parameters get the reference's dotted names."""


def _sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """Fixed 2-D sin-cos table (models.py:274-321; 'w goes first' meshgrid)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    return np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)


class _DiTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, y, drop):
        out = model.engine.forward(x.contiguous(), t.contiguous(), y.contiguous(), drop)
        ctx.model, ctx.y, ctx.drop = model, y.contiguous(), drop
        return out

    @staticmethod
    def backward(ctx, d_out):
        m = ctx.model
        m.engine.backward(d_out.contiguous(), ctx.y, ctx.drop)
        m.publish_grads()
        return None, None, None, None, None, None


class DiT(nn.Module):
    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, batch_size=32,
                 device="cuda"):
        super().__init__()
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.num_classes = num_classes
        self.class_dropout_prob = class_dropout_prob
        self._engine_args = dict(input_size=input_size, patch_size=patch_size, in_channels=in_channels,
                                 hidden_size=hidden_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                 num_classes=num_classes, learn_sigma=learn_sigma, device=device)
        self.engine = DitEngine(batch_size, **self._engine_args)
        self._anchor = torch.zeros((), device=self.engine.device, requires_grad=True)
        self._register_views()
        self.initialize_weights()

    # ------------------------------------------------------------------ parameters as arena views
    def _register_views(self):
        eng = self.engine
        for name, (off, shape, trainable) in eng.index.items():
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Holder())
                node = getattr(node, part)
            node.register_parameter(parts[-1], nn.Parameter(eng.view(eng.params, name), requires_grad=trainable))

    def publish_grads(self):
        eng = self.engine
        for name, p in self.named_parameters():
            if p.requires_grad:
                p.grad = eng.view(eng.grads, name)

    def initialize_weights(self):
        """Same distributions as models.py:182-216 (xavier on Linear, N(0,.02) embeddings, zero adaLN / final)."""
        eng = self.engine
        D = eng.cfg.hidden
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".bias"):
                    p.zero_()
                elif name in ("t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight", "y_embedder.embedding_table.weight"):
                    nn.init.normal_(p, std=0.02)
                elif "adaLN_modulation" in name or name.startswith("final_layer.linear"):
                    p.zero_()
                elif name.endswith(".weight"):
                    nn.init.xavier_uniform_(p.view(p.shape[0], -1))
            pe = get_2d_sincos_pos_embed(D, int(eng.tokens ** 0.5))
            self.pos_embed.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        eng.sync_bf16()

    def state_dict(self, *a, **kw):
        # the parameters are views into the arena: order this stream behind a block sweep a runner may have left in flight
        # (DiTSFRon.sweep_across_steps) before anybody copies them
        self.engine.drain_sweep()
        return super().state_dict(*a, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        self.engine.drain_sweep()
        # checkpoints written by the reference come from an nn.DataParallel wrapper (DiT/forget.py:193,347): "module." keys
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.engine.sync_bf16()
        return r

    def set_batch_size(self, batch_size):
        """Re-create the workspace for another per-GPU batch (parameters are kept)."""
        if batch_size == self.engine.cfg.batch:
            return
        old = self.engine
        old.drain_sweep()
        new = DitEngine(batch_size, share=old, grads=old.grads, **self._engine_args)     # same parameter AND gradient arenas
        new.probe, old.probe = old.probe, None
        new._share_fp8(old)
        old.close()
        self.engine = new

    # ------------------------------------------------------------------ forward
    def _draw_drop(self, n):
        # LabelEmbedder.token_drop (models.py:78-87): torch.rand(N) < p in train mode
        if self.training and self.class_dropout_prob > 0:
            return (torch.rand(n, device=self.engine.device) < self.class_dropout_prob).to(torch.uint8)
        return None

    def forward(self, x, t, y, force_drop_ids=None):
        if force_drop_ids is not None:
            drop = (force_drop_ids == 1).to(torch.uint8).contiguous()
        else:
            drop = self._draw_drop(x.shape[0])
        self.set_batch_size(x.shape[0])
        if torch.is_grad_enabled():
            return _DiTFn.apply(self._anchor, self, x, t, y, drop)
        return self.engine.forward(x.contiguous(), t.contiguous(), y.contiguous(), drop)

    def forward_with_cfg(self, x, t, y, cfg_scale):
        # models.py:250-266 (guidance on the first three channels, as the reference does); the mix runs in place on the
        # model output through sfron_cfg_combine
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0)
        with torch.no_grad():
            out = self.forward(combined, t, y).contiguous()
            check(_lib.lib().sfron_cfg_combine(ptr(out), out.shape[0], out.shape[1], out[0, 0].numel(), 3, float(cfg_scale),
                                               stream_ptr()), "cfg_combine")
        return out


def _cfg(depth, hidden_size, patch_size, num_heads):
    return dict(depth=depth, hidden_size=hidden_size, patch_size=patch_size, num_heads=num_heads)


_CONFIGS = {
    "DiT-XL/2": _cfg(28, 1152, 2, 16), "DiT-XL/4": _cfg(28, 1152, 4, 16), "DiT-XL/8": _cfg(28, 1152, 8, 16),
    "DiT-L/2": _cfg(24, 1024, 2, 16), "DiT-L/4": _cfg(24, 1024, 4, 16), "DiT-L/8": _cfg(24, 1024, 8, 16),
    "DiT-B/2": _cfg(12, 768, 2, 12), "DiT-B/4": _cfg(12, 768, 4, 12), "DiT-B/8": _cfg(12, 768, 8, 12),
    "DiT-S/2": _cfg(12, 384, 2, 6), "DiT-S/4": _cfg(12, 384, 4, 6), "DiT-S/8": _cfg(12, 384, 8, 6),
}


def _make(name):
    def ctor(**kwargs):
        c = dict(_CONFIGS[name])
        c.update(kwargs)
        return DiT(**c)
    ctor.__name__ = name.replace("-", "_").replace("/", "_")
    return ctor


# (the */8 configurations need input_size >= 64: the attention kernels take token counts that are multiples of 64, and
#  sfron_dit_param_layout rejects the 16-token grid a 32 x 32 latent gives them)
DiT_models = {name: _make(name) for name in _CONFIGS}


def randomize_zero_init(model, std=0.02, seed=0):
    """re-draw every all-zero trainable tensor so synthetic runs exercise every kernel."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.requires_grad and not bool(p.any()):
                p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))
    model.engine.sync_bf16()
    return model
