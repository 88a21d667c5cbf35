"""Oracle (test infrastructure): the Stable-Diffusion / LDM U-Net and the NSFW-removal SFR-on loop, restated in plain PyTorch fp32.

Follows /root/reference/SD:
  ldm/modules/diffusionmodules/util.py:21-53 (make_beta_schedule "linear" = linspace of sqrt(beta) squared),
      :173-197 (timestep_embedding: cos || sin, freq_i = exp(-ln(1e4) i / half)), :225-242 (GroupNorm32: fp32 GroupNorm(32), eps 1e-5)
  ldm/modules/diffusionmodules/openaimodel.py:77-91 (TimestepEmbedSequential dispatch), :94-128 (Upsample: nearest x2 + conv3x3),
      :147-174 (Downsample: conv3x3 stride 2 pad 1), :177-288 (ResBlock), :428-846 (UNetModel with use_spatial_transformer)
  ldm/modules/attention.py:37-63 (GEGLU, FeedForward), :74-77 (Normalize: GroupNorm(32, eps 1e-6)), :149-193 (CrossAttention),
      :196-250 (BasicTransformerBlock), :253-303 (SpatialTransformer)
  ldm/models/diffusion/ddpm.py:153-240 (register_schedule), :424-429 (q_sample), :1286-1319 (p_losses, eps / l2, logvar = 0)
  train-scripts/nsfw_removal.py:108-173 (loop body: forget = MSE(eps(x_f, c_forget), stopgrad eps(x_p, c_pseudo)) with shared t and
      noise; mask application as written is a no-op (a str is tested against a list of Parameters);
      Adam x2 on one state, no clipping, no EMA)
state_dict keys match the reference class (tests/golden/sd_unet.npz holds the key list and the outputs of the imported reference;
tests/test_oracle_golden.py checks this restatement against it).  Gradient checkpointing (util.py:118-170) recomputes the same
function: it does not change values and is not restated.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def normalization(c):
    return GroupNorm32(32, c)                       # torch default eps 1e-5


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class Upsample(nn.Module):
    def __init__(self, c, use_conv, out_channels=None):
        super().__init__()
        self.use_conv = use_conv
        if use_conv:
            self.conv = nn.Conv2d(c, out_channels or c, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    def __init__(self, c, use_conv, out_channels=None):
        super().__init__()
        self.op = nn.Conv2d(c, out_channels or c, 3, stride=2, padding=1) if use_conv else nn.AvgPool2d(2, 2)

    def forward(self, x):
        return self.op(x)


class ResBlock(nn.Module):
    def __init__(self, c, emb_c, dropout, out_channels=None):
        super().__init__()
        self.c, self.oc = c, out_channels or c
        self.in_layers = nn.Sequential(normalization(c), nn.SiLU(), nn.Conv2d(c, self.oc, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_c, self.oc))
        self.out_layers = nn.Sequential(normalization(self.oc), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.oc, self.oc, 3, padding=1)))
        self.skip_connection = nn.Identity() if self.oc == c else nn.Conv2d(c, self.oc, 1)

    def forward(self, x, emb):
        h = self.in_layers(x)
        h = h + self.emb_layers(emb)[..., None, None]
        h = self.out_layers(h)
        return self.skip_connection(x) + h


class GEGLU(nn.Module):
    def __init__(self, di, do):
        super().__init__()
        self.proj = nn.Linear(di, do * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim if context_dim is not None else query_dim
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def forward(self, x, context=None):
        h = self.heads
        context = x if context is None else context
        q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
        b, n, _ = q.shape

        def split(t):                                # 'b n (h d) -> (b h) n d'
            return t.view(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)
        q, k, v = split(q), split(k), split(v)
        attn = (torch.einsum("bid,bjd->bij", q, k) * self.scale).softmax(dim=-1)
        out = torch.einsum("bij,bjd->bid", attn, v)
        out = out.view(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, c, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, c, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(c, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim) for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, c, 1))

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        x_in = x
        x = self.proj_in(self.norm(x))
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            x = blk(x, context=context)
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(x) + x_in


class TimestepEmbedSequential(nn.Sequential):
    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    """openaimodel.py:428-846 for the options v1-inference.yaml uses: use_spatial_transformer, conv_resample, num_heads (head
    dim = channels / heads), no class conditioning, no scale-shift norm, no resblock_updown."""

    def __init__(self, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1), dropout=0.0,
                 channel_mult=(1, 2, 4, 4), num_heads=8, transformer_depth=1, context_dim=768, **unused):
        super().__init__()
        self.model_channels = model_channels
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, depth=transformer_depth, context_dim=context_dim))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, True, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, ted, dropout), SpatialTransformer(ch, num_heads, ch // num_heads, depth=transformer_depth, context_dim=context_dim),
            ResBlock(ch, ted, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, depth=transformer_depth, context_dim=context_dim))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, True, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

    def forward(self, x, timesteps=None, context=None):
        hs = []
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        h = x
        for m in self.input_blocks:
            h = m(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for m in self.output_blocks:
            h = m(torch.cat([h, hs.pop()], dim=1), emb, context)
        return self.out(h)


def randomize_zero_init(model, std=0.02, seed=0):
    """Re-draw every all-zero parameter tensor (the zero_module convolutions): with them at zero the output is identically zero
    and most gradients vanish (the same degeneracy as a freshly constructed DiT)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if not bool(p.any()):
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


# ----------------------------------------------------------------------------------------------- LDM schedule and losses
class LDMSchedule:
    """ddpm.py:153-240 for v1-inference.yaml (linear_start 0.00085, linear_end 0.012, 1000 steps): fp64 numpy tables, fp32 buffers."""

    def __init__(self, timesteps=1000, linear_start=0.00085, linear_end=0.012):
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = timesteps
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac), dtype=torch.float32)
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)

    def q_sample(self, x0, t, noise):
        a = self.sqrt_alphas_cumprod.to(x0.device)[t].view(-1, 1, 1, 1)
        s = self.sqrt_one_minus_alphas_cumprod.to(x0.device)[t].view(-1, 1, 1, 1)
        return a * x0 + s * noise

    def p_losses(self, unet, x0, context, t, noise):
        """eps-parameterisation, l2, logvar = 0, l_simple_weight 1, original_elbo_weight 0 (ddpm.py:1286-1319)."""
        out = unet(self.q_sample(x0, t, noise), t, context=context)
        return F.mse_loss(noise, out, reduction="none").mean([1, 2, 3]).mean()


class SDSfronOracle:
    """train-scripts/nsfw_removal.py:108-173 over latents and prompt embeddings that are already resident (the VAE / CLIP front-end
    of model.get_input is outside the path).  ``trainable``: names of the parameters handed to Adam (train_method "full": all;
    "xattn": those containing "attn2", :66-77).  ``mask_mode``: "as_written" (the reference's `n in parameters` test is always
    False: no masking) or "intended" (grad *= mask[name])."""

    def __init__(self, unet, schedule, lr=1e-5, forget_alpha=1.0, remain_alpha=1.0, train_method="full", mask=None, mask_mode="as_written"):
        self.unet, self.s = unet, schedule
        self.names = [n for n, _ in unet.named_parameters() if train_method == "full" or "attn2" in n]
        params = dict(unet.named_parameters())
        self.opt = torch.optim.Adam([params[n] for n in self.names], lr=lr)
        self.fa, self.ra, self.mask, self.mask_mode = forget_alpha, remain_alpha, mask, mask_mode
        unet.train()

    def step(self, forget, remain):
        """forget: x_f, x_p (latents of the forget / pseudo images), c_f, c_p (contexts), t, noise; remain: x, c, t, noise."""
        u, s = self.unet, self.s
        self.opt.zero_grad()
        f_out = u(s.q_sample(forget["x_f"], forget["t"], forget["noise"]), forget["t"], context=forget["c_f"])
        p_out = u(s.q_sample(forget["x_p"], forget["t"], forget["noise"]), forget["t"], context=forget["c_p"]).detach()
        ori_forget = F.mse_loss(f_out, p_out)
        (self.fa * ori_forget).backward()
        if self.mask is not None and self.mask_mode == "intended":
            for n, p in u.named_parameters():
                if p.grad is not None and n in self.names:
                    p.grad *= self.mask[n]
        self.opt.step()
        self.opt.zero_grad()
        ori_remain = s.p_losses(u, remain["x"], remain["c"], remain["t"], remain["noise"])
        (self.ra * ori_remain).backward()
        self.opt.step()
        return {"forget_loss": ori_forget.item(), "remain_loss": ori_remain.item()}


def sd_fisher(unet, schedule, batches, c_guidance):
    """train-scripts/generate_fisher.py:36-79 (and the remain twin :87-128) over resident latents / prompt embeddings:
    preds = (1 + c) eps(x_t, c_prompt) - c eps(x_t, c_null), loss = -MSELoss(noise, preds), F[name] += grad^2 / len(batches); eval
    mode (:25).  batches: list of dict(x, c, c_null, t, noise).  Returns name -> fp32 tensor."""
    unet.eval()
    fisher = {n: 0 for n, _ in unet.named_parameters()}
    for b in batches:
        unet.zero_grad()
        x_t = schedule.q_sample(b["x"], b["t"], b["noise"])
        preds = (1 + c_guidance) * unet(x_t, b["t"], context=b["c"]) - c_guidance * unet(x_t, b["t"], context=b["c_null"])
        (-F.mse_loss(b["noise"], preds)).backward()
        with torch.no_grad():
            for n, p in unet.named_parameters():
                if p.grad is not None:
                    fisher[n] = fisher[n] + p.grad.detach() ** 2 / len(batches)
    return fisher
