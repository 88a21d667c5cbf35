"""Oracle (test infrastructure): the DDPM conditional U-Net of BASELINE config 0, restated in plain PyTorch fp32.

Follows /root/reference/DDPM/models/diffusion.py:
  prob_mask_like :8-14, get_timestep_embedding :17-35 (sin || cos, log(1e4)/(half-1)), swish :38-40,
  GroupNorm(32, eps 1e-6) :43-46, Upsample :49-63 (nearest x2 + conv3x3), Downsample :66-82 (pad (0,1,0,1) + conv3x3 s2),
  ResnetBlock :85-145 (temb||cemb projection, dropout), AttnBlock :148-192 (single head, scale C^-0.5),
  Conditional_Model :195-413 (mode "train" -> cond_drop_prob, mode "test" -> (1+s)*cond - s*null).
state_dict keys match the reference (checked by tests/test_oracle_golden.py against the imported class).
Every random draw keeps the reference's order (CFG keep-mask first, then one dropout per ResnetBlock in
execution order), so a seeded forward is reproducible against the reference; ``keep_mask`` can also be given.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(t, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * -e)
    e = t.float()[:, None] * e[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, (0, 1, 0, 0))
    return e


def swish(x):
    return x * torch.sigmoid(x)


def gn(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, c, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    def __init__(self, c, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        if self.with_conv:
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return F.avg_pool2d(x, 2, 2)


class ResnetBlock(nn.Module):
    def __init__(self, cin, cout, dropout, emb_ch):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = gn(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        # reference quirk (:93-110): cemb_channels keeps its default 512 whatever `ch` is, so the projection is
        # Linear(temb_ch + 512, cout) and the model only runs for ch = 128 (temb_ch = cemb_ch = 512)
        self.temb_cemb_proj = nn.Linear(emb_ch + 512, cout)
        self.norm2 = gn(cout)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)

    def forward(self, x, temb, cemb):
        h = self.conv1(swish(self.norm1(x)))
        h = h + self.temb_cemb_proj(swish(torch.cat([temb, cemb], dim=-1)))[:, :, None, None]
        h = self.conv2(self.dropout(swish(self.norm2(h))))
        if self.cin != self.cout:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        h = self.norm(x)
        q, k, v = self.q(h), self.k(h), self.v(h)
        b, c, hh, ww = q.shape
        w = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
        w = F.softmax(w, dim=2)
        h = torch.bmm(v.reshape(b, c, hh * ww), w.permute(0, 2, 1)).reshape(b, c, hh, ww)
        return x + self.proj_out(h)


class ConditionalUNet(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), dropout=0.1,
                 in_channels=3, resolution=32, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1):
        super().__init__()
        self.ch, self.resolution, self.cond_drop_prob = ch, resolution, cond_drop_prob
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        emb = ch * 4
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, emb), nn.Linear(emb, emb)])
        self.classes_emb = nn.Embedding(n_classes, ch)
        self.null_classes_emb = nn.Parameter(torch.randn(ch))
        self.cemb = nn.Module()
        self.cemb.dense = nn.ModuleList([nn.Linear(ch, emb), nn.Linear(emb, emb)])
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        res = resolution
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        bin_ = None
        for lvl in range(self.num_resolutions):
            blocks, attns = nn.ModuleList(), nn.ModuleList()
            bin_, bout = ch * in_mult[lvl], ch * ch_mult[lvl]
            for _ in range(num_res_blocks):
                blocks.append(ResnetBlock(bin_, bout, dropout, emb))
                bin_ = bout
                if res in attn_resolutions:
                    attns.append(AttnBlock(bin_))
            d = nn.Module()
            d.block, d.attn = blocks, attns
            if lvl != self.num_resolutions - 1:
                d.downsample = Downsample(bin_, resamp_with_conv)
                res //= 2
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(bin_, bin_, dropout, emb)
        self.mid.attn_1 = AttnBlock(bin_)
        self.mid.block_2 = ResnetBlock(bin_, bin_, dropout, emb)
        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            blocks, attns = nn.ModuleList(), nn.ModuleList()
            bout, skip = ch * ch_mult[lvl], ch * ch_mult[lvl]
            for ib in range(num_res_blocks + 1):
                if ib == num_res_blocks:
                    skip = ch * in_mult[lvl]
                blocks.append(ResnetBlock(bin_ + skip, bout, dropout, emb))
                bin_ = bout
                if res in attn_resolutions:
                    attns.append(AttnBlock(bin_))
            u = nn.Module()
            u.block, u.attn = blocks, attns
            if lvl != 0:
                u.upsample = Upsample(bin_, resamp_with_conv)
                res *= 2
            self.up.insert(0, u)
        self.norm_out = gn(bin_)
        self.conv_out = nn.Conv2d(bin_, out_ch, 3, 1, 1)

    def forward(self, x, t, c, mode="train", cond_drop_prob=None, cond_scale=None, keep_mask=None):
        if mode == "train":
            return self._forward(x, t, c, cond_drop_prob, keep_mask)
        logits = self._forward(x, t, c, 0.0)
        if cond_scale == 0:
            return logits
        return (1 + cond_scale) * logits - cond_scale * self._forward(x, t, c, 1.0)

    def _forward(self, x, t, c, cond_drop_prob=None, keep_mask=None):
        b = x.shape[0]
        p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
        temb = self.temb.dense[1](swish(self.temb.dense[0](timestep_embedding(t, self.ch))))
        cemb = self.classes_emb(c)
        if p > 0:
            if keep_mask is None:          # prob_mask_like((b,), 1 - p)
                q = 1 - p
                dev = x.device
                keep_mask = (torch.ones(b, dtype=torch.bool, device=dev) if q == 1 else torch.zeros(b, dtype=torch.bool, device=dev)
                             if q == 0 else torch.zeros(b, device=dev).float().uniform_(0, 1) < q)
            cemb = torch.where(keep_mask[:, None], cemb, self.null_classes_emb[None, :].expand(b, -1))
        cemb = self.cemb.dense[1](swish(self.cemb.dense[0](cemb)))
        hs = [self.conv_in(x)]
        for lvl in range(self.num_resolutions):
            for ib in range(self.num_res_blocks):
                h = self.down[lvl].block[ib](hs[-1], temb, cemb)
                if len(self.down[lvl].attn) > 0:
                    h = self.down[lvl].attn[ib](h)
                hs.append(h)
            if lvl != self.num_resolutions - 1:
                hs.append(self.down[lvl].downsample(hs[-1]))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(hs[-1], temb, cemb)), temb, cemb)
        for lvl in reversed(range(self.num_resolutions)):
            for ib in range(self.num_res_blocks + 1):
                h = self.up[lvl].block[ib](torch.cat([h, hs.pop()], dim=1), temb, cemb)
                if len(self.up[lvl].attn) > 0:
                    h = self.up[lvl].attn[ib](h)
            if lvl != 0:
                h = self.up[lvl].upsample(h)
        return self.conv_out(swish(self.norm_out(h)))
