"""The Stable-Diffusion / LDM U-Net (BASELINE config 4: concept erasure, SD/train-scripts/nsfw_removal.py) over the HIP kernels.

Mirrors /root/reference/SD/ldm/modules/diffusionmodules/openaimodel.py:428-846 ``UNetModel`` for the options v1-inference.yaml
sets (use_spatial_transformer, conv_resample, num_heads, transformer_depth 1, no class conditioning): ``forward(x, timesteps,
context)``, identical ``state_dict()`` keys / shapes / order (checked against oracle.sd_ref, which tests/golden/sd_unet.npz pins
to the imported reference), so a CompVis checkpoint's ``model.diffusion_model.*`` entries load unchanged and
sfron.export.compvis_unet_to_diffusers re-keys them for UNet2DConditionModel.

Blocks (all on the kernels of csrc/conv.hip, no autograd inside -- the tape machinery of sfron.unet):
  ResBlock (openaimodel.py:177-288)            GroupNorm32(eps 1e-5)+SiLU -> conv3x3 (+ emb projection) -> GroupNorm32+SiLU -> conv3x3
  SpatialTransformer (attention.py:253-303)    GroupNorm(eps 1e-6) -> 1x1 proj_in -> BasicTransformerBlock -> 1x1 proj_out -> + x
  BasicTransformerBlock (:196-250)             LN -> self-attention + x;  LN -> cross-attention(context) + x;  LN -> GEGLU FF + x
  CrossAttention (:149-193)                    heads are column slices of the q / k / v matrices: one batched GEMM per product with an
                                               inner batch over heads; the 77 context tokens are padded to 80 rows (zero keys, softmax
                                               probability 0)
  Downsample / Upsample (:94-174)              stride-2 pad-1 conv3x3 / nearest x2 folded into the conv's im2col addressing
The reference wraps every block in gradient checkpointing (util.py:118-170); with 288 GB of HBM the activations are simply kept.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .unet import Act, _TapeNet, _L, _pad8, bgemm, cast_rows, count_flops


class UNetModel(_TapeNet):
    GN_EPS = 1e-5                                    # GroupNorm32 (util.py:225-242); SpatialTransformer.norm passes 1e-6
    RES_NAMES = (".in_layers.0", ".in_layers.2", ".out_layers.0", ".out_layers.3", ".skip_connection")

    def __init__(self, image_size=32, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
                 dropout=0, channel_mult=(1, 2, 4, 4), num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                 use_checkpoint=True, legacy=False, device="cuda", **unused):
        super().__init__()
        if not use_spatial_transformer or transformer_depth != 1 or legacy:
            raise NotImplementedError("only the v1-inference.yaml form (use_spatial_transformer, depth 1, legacy False) is built")
        if dropout:
            raise NotImplementedError("v1-inference.yaml trains the UNet with dropout 0")
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise _lib.SfronError("UNetModel needs a GPU (no CPU fallback)")
        self.in_channels, self.out_channels, self.mc = in_channels, out_channels, model_channels
        self.num_res_blocks, self.attn_res, self.channel_mult = num_res_blocks, tuple(attention_resolutions), tuple(channel_mult)
        self.heads, self.ctx_dim, self.ted = num_heads, context_dim, model_channels * 4
        self.dropout_p = 0.0
        self._plan()
        self._alloc()
        self._register_views()
        self.reset_parameters()

    # -------------------------------------------------------------------------------------------- structure
    def _plan(self):
        P, mc, ted = [], self.mc, self.ted

        def wb(name, *shape):
            P.extend([(name + ".weight", tuple(shape)), (name + ".bias", (shape[0],))])

        def res(name, cin, cout):
            wb(name + ".in_layers.0", cin); wb(name + ".in_layers.2", cout, cin, 3, 3); wb(name + ".emb_layers.1", cout, ted)
            wb(name + ".out_layers.0", cout); wb(name + ".out_layers.3", cout, cout, 3, 3)
            if cin != cout:
                wb(name + ".skip_connection", cout, cin, 1, 1)
            self.res_blocks.append((name, cin, cout))

        def st(name, c):
            wb(name + ".norm", c); wb(name + ".proj_in", c, c, 1, 1)
            t = name + ".transformer_blocks.0"
            for a, kd in ((".attn1", c), (".attn2", self.ctx_dim)):
                if a == ".attn2":
                    wb(t + ".ff.net.0.proj", 8 * c, c); wb(t + ".ff.net.2", c, 4 * c)
                P.extend([(t + a + ".to_q.weight", (c, c)), (t + a + ".to_k.weight", (c, kd)), (t + a + ".to_v.weight", (c, kd))])
                wb(t + a + ".to_out.0", c, c)
            for n in (".norm1", ".norm2", ".norm3"):
                wb(t + n, c)
            wb(name + ".proj_out", c, c, 1, 1)
            self.st_blocks.append((name, c))

        self.res_blocks, self.st_blocks = [], []
        wb("time_embed.0", ted, mc); wb("time_embed.2", ted, ted)
        wb("input_blocks.0.0", mc, self.in_channels, 3, 3)
        self.inp, self.outp = [("conv_in", None, None)], []          # (kind, names...) per input / output block
        chans, ch, ds, i = [mc], mc, 1, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks):
                res(f"input_blocks.{i}.0", ch, mult * mc)
                ch = mult * mc
                att = None
                if ds in self.attn_res:
                    att = f"input_blocks.{i}.1"
                    st(att, ch)
                self.inp.append(("res", f"input_blocks.{i}.0", att))
                chans.append(ch); i += 1
            if level != len(self.channel_mult) - 1:
                wb(f"input_blocks.{i}.0.op", ch, ch, 3, 3)
                self.inp.append(("down", f"input_blocks.{i}.0.op", None))
                chans.append(ch); ds *= 2; i += 1
        res("middle_block.0", ch, ch); st("middle_block.1", ch); res("middle_block.2", ch, ch)
        self.mid_c = ch
        j = 0
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for k in range(self.num_res_blocks + 1):
                ich = chans.pop()
                res(f"output_blocks.{j}.0", ch + ich, mc * mult)
                ch = mc * mult
                att, up, sub = None, None, 1
                if ds in self.attn_res:
                    att = f"output_blocks.{j}.{sub}"; st(att, ch); sub += 1
                if level and k == self.num_res_blocks:
                    up = f"output_blocks.{j}.{sub}.conv"; wb(up, ch, ch, 3, 3); ds //= 2
                self.outp.append((f"output_blocks.{j}.0", att, up))
                j += 1
        wb("out.0", ch); wb("out.2", self.out_channels, mc, 3, 3)
        self.final_c = ch
        self.param_specs = P

    def _alloc(self):
        """Contiguous groups: every ResBlock's emb_layers.1 weight (ONE [sum Cout][4 mc] projection GEMM of silu(emb) per pass) and
        bias; per transformer block attn1's to_q / to_k / to_v as one [3C][C] matrix and attn2's to_k / to_v as one [2C][ctx] matrix."""
        specs = OrderedDict(self.param_specs)
        rn = [n for n, _, _ in self.res_blocks]
        groups = [[n + ".emb_layers.1.weight" for n in rn], [n + ".emb_layers.1.bias" for n in rn]]
        for n, _ in self.st_blocks:
            t = n + ".transformer_blocks.0"
            groups.append([t + ".attn1.to_q.weight", t + ".attn1.to_k.weight", t + ".attn1.to_v.weight"])
            groups.append([t + ".attn2.to_k.weight", t + ".attn2.to_v.weight"])
        self._alloc_arena(specs, groups)
        self.proj_w_off = self.index[rn[0] + ".emb_layers.1.weight"][0]
        self.proj_b_off = self.index[rn[0] + ".emb_layers.1.bias"][0]
        self.proj_slices, c0 = {}, 0
        for n, _, cout in self.res_blocks:
            self.proj_slices[n] = (c0, cout)
            c0 += cout
        self.proj_total = c0

    def reset_parameters(self):
        """torch defaults of the layer types (kaiming_uniform(a = sqrt 5) + fan-in bias; norms 1 / 0) and the reference's zero_module
        on every ResBlock's out conv, every SpatialTransformer's proj_out and the final conv (openaimodel.py:240-244,787-791)."""
        import math
        import torch.nn as nn
        with torch.no_grad():
            for name, p in self.named_parameters():
                base = name.rsplit(".", 1)[0]
                is_norm = any(base.endswith(s) for s in (".in_layers.0", ".out_layers.0", ".norm", ".norm1", ".norm2", ".norm3")) or base == "out.0"
                if base.endswith(".out_layers.3") or base.endswith(".proj_out") or base == "out.2":
                    p.zero_()
                elif is_norm:
                    p.fill_(1.0 if name.endswith(".weight") else 0.0)
                elif name.endswith(".weight"):
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                else:
                    bound = 1 / math.sqrt(self.view(self.params, base + ".weight")[0].numel())
                    p.uniform_(-bound, bound)
        self.sync_bf16()

    def load_state_dict(self, state_dict, strict=True, **kw):
        pre = "model.diffusion_model."
        if state_dict and any(k.startswith(pre) for k in state_dict):            # a whole CompVis LatentDiffusion state dict
            state_dict = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # -------------------------------------------------------------------------------------------- blocks
    def _layernorm(self, x, name):
        dev = self.device_
        rows, D = x.rows, x.C
        y = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
        mean = torch.empty(rows, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        gam, bet = self._p(name + ".weight"), self._p(name + ".bias")
        trains = self._trains(name + ".weight")          # decided when the tape is built (a frozen norm's affine gradients are read by nobody)
        check(_L().sfron_layernorm_fwd(ptr(x.t), gam, bet, rows, D, 1e-5, ptr(y), ptr(mean), ptr(rstd), stream_ptr()), "layernorm_fwd")

        def bwd(dy, extra=None):
            """extra: fp32 [rows][D], the residual stream's share of x's gradient, added in the same pass (x.grad (+)= extra + d norm)."""
            g, acc = x.grad_buf()
            rpb = _L().sfron_layernorm_rows_per_block(rows)
            nblk = (rows + rpb - 1) // rpb
            pg = torch.empty(nblk, D, dtype=torch.float32, device=dev)
            pb = torch.empty_like(pg)
            check(_L().sfron_layernorm_bwd_res(ptr(dy), ptr(x.t), gam, ptr(mean), ptr(rstd), rows, D, ptr(g), acc, ptr(extra), ptr(pg), ptr(pb),
                                               stream_ptr()), "layernorm_bwd")
            if trains:
                self._reduce(pg, 1, nblk, D, self._g(name + ".weight"), D)
                self._reduce(pb, 1, nblk, D, self._g(name + ".bias"), D)
        return y, bwd

    def _mha(self, q, ldq, k, ldk, v, ldv, B, N, Lk, Lv, C, keep):
        """softmax(q k^T d^-0.5) v over `heads` column slices.  q [B*N][.], k / v [B*Lk][.] bf16 (addresses + leading dimensions);
        Lv = number of real keys (<= Lk, the rest is zero padding); keep = the tensors behind the addresses (held by the backward
        closure: the forward's temporaries must outlive it).  Returns (O bf16 [B*N][C], backward(dO bf16, dq, dk, dv addresses
        with the same leading dimensions))."""
        dev, h = self.device_, self.heads
        d = C // h
        scale = float(d ** -0.5)
        S = torch.empty(B * h * N, Lk, dtype=torch.float32, device=dev)
        bgemm(q, k, N, Lk, d, lda=ldq, ldb=ldk, batch=B, sa=N * ldq, sb=Lk * ldk, sc=h * N * Lk, batch2=h, sa2=d, sb2=d, sc2=N * Lk, c_f32=S, ldc=Lk)
        P = torch.empty(B * h * N, Lk, dtype=torch.bfloat16, device=dev)
        check(_L().sfron_softmax_fwd(ptr(S), B * h * N, Lk, Lv, scale, ptr(P), stream_ptr()), "softmax_fwd")
        del S
        O = torch.empty(B * N, C, dtype=torch.bfloat16, device=dev)
        bgemm(P, v, N, d, Lk, lda=Lk, ldb=ldv, b_t=True, batch=B, sa=h * N * Lk, sb=Lk * ldv, sc=N * C, batch2=h, sa2=N * Lk, sb2=d, sc2=d,
              c_bf16=O, ldc=C)

        def bwd(dO, dq, dk, dv, _keep=keep):
            dP = torch.empty(B * h * N, Lk, dtype=torch.float32, device=dev)
            bgemm(dO, v, N, Lk, d, lda=C, ldb=ldv, batch=B, sa=N * C, sb=Lk * ldv, sc=h * N * Lk, batch2=h, sa2=d, sb2=d, sc2=N * Lk, c_f32=dP, ldc=Lk)
            dS = torch.empty(B * h * N, Lk, dtype=torch.bfloat16, device=dev)
            check(_L().sfron_softmax_bwd(ptr(P), ptr(dP), B * h * N, Lk, scale, ptr(dS), stream_ptr()), "softmax_bwd")
            del dP
            bgemm(dS, k, N, d, Lk, lda=Lk, ldb=ldk, b_t=True, batch=B, sa=h * N * Lk, sb=Lk * ldk, sc=N * ldq, batch2=h, sa2=N * Lk, sb2=d, sc2=d,
                  c_bf16=dq, ldc=ldq)
            if dk is not None:
                bgemm(dS, q, Lk, d, N, lda=Lk, ldb=ldq, a_t=True, b_t=True, batch=B, sa=h * N * Lk, sb=N * ldq, sc=Lk * ldk, batch2=h, sa2=N * Lk,
                      sb2=d, sc2=d, c_bf16=dk, ldc=ldk)
                bgemm(P, dO, Lk, d, N, lda=Lk, ldb=C, a_t=True, b_t=True, batch=B, sa=h * N * Lk, sb=N * C, sc=Lk * ldv, batch2=h, sa2=N * Lk,
                      sb2=d, sc2=d, c_bf16=dv, ldc=ldv)
        return O, bwd

    def _flash_self_attention(self, qkv, B, N, C):
        """Self-attention through the fused flash-style kernels of csrc/attn.hip (qkv [B*N][3C], column = which * C + head * d + i)."""
        h = self.heads
        d = C // h
        dev = self.device_
        O = torch.empty(B * N, C, dtype=torch.bfloat16, device=dev)
        lse = torch.empty(B * h * N, dtype=torch.float32, device=dev)
        count_flops(4.0 * N * N * d * h * B)
        check(_L().sfron_attn_fwd(ptr(qkv), ptr(O), ptr(lse), B, N, h, d, stream_ptr()), "attn_fwd")

        def bwd(dO, dq, dk, dv):
            # the kernel writes the whole dqkv matrix: dq is its base address (dk = dq + C, dv = dq + 2C by construction)
            delta = torch.empty(B * h * N, dtype=torch.float32, device=dev)
            count_flops(8.0 * N * N * d * h * B)          # dS, dQ, dK, dV (the recomputed S is not algorithmic work)
            check(_L().sfron_attn_bwd(ptr(qkv), ptr(O), ptr(dO), ptr(lse), ptr(delta), dq, B, N, h, d, stream_ptr()), "attn_bwd")
        return O, bwd

    def _transformer(self, tape, name, x, ctx, Lp, Lv):
        """SpatialTransformer with one BasicTransformerBlock; ctx bf16 [B*Lp][ctx_dim] (rows >= Lv are zero)."""
        dev, B, C, N = self.device_, x.B, x.C, x.H * x.W
        rows, t = x.rows, name + ".transformer_blocks.0"
        hn, gn_b = self._gn(tape, x, name + ".norm", False, eps=1e-6)
        x0_t, pin_b = self._linear(hn, rows, name + ".proj_in", C, C)
        X0 = Act(x0_t, B, x.H, x.W, C)
        # ---- self-attention
        n1, ln1_b = self._layernorm(X0, t + ".norm1")
        qkv = torch.empty(rows, 3 * C, dtype=torch.bfloat16, device=dev)
        wqkv = self._w(t + ".attn1.to_q.weight")
        bgemm(n1, wqkv, rows, 3 * C, C, lda=C, ldb=C, c_bf16=qkv, ldc=3 * C)
        if (C // self.heads) <= 80 and (C // self.heads) % 8 == 0 and N % 64 == 0:
            O1, att1_b = self._flash_self_attention(qkv, B, N, C)       # scores never leave the chip (4096 tokens at 64x64)
        else:
            O1, att1_b = self._mha(qkv.data_ptr(), 3 * C, qkv.data_ptr() + 2 * C, 3 * C, qkv.data_ptr() + 4 * C, 3 * C, B, N, N, N, C, keep=(qkv,))
        x1_t, o1_b = self._linear(O1, rows, t + ".attn1.to_out.0", C, C, resid=X0.t)
        X1 = Act(x1_t, B, x.H, x.W, C)
        # ---- cross-attention: keys / values from the context
        n2, ln2_b = self._layernorm(X1, t + ".norm2")
        q2 = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
        bgemm(n2, self._w(t + ".attn2.to_q.weight"), rows, C, C, lda=C, ldb=C, c_bf16=q2, ldc=C)
        kv = torch.empty(B * Lp, 2 * C, dtype=torch.bfloat16, device=dev)
        bgemm(ctx, self._w(t + ".attn2.to_k.weight"), B * Lp, 2 * C, self.ctx_dim, lda=self.ctx_dim, ldb=self.ctx_dim, c_bf16=kv, ldc=2 * C)
        O2, att2_b = self._mha(q2.data_ptr(), C, kv.data_ptr(), 2 * C, kv.data_ptr() + 2 * C, 2 * C, B, N, Lp, Lv, C, keep=(q2, kv))
        x2_t, o2_b = self._linear(O2, rows, t + ".attn2.to_out.0", C, C, resid=X1.t)
        X2 = Act(x2_t, B, x.H, x.W, C)
        # ---- GEGLU feed-forward
        n3, ln3_b = self._layernorm(X2, t + ".norm3")
        hff, ff0_b = self._linear(n3, rows, t + ".ff.net.0.proj", C, 8 * C)
        gg = torch.empty(rows, 4 * C, dtype=torch.bfloat16, device=dev)
        check(_L().sfron_geglu_fwd(ptr(hff), rows, 4 * C, ptr(gg), stream_ptr()), "geglu_fwd")
        x3_t, ff2_b = self._linear(gg, rows, t + ".ff.net.2", 4 * C, C, resid=X2.t)
        x3b = cast_rows(x3_t, C, rows, C, dev)
        out_t, pout_b = self._linear(x3b, rows, name + ".proj_out", C, C, resid=x.t)
        out = Act(out_t, B, x.H, x.W, C)

        trains_qkv1 = self._trains(t + ".attn1.to_q.weight")

        def bwd():
            # the residual stream's share of each gradient (x_in, X2, X1, X0) is added by the norm's backward pass that writes the same
            # tensor (sfron_layernorm_bwd_res / sfron_groupnorm_bwd_res), not by a pass of its own
            d_out = out.grad
            d3 = pout_b(d_out, C)                                      # d X3
            d_gg = ff2_b(d3, C)                                        # fp32 [rows][4C]
            dh = torch.empty(rows, 8 * C, dtype=torch.bfloat16, device=dev)
            check(_L().sfron_geglu_bwd(ptr(d_gg), ptr(hff), rows, 4 * C, ptr(dh), stream_ptr()), "geglu_bwd")
            ln3_b(ff0_b(None, 8 * C, d_bf=dh), d3)                     # -> X2.grad = d3 (ff residual) + d norm3
            d2 = X2.grad
            dO2b = o2_b(d2, C, dx_bf16=True)                           # bf16 [rows][C]: the attention backward's d_o operand
            dq2 = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
            dkv = torch.empty(B * Lp, 2 * C, dtype=torch.bfloat16, device=dev)
            att2_b(dO2b, dq2.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + 2 * C)
            bgemm(dkv, ctx, 2 * C, self.ctx_dim, B * Lp, lda=2 * C, ldb=self.ctx_dim, a_t=True, b_t=True, c_f32=self._g(t + ".attn2.to_k.weight"),
                  ldc=self.ctx_dim)
            bgemm(dq2, n2, C, C, rows, lda=C, ldb=C, a_t=True, b_t=True, c_f32=self._g(t + ".attn2.to_q.weight"), ldc=C)
            dn2 = torch.empty(rows, C, dtype=torch.float32, device=dev)
            bgemm(dq2, self._w(t + ".attn2.to_q.weight"), rows, C, C, lda=C, ldb=C, b_t=True, c_f32=dn2, ldc=C)
            ln2_b(dn2, d2)                                             # -> X1.grad = d2 + d norm2
            d1 = X1.grad
            dO1b = o1_b(d1, C, dx_bf16=True)
            dqkv = torch.empty(rows, 3 * C, dtype=torch.bfloat16, device=dev)
            att1_b(dO1b, dqkv.data_ptr(), dqkv.data_ptr() + 2 * C, dqkv.data_ptr() + 4 * C)
            if trains_qkv1:
                bgemm(dqkv, n1, 3 * C, C, rows, lda=3 * C, ldb=C, a_t=True, b_t=True, c_f32=self._g(t + ".attn1.to_q.weight"), ldc=C)
            dn1 = torch.empty(rows, C, dtype=torch.float32, device=dev)
            bgemm(dqkv, wqkv, rows, C, 3 * C, lda=3 * C, ldb=C, b_t=True, c_f32=dn1, ldc=C)
            ln1_b(dn1, d1)                                             # -> X0.grad = d1 + d norm1
            gn_b(pin_b(X0.grad, C), d_out, C)                          # -> x.grad (+)= d_out (+ x_in) + d norm
        tape.append(bwd)
        return out

    # -------------------------------------------------------------------------------------------- forward / backward
    def _run(self, x, timesteps, context, need_grad):
        L, dev = _L(), self.device_
        B, mc, ted = x.shape[0], self.mc, self.ted
        S0 = x.shape[2]
        tape = []
        self._prep_conv_weights()
        # context: [B][Lv][ctx] -> bf16 rows padded to a multiple of 8 tokens (zero rows: zero keys / values, probability 0)
        Lv = context.shape[1]
        Lp = _pad8(Lv)
        ctx = torch.zeros(B * Lp, self.ctx_dim, dtype=torch.bfloat16, device=dev)
        ctx.view(B, Lp, self.ctx_dim)[:, :Lv].copy_(context)
        # ---- time embedding (:817-818): emb = Linear(SiLU(Linear(timestep_embedding(t))))
        te = torch.empty(B, mc, dtype=torch.bfloat16, device=dev)
        check(L.sfron_timestep_embed(ptr(timesteps.to(torch.int64).contiguous()), B, mc, ptr(te), mc, stream_ptr()), "timestep_embed")
        e0, te0_b = self._linear(te, B, "time_embed.0", mc, ted)
        e0s = torch.empty(B, ted, dtype=torch.bfloat16, device=dev)
        check(L.sfron_silu_fwd(ptr(e0), B * ted, ptr(e0s), stream_ptr()), "silu_fwd")
        emb, te2_b = self._linear(e0s, B, "time_embed.2", ted, ted)
        semb = torch.empty(B, ted, dtype=torch.bfloat16, device=dev)
        check(L.sfron_silu_fwd(ptr(emb), B * ted, ptr(semb), stream_ptr()), "silu_fwd")
        PT = self.proj_total
        proj = torch.empty(B, PT, dtype=torch.float32, device=dev)           # emb_layers of every ResBlock, one GEMM
        bgemm(semb, self.params_bf16.data_ptr() + 2 * self.proj_w_off, B, PT, ted, lda=ted, ldb=ted, bias=self.params.data_ptr() + 4 * self.proj_b_off,
              c_f32=proj, ldc=PT)
        d_proj = torch.zeros(B, PT, dtype=torch.float32, device=dev) if need_grad else None
        res_c = {n: (cin, cout) for n, cin, cout in self.res_blocks}
        # ---- input blocks
        cip = self.conv3["input_blocks.0.0"]["cip"]
        xr = torch.empty(B * S0 * S0, cip, dtype=torch.bfloat16, device=dev)
        check(L.sfron_nchw_to_rows_bf16(ptr(x.float().contiguous()), B, self.in_channels, S0 * S0, cip, ptr(xr), stream_ptr()), "nchw_to_rows")
        h0_t, cin_b = self._conv3(xr, B, S0, S0, "input_blocks.0.0", S0, S0)
        hs = [Act(h0_t, B, S0, S0, mc)]
        tape.append(lambda a=hs[0]: cin_b(a.grad, want_dsrc=False))
        res = S0
        for kind, n0, n1 in self.inp[1:]:
            if kind == "res":
                cin, cout = res_c[n0]
                h = self._resblock(tape, n0, hs[-1], cin, cout, proj, d_proj, None)
                if n1:
                    h = self._transformer(tape, n1, h, ctx, Lp, Lv)
                hs.append(h)
            else:
                src = hs[-1]
                sb = cast_rows(src.t, src.C, src.rows, src.C, dev)
                o_t, db = self._conv3(sb, B, res, res, n0, res // 2, res // 2, stride=2, pad=1)
                res //= 2
                o = Act(o_t, B, res, res, src.C)
                hs.append(o)

                def down_bwd(o=o, src=src, db=db):
                    ds = db(o.grad)
                    g, acc = src.grad_buf()
                    check(L.sfron_copy_cols(ptr(ds), src.C, src.rows, src.C, ptr(g), src.C, acc, stream_ptr()), "copy_cols")
                tape.append(down_bwd)
        # ---- middle
        C = self.mid_c
        h = self._resblock(tape, "middle_block.0", hs[-1], C, C, proj, d_proj, None)
        h = self._transformer(tape, "middle_block.1", h, ctx, Lp, Lv)
        h = self._resblock(tape, "middle_block.2", h, C, C, proj, d_proj, None)
        # ---- output blocks
        for n0, att, up in self.outp:
            cin, cout = res_c[n0]
            skip = hs.pop()
            c1, c2 = h.C, skip.C
            cat_t = torch.empty(h.rows, c1 + c2, dtype=torch.float32, device=dev)
            check(L.sfron_copy_cols2(ptr(h.t), c1, c1, ptr(cat_t), c1 + c2, 0, ptr(skip.t), c2, c2, cat_t.data_ptr() + 4 * c1, c1 + c2, 0, h.rows,
                                       stream_ptr()), "copy_cols2")
            ca = Act(cat_t, B, h.H, h.W, c1 + c2)

            def cat_bwd(ca=ca, a=h, s=skip, c1=c1, c2=c2):
                ga, acc_a = a.grad_buf()
                gs, acc_s = s.grad_buf()
                check(L.sfron_copy_cols2(ptr(ca.grad), c1 + c2, c1, ptr(ga), c1, acc_a, ca.grad.data_ptr() + 4 * c1, c1 + c2, c2, ptr(gs), c2, acc_s,
                                           a.rows, stream_ptr()), "copy_cols2")
            tape.append(cat_bwd)
            h = self._resblock(tape, n0, ca, cin, cout, proj, d_proj, None)
            if att:
                h = self._transformer(tape, att, h, ctx, Lp, Lv)
            if up:
                src = h
                sb = cast_rows(src.t, src.C, src.rows, src.C, dev)
                o_t, ub = self._conv3(sb, B, src.H, src.W, up, 2 * src.H, 2 * src.W, up=1)
                h = Act(o_t, B, 2 * src.H, 2 * src.W, src.C)

                def up_bwd(o=h, src=src, ub=ub):
                    ds = ub(o.grad)
                    g, acc = src.grad_buf()
                    check(L.sfron_copy_cols(ptr(ds), src.C, src.rows, src.C, ptr(g), src.C, acc, stream_ptr()), "copy_cols")
                tape.append(up_bwd)
        # ---- out (:841-846)
        a, gn_b = self._gn(tape, h, "out.0", True)
        v = self.conv3["out.2"]
        o_t, co_b = self._conv3(a, B, h.H, h.W, "out.2", h.H, h.W)
        out = torch.empty(B, self.out_channels, h.H, h.W, dtype=torch.float32, device=dev)
        check(L.sfron_rows_to_nchw(ptr(o_t), v["cop"], B, self.out_channels, h.H * h.W, ptr(out), stream_ptr()), "rows_to_nchw")
        if not need_grad:
            return out, None

        def backward(d_out):
            dr = torch.empty(B * h.H * h.W, v["cop"], dtype=torch.float32, device=dev)
            check(L.sfron_nchw_to_rows_f32(ptr(d_out.float().contiguous()), B, self.out_channels, h.H * h.W, v["cop"], ptr(dr), stream_ptr()),
                  "nchw_to_rows_f32")
            self._reduce_begin()
            try:
                gn_b(co_b(dr))
                hook = getattr(self, "_tape_hook", None)           # debugging aid: called after every backward step
                for i, step in enumerate(reversed(tape)):
                    step()
                    if hook is not None:
                        hook(i, step)
            except BaseException:
                self._red = self._scat = None
                raise
            self._reduce_flush()                                   # every collected parameter-gradient finish, d_proj's slices included
            dpb = cast_rows(d_proj, PT, B, PT, dev)
            from .unet import colsum_f32
            colsum_f32(d_proj, B, PT, PT, self.grads.data_ptr() + 4 * self.proj_b_off, self._cs)
            bgemm(dpb, semb, PT, ted, B, lda=PT, ldb=ted, a_t=True, b_t=True, c_f32=self.grads.data_ptr() + 4 * self.proj_w_off, ldc=ted)
            d_semb = torch.empty(B, ted, dtype=torch.float32, device=dev)
            bgemm(dpb, self.params_bf16.data_ptr() + 2 * self.proj_w_off, B, ted, PT, lda=PT, ldb=ted, b_t=True, c_f32=d_semb, ldc=ted)
            d_emb = torch.empty(B, ted, dtype=torch.float32, device=dev)
            check(L.sfron_silu_bwd(ptr(d_semb), ptr(emb), B * ted, None, ptr(d_emb), stream_ptr()), "silu_bwd")
            d_e0s = te2_b(d_emb, ted)
            d_e0 = torch.empty(B, ted, dtype=torch.float32, device=dev)
            check(L.sfron_silu_bwd(ptr(d_e0s), ptr(e0), B * ted, None, ptr(d_e0), stream_ptr()), "silu_bwd")
            te0_b(d_e0, ted, want_dx=False)
        return out, backward

    def _anchor(self):
        if not hasattr(self, "_anc"):
            self._anc = torch.zeros((), device=self.device_, requires_grad=True)
        return self._anc

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        assert y is None, "the v1 UNet is not class-conditional"
        if torch.is_grad_enabled():
            return _SDFn.apply(self._anchor(), self, x, timesteps, context)
        out, _ = self._run(x, timesteps, context, need_grad=False)
        return out


class _SDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, context):
        keep, model.wgrad_filter = model.wgrad_filter, None        # the autograd surface always forms every gradient
        try:
            out, bwd = model._run(x, t, context, need_grad=True)
        finally:
            model.wgrad_filter = keep
        ctx.model, ctx.bwd = model, bwd
        return out

    @staticmethod
    def backward(ctx, d_out):
        ctx.bwd(d_out)
        ctx.model.publish_grads()
        ctx.bwd = None
        return None, None, None, None, None
