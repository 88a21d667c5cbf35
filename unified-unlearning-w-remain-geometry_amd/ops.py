"""Thin Python wrappers over single C-ABI kernels (used by tests and by the autograd-facing modules).
The whole-model engine (engine.py) drives the same entry points."""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def gemm(A, B, M, N, K, *, a_t=False, b_t=False, epilogue=_lib.EPI_BF16, alpha=1.0, bias=None, c_bf16=None,
         c_f32=None, aux=None, gate=None, pos=None, tokens=1, accumulate=False, resid=None, split_k=1, split_stride=0, tile_hint=0,
         lda=None, ldb=None, ldc_bf16=None, ldc_f32=None, ldaux=None, ldgate=None, a_rowsum=None, col_partials=None,
         sumsq_partials=None, sumsq_mask=None):
    """C[M,N] = alpha * op(A) op(B) with the epilogues of include/sfron.h.  A/B are bf16 2-D tensors (or views
    described by explicit leading dimensions)."""
    d = _lib.GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else A.stride(0)
    d.ldb = ldb if ldb is not None else B.stride(0)
    d.a_transposed, d.b_transposed = int(a_t), int(b_t)
    d.epilogue, d.alpha = epilogue, alpha
    d.bias = ptr(bias)
    if c_bf16 is not None:
        d.c_bf16, d.ldc_bf16 = c_bf16.data_ptr(), ldc_bf16 if ldc_bf16 is not None else c_bf16.stride(0)
    if c_f32 is not None:
        d.c_f32, d.ldc_f32 = c_f32.data_ptr(), ldc_f32 if ldc_f32 is not None else c_f32.stride(0)
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), ldaux if ldaux is not None else aux.stride(0)
    if gate is not None:
        d.gate, d.ldgate = gate.data_ptr(), ldgate if ldgate is not None else gate.stride(0)
    if pos is not None:
        d.pos = pos.data_ptr()
    d.tokens, d.accumulate = tokens, int(accumulate)
    if resid is not None:
        d.resid = resid.data_ptr()
    d.split_k, d.split_stride, d.tile_hint = split_k, split_stride, tile_hint
    ws = None
    if a_rowsum is not None:
        import torch as _t
        ws = _t.empty((N // 192) * M, dtype=_t.float32, device=a_rowsum.device)      # partial sums per tile column
        d.a_rowsum, d.rowsum_ws = a_rowsum.data_ptr(), ws.data_ptr()
    if col_partials is not None:
        d.col_partials = col_partials.data_ptr()
    if sumsq_partials is not None:         # weight-gradient layout: masked sums of squares of the output tiles (sfron_gemm_sumsq_partials)
        d.sumsq_partials, d.sumsq_mask = sumsq_partials.data_ptr(), ptr(sumsq_mask)
    for t in (A, B, c_bf16, c_f32, aux, gate, pos, a_rowsum, col_partials, sumsq_partials, sumsq_mask):
        if t is not None and not t.is_cuda:
            raise _lib.SfronError("sfron ops need GPU tensors (no CPU fallback)")
    check(_lib.lib().sfron_gemm_bf16(ctypes.byref(d), stream_ptr()), "gemm_bf16")


def _L():
    return _lib.lib()


def rows_per_chunk(tokens):
    r = _L().sfron_rows_per_chunk(int(tokens))
    if r == 0:
        raise _lib.SfronError(f"tokens per sample = {tokens} must be a multiple of 4")
    return r


def ln_modulate_fwd(x, shift, scale, ldmod, tokens):
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(_L().sfron_ln_modulate_fwd(ptr(x), shift.data_ptr(), scale.data_ptr(), ldmod, tokens, M, D, ptr(out), ptr(mean),
                                     ptr(rstd), stream_ptr()), "ln_modulate_fwd")
    return out, mean, rstd


def ln_modulate_bwd(d_out, x, mean, rstd, scale, ldmod, tokens, dx, accumulate):
    M, D = x.shape
    nch = M // rows_per_chunk(tokens)
    p_shift = torch.empty(nch, D, dtype=torch.float32, device=x.device)
    p_scale = torch.empty_like(p_shift)
    check(_L().sfron_ln_modulate_bwd(ptr(d_out), ptr(x), ptr(mean), ptr(rstd), scale.data_ptr(), ldmod, tokens, M, D, ptr(dx),
                                     int(accumulate), ptr(p_shift), ptr(p_scale), stream_ptr()), "ln_modulate_bwd")
    return p_shift, p_scale


def gate_bwd(dy, branch, gate, ldmod, tokens):
    M, D = dy.shape
    nch = M // rows_per_chunk(tokens)
    d_branch = torch.empty(M, D, dtype=torch.bfloat16, device=dy.device)
    p_gate = torch.empty(nch, D, dtype=torch.float32, device=dy.device)
    p_dy = torch.empty_like(p_gate)
    check(_L().sfron_gate_bwd(ptr(dy), ptr(branch), gate.data_ptr(), ldmod, tokens, M, D, ptr(d_branch), ptr(p_gate), ptr(p_dy),
                              stream_ptr()), "gate_bwd")
    return d_branch, p_gate, p_dy


def ln_gate_bwd(d_out, x, mean, rstd, scale, ldmod, tokens, dx, accumulate, branch, gate, ldgate):
    """ln_modulate_bwd then gate_bwd of the next branch on the updated dx rows, one pass."""
    M, D = x.shape
    nch = M // rows_per_chunk(tokens)
    parts = [torch.empty(nch, D, dtype=torch.float32, device=x.device) for _ in range(4)]
    d_branch = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    check(_L().sfron_ln_gate_bwd(ptr(d_out), ptr(x), ptr(mean), ptr(rstd), scale.data_ptr(), ldmod, tokens, M, D, ptr(dx),
                                 int(accumulate), ptr(parts[0]), ptr(parts[1]), ptr(branch), gate.data_ptr(), ldgate,
                                 ptr(d_branch), ptr(parts[2]), ptr(parts[3]), stream_ptr()), "ln_gate_bwd")
    return parts[0], parts[1], d_branch, parts[2], parts[3]


def reduce_chunks(partials, groups, per_group, D, out, ldout, accumulate=False):
    check(_L().sfron_reduce_chunks(ptr(partials), groups, per_group, D, out.data_ptr(), ldout, int(accumulate), stream_ptr()),
          "reduce_chunks")


def weighted_reduce(partials, groups, per_group, D, w, ldw, out):
    check(_L().sfron_weighted_reduce(ptr(partials), groups, per_group, D, w.data_ptr(), ldw, out.data_ptr(), stream_ptr()),
          "weighted_reduce")


def colsum(X, out=None):
    M, N = X.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    partials = torch.empty(64, N, dtype=torch.float32, device=X.device)
    check(_L().sfron_colsum(ptr(X), int(X.dtype == torch.bfloat16), M, N, X.stride(0), ptr(partials), 64, out.data_ptr(),
                            stream_ptr()), "colsum")
    return out


def timestep_embed(t, dim=256):
    out = torch.empty(t.shape[0], dim, dtype=torch.bfloat16, device=t.device)
    check(_L().sfron_timestep_embed(ptr(t), t.shape[0], dim, ptr(out), dim, stream_ptr()), "timestep_embed")
    return out


def silu_fwd(x):
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_L().sfron_silu_fwd(ptr(x), x.numel(), ptr(y), stream_ptr()), "silu_fwd")
    return y


def silu_bwd(dy, x, want_bf16=True, want_f32=False):
    a = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    b = torch.empty_like(x) if want_f32 else None
    check(_L().sfron_silu_bwd(ptr(dy), ptr(x), x.numel(), ptr(a), ptr(b), stream_ptr()), "silu_bwd")
    return a, b


def cond_fwd(t_emb, table, y, drop, num_classes):
    n, D = t_emb.shape
    c = torch.empty_like(t_emb)
    sc = torch.empty(n, D, dtype=torch.bfloat16, device=t_emb.device)
    check(_L().sfron_cond_fwd(ptr(t_emb), ptr(table), ptr(y), ptr(drop), num_classes, n, D, ptr(c), ptr(sc), stream_ptr()),
          "cond_fwd")
    return c, sc


def cond_bwd(d_silu_c, c, y, drop, num_classes, d_table):
    n, D = c.shape
    d_c = torch.empty_like(c)
    check(_L().sfron_cond_bwd(ptr(d_silu_c), ptr(c), ptr(y), ptr(drop), num_classes, n, D, ptr(d_c), ptr(d_table),
                              stream_ptr()), "cond_bwd")
    return d_c


def patchify(img, p, chan_last=False):
    n, C, H, W = img.shape
    T, K = (H // p) * (W // p), C * p * p
    rows = torch.empty(n * T, K, dtype=torch.bfloat16, device=img.device)
    check(_L().sfron_patchify(ptr(img), n, C, H, W, p, int(chan_last), ptr(rows), K, stream_ptr()), "patchify")
    return rows


def unpatchify(rows, n, C, H, W, p):
    img = torch.empty(n, C, H, W, dtype=torch.float32, device=rows.device)
    check(_L().sfron_unpatchify(ptr(rows), rows.stride(0), n, C, H, W, p, ptr(img), stream_ptr()), "unpatchify")
    return img


def attn_fwd(qkv, B, T, H, hd):
    o = torch.empty(B * T, H * hd, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    check(_L().sfron_attn_fwd(ptr(qkv), ptr(o), ptr(lse), B, T, H, hd, stream_ptr()), "attn_fwd")
    return o, lse


def attn_bwd(qkv, o, d_o, lse, B, T, H, hd):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * T, dtype=torch.float32, device=qkv.device)
    check(_L().sfron_attn_bwd(ptr(qkv), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dqkv), B, T, H, hd, stream_ptr()), "attn_bwd")
    return dqkv


def attn_bwd_bias(qkv, o, d_o, lse, B, T, H, hd):
    """Attention backward that also returns the qkv.bias gradient partials [B][3*H*hd] (token sums per sample, fp32)."""
    dqkv = torch.empty_like(qkv)
    part = torch.empty(B, 3 * H * hd, dtype=torch.float32, device=qkv.device)
    check(_L().sfron_attn_bwd_bias(ptr(qkv), ptr(o), ptr(d_o), ptr(lse), ptr(dqkv), ptr(part), B, T, H, hd, stream_ptr()), "attn_bwd_bias")
    return dqkv, part
