"""GPU parity: adaLN elementwise kernels, conditioning/layout kernels and fused attention vs plain torch fp32
(on the same bf16-rounded inputs), through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("B,T,D", [(2, 64, 128), (3, 16, 144), (3, 6, 144), (2, 12, 64), (2, 256, 1152)])
def test_ln_modulate_fwd_bwd(B, T, D):
    from sfron import ops
    gen = torch.Generator().manual_seed(B * T + D)
    M = B * T
    x = torch.randn(M, D, generator=gen) * 2 + 0.3
    mod = torch.randn(B, 6 * D, generator=gen) * 0.5
    shift, scale = mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D]
    xr = x.clone().requires_grad_(True)
    sh, sc = shift.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), eps=1e-6).view(B, T, D) * (1 + sc.unsqueeze(1)) + sh.unsqueeze(1)
    dout = (torch.randn(M, D, generator=gen) * 0.1).to(torch.bfloat16)
    ref.backward(dout.float().view(B, T, D))
    xd, md = x.to(DEV), mod.to(DEV)
    out, mean, rstd = ops.ln_modulate_fwd(xd, md[:, 3 * D:], md[:, 4 * D:], 6 * D, T)
    close(out, ref.view(M, D), 1e-2, 1e-2)
    dx = torch.full((M, D), 0.5, device=DEV)
    ps, pc = ops.ln_modulate_bwd(dout.to(DEV), xd, mean, rstd, md[:, 4 * D:], 6 * D, T, dx, accumulate=True)
    close(dx - 0.5, xr.grad, 1e-3, 2e-5)
    dmod = torch.zeros(B, 6 * D, device=DEV)
    per = T // ops.rows_per_chunk(T)
    ops.reduce_chunks(ps, B, per, D, dmod[:, 3 * D:], 6 * D)
    ops.reduce_chunks(pc, B, per, D, dmod[:, 4 * D:], 6 * D)
    close(dmod[:, 3 * D:4 * D], sh.grad, 1e-4, 1e-4)
    close(dmod[:, 4 * D:5 * D], sc.grad, 1e-4, 2e-4)
    dx2 = torch.empty(M, D, device=DEV)
    ops.ln_modulate_bwd(dout.to(DEV), xd, mean, rstd, md[:, 4 * D:], 6 * D, T, dx2, accumulate=False)
    close(dx2, xr.grad, 1e-3, 2e-5)


@pytest.mark.parametrize("B,T,D", [(2, 64, 128), (2, 256, 1152)])
def test_gate_bwd_and_reductions(B, T, D):
    from sfron import ops
    gen = torch.Generator().manual_seed(D)
    M = B * T
    dy = torch.randn(M, D, generator=gen) * 0.1
    br = (torch.randn(M, D, generator=gen)).to(torch.bfloat16)
    mod = torch.randn(B, 6 * D, generator=gen)
    gate = mod[:, 2 * D:3 * D]
    d_branch, p_gate, p_dy = ops.gate_bwd(dy.to(DEV), br.to(DEV), mod.to(DEV)[:, 2 * D:], 6 * D, T)
    close(d_branch, (dy.view(B, T, D) * gate.unsqueeze(1)).reshape(M, D), 1e-2, 1e-3)
    per = T // ops.rows_per_chunk(T)
    dgate = torch.empty(B, D, device=DEV)
    ops.reduce_chunks(p_gate, B, per, D, dgate, D)
    close(dgate, (dy * br.float()).view(B, T, D).sum(1), 1e-4, 1e-4)
    dbias = torch.empty(D, device=DEV)
    ops.weighted_reduce(p_dy, B, per, D, mod.to(DEV)[:, 2 * D:], 6 * D, dbias)
    close(dbias, (dy.view(B, T, D) * gate.unsqueeze(1)).sum((0, 1)), 1e-4, 1e-4)
    # colsum, both dtypes
    close(ops.colsum(br.to(DEV)), br.float().sum(0), 1e-4, 1e-3)
    close(ops.colsum(dy.to(DEV)), dy.sum(0), 1e-4, 1e-4)


@pytest.mark.parametrize("B,T,D", [(2, 64, 128), (3, 6, 144), (2, 12, 64), (3, 24, 144), (2, 256, 1152)])
def test_ln_gate_bwd_fused_equals_separate(B, T, D):
    """The fused LN-backward + gate-backward pass is bit-identical to the two separate kernels (same per-row arithmetic,
    same fixed-order partial sums), for every chunking plan (16 / 8 / 4 rows combined, 2 rows per wave)."""
    from sfron import ops
    gen = torch.Generator().manual_seed(B * T * D)
    M = B * T
    x = (torch.randn(M, D, generator=gen) * 2 + 0.3).to(DEV)
    mod = (torch.randn(B, 6 * D, generator=gen) * 0.5).to(DEV)
    dout = (torch.randn(M, D, generator=gen) * 0.1).to(torch.bfloat16).to(DEV)
    br = torch.randn(M, D, generator=gen).to(torch.bfloat16).to(DEV)
    dx0 = (torch.randn(M, D, generator=gen) * 0.1).to(DEV)
    _, mean, rstd = ops.ln_modulate_fwd(x, mod[:, 3 * D:], mod[:, 4 * D:], 6 * D, T)
    for acc in (True, False):
        dxa = dx0.clone()
        ps, pc = ops.ln_modulate_bwd(dout, x, mean, rstd, mod[:, 4 * D:], 6 * D, T, dxa, accumulate=acc)
        db, pg, pd = ops.gate_bwd(dxa, br, mod[:, 2 * D:], 6 * D, T)
        dxb = dx0.clone()
        ps2, pc2, db2, pg2, pd2 = ops.ln_gate_bwd(dout, x, mean, rstd, mod[:, 4 * D:], 6 * D, T, dxb, acc, br, mod[:, 2 * D:], 6 * D)
        for u, v in ((dxa, dxb), (ps, ps2), (pc, pc2), (db, db2), (pg, pg2), (pd, pd2)):
            assert torch.equal(u, v)
        assert ps.shape[0] == M // ops.rows_per_chunk(T)


def test_conditioning_and_layout():
    from sfron import ops
    from oracle import dit_ref
    gen = torch.Generator().manual_seed(1)
    n, D, ncls = 5, 128, 10
    t = torch.tensor([0, 1, 500, 999, 37])
    emb = ops.timestep_embed(t.to(DEV), 256)
    close(emb, dit_ref.TimestepEmbedder.timestep_embedding(t, 256), 1e-2, 1e-2)
    t_emb = torch.randn(n, D, generator=gen)
    table = torch.randn(ncls + 1, D, generator=gen)
    y = torch.tensor([3, 3, 9, 0, 3])
    drop = torch.tensor([0, 1, 0, 0, 1], dtype=torch.uint8)
    c, sc = ops.cond_fwd(t_emb.to(DEV), table.to(DEV), y.to(DEV), drop.to(DEV), ncls)
    lab = torch.where(drop.bool(), ncls, y)
    c_ref = (t_emb + table[lab])
    close(c, c_ref, 1e-6, 1e-6)
    close(sc, F.silu(c_ref), 1e-2, 1e-2)
    dsc = torch.randn(n, D, generator=gen)
    cr = c_ref.clone().requires_grad_(True)
    F.silu(cr).backward(dsc)
    dtab = torch.zeros(ncls + 1, D, device=DEV)
    d_c = ops.cond_bwd(dsc.to(DEV), c, y.to(DEV), drop.to(DEV), ncls, dtab)
    close(d_c, cr.grad, 1e-4, 1e-5)
    want = torch.zeros(ncls + 1, D).index_add_(0, lab, cr.grad)
    close(dtab, want, 1e-4, 1e-5)
    a, b = ops.silu_bwd(dsc.to(DEV), c, True, True)
    close(b, cr.grad, 1e-4, 1e-5)
    close(ops.silu_fwd(c), F.silu(c_ref), 1e-2, 1e-2)
    # patchify == Conv2d(k=s=p) im2col; unpatchify == reference einsum
    img = torch.randn(2, 4, 8, 8, generator=gen)
    conv = torch.nn.Conv2d(4, 16, 2, 2)
    rows = ops.patchify(img.to(DEV), 2).float().cpu()
    ref = conv(img.to(torch.bfloat16).float()).flatten(2).transpose(1, 2).reshape(-1, 16)
    close(rows @ conv.weight.view(16, -1).t() + conv.bias, ref, 1e-5, 1e-5)
    m = dit_ref.DiT(input_size=8, patch_size=2, hidden_size=64, depth=1, num_heads=1, num_classes=2)
    tok = torch.randn(2, 16, 2 * 2 * 8, generator=gen)
    close(ops.unpatchify(tok.view(32, 32).to(DEV), 2, 8, 8, 8, 2), m.unpatchify(tok), 0, 0)
    # chan_last patchify is the exact adjoint layout of unpatchify
    im8 = torch.randn(2, 8, 8, 8, generator=gen)
    r2 = ops.patchify(im8.to(DEV), 2, chan_last=True)
    close(ops.unpatchify(r2.float(), 2, 8, 8, 8, 2), im8.to(torch.bfloat16).float(), 0, 0)


@pytest.mark.parametrize("B,T,H,hd", [(2, 256, 16, 72), (1, 128, 3, 72), (1, 256, 4, 64), (1, 1024, 8, 40), (2, 256, 4, 80), (1, 384, 2, 48), (3, 512, 2, 40)])
def test_attention_forward_eight_wave_form_is_bit_identical(B, T, H, hd):
    """k_attn_fwd8 (round 4: eight waves of 16 query rows per workgroup, sixteen waves per CU) and the whole-head form (eight waves of 32
    rows: form 16, taken where the sequence is a multiple of 256 rows) against the four-wave kernel: the same products and the same softmax
    arithmetic per row -> the same bits in O and LSE.  Round 6: form 2 -- where T is a multiple of 256 one workgroup walks both 128-row query
    blocks of a head as one chunk stream (k_attn_fwd<..., NIT = 2>) -- and form 0 (the rule) are in the comparison too."""
    from sfron import _lib, ops
    L = _lib.lib()
    gen = torch.Generator().manual_seed(T * H + hd)
    D = H * hd
    qkv = (torch.randn(B * T, 3 * D, generator=gen) * 1.5).to(torch.bfloat16).to(DEV)
    res = []
    for form in (4, 8, 16, 2, 0):
        old = L.sfron_attn_fwd_form(form)
        try:
            o, lse = ops.attn_fwd(qkv, B, T, H, hd)
            torch.cuda.synchronize()
            res.append((o.clone(), lse.clone()))
        finally:
            L.sfron_attn_fwd_form(old)
    assert torch.isfinite(res[0][0].float()).all() and torch.isfinite(res[0][1]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    assert torch.equal(res[0][0], res[3][0]) and torch.equal(res[0][1], res[3][1])
    assert torch.equal(res[0][0], res[4][0]) and torch.equal(res[0][1], res[4][1])


@pytest.mark.parametrize("B,T,H,hd", [(2, 64, 2, 64), (1, 128, 3, 72), (2, 256, 2, 72), (1, 256, 4, 64), (1, 192, 1, 64),
                                      (1, 1024, 8, 40), (2, 256, 4, 80), (1, 320, 2, 40), (1, 128, 2, 48)])
def test_attention_fwd_bwd(B, T, H, hd):
    from sfron import ops
    gen = torch.Generator().manual_seed(T + H + hd)
    D = H * hd
    qkv = (torch.randn(B * T, 3 * D, generator=gen) * 1.5).to(torch.bfloat16)
    qkv[0, :hd] *= 6.0            # a spiky query row: exercises the online-softmax rescale branch
    qkv[5, D:D + hd] *= 6.0       # and a spiky key
    d_o = (torch.randn(B * T, D, generator=gen) * 0.2).to(torch.bfloat16)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, D)
    ref.backward(d_o.float())
    o, lse = ops.attn_fwd(qkv.to(DEV), B, T, H, hd)
    close(o, ref, 2e-2, 2e-2)
    lse_ref = torch.logsumexp((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1)
    close(lse, lse_ref, 1e-4, 1e-3)
    dqkv = ops.attn_bwd(qkv.to(DEV), o, d_o.to(DEV), lse, B, T, H, hd)
    g = x.grad
    scale = g.abs().max().item()
    close(dqkv, g, 5e-2, 2e-2 * scale)
    # aggregate error must be far below bf16 rounding noise of the reference itself
    err = (dqkv.float().cpu() - g).norm() / g.norm()
    assert err < 2e-2, err


@pytest.mark.parametrize("B,T,H,hd", [(2, 256, 2, 64), (3, 256, 3, 72), (2, 128, 2, 72), (1, 128, 4, 64), (32, 256, 16, 72)])
def test_attention_backward_fused_vs_two_kernel_and_torch(B, T, H, hd):
    """The one-workgroup-per-(batch, head) backward (T = 128 / 256: S and dP formed once, dQ from the dS image in LDS) against
    the two-kernel form (dQ kernel + dK/dV kernel, forced through sfron_attn_bwd_form) and against torch autograd; the last
    case is the DiT-XL/2 batch-32 launch itself (512 workgroups).  Run twice: bitwise reproducible (no float atomics)."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(T * 7 + H + hd)
    D = H * hd
    qkv = (torch.randn(B * T, 3 * D, generator=gen) * 1.2).to(torch.bfloat16)
    qkv[3, :hd] *= 5.0
    qkv[7, D:D + hd] *= 5.0
    d_o = (torch.randn(B * T, D, generator=gen) * 0.2).to(torch.bfloat16)
    qd, dd = qkv.to(DEV), d_o.to(DEV)
    o, lse = ops.attn_fwd(qd, B, T, H, hd)
    L = _lib.lib()
    fused = ops.attn_bwd(qd, o, dd, lse, B, T, H, hd)
    again = ops.attn_bwd(qd, o, dd, lse, B, T, H, hd)
    assert torch.equal(fused, again)
    old = L.sfron_attn_bwd_form(2)
    try:
        two = ops.attn_bwd(qd, o, dd, lse, B, T, H, hd)
    finally:
        L.sfron_attn_bwd_form(old)
    # reference: torch autograd in fp32 on the GPU (same bf16 inputs)
    x = qd.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    (att @ v).transpose(1, 2).reshape(B * T, D).backward(dd.float())
    g = x.grad
    for name, got in (("fused", fused), ("two-kernel", two)):
        err = ((got.float() - g).norm() / g.norm()).item()
        assert err < 2e-2, (name, err)
        for w, nm in ((0, "dQ"), (1, "dK"), (2, "dV")):
            e = ((got.float()[:, w * D:(w + 1) * D] - g[:, w * D:(w + 1) * D]).norm() / g[:, w * D:(w + 1) * D].norm()).item()
            assert e < 2e-2, (name, nm, e)
    # the two forms round P and dS to bf16 at the same points: they agree far inside the bound against fp32
    assert ((fused.float() - two.float()).norm() / two.float().norm()).item() < 6e-3


@pytest.mark.parametrize("B,T,H,hd", [(2, 256, 2, 64), (3, 256, 3, 72), (2, 128, 2, 72), (2, 256, 4, 40), (32, 256, 16, 72)])
def test_attention_backward_with_qkv_bias_partials(B, T, H, hd):
    """sfron_attn_bwd_bias: the fused backward also leaves one partial row per sample of the qkv.bias gradient (token sums of
    dQ | dK | dV in fp32 before the bf16 rounding).  dqkv is bit-identical to the plain call; the rows match the column sums of
    the fp32 autograd gradient; two runs agree bit for bit (fixed order, no atomics)."""
    from sfron import ops, _lib
    assert _lib.lib().sfron_attn_bwd_bias_supported(T) == 1 and _lib.lib().sfron_attn_bwd_bias_supported(192) == 0
    gen = torch.Generator().manual_seed(T * 11 + H + hd)
    D = H * hd
    qd = (torch.randn(B * T, 3 * D, generator=gen) * 1.2).to(torch.bfloat16).to(DEV)
    dd = (torch.randn(B * T, D, generator=gen) * 0.2 + 0.05).to(torch.bfloat16).to(DEV)      # non-zero mean: sums do not cancel
    o, lse = ops.attn_fwd(qd, B, T, H, hd)
    plain = ops.attn_bwd(qd, o, dd, lse, B, T, H, hd)
    dqkv, part = ops.attn_bwd_bias(qd, o, dd, lse, B, T, H, hd)
    dqkv2, part2 = ops.attn_bwd_bias(qd, o, dd, lse, B, T, H, hd)
    assert torch.equal(dqkv, plain) and torch.equal(dqkv, dqkv2) and torch.equal(part, part2)
    x = qd.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1)
    (att @ v).transpose(1, 2).reshape(B * T, D).backward(dd.float())
    want = x.grad.view(B, T, 3 * D).sum(1)
    assert torch.isfinite(part).all()
    for w, nm in ((0, "dQ"), (1, "dK"), (2, "dV")):
        a, b = part[:, w * D:(w + 1) * D], want[:, w * D:(w + 1) * D]
        # dV sums do not cancel (tight); dQ / dK rows sum to ~0 over the softmax, so they are compared on the scale of the column norms
        scale = x.grad.view(B, T, 3 * D)[:, :, w * D:(w + 1) * D].abs().sum(1)
        assert ((a - b).abs() <= 2e-2 * scale + 1e-6).all(), (nm, ((a - b).abs() / (scale + 1e-9)).max().item())
    # against the column sums of the bf16 output itself (what the old column-sum launch formed): same numbers up to the rounding
    old = plain.float().view(B, T, 3 * D).sum(1)
    assert ((part - old).abs() <= 2.0 ** -8 * plain.float().abs().view(B, T, 3 * D).sum(1) + 1e-6).all()


@pytest.mark.parametrize("groups,per,D", [(1, 512, 128), (3, 64, 200), (1, 100, 1280), (2, 63, 128), (1, 7, 4608), (64, 96, 320)])
def test_reduce_chunks_both_forms(groups, per, D):
    """out[g][c] (+)= sum_j P[g][j][c]: the one-thread-per-column form and the 16-waves-per-64-columns form that many partials of few
    columns take (csrc/norm.hip reduce_chunks_wide: per_group >= 64 and at most 512 workgroups); each sums in a fixed order -- the same
    bits on every call -- and agrees with a float64 sum to fp32 rounding of the running sums."""
    from sfron import ops
    gen = torch.Generator().manual_seed(groups * per + D)
    P = torch.randn(groups, per, D, generator=gen).to(DEV)
    ref = P.double().sum(1)
    ld = D + 4
    for acc in (False, True):
        out = torch.full((groups, ld), 2.0, dtype=torch.float32, device=DEV)
        ops.reduce_chunks(P, groups, per, D, out, ld, accumulate=acc)
        again = torch.full((groups, ld), 2.0, dtype=torch.float32, device=DEV)
        ops.reduce_chunks(P, groups, per, D, again, ld, accumulate=acc)
        assert torch.equal(out, again)
        assert float((out[:, D:] - 2.0).abs().max()) == 0.0                      # columns beyond D untouched
        want = ref + (2.0 if acc else 0.0)
        close(out[:, :D], want, rtol=1e-5, atol=1e-5 * per ** 0.5)


@pytest.mark.parametrize("D,per,groups,layers", [(1152, 16, 4, 2), (2304, 4, 3, 1), (4104, 5, 2, 1)])
def test_reduce_slots_any_width(D, per, groups, layers):
    """The one-launch token reduction of a backward pass (csrc/norm.hip k_reduce_slots): slot (layer, kind, buf) holds [groups * per][D]
    partials; out[kind, buf][layer][b][c] = sum_j partial[b * per + j][c] in index order.  Rows wider than 2048 columns (no registry model has
    them) loop over column groups instead of failing in the middle of a backward pass."""
    import ctypes
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    gen = torch.Generator().manual_seed(D + per)
    n_slots = layers * 8
    parts = torch.randn(n_slots, groups * per, D, generator=gen).to(DEV)
    ld = D + 8
    outs = [torch.full((layers, groups, ld), 3.0, dtype=torch.float32, device=DEV) for _ in range(8)]
    base = (ctypes.c_void_p * 8)(*[o.data_ptr() for o in outs])
    lstride = (ctypes.c_long * 8)(*[groups * ld] * 8)
    lds = (ctypes.c_int * 8)(*[ld] * 8)
    check(_lib.lib().sfron_reduce_slots(ptr(parts), groups * per * D, n_slots, groups, per, D, base, lstride, lds, stream_ptr()), "reduce_slots")
    torch.cuda.synchronize()
    want = parts.view(layers, 8, groups, per, D).double().sum(3)
    for kb in range(8):
        close(outs[kb][:, :, :D], want[:, kb], rtol=1e-5, atol=1e-5 * per ** 0.5)
        assert float((outs[kb][:, :, D:] - 3.0).abs().max()) == 0.0


def test_attention_rejects_head_widths_without_a_kernel():
    """Head widths 88 / 96 at T >= 64 would need a sixth output d-tile: rounds 1-3 accepted them and left columns 80.. unwritten (found in
    round 4 by the bit-identity test above).  They are refused now; the LDM UNet's wider heads (160) take the batched-GEMM path."""
    from sfron import ops
    from sfron._lib import SfronError
    for hd in (88, 96):
        qkv = torch.zeros(128, 3 * hd, dtype=torch.bfloat16, device=DEV)
        with pytest.raises(SfronError):
            ops.attn_fwd(qkv, 1, 128, 1, hd)
