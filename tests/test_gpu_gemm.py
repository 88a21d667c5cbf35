"""GPU parity: bf16 MFMA GEMM (all three operand layouts + epilogues) vs torch fp32 on the same bf16 inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(torch.bfloat16)


def test_fragment_layout_exact_integers():
    """A = I with an ASYMMETRIC integer B catches swapped row/col maps and wrong transposed-read addressing:
    all values are small integers, exactly representable, so the result must be bit-exact."""
    from sfron import ops, _lib
    M = N = K = 128
    eye = torch.eye(K, dtype=torch.bfloat16, device=DEV)
    r = torch.arange(N, dtype=torch.float32).view(N, 1)
    c = torch.arange(K, dtype=torch.float32).view(1, K)
    Bm = ((3 * r + 5 * c) % 61 - 30).to(torch.bfloat16).to(DEV)          # [N, K], asymmetric
    # forward layout: C = A · B^T, A = I  ->  C = B^T
    C = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(eye, Bm, M, N, K, epilogue=_lib.EPI_F32, c_f32=C)
    assert torch.equal(C, Bm.float().t())
    # dgrad layout: C = A · B (B stored [K][N]), A = I -> C = B
    ops.gemm(eye, Bm, M, N, K, b_t=True, epilogue=_lib.EPI_F32, c_f32=C)
    assert torch.equal(C, Bm.float())
    # wgrad layout: C = A^T · B with A stored [K][M]; A = asymmetric, B = I -> C = A^T
    ops.gemm(Bm, eye, M, N, K, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C)
    assert torch.equal(C, Bm.float().t())
    # wgrad with A = I, B asymmetric -> C = B
    ops.gemm(eye, Bm, M, N, K, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C)
    assert torch.equal(C, Bm.float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (96, 160, 72), (32, 6912, 1152), (8192, 32, 1152),
                                   (1024, 1152, 1152)])
def test_forward_layout(M, N, K):
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M + N + K)
    A, B = _rand((M, K), gen), _rand((N, K), gen, 0.05)
    bias = torch.randn(N, generator=gen)
    want = A.float() @ B.float().t() + bias
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, bias=bias.to(DEV), c_bf16=Cb)
    np.testing.assert_allclose(Cb.float().cpu().numpy(), want.numpy(), rtol=1e-2, atol=2e-2)
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, bias=bias.to(DEV), epilogue=_lib.EPI_F32, c_f32=Cf)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (512, 1152, 3456), (96, 72, 32), (8192, 1152, 32)])
def test_dgrad_layout(M, N, K):
    """dX[M, N] = dY[M, K] · W[K, N]  (contraction K = forward out-features; W row-major [K][N])."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M * 3 + N + K)
    dY, W = _rand((M, K), gen), _rand((K, N), gen, 0.05)
    want = dY.float() @ W.float()
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(dY.to(DEV), W.to(DEV), M, N, K, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cf)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 256), (1152, 1152, 2048), (3456, 1152, 512), (6912, 1152, 32), (32, 1152, 1024),
                                   (72, 64, 40)])
def test_wgrad_layout(M, N, K):
    """dW[M, N] = dY[K, M]^T · X[K, N]  (contraction K = token rows)."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    dY, X = _rand((K, M), gen, 0.1), _rand((K, N), gen)
    want = dY.float().t() @ X.float()
    Cf = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm(dY.to(DEV), X.to(DEV), M, N, K, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cf)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)
    ops.gemm(dY.to(DEV), X.to(DEV), M, N, K, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cf, accumulate=True)
    np.testing.assert_allclose(Cf.cpu().numpy(), 2 * want.numpy(), rtol=4e-4, atol=4e-4 * K ** 0.5)


def test_epilogues():
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(5)
    Bsz, T, D, Hd = 3, 64, 128, 256
    M = Bsz * T
    X, W1, W2 = _rand((M, D), gen), _rand((Hd, D), gen, 0.08), _rand((D, Hd), gen, 0.08)
    b1, b2 = torch.randn(Hd, generator=gen) * 0.1, torch.randn(D, generator=gen) * 0.1
    # GELU: aux = pre-activation, c = gelu_tanh
    h_pre = X.float() @ W1.float().t() + b1
    aux = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    H = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    ops.gemm(X.to(DEV), W1.to(DEV), M, Hd, D, epilogue=_lib.EPI_GELU, bias=b1.to(DEV), c_bf16=H, aux=aux)
    np.testing.assert_allclose(aux.float().cpu().numpy(), h_pre.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(H.float().cpu().numpy(), torch.nn.functional.gelu(h_pre, approximate="tanh").numpy(),
                               rtol=1e-2, atol=1e-2)
    # gated residual: x += gate[b] * (H W2^T + b2), aux = branch output
    Hh = H.cpu()
    gate = torch.randn(Bsz, 6 * D, generator=gen)
    x0 = torch.randn(M, D, generator=gen)
    branch = Hh.float() @ W2.float().t() + b2
    want_x = x0 + gate[:, 2 * D:3 * D].repeat_interleave(T, dim=0) * branch
    xd = x0.to(DEV)
    a2 = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    gd = gate.to(DEV)
    ops.gemm(H, W2.to(DEV), M, D, Hd, epilogue=_lib.EPI_GATE_RES, bias=b2.to(DEV), c_f32=xd, aux=a2,
             gate=gd[:, 2 * D:], ldgate=6 * D, tokens=T)
    np.testing.assert_allclose(xd.cpu().numpy(), want_x.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(a2.float().cpu().numpy(), branch.numpy(), rtol=1e-2, atol=1e-2)
    # dGELU: d_hpre = (dA W2) * gelu'(h_pre)
    dA = _rand((M, D), gen, 0.1)
    hp = aux.cpu().float().requires_grad_(True)
    torch.nn.functional.gelu(hp, approximate="tanh").backward(dA.float() @ W2.float())
    dH = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dA.to(DEV), W2.to(DEV), M, Hd, D, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=dH, aux=aux)
    np.testing.assert_allclose(dH.float().cpu().numpy(), hp.grad.numpy(), rtol=1e-2, atol=2e-3)
    # POS: x = patches W^T + b + pos[row % T]
    pos = torch.randn(T, D, generator=gen)
    P, Wp = _rand((M, 64), gen), _rand((D, 64), gen, 0.1)
    xo = torch.empty(M, D, dtype=torch.float32, device=DEV)
    ops.gemm(P.to(DEV), Wp.to(DEV), M, D, 64, epilogue=_lib.EPI_POS, bias=b2.to(DEV), c_f32=xo, pos=pos.to(DEV), tokens=T)
    want = P.float() @ Wp.float().t() + b2 + pos.repeat(Bsz, 1)
    np.testing.assert_allclose(xo.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,F,D", [(512, 384, 256), (1024, 4608, 1152), (2048, 3072, 768)])
def test_gelu_pair_with_the_derivative_as_one_byte(M, F, D):
    """Round 6: SFRON_EPI_GELU_Q / SFRON_EPI_DGELU_Q (include/sfron.h) -- fc1 + GELU whose second output is gelu_tanh'(pre-activation) as ONE
    byte per element, code = round((g' + 0.15) * 196), and the fc2 dgrad that multiplies by the decoded byte.
      * the GELU output is bit-identical to SFRON_EPI_GELU's (same operations on the same accumulators);
      * every code decodes to within one half step (0.0026) + fp32 noise of gelu_tanh'(fp32 pre-activation);
      * the dgrad against torch autograd on the fp32 pre-activation: the same bound as the bf16-pre-activation form states in
        test_fused_epilogues, and against that form on the same inputs far inside it;
      * the dgrad's column partials (fc1.bias gradient) are offered as for SFRON_EPI_DGELU;
      * shapes off the 256 x 192 pipelined tile are refused, not mis-computed."""
    import torch.nn.functional as Fn
    from sfron import _lib, ops
    from sfron._lib import SfronError
    L = _lib.lib()
    assert L.sfron_gemm_gelu_q_supported(M, F, D) == 1 and L.sfron_gemm_gelu_q_supported(128, F, D) == 0 and L.sfron_gemm_gelu_q_supported(M, F + 64, D) == 0
    gen = torch.Generator().manual_seed(M + F)
    X, W1 = _rand((M, D), gen).to(DEV), _rand((F, D), gen, 0.06).to(DEV)
    b1 = (torch.randn(F, generator=gen) * 0.2).to(DEV)
    pre = X.float() @ W1.float().t() + b1                                    # fp32 pre-activation (what the accumulators hold)
    H0, aux0 = torch.empty(M, F, dtype=torch.bfloat16, device=DEV), torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    ops.gemm(X, W1, M, F, D, epilogue=_lib.EPI_GELU, bias=b1, c_bf16=H0, aux=aux0)
    H1, codes = torch.empty(M, F, dtype=torch.bfloat16, device=DEV), torch.full((M, F), 255, dtype=torch.uint8, device=DEV)
    ops.gemm(X, W1, M, F, D, epilogue=_lib.EPI_GELU_Q, bias=b1, c_bf16=H1, aux=codes)
    torch.cuda.synchronize()
    assert torch.equal(H0, H1)
    # repeatable bit for bit, also with another stream's traffic beside it (round 6: the EPI_GELUQ instantiation was the one in which hipcc
    # hoisted the epilogue's arithmetic above the padding behind the last asm MFMAs -- wrong elements whenever the MFMAs were delayed)
    side, junk = torch.cuda.Stream(), torch.empty(1 << 26, dtype=torch.float32, device=DEV)
    for _ in range(3):
        H2, codes2 = torch.empty_like(H1), torch.empty_like(codes)
        with torch.cuda.stream(side):
            junk.add_(1.0)
        ops.gemm(X, W1, M, F, D, epilogue=_lib.EPI_GELU_Q, bias=b1, c_bf16=H2, aux=codes2)
        torch.cuda.synchronize()
        assert torch.equal(H2, H1) and torch.equal(codes2, codes)
    p64 = pre.double().requires_grad_(True)
    Fn.gelu(p64, approximate="tanh").sum().backward()
    gprime = p64.grad.float()
    dec = codes.float() / 196.0 - 0.15
    assert int(codes.max()) <= 251 and float((dec - gprime).abs().max()) < 0.5 / 196.0 + 2e-4
    # backward: dX = (dY W2) * gelu'(pre)
    dY, W2 = _rand((M, D), gen, 0.1).to(DEV), _rand((D, F), gen, 0.06).to(DEV)
    want = (dY.float() @ W2.float()) * gprime
    dq = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    rows = L.sfron_gemm_dgelu_colpart_rows(M, F, D)
    part = torch.zeros(max(rows, 1), F, dtype=torch.float32, device=DEV)
    ops.gemm(dY, W2, M, F, D, b_t=True, epilogue=_lib.EPI_DGELU_Q, c_bf16=dq, aux=codes, col_partials=part if rows else None)
    d0 = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dY, W2, M, F, D, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=d0, aux=aux0)
    torch.cuda.synchronize()
    # element-wise: bf16 rounding of the result (rtol) + the half step of the code times the largest |dY W2| it multiplies (atol)
    vmax = float((dY.float() @ W2.float()).abs().max())
    np.testing.assert_allclose(dq.float().cpu().numpy(), want.cpu().numpy(), rtol=1e-2, atol=0.5 / 196.0 * vmax + 1e-3)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print(f"dgrad with one-byte GELU' [{M}x{F}x{D}]: rel-L2 vs fp32 autograd {rel(dq, want):.2e} (bf16 pre-activation form {rel(d0, want):.2e}); "
          f"the two forms differ by {rel(dq, d0):.2e}")
    assert rel(dq, want) < 6e-3 and rel(dq, d0) < 6e-3
    if rows:
        np.testing.assert_allclose(part.sum(0).cpu().numpy(), want.sum(0).cpu().numpy(), rtol=2e-2, atol=2e-2 * float(want.abs().sum(0).mean()))
    for bad in (dict(M=128), dict(F=F + 64)):
        m2, f2 = bad.get("M", M), bad.get("F", F)
        with pytest.raises(SfronError):
            ops.gemm(X[:m2], _rand((f2, D), gen, 0.06).to(DEV), m2, f2, D, epilogue=_lib.EPI_GELU_Q, bias=None,
                     c_bf16=torch.empty(m2, f2, dtype=torch.bfloat16, device=DEV), aux=torch.empty(m2, f2, dtype=torch.uint8, device=DEV))


def test_gemm_rejects_bad_args():
    from sfron import ops, _lib
    A = torch.zeros(128, 60, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.SfronError):
        ops.gemm(A, A, 128, 128, 60, c_bf16=torch.zeros(128, 128, dtype=torch.bfloat16, device=DEV))   # K % 8 != 0


@pytest.mark.parametrize("hint", [1, 2, 3, 32, 42])
@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
def test_fast_tiles_all_layouts(hint, layout):
    """LDS-DMA fast path (tile_hint 1: 128x128, 2: 256x128, 3: 256x256) vs the generic kernel and torch."""
    from sfron import ops, _lib
    # K/64 = 5 tiles: exercises the odd tail of the 2-deep ring; the interleaved schedule (42) needs an even tile count
    M, N, K = 512, 768, (384 if hint == 42 else 320)
    gen = torch.Generator().manual_seed(hint * 10 + len(layout))
    if layout == "fwd":
        A, B = _rand((M, K), gen), _rand((N, K), gen, 0.1)
        want = A.float() @ B.float().t()
        kw = dict()
    elif layout == "dgrad":
        A, B = _rand((M, K), gen), _rand((K, N), gen, 0.1)
        want = A.float() @ B.float()
        kw = dict(b_t=True)
    else:
        A, B = _rand((K, M), gen, 0.1), _rand((K, N), gen)
        want = A.float().t() @ B.float()
        kw = dict(a_t=True, b_t=True)
    Cf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    Cg = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cf, tile_hint=hint, **kw)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cg, tile_hint=-1, **kw)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)
    assert torch.equal(Cf, Cg), "fast and generic kernels accumulate in the same order: results must be identical"


@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
def test_tile_192_variants(layout):
    """192x192 tile on a 384 x 576 output, K = 96 (generic kernel: K % 64 != 0) and K = 1024: compiler-scheduled (5),
    hand-pipelined (35), interleaved (45) and three-slot (55) forms all accumulate in the generic kernel's order."""
    from sfron import ops, _lib
    for K in (96, 1024):
        M, N = 384, 576
        gen = torch.Generator().manual_seed(K + len(layout))
        if layout == "fwd":
            A, B, kw = _rand((M, K), gen), _rand((N, K), gen, 0.1), dict()
            want = A.float() @ B.float().t()
        elif layout == "dgrad":
            A, B, kw = _rand((M, K), gen), _rand((K, N), gen, 0.1), dict(b_t=True)
            want = A.float() @ B.float()
        else:
            A, B, kw = _rand((K, M), gen, 0.1), _rand((K, N), gen), dict(a_t=True, b_t=True)
            want = A.float().t() @ B.float()
        Cf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        Cg = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cf, tile_hint=5, **kw)
        Cp = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cp, tile_hint=35, **kw)
        ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cg, tile_hint=-1, **kw)
        np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)
        assert torch.equal(Cf, Cg)
        if K % 64 == 0:
            assert torch.equal(Cp, Cg), "hand-pipelined 192x192 tile"
            for Kx in (K, 64, 128, 192, 256):                  # interleaved schedule: 2, 4, 16 K tiles; odd counts fall back
                Ci = torch.zeros(M, N, dtype=torch.float32, device=DEV)
                Cr = torch.zeros(M, N, dtype=torch.float32, device=DEV)
                Ax = A[:, :Kx].contiguous() if layout != "wgrad" else A[:Kx].contiguous()
                Bx = B[:, :Kx].contiguous() if layout == "fwd" else B[:Kx].contiguous()
                ops.gemm(Ax.to(DEV), Bx.to(DEV), M, N, Kx, epilogue=_lib.EPI_F32, c_f32=Ci, tile_hint=45, **kw)
                ops.gemm(Ax.to(DEV), Bx.to(DEV), M, N, Kx, epilogue=_lib.EPI_F32, c_f32=Cr, tile_hint=-1, **kw)
                assert torch.equal(Ci, Cr), f"interleaved 192x192 tile, K={Kx}"
            if layout == "wgrad":                              # three LDS slots: 2 + 3j tiles (K = 128, 320, 512), else falls back
                for Kx in (128, 320, 512, 1024):
                    g3 = torch.Generator().manual_seed(Kx)
                    A3, B3 = _rand((Kx, M), g3, 0.1).to(DEV), _rand((Kx, N), g3).to(DEV)
                    C3 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
                    Cr = torch.zeros(M, N, dtype=torch.float32, device=DEV)
                    ops.gemm(A3, B3, M, N, Kx, epilogue=_lib.EPI_F32, c_f32=C3, tile_hint=55, **kw)
                    ops.gemm(A3, B3, M, N, Kx, epilogue=_lib.EPI_F32, c_f32=Cr, tile_hint=-1, **kw)
                    assert torch.equal(C3, Cr), f"three-slot 192x192 tile, K={Kx}"


@pytest.mark.parametrize("M,N,K,splits,hint", [(384, 192, 2048, 4, 0), (192, 384, 1024 + 64, 4, 0), (192, 192, 512, 3, 35),
                                               (160, 96, 512, 4, 0)])
def test_wgrad_split_k_slabs(M, N, K, splits, hint):
    """split-K weight gradient: every split writes its own fp32 slab (pipelined 192x192 tile when the shape fits, generic
    kernel otherwise, uneven last split included); the fixed-order slab sum equals the unsplit product."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M + N + K)
    A, Bm = _rand((K, M), gen).to(DEV), _rand((K, N), gen).to(DEV)          # C[M,N] = A^T B
    slabs = torch.full((splits, M, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(A, Bm, M, N, K, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=slabs, ldc_f32=N, split_k=splits,
             split_stride=M * N, tile_hint=hint)
    kchunk = -(-(-(-K // splits)) // 64) * 64          # ceil(ceil(K / splits) / 64) * 64, as the library plans it
    used = -(-K // kchunk)
    ref = A.float().t() @ Bm.float()
    got = slabs[:used].sum(0)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-3, atol=2e-2)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.reduce_chunks(slabs, 1, used, M * N, out, M * N)
    s = slabs[0].clone()
    for j in range(1, used):
        s += slabs[j]
    assert torch.equal(out, s)


@pytest.mark.parametrize("layout", ["fwd", "dgrad"])
@pytest.mark.parametrize("M,N,K", [(256, 144, 192), (512, 288, 384), (512, 1152, 576), (1024, 1152, 3456)])
def test_tile_256x144_three_slots(layout, M, N, K):
    """hint 62: 8 x 1 waves of 32 x 144, three LDS slots, uneven DMA plan (no-op shares), third A slot past the 16-bit ds
    offset, transposed-read B image padded to 160 columns (dgrad).  Same accumulation order as the generic kernel."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M + N + K)
    if layout == "fwd":
        A, B, kw = _rand((M, K), gen), _rand((N, K), gen, 0.1), dict()
        want = A.float() @ B.float().t()
    else:
        A, B, kw = _rand((M, K), gen), _rand((K, N), gen, 0.1), dict(b_t=True)
        want = A.float() @ B.float()
    Cf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    Cg = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cf, tile_hint=62, **kw)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=Cg, tile_hint=-1, **kw)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)
    assert torch.equal(Cf, Cg)


def test_tile_256x144_epilogues():
    """GELU (+pre-activation), gated residual, GELU' and pos-embed epilogues through the 256x144 tile vs torch."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(62)
    Bsz, T, D, Hd = 4, 64, 576, 1152
    M = Bsz * T
    X, W1, W2 = _rand((M, D), gen), _rand((Hd, D), gen, 0.05), _rand((D, Hd), gen, 0.05)
    b1, b2 = torch.randn(Hd, generator=gen) * 0.1, torch.randn(D, generator=gen) * 0.1
    h_pre = X.float() @ W1.float().t() + b1
    aux = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    H = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    ops.gemm(X.to(DEV), W1.to(DEV), M, Hd, D, epilogue=_lib.EPI_GELU, bias=b1.to(DEV), c_bf16=H, aux=aux, tile_hint=62)
    np.testing.assert_allclose(aux.float().cpu().numpy(), h_pre.numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(H.float().cpu().numpy(), torch.nn.functional.gelu(h_pre, approximate="tanh").numpy(), rtol=1e-2, atol=1e-2)
    gate = torch.randn(Bsz, 6 * D, generator=gen)
    x0 = torch.randn(M, D, generator=gen)
    branch = H.cpu().float() @ W2.float().t() + b2
    want_x = x0 + gate[:, 2 * D:3 * D].repeat_interleave(T, dim=0) * branch
    x1 = torch.empty(M, D, dtype=torch.float32, device=DEV)
    a2 = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    ops.gemm(H, W2.to(DEV), M, D, Hd, epilogue=_lib.EPI_GATE_RES, bias=b2.to(DEV), c_f32=x1, resid=x0.to(DEV), aux=a2,
             gate=gate.to(DEV)[:, 2 * D:], ldgate=6 * D, tokens=T, tile_hint=62)
    np.testing.assert_allclose(x1.cpu().numpy(), want_x.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(a2.float().cpu().numpy(), branch.numpy(), rtol=1e-2, atol=1e-2)
    dA = _rand((M, D), gen, 0.1)
    hp = aux.cpu().float().requires_grad_(True)
    torch.nn.functional.gelu(hp, approximate="tanh").backward(dA.float() @ W2.float())
    dH = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dA.to(DEV), W2.to(DEV), M, Hd, D, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=dH, aux=aux, tile_hint=62)
    np.testing.assert_allclose(dH.float().cpu().numpy(), hp.grad.numpy(), rtol=1e-2, atol=2e-3)
    pos = torch.randn(T, D, generator=gen)
    P, Wp = _rand((M, 192), gen), _rand((D, 192), gen, 0.1)
    xo = torch.empty(M, D, dtype=torch.float32, device=DEV)
    ops.gemm(P.to(DEV), Wp.to(DEV), M, D, 192, epilogue=_lib.EPI_POS, bias=b2.to(DEV), c_f32=xo, pos=pos.to(DEV), tokens=T, tile_hint=62)
    want = P.float() @ Wp.float().t() + b2 + pos.repeat(Bsz, 1)
    np.testing.assert_allclose(xo.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(32, 6912, 1152), (32, 1152, 256), (5, 48, 64), (17, 16, 32), (32, 195840, 1152)])
def test_skinny_rows_kernel(M, N, K):
    """At most 32 output rows, fp32 out (csrc/gemm.hip k_gemm_skinny: the conditioning path of the DiT forward pass): exact on small
    integers (catches a swapped row / column map or a wrong fragment address), and against torch fp32 and the generic kernel on random
    bf16 data; rows beyond M are neither read nor written."""
    from sfron import ops, _lib
    r = torch.arange(N, dtype=torch.float32).view(N, 1)
    c = torch.arange(K, dtype=torch.float32).view(1, K)
    m = torch.arange(M, dtype=torch.float32).view(M, 1)
    Bi = ((3 * r + 5 * c) % 7 - 3).to(torch.bfloat16)                    # [N, K]
    Ai = ((2 * m + c) % 5 - 2).to(torch.bfloat16)                        # [M, K]
    want = Ai.float() @ Bi.float().t()                                   # |sum| < 2^24: exact in fp32
    guard = torch.full((M + 3, N), 777.0, device=DEV)
    ops.gemm(Ai.to(DEV), Bi.to(DEV), M, N, K, epilogue=_lib.EPI_F32, c_f32=guard[:M])
    assert torch.equal(guard[:M].cpu(), want)
    assert torch.all(guard[M:] == 777.0)
    gen = torch.Generator().manual_seed(M + N + K)
    A, B = _rand((M, K), gen), _rand((N, K), gen, 0.05)
    bias = torch.randn(N, generator=gen)
    want = A.float() @ B.float().t() + bias
    Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, bias=bias.to(DEV), epilogue=_lib.EPI_F32, c_f32=Cf)
    np.testing.assert_allclose(Cf.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * K ** 0.5)
    Cg = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, bias=bias.to(DEV), epilogue=_lib.EPI_F32, c_f32=Cg, tile_hint=-1)   # the generic tile
    np.testing.assert_allclose(Cf.cpu().numpy(), Cg.cpu().numpy(), rtol=1e-5, atol=1e-5 * K ** 0.5)


@pytest.mark.parametrize("M,N,K,T", [(8192, 1152, 16, 256), (96, 48, 16, 32), (40, 16, 8, 8), (512, 768, 32, 64)])
def test_short_contraction_kernel(M, N, K, T):
    """Patch-embedding shape (csrc/gemm.hip k_gemm_shortk: K <= 32, bias + positional table in the epilogue): bit-identical to the generic
    kernel, exact on small integers, ragged row counts."""
    from sfron import ops, _lib
    gen = torch.Generator().manual_seed(M + N + K)
    A, B = _rand((M, K), gen), _rand((N, K), gen, 0.25)
    bias, pos = torch.randn(N, generator=gen), torch.randn(T, N, generator=gen)
    want = A.float() @ B.float().t() + bias + pos.repeat(M // T, 1)
    guard = torch.full((M + 2, N), 777.0, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_POS, bias=bias.to(DEV), c_f32=guard[:M], pos=pos.to(DEV), tokens=T)
    np.testing.assert_allclose(guard[:M].cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    assert torch.all(guard[M:] == 777.0)
    ref = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), M, N, K, epilogue=_lib.EPI_POS, bias=bias.to(DEV), c_f32=ref, pos=pos.to(DEV), tokens=T, tile_hint=-1)
    assert torch.equal(guard[:M], ref)


@pytest.mark.parametrize("N,K,M,masked", [(384, 192, 512, True), (1152, 4608, 8192, True), (3456, 1152, 8192, False), (768, 768, 2048, True)])
def test_wgrad_leaves_masked_sum_of_squares_of_its_tiles(N, K, M, masked):
    """sfron_gemm_desc.sumsq_partials: the weight-gradient product dW[N][K] = dY[M][N]^T X[M][K] also leaves, per 192 x 192 output tile, the sum of
    (mask ? dW : 0)^2 -- the masked sum of squares that clip_grad_norm_ needs after `grad *= mask` (DiT/forget.py:289-298) -- with dW itself
    unchanged bit for bit.  Against torch on the stored fp32 result (same values, another summation order: 1e-6) and run twice (same bits)."""
    from sfron import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(N + K)
    dY = (torch.randn(M, N, generator=g, device=DEV) * 0.1).to(torch.bfloat16)
    X = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    mask = (torch.rand(N, K, generator=g, device=DEV) < 0.5).to(torch.uint8) if masked else None
    n = _lib.lib().sfron_gemm_sumsq_partials(N, K, M)
    assert n == (N // 192) * (K // 192)
    ref = torch.empty(N, K, dtype=torch.float32, device=DEV)
    ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=ref)
    outs = []
    for _ in range(2):
        C = torch.empty(N, K, dtype=torch.float32, device=DEV)
        part = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
        ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C, sumsq_partials=part, sumsq_mask=mask)
        torch.cuda.synchronize()
        assert torch.equal(C, ref)
        outs.append(part.clone())
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()
    want = (ref.double() * (mask.double() if masked else 1.0)).pow(2).sum().item()
    assert abs(outs[0].sum().item() - want) <= 1e-6 * want
    # per tile: every partial belongs to exactly one 192 x 192 tile (any order)
    tiles = (ref.double() * (mask.double() if masked else 1.0)).pow(2).view(N // 192, 192, K // 192, 192).sum((1, 3)).flatten()
    assert torch.allclose(outs[0].sort().values, tiles.sort().values, rtol=1e-6)


def test_sumsq_partials_refused_where_the_shape_takes_another_tile():
    from sfron import _lib, ops
    assert _lib.lib().sfron_gemm_sumsq_partials(128, 192, 256) == 0
    dY, X = torch.zeros(256, 128, dtype=torch.bfloat16, device=DEV), torch.zeros(256, 192, dtype=torch.bfloat16, device=DEV)
    C, part = torch.empty(128, 192, device=DEV), torch.zeros(4, dtype=torch.float64, device=DEV)
    with pytest.raises(_lib.SfronError):
        ops.gemm(dY, X, 128, 192, 256, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C, sumsq_partials=part)


@pytest.mark.parametrize("K,S", [(768, 6), (3072, 8), (2304, 6)])
def test_split_k_products_with_their_finish_kernels(K, S):
    """Few-tile products of a small batch * tokens (BASELINE config 2: [2048 x 768] outputs = 32 tiles of 256 x 192): the contraction split over S
    workgroups per tile on the pipelined tile + sfron_split_sum_bf16 (an input gradient as the next product's bf16 operand) /
    sfron_split_gate_res (the gated-residual epilogue of the forward proj / fc2 products, DiT/models.py:120-121), against the one-launch
    products with the same epilogues: the same sums in another order (fp32 1e-5; bf16 one rounding), twice the same bits."""
    from sfron import _lib, ops
    from sfron._lib import check, ptr, stream_ptr
    M, N, T = 2048, 768, 64
    g = torch.Generator(device=DEV).manual_seed(K)
    X = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    Wt = (torch.randn(K, N, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=DEV) * 0.1
    gate = torch.randn(M // T, N, generator=g, device=DEV)
    resid = torch.randn(M, N, generator=g, device=DEV)
    L = _lib.lib()
    slabs = torch.empty(S, M, N, dtype=torch.float32, device=DEV)
    # (a) forward layout + gated residual
    x1, a1 = torch.empty(M, N, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_GATE_RES, bias=bias, c_f32=x1, resid=resid, aux=a1, gate=gate, ldgate=N, tokens=T)
    outs = []
    for _ in range(2):
        slabs.fill_(float("nan"))
        ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_F32, c_f32=slabs, ldc_f32=N, split_k=S, split_stride=M * N)
        x2, a2 = torch.empty(M, N, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        check(L.sfron_split_gate_res(ptr(slabs), S, M * N, ptr(bias), ptr(gate), N, T, ptr(resid), ptr(x2), ptr(a2), M, N, stream_ptr()), "split_gate_res")
        torch.cuda.synchronize()
        outs.append((x2, a2))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()
    assert ((outs[0][0] - x1).norm() / x1.norm()).item() < 1e-5
    assert ((outs[0][1].float() - a1.float()).norm() / a1.float().norm()).item() < 3e-3
    # (b) input-gradient layout (W read transposed) + bf16 finish
    dY = X
    want = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dY, Wt, M, N, K, b_t=True, c_bf16=want)
    slabs.fill_(float("nan"))
    ops.gemm(dY, Wt, M, N, K, b_t=True, epilogue=_lib.EPI_F32, c_f32=slabs, ldc_f32=N, split_k=S, split_stride=M * N)
    got = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    check(L.sfron_split_sum_bf16(ptr(slabs), S, M * N, M * N, ptr(got), stream_ptr()), "split_sum_bf16")
    ref32 = dY.float() @ Wt.float()
    assert ((got.float() - ref32).norm() / ref32.norm()).item() < 4e-3
    assert ((got.float() - want.float()).norm() / want.float().norm()).item() < 4e-3
