"""GPU parity of the convolutional U-Net path (csrc/conv.hip + sfron.unet) -- the DDPM Conditional_Model of BASELINE config 0:
(1) each kernel against torch fp32 on the same bf16-rounded inputs (implicit-GEMM convolution forward / input gradient / weight
    gradient incl. the (0,1,0,1)-pad stride-2 Downsample and the nearest-x2 Upsample forms, GroupNorm(32)+swish+dropout forward /
    backward, batched GEMM + softmax of the single-head AttnBlock);
(2) the whole model forward + backward against the oracle (oracle/ddpm_ref.py, pinned to DDPM/models/diffusion.py by
    tests/golden/ddpm_model.npz) with the SAME classifier-free keep mask and dropout masks, at a reduced depth and at
    cifar10_sfron.yml's size (38.6 M parameters, batch 64);
(3) the SFR-on loop body DDPM/runners/diffusion.py:1075-1180 (adaga / ga / rl) on the native denoiser against DDPMSfronOracle, and
    BASELINE config 1 itself: 50 SFR-on steps, batch 64, on the HIP path.
Tolerances: fp32 kernels 1e-5 .. 1e-4; bf16-operand GEMMs 2^-9 relative per element (bounds per test)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _rows(x):      # NCHW -> NHWC rows
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _nchw(r, B, H, W):
    return r.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


# (batch, image side, c_in, c_out).  The *_big / wide / ragged forms have pixels % 256 == 0 and channels % 64 == 0: forward and input
# gradient run on the LDS-DMA pipelined tile (k_cgemm: im2col as DMA address arithmetic -- stride 2, nearest-upsampled and
# zero-dilated sources, flipped taps; 160-column tiles for the 320-wide output, 128-column tiles with a ragged last tile for 192)
CONV_FORMS = {"same": (3, 8, 32, 64), "down": (3, 8, 32, 32), "up": (3, 8, 32, 32), "rgb_in": (3, 8, 3, 32), "rgb_out": (3, 8, 32, 3),
              "same_big": (16, 16, 64, 128),                     # 4096 pixels -> also the weight gradient's split-K path
              "down_big": (4, 16, 64, 64), "up_big": (4, 8, 64, 128), "wide": (4, 16, 128, 320), "ragged": (4, 8, 64, 192),
              # SD v1's hot shapes (timed in tools/bench_conv.py, now compared): 320 -> 320 at 64 x 64, 1280 -> 1280 at
              # 8 x 8 (split-K), and the concatenation convolutions 2560 -> 1280 at 16 x 16 / 1920 -> 640 at 32 x 32
              "sd320": (1, 64, 320, 320), "sd1280": (2, 8, 1280, 1280), "sdcat16": (1, 16, 2560, 1280), "sdcat32": (1, 32, 1920, 640),
              "sd_down": (1, 32, 640, 640), "sd_up": (1, 16, 1280, 1280)}


@pytest.mark.parametrize("form", list(CONV_FORMS))
def test_conv3x3_forward_dgrad_wgrad(form):
    import ctypes
    from sfron import _lib, unet
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(len(form))
    B, H, ci, co = CONV_FORMS[form]
    form = form.split("_")[0] if form.endswith("_big") else form.split("_")[1] if form.startswith("sd_") else form
    cip, cop = unet._pad8(ci), unet._pad8(co)
    x = torch.randn(B, ci, H, H, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(co, ci, 3, 3, generator=g) * 0.1)
    bias = torch.randn(co, generator=g) * 0.1
    wq = w.to(torch.bfloat16).float()
    xr = torch.zeros(B * H * H, cip)
    xr[:, :ci] = _rows(x)
    xr = xr.to(torch.bfloat16).to(DEV)
    wf = torch.empty(cop * 9 * cip, dtype=torch.bfloat16, device=DEV)
    wd = torch.empty(ci * 9 * cop, dtype=torch.bfloat16, device=DEV) if ci % 8 == 0 else None
    w_d = w.to(DEV)
    check(L.sfron_conv_wprep(ptr(w_d), co, ci, 9, cop, cip, ptr(wf), ptr(wd), stream_ptr()), "wprep")
    xt = x.clone().requires_grad_(True)
    wt = wq.clone().requires_grad_(True)
    if form == "down":          # Downsample (models/diffusion.py:76-80)
        ref = F.conv2d(F.pad(xt, (0, 1, 0, 1)), wt, bias, stride=2, padding=0)
        ho, kw = H // 2, dict(stride=2, pad=0, up=0)
    elif form == "up":          # Upsample (:56-60)
        ref = F.conv2d(F.interpolate(xt, scale_factor=2.0, mode="nearest"), wt, bias, padding=1)
        ho, kw = 2 * H, dict(stride=1, pad=1, up=1)
    else:
        ref = F.conv2d(xt, wt, bias, padding=1)
        ho, kw = H, dict(stride=1, pad=1, up=0)
    rows = B * ho * ho
    out = torch.empty(rows, cop, dtype=torch.float32, device=DEV)
    bp = torch.zeros(cop)
    bp[:co] = bias
    d = unet._conv_desc(B, H, H, cip, ho, ho, cop, 9, kw["stride"], kw["pad"], kw["up"], 0, bias=bp.to(DEV), out_f32=out, ld_out=cop)
    check(L.sfron_conv_fwd(ctypes.byref(d), ptr(xr), ptr(wf), stream_ptr()), "conv_fwd")
    got = _nchw(out.cpu(), B, ho, ho)[:, :co]
    np.testing.assert_allclose(got.numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-4 * math.sqrt(9 * ci))
    if cop != co:
        assert float(out[:, co:].abs().max()) == 0.0
    # backward: d_out random (bf16-rounded)
    dy = (torch.randn(ref.shape, generator=g) * 0.1).to(torch.bfloat16).float()
    ref.backward(dy)
    dyr = torch.zeros(rows, cop)
    dyr[:, :co] = _rows(dy)
    dyb = dyr.to(torch.bfloat16).to(DEV)
    wdsc = unet._conv_desc(B, H, H, cip, ho, ho, cop, 9, kw["stride"], kw["pad"], kw["up"], 0)
    nsl = L.sfron_conv_wgrad_splits(ctypes.byref(wdsc))
    dwg = torch.full((nsl * cop * 9 * cip,), float("nan"), dtype=torch.float32, device=DEV)
    check(L.sfron_conv_wgrad(ctypes.byref(wdsc), ptr(dyb), cop, ptr(xr), ptr(dwg), stream_ptr()), "conv_wgrad")
    dw = torch.empty(co, ci, 3, 3, dtype=torch.float32, device=DEV)
    check(L.sfron_conv_wgrad_scatter(ptr(dwg), co, ci, 9, cip, nsl, cop * 9 * cip, ptr(dw), stream_ptr()), "scatter")
    np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=3e-4, atol=3e-4 * math.sqrt(rows))
    if wd is None:
        return
    if form == "down":
        ds = torch.empty(B * H * H, ci, dtype=torch.float32, device=DEV)
        dd = unet._conv_desc(B, ho, ho, cop, H, H, ci, 9, 1, 2, 0, 1, out_f32=ds, ld_out=ci)
        check(L.sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(wd), stream_ptr()), "dgrad")
    elif form == "up":
        du = torch.empty(rows, ci, dtype=torch.float32, device=DEV)
        dd = unet._conv_desc(B, ho, ho, cop, ho, ho, ci, 9, 1, 1, 0, 0, out_f32=du, ld_out=ci)
        check(L.sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(wd), stream_ptr()), "dgrad")
        ds = torch.empty(B * H * H, ci, dtype=torch.float32, device=DEV)
        check(L.sfron_pool2_sum(ptr(du), B, H, H, ci, ptr(ds), 0, stream_ptr()), "pool2")
    else:
        ds = torch.empty(rows, ci, dtype=torch.float32, device=DEV)
        dd = unet._conv_desc(B, ho, ho, cop, ho, ho, ci, 9, 1, 1, 0, 0, out_f32=ds, ld_out=ci)
        check(L.sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(wd), stream_ptr()), "dgrad")
    np.testing.assert_allclose(_nchw(ds.cpu(), B, H, H).numpy(), xt.grad.numpy(), rtol=3e-4, atol=3e-4 * math.sqrt(9 * co))


@pytest.mark.parametrize("form", ["rows", "slabs", "rows64"])
@pytest.mark.parametrize("C,HW,swish,drop", [(128, 64, 1, False), (256, 16, 1, True), (64, 256, 0, False), (384, 64, 1, True), (320, 1024, 1, False),
                                             (1280, 64, 1, True), (2560, 16, 1, False)])
def test_groupnorm_swish_dropout_fwd_bwd(C, HW, swish, drop, form):
    """form "rows": the row-coalesced two-phase kernels (scratch given); "slabs": the per-(sample, group) kernels (no scratch); "rows64": batch 64,
    where the rule takes the one-launch form (k_gn3_*: a workgroup per sample and block of whole groups; round 6) for every shape here.
    Channels per group 2 .. 80, including the LDM widths (10, 40, 80: not powers of two) and more than 1024 channels."""
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(C + HW)
    B = 64 if form == "rows64" else 3
    form = "rows" if form == "rows64" else form
    x = torch.randn(B * HW, C, generator=g) * 1.5 + 0.3
    gamma, beta = torch.randn(C, generator=g) * 0.5 + 1.0, torch.randn(C, generator=g) * 0.2
    mask = (torch.rand(B * HW, C, generator=g) > 0.1).to(torch.uint8) if drop else None
    scale = 1.0 / 0.9 if drop else 1.0
    xt = x.clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    xn = xt.view(B, HW, C).permute(0, 2, 1)                   # [B, C, HW]
    z = F.group_norm(xn, 32, gt, bt, eps=1e-6)
    if swish:
        z = z * torch.sigmoid(z)
    z = z.permute(0, 2, 1).reshape(B * HW, C)
    if drop:
        z = z * mask.float() * scale
    y = torch.empty(B * HW, C, dtype=torch.bfloat16, device=DEV)
    mean = torch.empty(B * 32, dtype=torch.float32, device=DEV)
    rstd = torch.empty_like(mean)
    xd, gd, bd = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    md = mask.to(DEV) if drop else None
    ws = torch.empty(L.sfron_groupnorm_scratch_bytes(B, HW, C, 32) // 8 + 2, dtype=torch.float64, device=DEV) if form == "rows" else None
    check(L.sfron_groupnorm_fwd(ptr(xd), C, ptr(gd), ptr(bd), B, HW, C, 32, 1e-6, swish, ptr(md), scale, ptr(y), ptr(mean), ptr(rstd),
                                ptr(ws), stream_ptr()), "gn_fwd")
    np.testing.assert_allclose(y.float().cpu().numpy(), z.detach().numpy(), rtol=1e-2, atol=1e-2)
    dy = torch.randn(B * HW, C, generator=g) * 0.1
    z.backward(dy)
    dx = torch.full((B * HW, C), 0.5, dtype=torch.float32, device=DEV)
    pg = torch.empty(B, C, dtype=torch.float32, device=DEV)
    pb = torch.empty_like(pg)
    dy_d = dy.to(DEV)
    check(L.sfron_groupnorm_bwd(ptr(dy_d), ptr(xd), C, ptr(gd), ptr(bd), ptr(mean), ptr(rstd), B, HW, C, 32, swish, ptr(md), scale,
                                ptr(dx), C, 1, ptr(pg), ptr(pb), ptr(ws), stream_ptr()), "gn_bwd")
    np.testing.assert_allclose(dx.cpu().numpy() - 0.5, xt.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pg.sum(0).cpu().numpy(), gt.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(pb.sum(0).cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-4)
    # the residual form: dx (+)= extra + gradient in ONE pass gives the bits of  dx (+)= extra  followed by  dx += gradient
    # (extra read with its own row stride: a column slice of a wider buffer)
    wide = torch.randn(B * HW, C + 8, generator=g).to(DEV)
    extra = wide[:, 8:]
    for acc in (0, 1):
        two = torch.full((B * HW, C), 0.25, dtype=torch.float32, device=DEV)
        one = two.clone()
        check(L.sfron_copy_cols(extra.data_ptr(), C + 8, B * HW, C, ptr(two), C, acc, stream_ptr()), "copy_cols")
        check(L.sfron_groupnorm_bwd(ptr(dy_d), ptr(xd), C, ptr(gd), ptr(bd), ptr(mean), ptr(rstd), B, HW, C, 32, swish, ptr(md), scale,
                                    ptr(two), C, 1, ptr(pg), ptr(pb), ptr(ws), stream_ptr()), "gn_bwd")
        check(L.sfron_groupnorm_bwd_res(ptr(dy_d), ptr(xd), C, ptr(gd), ptr(bd), ptr(mean), ptr(rstd), B, HW, C, 32, swish, ptr(md), scale,
                                        ptr(one), C, acc, extra.data_ptr(), C + 8, ptr(pg), ptr(pb), ptr(ws), stream_ptr()), "gn_bwd_res")
        assert torch.equal(one, two)
        ref = xt.grad.numpy() + extra.cpu().numpy() + (0.25 if acc else 0.0)
        np.testing.assert_allclose(one.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
    # the operand form: dx as bf16 (the rounding of the fp32 result, bit for bit) + its column sums per (sample, chunk)
    if form == "rows" and L.sfron_groupnorm_bwd_cast_ok(C, C, 32):
        plain = torch.empty(B * HW, C, dtype=torch.float32, device=DEV)
        check(L.sfron_groupnorm_bwd(ptr(dy_d), ptr(xd), C, ptr(gd), ptr(bd), ptr(mean), ptr(rstd), B, HW, C, 32, swish, ptr(md), scale,
                                    ptr(plain), C, 0, ptr(pg), ptr(pb), ptr(ws), stream_ptr()), "gn_bwd")
        nch = L.sfron_groupnorm_chunks(B, HW)
        d16 = torch.empty(B * HW, C, dtype=torch.bfloat16, device=DEV)
        cpart = torch.full((B * nch, C), float("nan"), dtype=torch.float32, device=DEV)
        pg2, pb2 = torch.empty_like(pg), torch.empty_like(pb)
        check(L.sfron_groupnorm_bwd_cast(ptr(dy_d), ptr(xd), C, ptr(gd), ptr(bd), ptr(mean), ptr(rstd), B, HW, C, 32, swish, ptr(md), scale,
                                         ptr(d16), ptr(cpart), ptr(pg2), ptr(pb2), ptr(ws), stream_ptr()), "gn_bwd_cast")
        assert torch.equal(d16, plain.to(torch.bfloat16)) and torch.equal(pg2, pg) and torch.equal(pb2, pb)
        per_sample = cpart.view(B, nch, C).double().sum(1)
        np.testing.assert_allclose(per_sample.cpu().numpy(), plain.view(B, HW, C).double().sum(1).cpu().numpy(), rtol=1e-5, atol=1e-5 * HW ** 0.5)
    else:
        assert form != "rows" or C > 2048


@pytest.mark.parametrize("M,N,K,out", [(512, 320, 1280, "f32"), (1024, 640, 320, "bf16"), (256, 200, 128, "f32"), (768, 2560, 64, "bf16")])
def test_pipelined_tile_plain_products(M, N, K, out):
    """C = A B^T (+ bias, + per-sample vector, + residual | accumulate) on the LDS-DMA pipelined tile (M % 256 == 0, K % 64 == 0):
    160-column tiles (N % 160 == 0) and 128-column tiles with a ragged last tile, both epilogues, against torch fp32."""
    from sfron import unet
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    ref = a.float() @ b.float().t() + bias
    ad, bd, biasd = a.to(DEV), b.to(DEV), bias.to(DEV)
    if out == "bf16":
        c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        unet.bgemm(ad, bd, M, N, K, lda=K, ldb=K, c_bf16=c, ldc=N, bias=biasd)
        np.testing.assert_allclose(c.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2 * math.sqrt(K) * 0.25)
        return
    T = 64
    vec = torch.randn(M // T, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref2 = ref + vec.repeat_interleave(T, dim=0) + resid
    c = torch.empty(M, N, dtype=torch.float32, device=DEV)
    vd, rd = vec.to(DEV), resid.to(DEV)
    unet.bgemm(ad, bd, M, N, K, lda=K, ldb=K, c_f32=c, ldc=N, bias=biasd, vec=vd, ld_vec=N, rows_per_sample=T, resid=rd)
    np.testing.assert_allclose(c.cpu().numpy(), ref2.numpy(), rtol=2e-4, atol=2e-4 * math.sqrt(K))
    unet.bgemm(ad, bd, M, N, K, lda=K, ldb=K, c_f32=c, ldc=N, accumulate=True)          # c += A B^T
    np.testing.assert_allclose(c.cpu().numpy(), (ref2 + ref - bias).numpy(), rtol=2e-4, atol=4e-4 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(512, 320, 1280), (256, 200, 192), (1024, 2560, 320), (256, 768, 64), (4096, 320, 320)])
def test_pipelined_tile_transposed_read_products(M, N, K):
    """The transposed-read tile (k_cgemm_t) on plain products: Linear input gradient dX = dY W (A direct, B read transposed) and the
    weight gradient dW = dY^T X (both read transposed, contraction over the M rows; through the split-K slabs when the caller gives
    scratch), ragged last tiles in both output dimensions, fp32 and bf16 results, against torch fp32."""
    from sfron import unet
    g = torch.Generator().manual_seed(M * 3 + N + K)
    dy = (torch.randn(M, N, generator=g) * 0.5).to(torch.bfloat16)          # [rows][out]
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)           # [out][in]
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)           # [rows][in]
    dyd, wd, xd = dy.to(DEV), w.to(DEV), x.to(DEV)
    # input gradient [M][K] = dY [M][N] . W [N][K]: contraction over N, B read transposed
    ref_dx = dy.float() @ w.float()
    dx = torch.full((M, K), float("nan"), dtype=torch.float32, device=DEV)
    unet.bgemm(dyd, wd, M, K, N, lda=N, ldb=K, b_t=True, c_f32=dx, ldc=K)
    np.testing.assert_allclose(dx.cpu().numpy(), ref_dx.numpy(), rtol=2e-4, atol=2e-4 * math.sqrt(N))
    dxb = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    unet.bgemm(dyd, wd, M, K, N, lda=N, ldb=K, b_t=True, c_bf16=dxb, ldc=K)
    np.testing.assert_allclose(dxb.float().cpu().numpy(), ref_dx.numpy(), rtol=1e-2, atol=1e-2 * math.sqrt(N) * 0.25)
    # weight gradient [N][K] = dY^T X: contraction over the M rows, both operands read transposed
    ref_dw = dy.float().t() @ x.float()
    dw = torch.full((N, K), float("nan"), dtype=torch.float32, device=DEV)
    unet.bgemm(dyd, xd, N, K, M, lda=N, ldb=K, a_t=True, b_t=True, c_f32=dw, ldc=K)
    np.testing.assert_allclose(dw.cpu().numpy(), ref_dw.numpy(), rtol=2e-4, atol=2e-4 * math.sqrt(M))
    unet.bgemm(dyd, xd, N, K, M, lda=N, ldb=K, a_t=True, b_t=True, c_f32=dw, ldc=K, accumulate=True)
    np.testing.assert_allclose(dw.cpu().numpy(), 2 * ref_dw.numpy(), rtol=2e-4, atol=4e-4 * math.sqrt(M))


def test_head_batched_products_split_over_the_contraction():
    """dK / dV of a cross-attention: per (sample, head) out [Lk][d] = A[tokens][Lk]^T B[tokens][d] with the head's columns taken from a
    [tokens][heads * d] matrix -- one output tile per (sample, head), contraction over 2048 tokens, split into fp32 slabs and added
    in a fixed order (bf16 result) -- against torch fp32, twice the same bits."""
    from sfron import unet
    g = torch.Generator().manual_seed(91)
    Bn, H, T, Lk, d = 2, 4, 2048, 80, 40
    P = (torch.rand(Bn * H * T, Lk, generator=g)).to(torch.bfloat16)                    # [b][h][token][key]
    dO = (torch.randn(Bn * T, H * d, generator=g) * 0.3).to(torch.bfloat16)             # [b][token][h * d]
    ref = torch.einsum("bhtk,bthd->bkhd", P.float().view(Bn, H, T, Lk), dO.float().view(Bn, T, H, d)).reshape(Bn * Lk, H * d)
    Pd, dOd = P.to(DEV), dO.to(DEV)
    outs = []
    for _ in range(2):
        dv = torch.full((Bn * Lk, H * d), float("nan"), dtype=torch.bfloat16, device=DEV)
        unet.bgemm(Pd, dOd, Lk, d, T, lda=Lk, ldb=H * d, a_t=True, b_t=True, batch=Bn, sa=H * T * Lk, sb=T * H * d, sc=Lk * H * d, batch2=H,
                   sa2=T * Lk, sb2=d, sc2=d, c_bf16=dv, ldc=H * d)
        outs.append(dv.clone())
    np.testing.assert_allclose(outs[0].float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2 * math.sqrt(T) * 0.3)
    assert torch.equal(outs[0], outs[1])


def test_few_rows_deep_contraction_split():
    """dX = dY W with 8 rows over a contraction of 8192 (the embedding projection's input gradient): split over the chip into fp32
    slabs added in a fixed order -- against torch fp32, and twice the same bits."""
    from sfron import unet
    g = torch.Generator().manual_seed(77)
    M, N, K = 8, 1280, 8192
    dy = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g) * 0.1).to(torch.bfloat16)
    ref = dy.float() @ w.float()
    dyd, wd = dy.to(DEV), w.to(DEV)
    outs = []
    for _ in range(2):
        dx = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        unet.bgemm(dyd, wd, M, N, K, lda=K, ldb=N, b_t=True, c_f32=dx, ldc=N)
        outs.append(dx.clone())
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * math.sqrt(K))
    assert torch.equal(outs[0], outs[1])


def test_batched_gemm_and_softmax():
    from sfron import _lib, unet
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    B, T, C = 5, 48, 64
    qkv = (torch.randn(B * T, 3 * C, generator=g)).to(torch.bfloat16)
    qd = qkv.to(DEV)
    q, k, v = qkv.float().view(B, T, 3, C).unbind(2)
    S = torch.empty(B * T, T, dtype=torch.float32, device=DEV)
    unet.bgemm(qd.data_ptr(), qd.data_ptr() + 2 * C, T, T, C, lda=3 * C, ldb=3 * C, batch=B, sa=T * 3 * C, sb=T * 3 * C, sc=T * T, c_f32=S, ldc=T)
    ref = q @ k.transpose(1, 2)
    np.testing.assert_allclose(S.cpu().view(B, T, T).numpy(), ref.numpy(), rtol=1e-4, atol=1e-3)
    P = torch.empty(B * T, T, dtype=torch.bfloat16, device=DEV)
    scale = C ** -0.5
    check(L.sfron_softmax_fwd(ptr(S), B * T, T, T, scale, ptr(P), stream_ptr()), "softmax")
    pr = torch.softmax(ref * scale, dim=-1)
    np.testing.assert_allclose(P.float().cpu().view(B, T, T).numpy(), pr.numpy(), rtol=1e-2, atol=2e-3)
    O = torch.empty(B * T, C, dtype=torch.bfloat16, device=DEV)
    unet.bgemm(P, qd.data_ptr() + 4 * C, T, C, T, lda=T, ldb=3 * C, b_t=True, batch=B, sa=T * T, sb=T * 3 * C, sc=T * C, c_bf16=O, ldc=C)
    np.testing.assert_allclose(O.float().cpu().view(B, T, C).numpy(), (P.float().cpu().view(B, T, T) @ v).numpy(), rtol=1e-2, atol=1e-2)
    dP = torch.randn(B * T, T, generator=g)
    dS = torch.empty(B * T, T, dtype=torch.bfloat16, device=DEV)
    dP_d = dP.to(DEV)
    check(L.sfron_softmax_bwd(ptr(P), ptr(dP_d), B * T, T, scale, ptr(dS), stream_ptr()), "softmax_bwd")
    pf = P.float().cpu()
    want = scale * pf * (dP - (pf * dP).sum(-1, keepdim=True))
    np.testing.assert_allclose(dS.float().cpu().numpy(), want.numpy(), rtol=1e-2, atol=2e-3)
    # transposed forms: dV = P^T dO, dK = dS^T Q
    dO = torch.randn(B * T, C, generator=g).to(torch.bfloat16).to(DEV)
    dV = torch.empty(B * T, C, dtype=torch.bfloat16, device=DEV)
    unet.bgemm(P, dO, T, C, T, lda=T, ldb=C, a_t=True, b_t=True, batch=B, sa=T * T, sb=T * C, sc=T * C, c_bf16=dV, ldc=C)
    want = pf.view(B, T, T).transpose(1, 2) @ dO.float().cpu().view(B, T, C)
    np.testing.assert_allclose(dV.float().cpu().view(B, T, C).numpy(), want.numpy(), rtol=1e-2, atol=1e-2)


class _FixedDrop(torch.nn.Module):
    """Stands in for a ResnetBlock's nn.Dropout inside the ORACLE so that it applies the mask the HIP path was given."""

    def __init__(self, mask_nchw, p):
        super().__init__()
        self.m, self.s = mask_nchw, 1.0 / (1.0 - p)

    def forward(self, x):
        return x * self.m * self.s


def _pair(cfg, seed):
    from oracle import ddpm_ref
    from sfron import unet
    torch.manual_seed(seed)
    ref = ddpm_ref.ConditionalUNet(**cfg)
    model = unet.Conditional_Model(**{k: v for k, v in cfg.items()})
    assert [n for n, _ in model.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    model.load_state_dict({"module." + k: v for k, v in ref.state_dict().items()})        # DataParallel-style keys, as ckpt.pth has them
    return ref, model


def _resblocks_in_execution_order(ref):
    blocks = []
    for lvl in range(ref.num_resolutions):
        blocks += list(ref.down[lvl].block)
    blocks += [ref.mid.block_1, ref.mid.block_2]
    for lvl in reversed(range(ref.num_resolutions)):
        blocks += list(ref.up[lvl].block)
    return blocks


def _fwd_bwd_both(ref, model, cfg, B, seed, p_drop):
    g = torch.Generator().manual_seed(seed)
    S = cfg["resolution"]
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    c = torch.randint(0, 10, (B,), generator=g)
    keep = (torch.rand(B, generator=g) > 0.3)
    w = torch.randn(B, 3, S, S, generator=g) * 0.1
    # dropout masks per ResnetBlock in execution order (NHWC rows for the HIP path, NCHW for the oracle)
    ref.train(); model.train()
    masks_rows, blocks = [], _resblocks_in_execution_order(ref)
    shapes = []
    hook_handles = [blk.conv2.register_forward_pre_hook(lambda m, inp, sh=shapes: sh.append(tuple(inp[0].shape))) for blk in blocks]
    with torch.no_grad():
        ref(x, t.float(), c, mode="train", keep_mask=keep)
    for h in hook_handles:
        h.remove()
    for blk, shp in zip(blocks, shapes):
        if p_drop > 0:
            m = (torch.rand(shp[0], shp[2], shp[3], shp[1], generator=g) >= p_drop)
            masks_rows.append(m.reshape(-1, shp[1]).to(torch.uint8))
            blk.dropout = _FixedDrop(m.permute(0, 3, 1, 2).float(), p_drop)
        else:
            masks_rows.append(None)
            blk.dropout = torch.nn.Identity()
    ref.zero_grad()
    out_ref = ref(x, t.float(), c, mode="train", keep_mask=keep)
    (out_ref * w).sum().backward()
    out = model(x.to(DEV), t.float().to(DEV), c.to(DEV), mode="train", keep_mask=keep, dropout_masks=masks_rows)
    assert out.requires_grad and out.shape == out_ref.shape
    (out * w.to(DEV)).sum().backward()
    return out, out_ref


SMALL = dict(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.1, resolution=16, n_classes=10)
CIFAR = dict(ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), dropout=0.1, resolution=32, n_classes=10)


@pytest.mark.parametrize("cfg,B,p_drop", [(SMALL, 4, 0.0), (SMALL, 3, 0.1), (CIFAR, 8, 0.1)])
def test_unet_forward_backward_vs_oracle(cfg, B, p_drop):
    ref, model = _pair(cfg, seed=B)
    model.dropout_p = p_drop if p_drop > 0 else 0.0
    out, out_ref = _fwd_bwd_both(ref, model, cfg, B, seed=11, p_drop=p_drop)
    e_out = _rel(out, out_ref)
    worst, wname = 0.0, ""
    dots = na = nb = 0.0
    gmax = max(q.grad.norm().item() for q in ref.parameters())
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert q.grad is not None, n
        ga, gb = p.grad.detach().cpu().flatten(), q.grad.flatten()
        assert torch.isfinite(ga).all(), n
        if n.endswith(".k.bias"):
            # softmax over the keys is invariant to a constant added to every key's score: d k.bias = 0 exactly; both sides hold
            # rounding noise only
            assert ga.norm().item() < 1e-3 * gmax and gb.norm().item() < 1e-3 * gmax, n
            continue
        e = ((ga - gb).norm() / (gb.norm() + 1e-30)).item()
        if e > worst:
            worst, wname = e, n
        dots += torch.dot(ga.double(), gb.double()).item(); na += ga.double().pow(2).sum().item(); nb += gb.double().pow(2).sum().item()
    cos = dots / math.sqrt(na * nb)
    print(f"U-Net {cfg['ch_mult']} B={B} dropout {p_drop}: out rel-L2 {e_out:.3e}, worst grad rel-L2 {worst:.3e} ({wname}), cosine {cos:.6f}")
    assert e_out < 1.5e-2, e_out
    assert worst < 5e-2, (wname, worst)
    assert cos > 0.9995, cos


def test_unet_cifar10_batch64_forward_backward_vs_oracle():
    """cifar10_sfron.yml's model (38.6 M parameters) at the batch of BASELINE config 1."""
    ref, model = _pair(CIFAR, seed=64)
    assert sum(p.numel() for p in model.parameters()) == 38_632_323
    out, out_ref = _fwd_bwd_both(ref, model, CIFAR, 64, seed=12, p_drop=0.1)
    assert _rel(out, out_ref) < 1.5e-2
    dots = na = nb = 0.0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        ga, gb = p.grad.detach().cpu().flatten().double(), q.grad.flatten().double()
        if not n.endswith(".k.bias"):
            assert ((ga - gb).norm() / (gb.norm() + 1e-30)).item() < 5e-2, n
        dots += torch.dot(ga, gb).item(); na += ga.pow(2).sum().item(); nb += gb.pow(2).sum().item()
    assert dots / math.sqrt(na * nb) > 0.9995


def test_dropout_mask_kernel_statistics_and_keys():
    """nn.Dropout's keep mask in one launch: keep rate 1 - p, identical for identical (seed, counter, salt), different when any of the
    three changes, no run structure along rows (a counter-based draw, not the torch stream: tests that compare with the oracle hand
    the masks over)."""
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    n, p = 1 << 20, 0.1
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)

    def draw(seed, salt, size=n):
        m = torch.empty(size, dtype=torch.uint8, device=DEV)
        check(L.sfron_dropout_mask(seed, ptr(cnt), salt, size, p, ptr(m), stream_ptr()), "dropout_mask")
        return m
    a, a2, b, c = draw(7, 1), draw(7, 1), draw(7, 2), draw(8, 1)
    cnt.add_(1)
    d = draw(7, 1)
    assert torch.equal(a, a2) and not torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    for m in (a, b, c, d):
        assert set(m.unique().tolist()) <= {0, 1}
        assert abs(m.float().mean().item() - (1 - p)) < 2e-3                                     # sigma = 2.9e-4
    assert abs((a ^ b).float().mean().item() - 2 * p * (1 - p)) < 3e-3                             # independent draws
    x = a.view(1024, 1024).float()
    assert abs(torch.corrcoef(torch.stack([x[:, :-1].flatten(), x[:, 1:].flatten()]))[0, 1].item()) < 5e-3
    odd = draw(7, 3, size=1001)                                                                   # tail that is not a multiple of four
    assert odd.shape[0] == 1001 and set(odd.unique().tolist()) <= {0, 1}
    # every mask of a pass in one launch (sfron_dropout_mask_batch): the same bits as the launches it replaces
    sizes = [64 * 32 * 32 * 128, 1001, 4, 7, 64 * 8 * 8 * 256]
    items, off = [], 0
    for k, sz in enumerate(sizes):
        items.append((k + 1, sz, off))
        off += (sz + 3) // 4 * 4
    table = torch.tensor([v for it in items for v in it], dtype=torch.int64, device=DEV)
    buf = torch.full((off,), 9, dtype=torch.uint8, device=DEV)
    check(L.sfron_dropout_mask_batch(7, ptr(cnt), ptr(table), len(items), max(sizes), p, ptr(buf), stream_ptr()), "dropout_mask_batch")
    for salt, sz, o in items:
        assert torch.equal(buf[o:o + sz], draw(7, salt, size=sz)), (salt, sz)
        assert bool((buf[o + sz:o + (sz + 3) // 4 * 4] == 9).all())                             # nothing written past a mask's end


def test_unet_draws_its_dropout_masks_in_one_launch_from_the_second_pass_on():
    """Conditional_Model in training mode draws its own nn.Dropout masks (models/diffusion.py:131): the first pass at a batch size asks mask by
    mask and records the requests, later passes draw them with ONE launch -- the masks of a pass depend on (seed, pass counter, block) only,
    so two models with the same seed see the same masks whichever way they were drawn: same outputs bit for bit."""
    from sfron import unet
    outs = []
    for warm in (0, 1):
        torch.manual_seed(123)
        model = unet.Conditional_Model(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.3, resolution=16, n_classes=10)
        model.train()
        g = torch.Generator().manual_seed(5)
        x, t, c = torch.randn(4, 3, 16, 16, generator=g).to(DEV), torch.randint(0, 1000, (4,), generator=g).float().to(DEV), torch.randint(0, 10, (4,), generator=g).to(DEV)
        keep = torch.ones(4, dtype=torch.uint8, device=DEV)
        if warm:                                   # an extra pass first: the compared pass is then a batched one -- rewind its counter
            model(x, t, c, keep_mask=keep)
            assert model._drop_plans and 4 in model._drop_plans
            model._drop_counter.zero_()
        outs.append(model(x, t, c, keep_mask=keep).detach().clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C,HW,nsl,drop", [(256, 64, 3, True), (256, 256, 2, False), (512, 16, 6, False), (128, 256, 1, True), (384, 64, 4, False)])
def test_groupnorm_finishes_a_split_convolution_itself(C, HW, nsl, drop):
    """sfron_groupnorm_{fwd,bwd_res,bwd_cast}_src (round 6): the GroupNorm input handed over as the unfinished result of a split-K product
    (n slabs + bias + per-sample vector + residual).  The one-launch kernels form each element in their first pass and store it where the
    finish launch would have: x / dy, y, mean, rstd, dx (with accumulate and the extra term), the bf16 operand form, its column sums and the
    parameter-gradient partials all equal sfron_split_finish followed by the plain entry point, bit for bit."""
    import ctypes
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    B = 64
    assert L.sfron_groupnorm_one_launch(B, HW, C, 32) == 1 and L.sfron_groupnorm_one_launch(2, 4096, 320, 32) == 0
    g = torch.Generator().manual_seed(C + HW + nsl)
    rows = B * HW
    slabs = (torch.randn(nsl, rows, C, generator=g) * 0.7).to(DEV)
    bias, vec, resid = torch.randn(C, generator=g).to(DEV), torch.randn(B, C + 8, generator=g).to(DEV), torch.randn(rows, C, generator=g).to(DEV)
    gamma, beta = (torch.randn(C, generator=g) * 0.5 + 1.0).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    mask = (torch.rand(rows, C, generator=g) > 0.1).to(torch.uint8).to(DEV) if drop else None
    scale = 1.0 / 0.9 if drop else 1.0

    def source(with_extras):
        src = _lib.SplitSrc()
        src.slabs, src.n_slabs, src.slab_stride = slabs.data_ptr(), nsl, rows * C
        if with_extras:
            src.bias, src.sample_vec, src.ld_vec, src.resid, src.ld_resid = bias.data_ptr(), vec.data_ptr(), C + 8, resid.data_ptr(), C
        return src

    ws = torch.empty(L.sfron_groupnorm_scratch_bytes(B, HW, C, 32) // 8 + 2, dtype=torch.float64, device=DEV)
    # ---- forward
    src = source(True)
    x0 = torch.full((rows, C), float("nan"), device=DEV)
    check(L.sfron_split_finish(ctypes.byref(src), rows, C, HW, ptr(x0), None, C, stream_ptr()), "split_finish")
    want = slabs.double().sum(0) + bias.double() + vec[:, :C].double().repeat_interleave(HW, 0) + resid.double()
    np.testing.assert_allclose(x0.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    y0, m0, r0 = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV), torch.empty(B * 32, device=DEV), torch.empty(B * 32, device=DEV)
    check(L.sfron_groupnorm_fwd(ptr(x0), C, ptr(gamma), ptr(beta), B, HW, C, 32, 1e-6, 1, ptr(mask), scale, ptr(y0), ptr(m0), ptr(r0), ptr(ws),
                                stream_ptr()), "gn_fwd")
    x1 = torch.full((rows, C), float("nan"), device=DEV)
    y1, m1, r1 = torch.empty_like(y0), torch.empty_like(m0), torch.empty_like(r0)
    check(L.sfron_groupnorm_fwd_src(ctypes.byref(src), ptr(x1), ptr(gamma), ptr(beta), B, HW, C, 32, 1e-6, 1, ptr(mask), scale, ptr(y1), ptr(m1), ptr(r1),
                                    stream_ptr()), "gn_fwd_src")
    for a, b in ((x0, x1), (y0, y1), (m0, m1), (r0, r1)):
        assert torch.equal(a, b)
    # ---- backward, fp32 form with accumulate + extra: dy = the slabs alone (an input-gradient convolution has no epilogue terms)
    src = source(False)
    dy0 = torch.full((rows, C), float("nan"), device=DEV)
    check(L.sfron_split_finish(ctypes.byref(src), rows, C, HW, ptr(dy0), None, C, stream_ptr()), "split_finish")
    extra = torch.randn(rows, C + 4, generator=g).to(DEV)
    dx0, pg0, pb0 = torch.full((rows, C), 0.5, device=DEV), torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    check(L.sfron_groupnorm_bwd_res(ptr(dy0), ptr(x0), C, ptr(gamma), ptr(beta), ptr(m0), ptr(r0), B, HW, C, 32, 1, ptr(mask), scale, ptr(dx0), C, 1,
                                    ptr(extra), C + 4, ptr(pg0), ptr(pb0), ptr(ws), stream_ptr()), "gn_bwd_res")
    dy1 = torch.full((rows, C), float("nan"), device=DEV)
    dx1, pg1, pb1 = torch.full((rows, C), 0.5, device=DEV), torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    check(L.sfron_groupnorm_bwd_res_src(ctypes.byref(src), ptr(dy1), ptr(x0), C, ptr(gamma), ptr(beta), ptr(m0), ptr(r0), B, HW, C, 32, 1, ptr(mask), scale,
                                        ptr(dx1), C, 1, ptr(extra), C + 4, ptr(pg1), ptr(pb1), stream_ptr()), "gn_bwd_res_src")
    for a, b in ((dy0, dy1), (dx0, dx1), (pg0, pg1), (pb0, pb1)):
        assert torch.equal(a, b)
    # ---- backward, bf16 operand form + column sums
    nch = L.sfron_groupnorm_chunks(B, HW)
    d16a, cpa = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV), torch.full((B * nch, C), float("nan"), device=DEV)
    check(L.sfron_groupnorm_bwd_cast(ptr(dy0), ptr(x0), C, ptr(gamma), ptr(beta), ptr(m0), ptr(r0), B, HW, C, 32, 1, ptr(mask), scale, ptr(d16a), ptr(cpa),
                                     ptr(pg0), ptr(pb0), ptr(ws), stream_ptr()), "gn_bwd_cast")
    dy2 = torch.full((rows, C), float("nan"), device=DEV)
    d16b, cpb = torch.empty_like(d16a), torch.full((B * nch, C), float("nan"), device=DEV)
    check(L.sfron_groupnorm_bwd_cast_src(ctypes.byref(src), ptr(dy2), ptr(x0), C, ptr(gamma), ptr(beta), ptr(m0), ptr(r0), B, HW, C, 32, 1, ptr(mask), scale,
                                         ptr(d16b), ptr(cpb), ptr(pg1), ptr(pb1), stream_ptr()), "gn_bwd_cast_src")
    for a, b in ((dy0, dy2), (d16a, d16b), (cpa, cpb), (pg0, pg1), (pb0, pb1)):
        assert torch.equal(a, b)
    # a shape outside the one-launch rule: nothing launched, the caller falls back
    small = _lib.SplitSrc()
    small.slabs, small.n_slabs, small.slab_stride = slabs.data_ptr(), 1, 2 * HW * C
    assert L.sfron_groupnorm_fwd_src(ctypes.byref(small), ptr(x1), ptr(gamma), ptr(beta), 2, HW, C, 32, 1e-6, 1, None, 1.0, ptr(y1), ptr(m1), ptr(r1),
                                     stream_ptr()) == 1002


def test_resblock_groupnorms_absorb_the_split_finishes_bitwise():
    """ResnetBlocks at a DDPM width (256 channels, 8 x 8, batch 64: the 3 x 3 convolutions and their input-gradient convolutions split their
    contraction): with _TapeNet.FUSE_SPLIT_FINISH the finish launches between a convolution and the GroupNorm that follows are gone -- norm2
    forms conv1's output, norm2's / norm1's backward form the two input gradients -- and the output and every gradient equal the unfused
    tape bit for bit."""
    from sfron import unet
    cfg = dict(ch=128, ch_mult=(2,), num_res_blocks=1, attn_resolutions=(), dropout=0.0, resolution=8, n_classes=10)      # (the class only runs at ch = 128)
    _, model = _pair(cfg, seed=47)
    model.train()
    g = torch.Generator().manual_seed(4)
    B = 64
    x, t = torch.randn(B, 3, 8, 8, generator=g).to(DEV), torch.randint(0, 1000, (B,), generator=g).float().to(DEV)
    c, keep = torch.randint(0, 10, (B,), generator=g).to(DEV), torch.ones(B, dtype=torch.uint8, device=DEV)
    w = torch.randn(B, 3, 8, 8, generator=g).to(DEV)
    res, calls, real = {}, {"n": 0}, unet.split_source

    def counting(*a, **k):
        r = real(*a, **k)
        calls["n"] += r is not None
        return r
    try:
        unet.split_source = counting
        for fuse in (True, False, True):
            model.FUSE_SPLIT_FINISH = fuse
            model.grads.fill_(float("nan"))
            n0 = calls["n"]
            out, bwd = model._run(x, t, c, keep, None, need_grad=True)
            bwd(w.clone())
            res.setdefault(fuse, []).append((out.clone(), model.grads.clone(), calls["n"] - n0))
    finally:
        unet.split_source = real
        model.FUSE_SPLIT_FINISH = True
    assert res[True][0][2] >= 6 and res[False][0][2] == 0          # the fused tape did hand split results to GroupNorms
    for a, b in ((res[True][0], res[False][0]), (res[True][0], res[True][1])):
        assert torch.isfinite(a[0]).all() and torch.equal(a[0], b[0])
        bad = [n for n in model.index if not torch.equal(model.view(a[1], n), model.view(b[1], n))]
        assert not bad, bad[:8]


@pytest.mark.parametrize("B", [8, 64])
def test_parameter_gradient_finishes_in_one_launch_equal_the_separate_launches(B):
    """sfron_reduce_batch (round 6): the fixed-order sums that finish GroupNorm affine gradients, conv1 bias gradients and the per-sample
    projection gradients are collected by the tape and issued as one launch per backward pass (items by value in the kernel arguments).  Each
    item is summed in the order its own sfron_reduce_chunks launch would use: every gradient equals the unbatched pass bit for bit -- at
    batch 8 (two-phase GroupNorm, chunked partials) and at batch 64 (one-launch GroupNorm, the wide reduction form)."""
    _, model = _pair(dict(SMALL, dropout=0.0), seed=43)
    model.train()
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(B, 3, 16, 16, generator=g).to(DEV), torch.randint(0, 1000, (B,), generator=g).float().to(DEV)
    c, keep = torch.randint(0, 10, (B,), generator=g).to(DEV), torch.ones(B, dtype=torch.uint8, device=DEV)
    w = torch.randn(B, 3, 16, 16, generator=g).to(DEV)
    grads = {}
    assert model.BATCH_REDUCTIONS is True
    try:
        for batched in (True, False, True):
            model.BATCH_REDUCTIONS = batched
            model.grads.fill_(float("nan"))
            out, bwd = model._run(x, t, c, keep, None, need_grad=True)
            bwd(w.clone())
            assert model._red is None
            grads.setdefault(batched, []).append(model.grads.clone())
    finally:
        model.BATCH_REDUCTIONS = True
    for n in model.index:
        a, b, a2 = (model.view(v, n) for v in (grads[True][0], grads[False][0], grads[True][1]))
        assert torch.isfinite(a).all(), n
        assert torch.equal(a, b) and torch.equal(a, a2), n


def test_reduce_batch_many_items_of_both_forms():
    """More items than one launch carries (120), both summation forms, odd widths, strided outputs: against sfron_reduce_chunks item by item."""
    import ctypes
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    shapes = [(1, 64, 128), (1, 512, 320), (64, 8, 256), (3, 5, 7), (1, 1, 1280), (2, 130, 66), (1, 96, 1000)] * 20      # 140 items
    items, want, keep = (_lib.ReduceItem * len(shapes))(), [], []
    for it, (groups, per, D) in zip(items, shapes):
        P = torch.randn(groups * per, D, generator=g).to(DEV)
        ld = D + 4
        o1 = torch.full((groups, ld), 7.0, device=DEV)
        o2 = torch.full((groups, ld), 7.0, device=DEV)
        check(L.sfron_reduce_chunks(ptr(P), groups, per, D, o1.data_ptr(), ld, 0, stream_ptr()), "reduce_chunks")
        it.partials, it.out, it.groups, it.per_group, it.D, it.ldout = P.data_ptr(), o2.data_ptr(), groups, per, D, ld
        want.append(o1); keep.append((P, o2))
    check(L.sfron_reduce_batch(ctypes.cast(items, ctypes.c_void_p), len(shapes), stream_ptr()), "reduce_batch")
    torch.cuda.synchronize()
    for o1, (_, o2) in zip(want, keep):
        assert torch.equal(o1, o2)
    assert L.sfron_reduce_batch(None, 1, stream_ptr()) != 0


def test_unet_forward_backward_bitwise_reproducible():
    """Three repetitions of the same forward + backward with different garbage in freed memory in between: every output and every
    gradient bit agrees (no floating-point atomics, no read of uninitialised scratch; the up-path GroupNorms have 12 channels per
    group, the case that used to go through an LDS atomic)."""
    _, model = _pair(dict(SMALL, dropout=0.0), seed=40)
    model.train()
    g = torch.Generator().manual_seed(1)
    B = 8
    x, t = torch.randn(B, 3, 16, 16, generator=g).to(DEV), torch.randint(0, 1000, (B,), generator=g).float().to(DEV)
    c, keep = torch.randint(0, 10, (B,), generator=g).to(DEV), torch.ones(B, dtype=torch.uint8, device=DEV)
    w = torch.randn(B, 3, 16, 16, generator=g).to(DEV)
    outs, grads = [], []
    for rep in range(3):
        junk = torch.randn(1 << 22, device=DEV) * (rep + 1)
        del junk
        out, bwd = model._run(x, t, c, keep, None, need_grad=True)
        bwd(w.clone())
        outs.append(out.clone()); grads.append(model.grads.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for r in (1, 2):
        bad = [n for n in model.index if not torch.equal(model.view(grads[0], n), model.view(grads[r], n))]
        assert not bad, bad[:8]


def test_conv_tiles_loader_wave_form_is_bit_identical_to_the_shared_wave_form():
    """The pipelined convolution tiles (csrc/conv.hip k_cgemm / k_cgemm_t) run by default with four extra waves that issue every
    LDS-DMA and do the im2col address arithmetic; the eight multiplying waves see the same tiles in the same order, so a whole
    U-Net forward + backward (cifar10_sfron.yml model, batch 16: contractions of >= 8 K-tiles in all three
    products) gives the same bits as the form in which every wave issues its share (sfron_gemm_loader_waves(10))."""
    from sfron import _lib
    _, model = _pair(dict(CIFAR, dropout=0.0), seed=41)
    model.train()
    g = torch.Generator().manual_seed(2)
    B = 16
    x, t = torch.randn(B, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 1000, (B,), generator=g).float().to(DEV)
    c, keep = torch.randint(0, 10, (B,), generator=g).to(DEV), torch.ones(B, dtype=torch.uint8, device=DEV)
    w = torch.randn(B, 3, 32, 32, generator=g).to(DEV)
    L = _lib.lib()
    res = {}
    old = L.sfron_gemm_loader_waves(10)
    try:
        assert old == 4
        for form in (10, 4):
            L.sfron_gemm_loader_waves(form)
            out, bwd = model._run(x, t, c, keep, None, need_grad=True)
            bwd(w.clone())
            torch.cuda.synchronize()
            res[form] = (out.clone(), model.grads.clone())
    finally:
        L.sfron_gemm_loader_waves(old)
    assert torch.isfinite(res[4][0]).all() and torch.equal(res[10][0], res[4][0])
    bad = [n for n in model.index if not torch.equal(model.view(res[10][1], n), model.view(res[4][1], n))]
    assert not bad, bad[:8]


def test_unet_test_mode_guidance_matches_oracle():
    ref, model = _pair(SMALL, seed=5)
    ref.eval(); model.eval()
    g = torch.Generator().manual_seed(6)
    x, t, c = torch.randn(3, 3, 16, 16, generator=g), torch.tensor([0.0, 500.0, 999.0]), torch.tensor([1, 9, 4])
    with torch.no_grad():
        want = ref(x, t, c, mode="test", cond_scale=2.0)
        want0 = ref(x, t, c, mode="test", cond_scale=0)
    got = model(x.to(DEV), t.to(DEV), c.to(DEV), mode="test", cond_scale=2.0)
    got0 = model(x.to(DEV), t.to(DEV), c.to(DEV), mode="test", cond_scale=0)
    assert _rel(got, want) < 2e-2 and _rel(got0, want0) < 1.5e-2


def _synthetic(step, stream, B, g):
    x0 = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    e = torch.randn(B, 3, 32, 32, generator=g)
    t = torch.randint(0, 1000, (B // 2 + 1,), generator=g)
    t = torch.cat([t, 1000 - t - 1])[:B]                        # antithetic (runners/diffusion.py:1091-1094)
    c = torch.zeros(B, dtype=torch.int64) if stream == "forget" else torch.randint(1, 10, (B,), generator=g)
    return dict(x0=x0, e=e, t=t, c=c)


@pytest.mark.parametrize("loss", ["adaga", "ga", "rl"])
def test_ddpm_sfron_iterations_native_denoiser_vs_oracle(loss):
    """DDPM/runners/diffusion.py:1075-1180 on the NATIVE denoiser (no torch-autograd module anywhere on the HIP side): cosine-decayed
    alpha, forget loss ga / adaga / rl, mask -> clip -> Adam, remain loss -> clip -> Adam, EMAHelper; dropout off (p = 0) and
    the classifier-free keep masks handed to both sides."""
    from oracle import sfron_ref
    from sfron import ddpm
    cfg = dict(CIFAR, dropout=0.0)
    ref, model = _pair(cfg, seed=21)
    B, n_it = 8, 3
    gm = torch.Generator().manual_seed(22)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters()}
    betas = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    hp = dict(lr=1e-4, forget_alpha=10.0, grad_clip=1.0, mask=mask, unlearn_loss=loss, lambd=0.5, n_iters=n_it, decay_forget_alpha=True)
    fwd = ref.forward
    ref.forward = lambda x, tf, c, drop: fwd(x, tf, c, mode="train", keep_mask=~drop)      # the oracle loop's model signature
    orc = sfron_ref.DDPMSfronOracle(ref, betas, ema_mu=1e-4, label_to_forget=0, **hp)
    run = ddpm.DDPMSFRon(model, betas=betas.to(DEV), ema_rate=1e-4, label_to_forget=0, **hp)
    g = torch.Generator().manual_seed(23)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    for it in range(n_it):
        f, r = _synthetic(it, "forget", B, g), _synthetic(it, "remain", B, g)
        for b in (f, r):
            b["drop"] = (torch.rand(B, generator=g) < 0.1)              # True = label dropped (null embedding)
        want = orc.step(it, {k: v for k, v in f.items()}, {k: v for k, v in r.items()})
        fd = {k: v.to(DEV) for k, v in f.items()}; rd = {k: v.to(DEV) for k, v in r.items()}
        fd["keep_mask"], rd["keep_mask"] = ~f["drop"], ~r["drop"]
        got = run.step(it, fd, rd)
        assert got["forget_loss"].item() == pytest.approx(want["forget_loss"], rel=3e-2, abs=1e-3)
        assert got["remain_loss"].item() == pytest.approx(want["remain_loss"], rel=3e-2)
    same = tot = 0
    num = den = 0.0
    views = run.flat.named_views(run.flat.p)
    for n, q in ref.named_parameters():
        du_ref = (q.detach() - p0[n]).flatten()
        du = (views[n].cpu() - p0[n]).flatten()
        big = du_ref.abs() > 0.05 * du_ref.abs().max()
        same += int((torch.sign(du[big]) == torch.sign(du_ref[big])).sum()); tot += int(big.sum())
        num += (du - du_ref).double().pow(2).sum().item(); den += du_ref.double().pow(2).sum().item()
    print(f"DDPM {loss}: update sign agreement {same / tot:.4f}, bulk relative error {(num / den) ** 0.5:.3f}")
    assert same / tot > 0.99 and (num / den) ** 0.5 < 0.15          # measured 0.9967..0.9977 / 0.068..0.082
    sh = run.ema_state_dict()
    # EMAHelper (mu = 1e-4: the shadow follows the weights): same relation to the oracle's shadow as the weights have, and the
    # shadow is NOT the weights themselves (it lags by mu * the last update)
    for n in ("conv_in.weight", "mid.attn_1.q.weight", "up.0.block.2.temb_cemb_proj.weight"):
        moved = (orc.shadow[n] - p0[n]).norm().item()
        assert (sh[n].cpu() - orc.shadow[n]).norm().item() < 0.35 * moved, n
        assert not torch.equal(sh[n], views[n])


def test_ddpm_joint_method_vs_oracle():
    """--method joint of the DDPM runner (runners/diffusion.py:1160-1167): one clipped Adam step per iteration on remain_alpha * remain_loss +
    alpha * forget_loss, both graphs from the same weights, the mask loop a no-op (it runs on stale gradients in front of zero_grad()):
    three iterations on a small U-Net against the oracle restating those lines; one optimizer step per iteration; the result differs
    from the "ron" method's."""
    from oracle import sfron_ref
    from sfron import ddpm
    cfg = dict(SMALL, dropout=0.0)
    B, n_it = 8, 3
    betas = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    res = {}
    for method in ("joint", "ron"):
        ref, model = _pair(cfg, seed=51)
        gm = torch.Generator().manual_seed(52)
        mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters()}
        hp = dict(lr=1e-4, forget_alpha=10.0, grad_clip=1.0, mask=mask, unlearn_loss="adaga", lambd=0.5, n_iters=n_it, decay_forget_alpha=True)
        fwd = ref.forward
        ref.forward = lambda x, tf, c, drop, fwd=fwd: fwd(x, tf, c, mode="train", keep_mask=~drop)
        orc = sfron_ref.DDPMSfronOracle(ref, betas, ema_mu=1e-4, method=method, **hp)
        run = ddpm.DDPMSFRon(model, betas=betas.to(DEV), ema_rate=1e-4, method=method, **hp)
        g = torch.Generator().manual_seed(53)
        p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
        for it in range(n_it):
            f, r = _synthetic(it, "forget", B, g), _synthetic(it, "remain", B, g)
            for b in (f, r):
                b["x0"], b["e"] = b["x0"][:, :, :16, :16].contiguous(), b["e"][:, :, :16, :16].contiguous()
                b["drop"] = (torch.rand(B, generator=g) < 0.1)
            want = orc.step(it, dict(f), dict(r))
            fd = {k: v.to(DEV) for k, v in f.items()}; rd = {k: v.to(DEV) for k, v in r.items()}
            fd["keep_mask"], rd["keep_mask"] = ~f["drop"], ~r["drop"]
            got = run.step(it, fd, rd)
            assert got["forget_loss"].item() == pytest.approx(want["forget_loss"], rel=3e-2, abs=1e-3)
            assert got["remain_loss"].item() == pytest.approx(want["remain_loss"], rel=3e-2)
        assert run.opt.step_count == (n_it if method == "joint" else 2 * n_it)
        same = tot = 0
        views = run.flat.named_views(run.flat.p)
        for n, q in ref.named_parameters():
            if n.endswith(".k.bias"):
                continue
            du_ref, du = (q.detach() - p0[n]).flatten(), (views[n].cpu() - p0[n]).flatten()
            big = du_ref.abs() > 0.05 * du_ref.abs().max()
            same += int((torch.sign(du[big]) == torch.sign(du_ref[big])).sum()); tot += int(big.sum())
        assert same / tot > 0.985, (method, same / tot)
        res[method] = run.flat.p.clone()
    assert not torch.equal(res["joint"], res["ron"])
    with pytest.raises(ValueError):
        ddpm.DDPMSFRon(model, method="alternate")


@pytest.mark.parametrize("loss", ["adaga", "rl"])
def test_ddpm_graph_replay_matches_eager(loss):
    """The forget / remain stages replayed as HIP graphs (sfron.graphs; the decayed alpha as a device scalar) against the eager loop:
    dropout off and explicit keep masks, so both runs see identical inputs -- parameters, EMA and losses agree to fp32 rounding of
    alpha (a python double in the eager loop, an fp32 device scalar in the graph)."""
    from sfron import ddpm
    cfg = dict(SMALL, dropout=0.0)
    B, n_it = 8, 4
    g = torch.Generator().manual_seed(41)
    batches = []
    for it in range(n_it):
        pair = []
        for stream in ("forget", "remain"):
            b = _synthetic(it, stream, B, g)
            b["x0"], b["e"] = b["x0"][:, :, :16, :16].contiguous(), b["e"][:, :, :16, :16].contiguous()
            b["keep_mask"] = (torch.rand(B, generator=g) >= 0.1).to(torch.uint8)
            pair.append({k: v.to(DEV) for k, v in b.items()})
        batches.append(pair)
    res = []
    for use in (False, True):
        _, model = _pair(cfg, seed=40)
        run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, unlearn_loss=loss, n_iters=n_it, use_graphs=use)
        losses = [run.step(it, *batches[it]) for it in range(n_it)]
        if use:
            assert run._graphs["forget"].graph is not None and run._graphs["remain"].graph is not None
        res.append((run.flat.p.clone(), run.shadow.clone(), [(l["forget_loss"].item(), l["remain_loss"].item()) for l in losses]))
    for (fa, ra), (fb, rb) in zip(res[0][2], res[1][2]):
        assert fa == pytest.approx(fb, rel=1e-5) and ra == pytest.approx(rb, rel=1e-5)
    dp = (res[0][0] - res[1][0]).abs().max().item()
    assert dp < 2e-5, dp                                           # Adam steps of lr 1e-4: a flipped update would show as 2e-4
    assert (res[0][1] - res[1][1]).abs().max().item() < 1e-6


def test_ddpm_graph_recapture_at_a_new_batch_size_builds_no_dropout_plan_inside_the_capture():
    """graphs.StageGraph re-captures WITHOUT a warm-up pass when the input signature changes (a new batch size after the first
    capture), so the first pass at that batch size runs under stream capture -- where the dropout plan's device table (a pageable
    host-to-device copy) must not be built: the pass keeps its per-mask launches, no plan appears for the new size, the plan of the first
    size keeps its table, and the loop goes on with finite losses at both sizes."""
    from sfron import ddpm
    cfg = dict(SMALL, dropout=0.1)
    g = torch.Generator().manual_seed(43)

    def pair(it, B):
        out = []
        for stream in ("forget", "remain"):
            b = _synthetic(it, stream, B, g)
            b["x0"], b["e"] = b["x0"][:, :, :16, :16].contiguous(), b["e"][:, :, :16, :16].contiguous()
            out.append({k: v.to(DEV) for k, v in b.items()})
        return out
    _, model = _pair(cfg, seed=44)
    run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, unlearn_loss="ga", n_iters=12, use_graphs=True)
    for it in range(4):                                   # eager warm-up passes build the plan of batch 8, then capture + replay
        run.step(it, *pair(it, 8))
    assert run._graphs["forget"].graph is not None and 8 in model._drop_plans
    table8 = model._drop_plans[8]["table"].clone()
    outs = [run.step(4 + it, *pair(4 + it, 4)) for it in range(3)]          # new signature: re-capture with no warm-up, then two replays
    assert 4 not in model._drop_plans                                       # nothing was built while the stream was capturing
    outs += [run.step(7 + it, *pair(7 + it, 8)) for it in range(2)]         # and back
    torch.cuda.synchronize()
    assert torch.equal(model._drop_plans[8]["table"], table8)
    for o in outs:
        assert torch.isfinite(o["forget_loss"]).item() and torch.isfinite(o["remain_loss"]).item()
    assert torch.isfinite(run.flat.p).all().item()


def test_ddpm_fisher_clip_before_square_guided_forward_vs_oracle():
    """DDPM/runners/diffusion.py:1244-1299: Fisher diagonal from the cond_scale-guided mode="test" forward (gradients through the
    conditional AND the null branch), gradients clipped to norm 1 before they are squared; then the saliency mask
    (DDPM/generate_fisher_mask.py:39-46) from a forget / remain pair of such Fisher estimates."""
    from oracle import sfron_ref, sweep_ref
    from sfron import fisher
    ref, model = _pair(dict(SMALL, dropout=0.0), seed=31)
    betas = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    g = torch.Generator().manual_seed(32)

    def batch(stream):
        bt = _synthetic(0, stream, 6, g)
        return {k: (v[:, :, :16, :16].contiguous() if v.dim() == 4 else v) for k, v in bt.items()}
    sets = {s: [batch(s) for _ in range(2)] for s in ("forget", "remain")}
    F_ref = {s: sfron_ref.ddpm_fisher(ref, sets[s], betas, cond_scale=2.0, grad_clip=1.0) for s in sets}
    F_hip = {}
    for s in sets:
        acc = fisher.DDPMFisherAccumulator(model, betas.to(DEV), n_batches=2, cond_scale=2.0, grad_clip=1.0)
        for bt in sets[s]:
            acc.accumulate({k: v.to(DEV) for k, v in bt.items()})
        F_hip[s] = acc.state_dict(prefix="")
    num = den = 0.0
    for n, fr in F_ref["forget"].items():
        fh = F_hip["forget"][n]
        num += (fh - fr).double().pow(2).sum().item(); den += fr.double().pow(2).sum().item()
        if not n.endswith(".k.bias"):
            assert ((fh - fr).norm() / (fr.norm() + 1e-30)).item() < 0.12, n          # g^2 doubles the bf16 gradient error
    assert (num / den) ** 0.5 < 0.05
    masks = fisher.masks_from_fisher(F_hip["forget"], F_hip["remain"], 1.0)
    m_ref = sweep_ref.mask_from_fisher(F_ref["forget"]["mid.block_1.conv1.weight"], F_ref["remain"]["mid.block_1.conv1.weight"], 1.0)
    agree = (masks["mid.block_1.conv1.weight"] == m_ref).float().mean().item()
    print(f"DDPM Fisher: bulk relative error {(num / den) ** 0.5:.3f}; saliency mask agreement on mid.block_1.conv1.weight {agree:.4f}")
    assert agree > 0.9


def test_config1_fifty_sfron_steps_batch64_native():
    """BASELINE config 1 on the HIP path: DDPM CIFAR-10 class-forget, 50 SFR-on steps, batch 64, cifar10_sfron.yml model and
    hyper-parameters (adaga, lambd 0.5, alpha 10 cosine-decayed, Adam 1e-4, clip 1.0, EMA 1e-4), synthetic inputs."""
    import time
    from sfron import ddpm, unet
    torch.manual_seed(1234)
    model = unet.Conditional_Model(unet.config_namespace())
    gm = torch.Generator().manual_seed(0)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in model.named_parameters()}
    run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, mask=mask, unlearn_loss="adaga", lambd=0.5,
                         n_iters=50, decay_forget_alpha=True)
    g = torch.Generator().manual_seed(1)
    p0 = run.flat.p.clone()
    batches = [({k: v.to(DEV) for k, v in _synthetic(i, "forget", 64, g).items()}, {k: v.to(DEV) for k, v in _synthetic(i, "remain", 64, g).items()})
               for i in range(4)]
    losses = []
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(50):
        out = run.step(it, *batches[it % 4])
        losses.append((out["forget_loss"], out["remain_loss"]))
    torch.cuda.synchronize()
    dt = time.time() - t0
    fl = torch.stack([a for a, _ in losses]).cpu()
    rl = torch.stack([b for _, b in losses]).cpu()
    print(f"config 1: 50 SFR-on steps, batch 64: {dt / 50 * 1e3:.1f} ms/step; remain loss {rl[0].item():.1f} -> {rl[-1].item():.1f}; "
          f"forget loss {fl[0].item():.1f} -> {fl[-1].item():.1f}")
    assert torch.isfinite(fl).all() and torch.isfinite(rl).all()
    assert torch.isfinite(run.flat.p).all() and not torch.equal(run.flat.p, p0)
    assert rl[-5:].mean() < rl[:5].mean()             # the remain stage keeps fitting the remaining classes
    assert run.opt.step_count == 100


def test_unet_test_mode_is_differentiable_like_the_reference():
    """mode="test" (models/diffusion.py:340-357) carries gradients through BOTH guidance branches, as the reference's Fisher loop needs
    (runners/diffusion.py:1260-1276): every parameter gradient of sum(w * ((1 + s) f_cond - s f_null)) against the oracle's autograd,
    for s = 2 and for s = 0 (conditional branch alone); under no_grad the result carries no graph."""
    ref, model = _pair(SMALL, seed=6)
    ref.eval(); model.eval()
    g = torch.Generator().manual_seed(16)
    x, t, c = torch.randn(2, 3, 16, 16, generator=g), torch.tensor([3.0, 700.0]), torch.tensor([1, 2])
    w = torch.randn(2, 3, 16, 16, generator=g) * 0.1
    for s_ in (2.0, 0.0):
        ref.zero_grad()
        want = ref(x, t, c, mode="test", cond_scale=s_)
        (want * w).sum().backward()
        model.zero_grad()
        out = model(x.to(DEV), t.to(DEV), c.to(DEV), mode="test", cond_scale=s_)
        assert out.requires_grad and _rel(out, want) < (3e-2 if s_ else 1.5e-2)
        (out * w.to(DEV)).sum().backward()
        dots = na = nb = 0.0
        gmed = float(torch.tensor([q.grad.norm().item() for q in ref.parameters() if q.grad is not None]).median())
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            if q.grad is None:
                continue
            ga, gb = p.grad.detach().cpu().flatten().double(), q.grad.flatten().double()
            dots += torch.dot(ga, gb).item(); na += ga.pow(2).sum().item(); nb += gb.pow(2).sum().item()
            if gb.norm().item() < 2e-3 * gmed:       # analytically zero (a key bias, a constant in front of a normalisation): noise on both sides
                continue
            # the two branches' bf16 errors add with weights (1 + s) and s while the difference is no larger than one branch
            assert ((ga - gb).norm() / gb.norm()).item() < (0.15 if s_ else 6e-2), (s_, n, ((ga - gb).norm() / gb.norm()).item())
        assert dots / math.sqrt(na * nb) > (0.998 if s_ else 0.9995), (s_, dots / math.sqrt(na * nb))
    with torch.no_grad():
        assert not model(x.to(DEV), t.to(DEV), c.to(DEV), mode="test", cond_scale=2.0).requires_grad


def test_copy_cols2_is_cat_and_its_backward():
    """sfron_copy_cols2: torch.cat([h, skip], dim=1) of the up path (models/diffusion.py:403) in one launch, and the backward that splits the
    gradient into the two sources' buffers (one overwritten, one accumulated into) -- against torch, bit for bit; an unaligned width takes the
    two plain launches behind the same entry point."""
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for rows, c1, c2 in ((4096, 256, 128), (1000, 128, 256), (64, 6, 10)):
        a, b = torch.randn(rows, c1, generator=g).to(DEV), torch.randn(rows, c2, generator=g).to(DEV)
        cat = torch.full((rows, c1 + c2), float("nan"), device=DEV)
        check(L.sfron_copy_cols2(ptr(a), c1, c1, ptr(cat), c1 + c2, 0, ptr(b), c2, c2, cat.data_ptr() + 4 * c1, c1 + c2, 0, rows, stream_ptr()), "copy_cols2")
        assert torch.equal(cat, torch.cat([a, b], dim=1))
        dcat = torch.randn(rows, c1 + c2, generator=g).to(DEV)
        ga, gb0 = torch.full((rows, c1), float("nan"), device=DEV), torch.randn(rows, c2, generator=g).to(DEV)
        gb = gb0.clone()
        check(L.sfron_copy_cols2(ptr(dcat), c1 + c2, c1, ptr(ga), c1, 0, dcat.data_ptr() + 4 * c1, c1 + c2, c2, ptr(gb), c2, 1, rows, stream_ptr()), "copy_cols2")
        assert torch.equal(ga, dcat[:, :c1]) and torch.equal(gb, gb0 + dcat[:, c1:])
