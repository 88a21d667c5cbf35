"""GPU parity of the Stable-Diffusion path (sfron.sd_unet / sfron.sd over csrc/conv.hip) -- BASELINE config 4:
(1) the kernels the LDM UNet adds to the DDPM set -- LayerNorm (affine) forward / backward, GEGLU (erf GELU), the head-batched
    attention GEMMs with a padded context -- against torch fp32;
(2) UNetModel forward + backward against the oracle (oracle/sd_ref.py, pinned to SD/ldm/modules/... by tests/golden/sd_unet.npz):
    a small config and a three-level one with a 77-token context (padded to 80 inside);
(3) the loop body of SD/train-scripts/nsfw_removal.py:108-173 (train_method xattn / full, mask as written / intended) against
    SDSfronOracle.
Tolerances: fp32 kernels 1e-5 .. 2e-4; bf16-operand GEMM paths as in tests/test_gpu_unet.py."""
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


@pytest.mark.parametrize("rows,D", [(256, 320), (100, 64), (4096, 1280)])
def test_layernorm_affine_fwd_bwd(rows, D):
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    gam, bet = torch.randn(D, generator=g) * 0.3 + 1, torch.randn(D, generator=g) * 0.2
    xt, gt, bt = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.layer_norm(xt, (D,), gt, bt, eps=1e-5)
    dy = torch.randn(rows, D, generator=g) * 0.1
    ref.backward(dy)
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    xd, gd, bd = x.to(DEV), gam.to(DEV), bet.to(DEV)
    check(L.sfron_layernorm_fwd(ptr(xd), ptr(gd), ptr(bd), rows, D, 1e-5, ptr(y), ptr(mean), ptr(rstd), stream_ptr()), "ln_fwd")
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.detach().numpy(), rtol=1e-2, atol=1e-2)
    rpb = L.sfron_layernorm_rows_per_block(rows)
    nblk = (rows + rpb - 1) // rpb
    dx = torch.full((rows, D), 0.25, device=DEV)
    pg, pb = torch.empty(nblk, D, device=DEV), torch.empty(nblk, D, device=DEV)
    dyd = dy.to(DEV)
    check(L.sfron_layernorm_bwd(ptr(dyd), ptr(xd), ptr(gd), ptr(mean), ptr(rstd), rows, D, ptr(dx), 1, ptr(pg), ptr(pb), stream_ptr()), "ln_bwd")
    np.testing.assert_allclose(dx.cpu().numpy() - 0.25, xt.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pg.sum(0).cpu().numpy(), gt.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(pb.sum(0).cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-4)
    # the residual form: dx (+)= extra + gradient in one pass = the bits of dx (+)= extra, then dx += gradient; the parameter-gradient
    # partials finish on the 16-wave form of sfron_reduce2 when there are >= 64 of them
    extra = torch.randn(rows, D, generator=g).to(DEV)
    for acc in (0, 1):
        two = torch.full((rows, D), 0.5, device=DEV)
        one = two.clone()
        check(L.sfron_copy_cols(ptr(extra), D, rows, D, ptr(two), D, acc, stream_ptr()), "copy_cols")
        check(L.sfron_layernorm_bwd(ptr(dyd), ptr(xd), ptr(gd), ptr(mean), ptr(rstd), rows, D, ptr(two), 1, ptr(pg), ptr(pb), stream_ptr()), "ln_bwd")
        check(L.sfron_layernorm_bwd_res(ptr(dyd), ptr(xd), ptr(gd), ptr(mean), ptr(rstd), rows, D, ptr(one), acc, ptr(extra), ptr(pg), ptr(pb),
                                        stream_ptr()), "ln_bwd_res")
        assert torch.equal(one, two)
    dg, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    check(L.sfron_reduce2(ptr(pg), ptr(pb), 1, nblk, D, ptr(dg), D, ptr(db), D, stream_ptr()), "reduce2")
    np.testing.assert_allclose(dg.cpu().numpy(), gt.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_geglu_fwd_bwd():
    from sfron import _lib
    from sfron._lib import check, ptr, stream_ptr
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    rows, Fd = 300, 256
    h = torch.randn(rows, 2 * Fd, generator=g) * 1.5
    ht = h.clone().requires_grad_(True)
    a, gate = ht.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    d = torch.randn(rows, Fd, generator=g) * 0.2
    ref.backward(d)
    out = torch.empty(rows, Fd, dtype=torch.bfloat16, device=DEV)
    hd, dd = h.to(DEV), d.to(DEV)
    check(L.sfron_geglu_fwd(ptr(hd), rows, Fd, ptr(out), stream_ptr()), "geglu_fwd")
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.detach().numpy(), rtol=1e-2, atol=1e-2)
    dh = torch.empty(rows, 2 * Fd, dtype=torch.bfloat16, device=DEV)
    check(L.sfron_geglu_bwd(ptr(dd), ptr(hd), rows, Fd, ptr(dh), stream_ptr()), "geglu_bwd")
    np.testing.assert_allclose(dh.float().cpu().numpy(), ht.grad.numpy(), rtol=1e-2, atol=2e-3)


SMALL = dict(in_channels=4, out_channels=4, model_channels=32, attention_resolutions=(2, 1), num_res_blocks=1, channel_mult=(1, 2), num_heads=2,
             context_dim=24)
MID = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=(4, 2, 1), num_res_blocks=2, channel_mult=(1, 2, 4), num_heads=4,
           context_dim=40)


def _pair(cfg, seed):
    from oracle import sd_ref
    from sfron import sd_unet
    torch.manual_seed(seed)
    ref = sd_ref.UNetModel(**cfg)
    sd_ref.randomize_zero_init(ref, std=0.05, seed=seed + 1)
    model = sd_unet.UNetModel(**cfg)
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    model.load_state_dict({"model.diffusion_model." + k: v for k, v in ref.state_dict().items()})     # CompVis checkpoint keys
    return ref, model


# head width 160 (v1's 1280-channel levels: 1280 / 8 heads) is wider than the flash kernels take: scores through k_bgemm + k_softmax.
# One level of 320 channels with 2 heads reaches that path at a size the CPU oracle finishes in a second.
HD160 = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(1,), num_res_blocks=1, channel_mult=(1,), num_heads=2,
             context_dim=64)


def _compare_unet(ref, model, x, t, ctx, w, label, out_tol=1.5e-2, grad_tol=6e-2, cos_min=0.9995):
    out_ref = ref(x, timesteps=t, context=ctx)
    (out_ref * w).sum().backward()
    out = model(x.to(DEV), timesteps=t.to(DEV), context=ctx.to(DEV))
    assert out.requires_grad
    (out * w.to(DEV)).sum().backward()
    e_out = _rel(out, out_ref)
    worst, wname = 0.0, ""
    dots = na = nb = 0.0
    gmed = float(torch.tensor([q.grad.norm().item() for q in ref.parameters()]).median())
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        ga, gb = p.grad.detach().cpu().flatten(), q.grad.flatten()
        assert torch.isfinite(ga).all(), n
        dots += torch.dot(ga.double(), gb.double()).item(); na += ga.double().pow(2).sum().item(); nb += gb.double().pow(2).sum().item()
        if gb.norm().item() < 2e-3 * gmed:
            # analytically (near-)zero gradients: a per-channel constant in front of a GroupNorm whose groups are single channels
            # (model_channels 32 / 32 groups) is normalised away, a shift of every key leaves a softmax unchanged -- both sides hold
            # cancellation noise only
            assert ga.norm().item() < 3e-2 * gmed, (n, ga.norm().item(), gmed)
            continue
        e = ((ga - gb).norm() / (gb.norm() + 1e-30)).item()
        if e > worst:
            worst, wname = e, n
    cos = dots / math.sqrt(na * nb)
    print(f"{label}: out rel-L2 {e_out:.3e}, worst grad rel-L2 {worst:.3e} ({wname}), cosine {cos:.6f}")
    assert e_out < out_tol, e_out
    assert worst < grad_tol, (wname, worst)
    assert cos > cos_min, cos


def test_sd_v1_unet_forward_backward_vs_oracle_at_full_size():
    """BASELINE config 4 at its real widths: the v1-inference.yaml UNet -- 859 520 964 parameters, channels 320 / 640 / 1280,
    heads of 40 / 80 / 160, 4096-token self-attention at 64 x 64 (the two-kernel flash backward), the 2560 -> 1280 and 1920 -> 640
    concatenation convolutions -- batch 1, 64 x 64 latents, 77-token 768-wide context, forward + backward against oracle.sd_ref (pinned to
    SD/ldm/modules/diffusionmodules/openaimodel.py:428-846 by tests/golden/sd_unet.npz) on the CPU.  Same bounds as the small cases."""
    from oracle import sd_ref
    from sfron import sd_unet
    torch.manual_seed(77)
    ref = sd_ref.UNetModel()
    sd_ref.randomize_zero_init(ref, std=0.02, seed=78)
    assert sum(p.numel() for p in ref.parameters()) == 859_520_964
    model = sd_unet.UNetModel()
    model.load_state_dict({"model.diffusion_model." + k: v for k, v in ref.state_dict().items()})
    ref.train(); model.train()
    g = torch.Generator().manual_seed(79)
    x = torch.randn(1, 4, 64, 64, generator=g)
    t = torch.tensor([437])
    ctx = torch.randn(1, 77, 768, generator=g)
    w = torch.randn(1, 4, 64, 64, generator=g) * 0.1
    _compare_unet(ref, model, x, t, ctx, w, "SD v1 UNet (859.5 M parameters) B=1 64x64 ctx 77x768")


@pytest.mark.parametrize("cfg,B,S,Lc", [(SMALL, 3, 8, 5), (MID, 2, 32, 77), (HD160, 2, 8, 77)])
def test_sd_unet_forward_backward_vs_oracle(cfg, B, S, Lc):
    ref, model = _pair(cfg, seed=B)
    ref.train(); model.train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 4, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ctx = torch.randn(B, Lc, cfg["context_dim"], generator=g)
    w = torch.randn(B, 4, S, S, generator=g) * 0.1
    _compare_unet(ref, model, x, t, ctx, w, f"SD UNet mc={cfg['model_channels']} B={B} {S}x{S} ctx {Lc}")


# 2 x the worst per-tensor value measured on MI355X (round 5; printed by the test): xattn / as_written min cosine 0.9870, norm ratios within
# 1.7 %; full / intended min cosine 0.9117 (out.2.bias), norm ratios within 3.3 %
SD_ORACLE_UPDATE_COS_MIN = {"xattn": 0.974, "full": 0.823}
SD_ORACLE_UPDATE_NORM_TOL = {"xattn": 0.034, "full": 0.066}


@pytest.mark.parametrize("method,mask_mode", [("xattn", "as_written"), ("full", "intended")])
def test_sd_nsfw_removal_iterations_vs_oracle(method, mask_mode):
    from oracle import sd_ref
    from sfron import sd
    ref, model = _pair(SMALL, seed=14)
    gm = torch.Generator().manual_seed(15)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters()}
    hp = dict(lr=1e-4, forget_alpha=1.0, remain_alpha=1.0, train_method=method, mask=mask, mask_mode=mask_mode)
    orc = sd_ref.SDSfronOracle(ref, sd_ref.LDMSchedule(), **hp)
    run = sd.SDSFRon(model, **hp)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    g = torch.Generator().manual_seed(16)
    B, S, Lc = 4, 8, 6
    # tensors whose gradient is exactly zero analytically (a per-channel constant in front of a GroupNorm with one channel per group: the
    # 32-channel levels of SMALL): a probe backward pass of the oracle leaves fp32 cancellation noise there -- found by size, compared by size
    gp = torch.Generator().manual_seed(99)
    (ref(torch.randn(B, 4, S, S, generator=gp), timesteps=torch.randint(0, 1000, (B,), generator=gp),
         context=torch.randn(B, Lc, 24, generator=gp)) * torch.randn(B, 4, S, S, generator=gp)).sum().backward()
    gnorm = {n: p.grad.double().norm().item() for n, p in ref.named_parameters()}
    med = float(np.median(list(gnorm.values())))
    zero_grad_tensors = {n for n, v in gnorm.items() if v < 1e-4 * med}
    assert len(zero_grad_tensors) <= 13, sorted(zero_grad_tensors)
    for p in ref.parameters():
        p.grad = None
    c_f, c_p = torch.randn(1, Lc, 24, generator=g).expand(B, -1, -1).contiguous(), torch.randn(1, Lc, 24, generator=g).expand(B, -1, -1).contiguous()
    for it in range(3):
        xf = torch.randn(B, 4, S, S, generator=g)
        forget = dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, S, S, generator=g))
        remain = dict(x=torch.randn(B, 4, S, S, generator=g), c=c_p, t=torch.randint(0, 1000, (B,), generator=g), noise=torch.randn(B, 4, S, S, generator=g))
        want = orc.step(forget, remain)
        got = run.step({k: v.to(DEV) for k, v in forget.items()}, {k: v.to(DEV) for k, v in remain.items()})
        assert got["forget_loss"].item() == pytest.approx(want["forget_loss"], rel=4e-2, abs=1e-5)
        assert got["remain_loss"].item() == pytest.approx(want["remain_loss"], rel=3e-2)
    # EVERY trainable tensor's update against the oracle's: cosine + ratio of the norms (the pooled sign / bulk metric
    # could not fail on a wrong small tensor).  Bounds = 2 x the worst measured (printed below).
    worst_cos, worst_ratio = (2.0, None), (0.0, None)
    for n, q in ref.named_parameters():
        mine = model.view(model.params, n).cpu()
        du_ref, du = (q.detach() - p0[n]).flatten().double(), (mine - p0[n]).flatten().double()
        if method == "xattn" and "attn2" not in n:
            assert torch.equal(mine, p0[n]), n                  # untouched by the optimizer
            continue
        if mask_mode == "intended":
            # masked-out coordinates still move in the forget stage through momentum -- both paths' updates include that
            pass
        if n in zero_grad_tensors:
            assert du.abs().max().item() <= 6.5 * hp["lr"], n          # six Adam steps of at most lr along a direction the function ignores
            continue
        cos = float((du * du_ref).sum() / (du.norm() * du_ref.norm() + 1e-30))
        ratio = float(du.norm() / (du_ref.norm() + 1e-30))
        assert du_ref.norm().item() > 0, n
        if cos < worst_cos[0]:
            worst_cos = (cos, n)
        if abs(ratio - 1.0) > abs(worst_ratio[0] - 1.0) or worst_ratio[1] is None:
            worst_ratio = (ratio, n)
        assert cos >= SD_ORACLE_UPDATE_COS_MIN[method], (n, cos)
        assert abs(ratio - 1.0) < SD_ORACLE_UPDATE_NORM_TOL[method], (n, ratio)
    print(f"SD {method}/{mask_mode}: min per-tensor update cosine {worst_cos[0]:.4f} ({worst_cos[1]}), worst norm ratio {worst_ratio[0]:.4f} ({worst_ratio[1]})")


def test_latent_diffusion_surface_p_losses_vs_oracle():
    """LatentDiffusion.q_sample / apply_model / p_losses (ddpm.py:424-445,1121-1131,1286-1319) on the native UNet: loss and gradients
    through torch autograd against the oracle's schedule + UNet."""
    from oracle import sd_ref
    from sfron import sd
    ref, model = _pair(SMALL, seed=61)
    ref.train()
    ldm = sd.LatentDiffusion(model).train()
    assert ldm.model.diffusion_model is model and ldm.num_timesteps == 1000
    g = torch.Generator().manual_seed(62)
    B, S, Lc = 4, 8, 6
    x0, noise = torch.randn(B, 4, S, S, generator=g), torch.randn(B, 4, S, S, generator=g)
    t, c = torch.randint(0, 1000, (B,), generator=g), torch.randn(B, Lc, 24, generator=g)
    want = sd_ref.LDMSchedule().p_losses(ref, x0, c, t, noise)
    want.backward()
    loss, d = ldm.p_losses(x0.to(DEV), {"c_crossattn": [c.to(DEV)]}, t.to(DEV), noise.to(DEV))
    loss.backward()
    assert loss.item() == pytest.approx(want.item(), rel=2e-2) and "train/loss_simple" in d
    dots = na = nb = 0.0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        ga, gb = p.grad.detach().cpu().flatten().double(), q.grad.flatten().double()
        dots += torch.dot(ga, gb).item(); na += ga.pow(2).sum().item(); nb += gb.pow(2).sum().item()
    assert dots / math.sqrt(na * nb) > 0.9995
    with pytest.raises(NotImplementedError):
        ldm.get_input({"jpg": None}, "jpg")


def test_sd_unet_forward_backward_bitwise_reproducible():
    """As for the DDPM U-Net: repeated forward + backward passes of the LDM UNet agree bit for bit (GroupNorm over 1 .. 2 channels per
    group here, LayerNorm, GEGLU, both attention forms, split-K slabs)."""
    _, model = _pair(MID, seed=50)
    model.train()
    g = torch.Generator().manual_seed(2)
    B, S, Lc = 2, 32, 77
    x, t = torch.randn(B, 4, S, S, generator=g).to(DEV), torch.randint(0, 1000, (B,), generator=g).to(DEV)
    ctx, w = torch.randn(B, Lc, MID["context_dim"], generator=g).to(DEV), torch.randn(B, 4, S, S, generator=g).to(DEV)
    outs, grads = [], []
    for rep in range(3):
        junk = torch.randn(1 << 22, device=DEV) * (rep + 1)
        del junk
        out, bwd = model._run(x, t, ctx, need_grad=True)
        bwd(w.clone())
        outs.append(out.clone()); grads.append(model.grads.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for r in (1, 2):
        bad = [n for n in model.index if not torch.equal(model.view(grads[0], n), model.view(grads[r], n))]
        assert not bad, bad[:8]


def test_sd_graph_replay_matches_eager():
    """The two stages of an iteration replayed as HIP graphs (sfron.graphs) give bit-for-bit the parameters of the eager loop."""
    from sfron import sd
    hp = dict(lr=1e-4, train_method="full")
    g = torch.Generator().manual_seed(31)
    B, S, Lc = 4, 8, 6
    c_f, c_p = torch.randn(B, Lc, 24, generator=g).to(DEV), torch.randn(B, Lc, 24, generator=g).to(DEV)
    batches = []
    for it in range(4):
        xf = torch.randn(B, 4, S, S, generator=g).to(DEV)
        batches.append((dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.randint(0, 1000, (B,), generator=g).to(DEV), noise=torch.randn(B, 4, S, S, generator=g).to(DEV)),
                        dict(x=torch.randn(B, 4, S, S, generator=g).to(DEV), c=c_p, t=torch.randint(0, 1000, (B,), generator=g).to(DEV),
                             noise=torch.randn(B, 4, S, S, generator=g).to(DEV))))
    res = []
    for use in (False, True):
        _, model = _pair(SMALL, seed=30)
        run = sd.SDSFRon(model, use_graphs=use, **hp)
        losses = [run.step(*b) for b in batches]
        if use:
            assert run._graphs["forget"].graph is not None and run._graphs["remain"].graph is not None      # iterations 2.. were replays
        res.append((model.params.clone(), [(l["forget_loss"].item(), l["remain_loss"].item()) for l in losses]))
    assert res[0][1] == res[1][1]
    assert torch.equal(res[0][0], res[1][0])


def test_sd_fisher_and_mask_vs_oracle():
    """SD/train-scripts/generate_fisher.py:36-79: guided two-branch prediction, -MSE, F += g^2 / n; then the saliency mask."""
    from oracle import sd_ref
    from sfron import fisher, sd
    ref, model = _pair(SMALL, seed=21)
    g = torch.Generator().manual_seed(22)
    B, S, Lc, n = 3, 8, 6, 3
    c, c0 = torch.randn(1, Lc, 24, generator=g).expand(B, -1, -1).contiguous(), torch.randn(1, Lc, 24, generator=g).expand(B, -1, -1).contiguous()
    batches = [dict(x=torch.randn(B, 4, S, S, generator=g), c=c, c_null=c0, t=torch.randint(0, 1000, (B,), generator=g),
                    noise=torch.randn(B, 4, S, S, generator=g)) for _ in range(n)]
    want = sd_ref.sd_fisher(ref, sd_ref.LDMSchedule(), batches, c_guidance=7.5)
    acc = fisher.SDFisherAccumulator(model, sd.LDMSchedule(device=DEV), n_batches=n, c_guidance=7.5)
    for b in batches:
        loss = acc.accumulate({k: v.to(DEV) for k, v in b.items()})
    assert torch.isfinite(loss) and loss.item() < 0
    got = acc.state_dict()
    assert list(got.keys()) == [k for k, _ in ref.named_parameters()]
    num = den = 0.0
    tot_ref = sum(float(v.sum()) for v in want.values())
    for k, w in want.items():
        a = got[k]
        assert a.shape == w.shape and torch.isfinite(a).all() and (a >= 0).all(), k
        num += (a - w).double().pow(2).sum().item(); den += w.double().pow(2).sum().item()
        if float(w.sum()) > 1e-4 * tot_ref:                     # tensors that carry the Fisher mass: their totals agree
            assert float(a.sum()) == pytest.approx(float(w.sum()), rel=0.2), k
    print(f"SD Fisher: bulk relative error {(num / den) ** 0.5:.3f}")
    assert (num / den) ** 0.5 < 0.15
    # the mask from two Fisher dicts: the threshold rule is bit-exact on identical inputs (tests/test_gpu_fisher_and_acceptance.py);
    # here: the reference file format round-trips (UNet-relative names, bool tensors)
    m = fisher.masks_from_fisher(got, {k: v * 0.5 + 1e-9 for k, v in got.items()}, th=1.5)
    assert set(m.keys()) == set(got.keys()) and all(v.dtype == torch.bool for v in m.values())
    k0 = max(got, key=lambda k: got[k].numel())
    exp = ((got[k0] + 1e-15) / (got[k0] * 0.5 + 1e-9 + 1e-15)) >= 1.5
    assert torch.equal(m[k0], exp)


def test_sd_v1_full_size_nsfw_removal_steps():
    """BASELINE config 4 at its real size: the v1-inference.yaml UNet (859,520,964 parameters), 64x64 latents (512 px), 77-token
    context, batch 2 (SD/README.md:69), train_method full: three iterations of the nsfw_removal loop on the HIP path -- finite
    losses, weights move, timing printed (parity at this size is carried by the kernels' size-independent tests above and the
    two smaller whole-model comparisons)."""
    import time
    from sfron import sd, sd_unet
    torch.manual_seed(0)
    model = sd_unet.UNetModel()
    assert sum(p.numel() for p in model.parameters()) == 859_520_964
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.parameters():                                  # the zero_module tensors, re-drawn
            if not bool(p.any()):
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    model.sync_bf16()
    run = sd.SDSFRon(model, lr=1e-5, forget_alpha=1.0, remain_alpha=1.0, train_method="full")
    B = 2
    gd = torch.Generator(device=DEV).manual_seed(2)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=gd)
    c_f, c_p = rn(1, 77, 768).expand(B, -1, -1).contiguous(), rn(1, 77, 768).expand(B, -1, -1).contiguous()
    p0 = model.params.clone()
    times = []
    for it in range(3):
        xf = rn(B, 4, 64, 64)
        forget = dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.randint(0, 1000, (B,), device=DEV, generator=gd), noise=rn(B, 4, 64, 64))
        remain = dict(x=rn(B, 4, 64, 64), c=c_p, t=torch.randint(0, 1000, (B,), device=DEV, generator=gd), noise=rn(B, 4, 64, 64))
        torch.cuda.synchronize(); t0 = time.time()
        out = run.step(forget, remain)
        torch.cuda.synchronize(); times.append(time.time() - t0)
        assert torch.isfinite(out["forget_loss"]) and torch.isfinite(out["remain_loss"])
    print(f"SD v1 UNet, batch 2, 64x64 latents: {min(times) * 1e3:.0f} ms per SFR-on iteration (forget fwd+bwd, pseudo fwd, remain fwd+bwd, 2 Adam sweeps); "
          f"losses {out['forget_loss'].item():.4f} / {out['remain_loss'].item():.4f}")
    assert torch.isfinite(model.params).all() and not torch.equal(model.params, p0)
