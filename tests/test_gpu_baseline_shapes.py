"""GPU parity AT THE BASELINE SIZES (DiT-XL/2, 256 px, batch 32 per GPU -> M = 8192 token rows, D = 1152, F = 4608):
(a) every block GEMM of the engine (DiT/models.py:108-121 forward and backward: SURVEY.md appendix D) at its exact shape,
    through the tile / grouped-order choice the dispatcher makes for that shape (tile_hint = 0, as dit_engine.hip calls it),
    with the epilogue the engine fuses, against torch fp32 on the same bf16 inputs;
(b) one whole DiT-XL/2 forward + backward at batch 32 against the CPU oracle (oracle.dit_ref restates DiT/models.py:145-248;
    the timm Attention / Mlp / PatchEmbed boundary is "parity unpinned");
(c) DiT-B/4 (BASELINE config 2) at FULL depth: three SFR-on iterations (DiT/forget.py:256-322) against the oracle, and the
    north-star acceptance -- eps-pred MSE within 1e-4 of the reference path after 50 steps -- on DiT-B/4.
Tolerances: fp32 outputs differ from torch by accumulation order only (rel-L2 < 1e-5 on K <= 8192); bf16 outputs by one bf16
rounding (2^-9 relative per element); whole-model bounds as in tests/test_gpu_dit.py (bf16 GEMM operands).
The oracle's PatchEmbed / Attention / Mlp restate timm's published behaviour (timm is un-vendored and un-pinned by the reference):
the 4e-2 per-tensor gradient bounds below are against that restatement -- parity unpinned at timm."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
M, D, F = 8192, 1152, 4608
T = 256


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * scale).to(torch.bfloat16)


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("name,N,K,epi", [("qkv", 3 * D, D, "bf16"), ("proj", D, D, "gate_res"), ("fc1", F, D, "gelu"),
                                          ("fc2", D, F, "gate_res")])
def test_forward_gemms_at_baseline_shapes(name, N, K, epi):
    from sfron import ops, _lib
    X, W = _rand((M, K), 1), _rand((N, K), 2, 0.03)
    bias = torch.randn(N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) * 0.1
    pre = X.float() @ W.float().t() + bias
    if epi == "bf16":
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(X, W, M, N, K, bias=bias, c_bf16=C)
        assert _rel(C, pre) < 4e-3
        assert torch.allclose(C.float(), pre, rtol=1e-2, atol=2e-2)
    elif epi == "gelu":
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        aux = torch.empty_like(C)
        ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_GELU, bias=bias, c_bf16=C, aux=aux)
        assert _rel(aux, pre) < 4e-3
        want = torch.nn.functional.gelu(pre, approximate="tanh")
        assert _rel(C, want) < 5e-3
        assert torch.allclose(C.float(), want, rtol=1e-2, atol=1e-2)
    else:
        B = M // T
        gate = torch.randn(B, 6 * D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
        x0 = torch.randn(M, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
        x1 = torch.empty_like(x0)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm(X, W, M, N, K, epilogue=_lib.EPI_GATE_RES, bias=bias, c_f32=x1, resid=x0, aux=aux, gate=gate[:, 2 * D:],
                 ldgate=6 * D, tokens=T)
        want = x0 + gate[:, 2 * D:3 * D].repeat_interleave(T, dim=0) * pre
        assert _rel(aux, pre) < 4e-3
        assert _rel(x1, want) < 1e-5
        assert torch.allclose(x1, want, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("name,N,K,epi", [("qkv", D, 3 * D, "bf16"), ("proj", D, D, "bf16"), ("fc1", D, F, "bf16"),
                                          ("fc2+gelu'", F, D, "dgelu")])
def test_dgrad_gemms_at_baseline_shapes(name, N, K, epi):
    """dX[M, N] = dY[M, K] W[K, N] (W row-major [out = K][in = N]: the forward weight read transposed)."""
    from sfron import ops, _lib
    dY, W = _rand((M, K), 6, 0.1), _rand((K, N), 7, 0.03)
    want = dY.float() @ W.float()
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    if epi == "bf16":
        ops.gemm(dY, W, M, N, K, b_t=True, c_bf16=C)
        Cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(dY, W, M, N, K, b_t=True, epilogue=_lib.EPI_F32, c_f32=Cf)
        assert _rel(Cf, want) < 1e-5
    else:
        hpre = _rand((M, N), 8, 1.5)
        ops.gemm(dY, W, M, N, K, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=C, aux=hpre)
        h = hpre.float().requires_grad_(True)
        torch.nn.functional.gelu(h, approximate="tanh").backward(want)
        want = h.grad
        # the same product with the fc1 bias-gradient partials formed in its epilogue (fp32 column sums of each 256-row tile row,
        # before the bf16 rounding): d_hpre unchanged bit for bit, the partial rows are the column sums of the fp32 product
        rows = _lib.lib().sfron_gemm_dgelu_colpart_rows(M, N, K)
        assert rows == M // 256
        C2 = torch.empty_like(C)
        part = torch.full((rows, N), float("nan"), device=DEV)
        ops.gemm(dY, W, M, N, K, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=C2, aux=hpre, col_partials=part)
        assert torch.equal(C2, C)
        assert torch.allclose(part, want.view(rows, 256, N).sum(1), rtol=2e-3, atol=2e-3 * want.abs().max().item())
        assert _rel(part.sum(0), want.sum(0)) < 2e-3
        part2 = torch.empty_like(part)             # fixed summation order, no atomics: bitwise reproducible
        ops.gemm(dY, W, M, N, K, b_t=True, epilogue=_lib.EPI_DGELU, c_bf16=C2, aux=hpre, col_partials=part2)
        assert torch.equal(part, part2)
    assert _rel(C, want) < 5e-3
    assert torch.allclose(C.float(), want, rtol=1.5e-2, atol=2e-3 * float(want.abs().max()))


@pytest.mark.parametrize("name,N,K,rowsum", [("qkv", 3 * D, D, True), ("proj", D, D, False), ("fc1", F, D, True), ("fc2", D, F, False)])
def test_wgrad_gemms_at_baseline_shapes(name, N, K, rowsum):
    """dW[N, K] = dY[M, N]^T X[M, K] with the M = 8192 token rows as the contraction; qkv / fc1 also produce their bias
    gradient sum_rows dY inside the same launch (a_rowsum)."""
    from sfron import ops, _lib
    dY, X = _rand((M, N), 9, 0.1), _rand((M, K), 10)
    want = dY.float().t() @ X.float()
    dW = torch.full((N, K), float("nan"), dtype=torch.float32, device=DEV)
    db = torch.full((N,), float("nan"), dtype=torch.float32, device=DEV) if rowsum else None
    if rowsum:
        assert _lib.lib().sfron_gemm_rowsum_supported(N, K, M) == 1
    ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=dW, a_rowsum=db)
    assert _rel(dW, want) < 1e-5
    assert torch.allclose(dW, want, rtol=2e-4, atol=2e-4 * M ** 0.5)
    if rowsum:
        want_b = dY.float().sum(0)
        assert torch.allclose(db, want_b, rtol=1e-5, atol=1e-3)
        # same launch without the row sums: the weight gradient itself is unchanged, bit for bit
        dW2 = torch.empty_like(dW)
        ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=dW2)
        assert torch.equal(dW, dW2)


@pytest.mark.parametrize("M", [8192, 1024])
def test_loader_wave_gemms_are_bit_identical_to_the_shared_wave_form(M):
    """The three-slot tiles of the dgrad / weight-gradient layouts run with four loader waves by default (csrc/gemm.hip k_gemm_pipe NL:
    the multiplying waves issue no LDS-DMA).  Same products in the same order: every output bit equals the shared-wave form's."""
    from sfron import ops, _lib
    D, F_ = 1152, 4608
    g = torch.Generator(device=DEV).manual_seed(M)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16)
    cases = []
    for N, K in ((3 * D, D), (D, D), (F_, D)):                # dgrad: dX[M, K] = dY[M, N] W[N, K]
        dY, W = rnd(M, N), rnd(N, K)
        def run(dY=dY, W=W, N=N, K=K):
            C = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
            ops.gemm(dY, W, M, K, N, b_t=True, c_bf16=C)
            return C
        cases.append(run)
    for N, K in ((3 * D, D), (D, D), (F_, D), (D, F_)):       # weight gradient: dW[N, K] = dY[M, N]^T X[M, K]
        dY, X = rnd(M, N), rnd(M, K)
        def run(dY=dY, X=X, N=N, K=K):
            C = torch.empty(N, K, dtype=torch.float32, device=DEV)
            ops.gemm(dY, X, N, K, M, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=C)
            return C
        cases.append(run)
    L = _lib.lib()
    old = L.sfron_gemm_loader_waves(0)
    try:
        assert old == 4                                        # the default
        ref = [run() for run in cases]
        L.sfron_gemm_loader_waves(4)
        out = [run() for run in cases]
    finally:
        L.sfron_gemm_loader_waves(old)
    for a, b in zip(ref, out):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def test_rowsum_small_shapes_and_rejections():
    from sfron import ops, _lib
    L = _lib.lib()
    assert L.sfron_gemm_rowsum_supported(192, 192, 128) == 1 and L.sfron_gemm_rowsum_supported(192, 192, 192) == 0
    assert L.sfron_gemm_rowsum_supported(128, 192, 128) == 0
    for (Mw, Nw, Kw) in [(192, 192, 128), (384, 576, 320), (768, 192, 512)]:
        dY, X = _rand((Kw, Mw), Mw + Kw, 0.3), _rand((Kw, Nw), Nw)
        dW = torch.empty(Mw, Nw, dtype=torch.float32, device=DEV)
        db = torch.empty(Mw, dtype=torch.float32, device=DEV)
        ops.gemm(dY, X, Mw, Nw, Kw, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=dW, a_rowsum=db)
        ref = torch.empty_like(dW)
        ops.gemm(dY, X, Mw, Nw, Kw, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=ref, tile_hint=-1)
        assert torch.equal(dW, ref)
        assert torch.allclose(db, dY.float().sum(0), rtol=1e-6, atol=1e-4)
    dY, X = _rand((192, 192), 1), _rand((192, 192), 2)
    with pytest.raises(_lib.SfronError):          # 3 k-tiles: not a shape of the three-slot kernel
        ops.gemm(dY, X, 192, 192, 192, a_t=True, b_t=True, epilogue=_lib.EPI_F32,
                 c_f32=torch.empty(192, 192, dtype=torch.float32, device=DEV), a_rowsum=torch.empty(192, device=DEV))


def _pair(name_or_cfg, batch, seed, std=0.02):
    from oracle import dit_ref
    from sfron import dit
    torch.manual_seed(seed)
    if isinstance(name_or_cfg, str):
        ref = dit_ref.build(name_or_cfg, input_size=32)
        model = dit.DiT_models[name_or_cfg](input_size=32, num_classes=1000, batch_size=batch)
    else:
        ref = dit_ref.DiT(**name_or_cfg)
        model = dit.DiT(batch_size=batch, **name_or_cfg)
    dit_ref.randomize_zero_init(ref, std=std, seed=seed + 1)
    model.load_state_dict(ref.state_dict())
    return ref, model


def test_xl2_batch32_forward_backward_vs_oracle():
    """The headline configuration itself: DiT-XL/2 (28 blocks, 16 heads of 72), 256 px latents, batch 32, one forward +
    backward pass (DiT/models.py:233-248 and its autograd) against the CPU oracle; same bounds as the small cases."""
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    B = 32
    ref, model = _pair("DiT-XL/2", B, seed=21)
    gen = torch.Generator().manual_seed(22)
    x = torch.randn(B, 4, 32, 32, generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    t[0], t[1] = 0, 999
    y = torch.randint(0, 1000, (B,), generator=gen)
    drop = (torch.rand(B, generator=gen) < 0.1).long()
    w = torch.randn(B, 8, 32, 32, generator=gen) * 1e-2
    ref.train()
    out_ref = ref(x, t, y, force_drop_ids=drop)
    (out_ref * w).sum().backward()
    model.train()
    out = model(x.to(DEV), t.to(DEV), y.to(DEV), force_drop_ids=drop.to(DEV))
    e_out = _rel(out.cpu(), out_ref)
    model.zero_grad()
    (out * w.to(DEV)).sum().backward()
    worst, worst_name = 0.0, ""
    dots = norms_a = norms_b = 0.0
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if not q.requires_grad:
            continue
        ga, gb = p.grad.detach().cpu().flatten(), q.grad.flatten()
        e = ((ga - gb).norm() / (gb.norm() + 1e-30)).item()
        if e > worst:
            worst, worst_name = e, n
        dots += torch.dot(ga.double(), gb.double()).item()
        norms_a += ga.double().pow(2).sum().item()
        norms_b += gb.double().pow(2).sum().item()
    cos = dots / (norms_a ** 0.5 * norms_b ** 0.5)
    print(f"DiT-XL/2 B=32: output rel-L2 {e_out:.3e}; worst per-tensor gradient rel-L2 {worst:.3e} ({worst_name}); cosine {cos:.6f}")
    # bounds = 2 x measured on MI355X (output rel-L2 2.8e-3, worst per-tensor gradient 6.2e-3, 1 - cosine 6e-6; VERDICT r4 weak #2: the
    # bounds of the small cases, 1.5e-2 / 4e-2 / 0.9995, were 5-6 x what this size measures)
    assert e_out < 6e-3, e_out
    assert worst < 1.3e-2, (worst_name, worst)
    assert cos > 0.99998, cos


B4 = dict(input_size=32, patch_size=4, in_channels=4, hidden_size=768, depth=12, num_heads=12, num_classes=1000)


def _update_agreement(ref, eng, p0):
    """How the parameter UPDATES (p - p0) of the two paths compare: fraction of coordinates (with a non-negligible oracle
    update) that moved in the same direction, and the bulk relative error of the update vector."""
    same = tot = 0
    num = den = 0.0
    for n, q in ref.named_parameters():
        if not q.requires_grad:
            continue
        du_ref = (q.detach() - p0[n]).flatten()
        du = (eng.view(eng.params, n).cpu() - p0[n]).flatten()
        big = du_ref.abs() > 0.05 * du_ref.abs().max()
        same += int((torch.sign(du[big]) == torch.sign(du_ref[big])).sum())
        tot += int(big.sum())
        num += (du - du_ref).double().pow(2).sum().item()
        den += du_ref.double().pow(2).sum().item()
    return same / max(1, tot), (num / den) ** 0.5


def test_dit_b4_full_depth_sfron_iterations_vs_oracle():
    """BASELINE config 2 (DiT-B/4, 12 blocks, 12 heads of 64, 64 tokens) through three SFR-on iterations at batch 8."""
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    B = 8
    ref, model = _pair(B4, B, seed=31, std=0.02)
    model.train()
    gm = torch.Generator().manual_seed(32)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=mask, unlearn_loss="ga", forget_class=207)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)
    for it in range(3):
        f, r = data.synthetic_batch(3, it, "forget", **kw), data.synthetic_batch(3, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                        {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        assert got["forget_mse"].mean().item() == pytest.approx(want["forget_mse"], rel=3e-2)
        assert got["remain_mse"].mean().item() == pytest.approx(want["remain_mse"], rel=3e-2)
        assert got["stats"][0].item() == pytest.approx(want["forget_gnorm"], rel=5e-2)
    agree, bulk = _update_agreement(ref, model.engine, p0)
    print(f"DiT-B/4 depth 12, 3 iterations: update sign agreement {agree:.4f}, bulk relative error of the update {bulk:.3f}")
    assert agree > 0.995, agree          # measured 0.9998
    assert bulk < 0.08, bulk             # measured 0.023
    assert runner.opt.step_count == 6


def test_dit_b4_fifty_step_eps_mse_within_1e4_of_oracle():
    """North-star acceptance on BASELINE config 2: after 50 SFR-on steps of DiT-B/4 (full depth), the eps-pred MSE of the HIP
    path on a held-out batch is within 1e-4 of the CPU oracle's (same seeds, masks, hyper-parameters)."""
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    B = 8
    ref, model = _pair(B4, B, seed=41, std=0.02)
    model.train()
    gm = torch.Generator().manual_seed(42)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=mask, unlearn_loss="ga", forget_class=207)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)
    worst = 0.0
    for it in range(50):
        f, r = data.synthetic_batch(13, it, "forget", **kw), data.synthetic_batch(13, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                        {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        worst = max(worst, abs(got["remain_mse"].mean().item() - want["remain_mse"]),
                    abs(got["forget_mse"].mean().item() - want["forget_mse"]))
    ref.eval()
    hb = data.synthetic_batch(14, 0, "remain", global_batch=32, num_classes=1000, forget_class=207)
    tab = dref.DiffusionTables(1000)
    with torch.no_grad():
        t_ref = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), hb["x0"], hb["t"], dict(y=hb["y"]), hb["noise"])
    model.eval()
    d = runner.diffusion
    hbd = {k: v.to(DEV) for k, v in hb.items()}
    with torch.no_grad():
        out = model(d.q_sample(hbd["x0"], hbd["t"], hbd["noise"]), hbd["t"], hbd["y"])
    mse_hip, _, _ = d.loss_fwd_bwd(out.contiguous(), hbd["x0"], hbd["t"], hbd["noise"], 1.0)
    final_gap = abs(mse_hip.mean().item() - t_ref["mse"].mean().item())
    print(f"DiT-B/4: max per-step |mse gap| over 50 steps = {worst:.2e}; held-out eps-MSE gap after 50 steps = {final_gap:.2e} "
          f"(oracle {t_ref['mse'].mean().item():.5f})")
    assert final_gap < 1e-4, final_gap
    assert worst < 1e-3, worst        # per-step training-batch mse (bf16 forward noise on a loss of O(1)); measured bound in DESIGN.md


def test_xl2_ten_sfron_iterations_vs_oracle():
    """Multi-step acceptance AT THE HEADLINE GEOMETRY (DiT-XL/2: 28 blocks, 16 heads of 72, 256 tokens), batch 4: ten iterations of
    DiT/forget.py:256-322 on the fused HIP runner against DiTSfronOracle (same seeds, mask, hyper-parameters; about a minute of CPU):
    per-step losses and gradient norms, the direction of the parameter updates, and the eps-pred MSE on a held-out batch after the
    ten steps (north-star: within 1e-4 of the reference path)."""
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    B = 4
    ref, model = _pair("DiT-XL/2", B, seed=51, std=0.02)
    model.train()
    gm = torch.Generator().manual_seed(52)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=mask, unlearn_loss="ga", forget_class=207)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)
    hb = data.synthetic_batch(24, 0, "remain", global_batch=16, num_classes=1000, forget_class=207)
    hbd = {k: v.to(DEV) for k, v in hb.items()}
    d = runner.diffusion

    def heldout():
        """eps-pred MSE of both paths on the held-out batch, eval mode (no label dropout)"""
        ref.eval(); model.eval()
        with torch.no_grad():
            t_ref = dref.training_losses(dref.DiffusionTables(1000), lambda x, t, y: ref(x, t, y), hb["x0"], hb["t"], dict(y=hb["y"]), hb["noise"])
            out = model(d.q_sample(hbd["x0"], hbd["t"], hbd["noise"]), hbd["t"], hbd["y"])
        mse_hip, _, _ = d.loss_fwd_bwd(out.contiguous(), hbd["x0"], hbd["t"], hbd["noise"], 1.0)
        ref.train(); model.train()
        return mse_hip.mean().item(), t_ref["mse"].mean().item()
    h0, r0 = heldout()
    worst = 0.0
    for it in range(10):
        f, r = data.synthetic_batch(23, it, "forget", **kw), data.synthetic_batch(23, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                        {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        fm, rm = got["forget_mse"].mean().item(), got["remain_mse"].mean().item()
        assert fm == pytest.approx(want["forget_mse"], rel=3e-2), it
        assert rm == pytest.approx(want["remain_mse"], rel=3e-2), it
        assert got["stats"][0].item() == pytest.approx(want["forget_gnorm"], rel=5e-2), it
        worst = max(worst, abs(fm - want["forget_mse"]), abs(rm - want["remain_mse"]))
    runner.guard.poll(block=True)
    agree, bulk = _update_agreement(ref, model.engine, p0)
    h1, r1 = heldout()
    gap0, gap = abs(h0 - r0), abs(h1 - r1)
    drift = abs((h1 - h0) - (r1 - r0))
    print(f"DiT-XL/2 batch 4, 10 iterations: update sign agreement {agree:.4f}, bulk relative error of the update {bulk:.3f}, "
          f"max per-step |mse gap| {worst:.2e}; held-out eps-MSE: oracle {r0:.5f} -> {r1:.5f}, HIP {h0:.5f} -> {h1:.5f}; gap before the "
          f"first step {gap0:.2e}, after ten {gap:.2e}, |difference of the two paths' CHANGE| {drift:.2e}")
    assert agree > 0.995, agree
    assert bulk < 0.1, bulk
    # Random-init DiT-XL/2 is far from converged: the held-out MSE of BOTH paths falls by ~0.28 in these ten steps (measured 1.449 ->
    # 1.169).  The two paths start 1.6e-4 apart (bf16 rounding of weights / activations through 28 blocks, before any step) and their
    # ten-step CHANGE differs by 2.1e-4 = 0.07 % of the change: bound = 0.2 % of the oracle's change + the north-star's 1e-4.
    assert drift < 2e-3 * abs(r1 - r0) + 1e-4, (drift, r0, r1)
    # 2 x measured, fixed: gap before the first step 1.6e-4 (the bf16 weight rounding, profiles/r05_xl2_gap_bisect.txt), after ten
    # iterations up to 3.1e-4 over the round-5 builds, worst per-step training-batch gap 1.1e-3 (4 samples, loss of O(1))
    assert gap0 < 3.2e-4 and gap < 6.2e-4, (gap0, gap)
    assert worst < 2.2e-3, worst
    assert runner.opt.step_count == 20


def test_xl2_fifty_sfron_iterations_vs_oracles():
    """The north-star acceptance WHERE IT IS STATED -- DiT-XL/2, 50 SFR-on iterations (DiT/forget.py:256-322), held-out eps-pred MSE
    (gaussian_diffusion.py:746-783 "mse") against the reference path -- with the reason for what is found (bisect and
    sensitivity runs: tests/debug/xl2_gap_bisect.py, table in DESIGN.md section 3).

    Three trajectories from the same weights over the same 50 batches (batch 4; hyper-parameters of the ten-iteration test above):
      HIP      the fused runner (bf16 GEMM operands as north_star prescribes, fp32 everything else)
      fp32     oracle.sfron_ref.DiTSfronOracle, fp32 throughout -- the reference path
      bf16-op  the same oracle with bf16-rounded weights and product inputs in its forward / backward passes and fp32 masters in the
               optimizer (oracle/bf16_ref.py): what the prescribed operand type alone does to the reference
    The two oracles run on the GPU in fp32 (same torch modules as on the CPU; checked here: one forward pass on both devices agrees to 1e-6)
    so that 2 x 50 iterations of DiT-XL/2 take a minute instead of ten.

    Measured on MI355X (mean over 4 held-out batches of 16): the bf16-op oracle is 1.3e-4 from the fp32 oracle before any step (the WEIGHT
    rounding: 1.07e-4 of it) and 7e-4 ... 9e-4 after 50 steps of a trajectory along which the held-out MSE itself moves 1.44 -> 0.50; the HIP
    path follows the bf16-op oracle to 2e-6 ... 1.6e-4 over the checkpoints.  That is also how far two runs of the bf16-op ORACLE ITSELF end up
    from each other (-7.1e-4 and -9.1e-4 at step 50 in two runs that differ only in the order of torch's fp32 atomics in the embedding
    backward; tests/debug/xl2_gap_bisect.py --sensitivity perturbs one ulp deliberately): a 1e-7 perturbation flips bf16 roundings of
    weights, each flip is a 2^-8 relative step.  The fp32 trajectory is NOT chaotic (one ulp on every initial weight moves it by < 3e-7 at
    every checkpoint), so with fp32 operands the north-star's 1e-4 would be a meaningful bound at this size; with the bf16 operands it
    prescribes, the reference's own trajectory is only defined to ~2e-4 here, and 1e-4 is met where the loss moves slowly (BASELINE config 2,
    test_dit_b4_fifty_step_eps_mse_within_1e4_of_oracle: 1.1e-5; and at THIS geometry from a settled state:
    test_xl2_fifty_sfron_iterations_from_a_settled_state below).

    Asserted -- ONE set of bounds, fixed in round 5 from five oracle runs and three builds and not moved since (the derivation is the comment
    above the asserts; a build that breaks them is a finding, not a reason to rebase them):
      * before any step: HIP within 5e-5 of the bf16-operand oracle (no trajectory yet: pure forward-pass parity);
      * at iterations 10 / 20 / 30 / 50: HIP within 6e-4 of the bf16-operand oracle (2 x the 3e-4 by which that oracle differs from ITSELF run
        to run, = the largest build-to-build movement of the HIP path);
      * HIP and the bf16-operand oracle each within 1.5e-3 of the fp32 oracle (1.5 x the worst of either over all runs, 1.05e-3);
      * the operand type's own cost is not zero (> 5e-5 before any step, > 2e-4 along the trajectory): the claim above is checked, not assumed."""
    import copy
    from oracle import bf16_ref
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    B = 4
    ref, model = _pair("DiT-XL/2", B, seed=51, std=0.02)
    tab = dref.DiffusionTables(1000)
    hbs = [data.synthetic_batch(24 + i, 0, "remain", global_batch=16, num_classes=1000, forget_class=207) for i in range(4)]
    # the oracle on the CPU (the contract) and the same modules on the GPU
    ref.eval()
    with torch.no_grad():
        c4 = {k: v[:4] for k, v in hbs[0].items()}
        t_cpu = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), c4["x0"], c4["t"], dict(y=c4["y"]), c4["noise"])["mse"]
        ref.to(DEV)
        g4 = {k: v.to(DEV) for k, v in c4.items()}
        t_gpu = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), g4["x0"], g4["t"], dict(y=g4["y"]), g4["noise"])["mse"].cpu()
    assert (t_cpu - t_gpu).abs().max().item() < 1e-6, (t_cpu, t_gpu)
    ref.train()
    model.train()
    refb = copy.deepcopy(ref)
    gm = torch.Generator().manual_seed(52)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    mask_dev = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in mask.items()}
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, unlearn_loss="ga", forget_class=207)
    orc = sfron_ref.DiTSfronOracle(ref, tab, mask=mask_dev, **hp)
    orcb = sfron_ref.DiTSfronOracle(refb, tab, mask=mask_dev, **hp)
    d = diffusion.create_diffusion("")
    runner = step.DiTSFRon(model, d, mask=mask, **hp)
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)

    def held_oracle(m, rounded):
        m.eval()
        vals = []
        with torch.no_grad():
            for hb in hbs:
                g = {k: v.to(DEV) for k, v in hb.items()}
                fn = lambda: dref.training_losses(tab, lambda x, t, y: m(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])["mse"].mean().item()
                if rounded:
                    with bf16_ref.OperandRounding(m, ("W", "A", "E")):
                        vals.append(fn())
                else:
                    vals.append(fn())
        m.train()
        return sum(vals) / len(vals)

    def held_hip():
        model.eval()
        vals = []
        with torch.no_grad():
            for hb in hbs:
                g = {k: v.to(DEV) for k, v in hb.items()}
                out = model(d.q_sample(g["x0"], g["t"], g["noise"]), g["t"], g["y"])
                vals.append(d.loss_fwd_bwd(out.contiguous(), g["x0"], g["t"], g["noise"], 1.0)[0].mean().item())
        model.train()
        model.set_batch_size(B)
        return sum(vals) / len(vals)

    rows = []

    def checkpoint(it):
        f32, b16, hip = held_oracle(ref, False), held_oracle(refb, True), held_hip()
        rows.append((it, f32, b16 - f32, hip - f32, hip - b16))
        print(f"DiT-XL/2 after {it:2d} iterations: held-out eps-MSE fp32 oracle {f32:.5f} | bf16-operand oracle {b16 - f32:+.2e} | HIP {hip - f32:+.2e} | "
              f"HIP - bf16-operand oracle {hip - b16:+.2e}", flush=True)
    checkpoint(0)
    for it in range(50):
        f, r = data.synthetic_batch(23, it, "forget", **kw), data.synthetic_batch(23, it, "remain", **kw)
        fd, rd = {k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()}
        fo, ro = ({k: v.long() if k == "drop" else v for k, v in b.items()} for b in (fd, rd))
        orc.step(fo, ro)
        bf16_ref.sfron_step_bf16_operands(orcb, fo, ro)
        runner.step(fd, rd)
        if it + 1 in (10, 20, 30, 50):
            checkpoint(it + 1)
    runner.guard.poll(block=True)
    assert runner.opt.step_count == 100
    # Bounds: a bf16-operand trajectory at this size is defined to ~3e-4 only -- the bf16-operand ORACLE differs from itself
    # by 2e-4 between runs, and the HIP path (reproducible run to run for ONE build) moves by as much between builds that differ in a summation
    # order only: at iteration 50 it measured -7.5e-4, -6.2e-4 and -1.05e-3 against the fp32 oracle in three builds of round 5 (clip norm fused
    # into the weight-gradient epilogues; 8 instead of 4 rows per partial in the LayerNorm backward), the bf16-operand oracle -6.4e-4 ... -9.1e-4 in
    # five runs.  So: two such trajectories are compared at 6e-4 (measured: up to 3.5e-4), each against the fp32 oracle at 1.5e-3, and before any
    # step (no trajectory yet) HIP and the bf16-operand oracle at 5e-5.
    assert abs(rows[0][4]) < 5e-5, rows[0]
    for it, f32, d_b16, d_hip, d_hb in rows:
        assert abs(d_hb) < 6e-4, (it, d_hb)
        assert abs(d_hip) < 1.5e-3 and abs(d_b16) < 1.5e-3, (it, d_hip, d_b16)
    # the operand type's own cost is what the header says it is (not zero)
    assert abs(rows[0][2]) > 5e-5 and max(abs(r[2]) for r in rows) > 2e-4


def test_xl2_fifty_sfron_iterations_from_a_settled_state():
    """The north-star acceptance in the regime the reference runs in.  DiT/forget.py starts from PRETRAINED weights
    (forget.py:183-187) and moves them slowly (lr 1e-4, forget_alpha 1e-3, 500-1000 iterations, DiT/README.md:62-68); the random-init
    trajectory of the test above is the opposite regime (held-out loss 1.44 -> 0.50 in 50 iterations).  There is no checkpoint offline, so the
    settled state is MADE here: WARM fp32-oracle SFR-on iterations on the GPU from the same random init (the loss has flattened by then), then
    weights + both AdamW moments + step count + EMA go into the HIP runner through the reference's own checkpoint format
    (step.load_checkpoint: forget.py:346-353) and into the bf16-operand oracle, and the 50 COMPARED iterations run on all three.

    Also checked here: the GPU-resident fp32 oracle (torch on rocBLAS) against the CPU oracle -- the contract -- on one full TRAINING
    ITERATION (both losses and the clipped forget-gradient norm), not one forward pass."""
    import copy
    from oracle import bf16_ref
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    B, WARM = 4, 250
    ref, model = _pair("DiT-XL/2", B, seed=51, std=0.02)
    tab = dref.DiffusionTables(1000)
    gm = torch.Generator().manual_seed(52)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    mask_dev = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in mask.items()}
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, unlearn_loss="ga", forget_class=207)
    kw = dict(global_batch=B, num_classes=1000, forget_class=207)

    def batches(seed, it, dev):
        f, r = data.synthetic_batch(seed, it, "forget", **kw), data.synthetic_batch(seed, it, "remain", **kw)
        fd, rd = {k: v.to(dev) for k, v in f.items()}, {k: v.to(dev) for k, v in r.items()}
        return fd, rd, {k: v.long() if k == "drop" else v for k, v in fd.items()}, {k: v.long() if k == "drop" else v for k, v in rd.items()}

    # ---- the GPU-resident oracle against the CPU oracle on one full training iteration
    ref_cpu = copy.deepcopy(ref)
    orc_cpu = sfron_ref.DiTSfronOracle(ref_cpu, tab, mask=mask, **hp)
    _, _, fo, ro = batches(33, 0, "cpu")
    w_cpu = orc_cpu.step(fo, ro)
    ref.to(DEV)
    orc = sfron_ref.DiTSfronOracle(ref, tab, mask=mask_dev, **hp)
    _, _, fo, ro = batches(33, 0, DEV)
    w_gpu = orc.step(fo, ro)
    print(f"oracle on CPU vs on the GPU, one SFR-on iteration of DiT-XL/2: forget mse {w_cpu['forget_mse']:.6f} / {w_gpu['forget_mse']:.6f}, "
          f"remain mse {w_cpu['remain_mse']:.6f} / {w_gpu['remain_mse']:.6f}, forget gradient norm {w_cpu['forget_gnorm']:.6e} / "
          f"{w_gpu['forget_gnorm']:.6e}", flush=True)
    assert w_gpu["forget_mse"] == pytest.approx(w_cpu["forget_mse"], rel=2e-5)
    assert w_gpu["remain_mse"] == pytest.approx(w_cpu["remain_mse"], rel=2e-5)
    assert w_gpu["forget_gnorm"] == pytest.approx(w_cpu["forget_gnorm"], rel=2e-4)
    del ref_cpu, orc_cpu

    # ---- settle
    for it in range(1, WARM):
        _, _, fo, ro = batches(33, it, DEV)
        w = orc.step(fo, ro)
        if it % 50 == 0 or it == WARM - 1:
            print(f"  warm-up iteration {it}: forget mse {w['forget_mse']:.4f}, remain mse {w['remain_mse']:.4f}", flush=True)
    hbs = [data.synthetic_batch(24 + i, 0, "remain", global_batch=16, num_classes=1000, forget_class=207) for i in range(4)]

    # ---- the settled state into the two other paths
    refb = copy.deepcopy(ref)
    orcb = sfron_ref.DiTSfronOracle(refb, tab, mask=mask_dev, **hp)
    orcb.opt.load_state_dict(copy.deepcopy(orc.opt.state_dict()))
    orcb.ema = {n: v.clone() for n, v in orc.ema.items()}
    d = diffusion.create_diffusion("")
    runner = step.DiTSFRon(model, d, mask=mask, **hp)
    runner.load_checkpoint({"model": {k: v.detach().clone() for k, v in ref.state_dict().items()},
                            "ema": {n: v.clone() for n, v in orc.ema.items()}, "opt": orc.opt.state_dict()})
    assert runner.opt.step_count == 2 * WARM
    model.train()

    def held_oracle(m, rounded):
        m.eval()
        vals = []
        with torch.no_grad():
            for hb in hbs:
                g = {k: v.to(DEV) for k, v in hb.items()}
                fn = lambda: dref.training_losses(tab, lambda x, t, y: m(x, t, y), g["x0"], g["t"], dict(y=g["y"]), g["noise"])["mse"].mean().item()
                if rounded:
                    with bf16_ref.OperandRounding(m, ("W", "A", "E")):
                        vals.append(fn())
                else:
                    vals.append(fn())
        m.train()
        return sum(vals) / len(vals)

    def held_hip():
        model.eval()
        vals = []
        with torch.no_grad():
            for hb in hbs:
                g = {k: v.to(DEV) for k, v in hb.items()}
                out = model(d.q_sample(g["x0"], g["t"], g["noise"]), g["t"], g["y"])
                vals.append(d.loss_fwd_bwd(out.contiguous(), g["x0"], g["t"], g["noise"], 1.0)[0].mean().item())
        model.train()
        model.set_batch_size(B)
        return sum(vals) / len(vals)

    rows = []

    def checkpoint(it):
        f32, b16, hip = held_oracle(ref, False), held_oracle(refb, True), held_hip()
        rows.append((it, f32, b16 - f32, hip - f32, hip - b16))
        print(f"DiT-XL/2 settled ({WARM} warm-up iterations) + {it:2d}: held-out eps-MSE fp32 oracle {f32:.5f} | bf16-operand oracle {b16 - f32:+.2e} | "
              f"HIP {hip - f32:+.2e} | HIP - bf16-operand oracle {hip - b16:+.2e}", flush=True)
    checkpoint(0)
    for it in range(50):
        fd, rd, fo, ro = batches(34, it, DEV)
        orc.step(fo, ro)
        bf16_ref.sfron_step_bf16_operands(orcb, fo, ro)
        runner.step(fd, rd)
        if it + 1 in (10, 50):
            checkpoint(it + 1)
    runner.guard.poll(block=True)
    assert runner.opt.step_count == 2 * WARM + 100
    moved = abs(rows[-1][1] - rows[0][1])
    print(f"held-out eps-MSE of the fp32 oracle moved by {moved:.2e} over the 50 compared iterations", flush=True)
    # Measured on MI355X (round 6, gpurun_out/r06a/settled.log -> profiles/r06_xl2_settled.txt): the held-out eps-MSE of the fp32 oracle is 0.275
    # and moves by 7.5e-3 over the 50 compared iterations; HIP - fp32 oracle = +4.4e-5 / +6.1e-5 / +1.3e-5 at iterations 0 / 10 / 50, the
    # bf16-operand oracle +4.5e-5 / +6.9e-5 / +1.4e-5, HIP - bf16-operand oracle -1.0e-6 / -7.9e-6 / -1.2e-6.
    #   * the north-star's own number, as stated: |HIP - fp32 reference path| < 1e-4 at every checkpoint (not derived from the measurement);
    #   * HIP against the bf16-operand oracle: 2e-5 (2.5 x the worst measured) -- the implementation's share of the gap, an order of magnitude
    #     below the operand type's.
    for it, f32, d_b16, d_hip, d_hb in rows:
        assert abs(d_hip) < 1e-4, (it, d_hip, d_b16, d_hb)
        assert abs(d_hb) < 2e-5, (it, d_hb)
    assert moved < 0.05, moved                 # the regime this test is about: a loss that moves slowly


@pytest.mark.gpu
def test_xl2_headline_schedule_is_reproducible_bit_for_bit():
    """The headline runner exactly as bench.py configures it (DiT-XL/2, batch 32, block sweeps beside the forward pass and across the step
    boundary, weight gradients / the clip norm's adaLN share on their own streams, fused clip norm) for 12 steps, twice from the same weights
    and batches: parameters, both moments and the EMA come out bit-identical.  Every overlap in the step is ordered by events only; a missing
    one is a race, and a race shows as a difference (tools/soak_repro.py is the same run at any length: 60 and 300 steps measured identical)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("soak_repro", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                                                              "soak_repro.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.run(12)
    b = mod.run(12)
    assert a == b, (a, b)
    assert all(x == x for x in a[1])            # finite losses
