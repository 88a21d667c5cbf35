"""GPU parity against numbers the REFERENCE ITSELF produced -- no oracle between the HIP path and the fixture.

tests/golden/make_golden.py imported the reference classes in the build container (DiT/models.py:145-248 with the three
timm classes stood in -- parity unpinned there; DDPM/models/diffusion.py:195-413;
SD/ldm/modules/diffusionmodules/openaimodel.py:428-846) at shapes the HIP kernels accept and stored inputs + outputs:
  dit_gpu.npz   DiT 16x16 / patch 2 (64 tokens), D 128, 2 heads of 64, depth 2: forward (eval and with explicit drop ids),
                every gradient tensor (norm + a seeded random projection; ten in full), 3 SFR-on iterations (DiT/forget.py:256-322)
  ddpm_gpu.npz  Conditional_Model ch 128, 16x16, attention over 64 tokens, dropout 0: test-mode forward (cond_scale 2),
                gradients, 2 SFR-on iterations (DDPM/runners/diffusion.py:1075-1180: adaga, cosine alpha, clip twice, EMAHelper)
  sd_unet.npz   LDM UNetModel model_channels 32, 8x8 latents, 5-token context of width 24: forward + gradients, 2 iterations of
                the nsfw_removal.py:108-173 loop body (xattn)
The weights are inputs: regenerated here with the generator the fixture script used (oracle classes under a seed) and checked
against the fixture's per-tensor sums before use.  Tolerances are those of bf16 GEMM operands (2^-9 per element): outputs
1.5e-2 rel-L2, per-tensor gradients 1.2e-2 (DiT) / 3.5e-2 (DDPM) / 6e-2 (LDM UNet) = 2 x the worst measured, stated at each assert."""
import math
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _proj(t, name):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return float((t.detach().double().cpu().flatten() * torch.randn(t.numel(), generator=g, dtype=torch.float64)).sum())


def _check_grads(named_grads, G, tol, what):
    """every gradient tensor against the reference's norm and seeded projection (|<g - g_ref, r>| <= tol * |g_ref| * O(1): r is
    a unit-variance Gaussian vector, so the projection of an error of relative size e is ~ e * |g_ref|), ten tensors element-wise.
    Prints the worst measured value of each kind (what `tol` is set from: 2 x the worst)."""
    names, norms, projs = list(G["grad_names"]), G["grad_norms"], G["grad_proj"]
    gmax = float(norms.max())
    seen = 0
    w_norm = w_proj = w_elem = 0.0
    for n, gn, gp in zip(names, norms, projs):
        g = named_grads[str(n)]
        assert g is not None and torch.isfinite(g).all(), n
        if gn < 1e-3 * gmax:                      # (near-)zero gradients: compared by magnitude
            assert g.norm().item() < 3e-2 * gmax + 3 * gn, (what, n, g.norm().item(), gn)
            continue
        w_norm = max(w_norm, abs(g.double().norm().item() - gn) / gn)
        w_proj = max(w_proj, abs(_proj(g, str(n)) - gp) / (4.0 * gn))
        assert abs(g.double().norm().item() - gn) < tol * gn, (what, n, g.norm().item(), gn)
        assert abs(_proj(g, str(n)) - gp) < 4.0 * tol * gn, (what, n, _proj(g, str(n)), gp, gn)
        seen += 1
    assert seen > len(names) // 2
    for k in G.files:
        if k.startswith("grad::") or k.startswith("grad/"):
            n = k.split("::")[-1] if "::" in k else k[len("grad/"):]
            want = torch.from_numpy(G[k])
            if want.norm().item() < 1e-3 * gmax:
                continue
            w_elem = max(w_elem, _rel(named_grads[n], want))
            assert _rel(named_grads[n], want) < tol, (what, n, _rel(named_grads[n], want))
    print(f"{what} gradients vs the reference fixture: worst norm error {w_norm:.2e}, worst projection error / 4 {w_proj:.2e}, worst element-wise "
          f"rel-L2 {w_elem:.2e} (bound {tol:.1e})")



_LAST = {}          # tests/debug/print_update_cosines.py reads the per-tensor numbers of the last run from here


def _update_cosines(U, got_update, skip=()):
    """Per-tensor cosine between the HIP path's parameter update and the reference's (tests/golden/*_updates.npz: every trainable
    tensor's (p_final - p0) / lr as int8 in units of 1 / scale, tensors above `cap` coordinates as a strided sample), and the ratio of
    the update norms.  got_update(name) -> the HIP path's p_final - p0 (CPU tensor).  Returns {name: (cosine, norm ratio, ref norm / sqrt(n))}."""
    lr, scale, cap = float(U["lr"]), float(U["scale"]), int(U["cap"])
    out = {}
    for n in [str(x) for x in U["names"]]:
        if n in skip:
            continue
        du = (got_update(n).double().flatten() / lr)
        stride = -(-du.numel() // cap)
        ref = torch.from_numpy(U["upd::" + n].astype(np.float64)) / scale
        a = du[::stride]
        assert a.numel() == ref.numel(), n
        cos = float((a * ref).sum() / (a.norm() * ref.norm() + 1e-30))
        out[n] = (cos, float(du.norm()) / (float(U["norm::" + n]) + 1e-30), float(U["norm::" + n]) / du.numel() ** 0.5)
    return out

# ------------------------------------------------------------------------------------------------ DiT
# bounds = 2 x the worst value measured on MI355X (tests/debug/print_update_cosines.py: DiT min cosine 0.9966, norm ratios within 0.9 %;
# DDPM 146 tensors, min cosine 0.9908, norm ratios within 0.7 %)
DIT_UPDATE_COS_MIN, DIT_UPDATE_NORM_TOL = 0.993, 0.02
DDPM_UPDATE_COS_MIN, DDPM_UPDATE_NORM_TOL = 0.98, 0.02
GPU_DIT = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10)


def _dit_from_fixture(G, batch):
    from oracle import dit_ref            # weight GENERATOR only (inputs); every expected number below comes from the fixture
    from sfron import dit
    torch.manual_seed(4242)
    src = dit_ref.DiT(**GPU_DIT)
    dit_ref.randomize_zero_init(src, std=0.05, seed=4243)
    sd = src.state_dict()
    assert list(sd.keys()) == [str(k) for k in G["param_names"]]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], G["param_sums"], rtol=1e-9, atol=1e-9)
    model = dit.DiT(batch_size=batch, **GPU_DIT)
    model.load_state_dict(sd)
    return model, sd


def test_dit_forward_backward_vs_reference_fixture():
    G = np.load(os.path.join(GOLD, "dit_gpu.npz"))
    model, _ = _dit_from_fixture(G, 4)
    x, t, y = (torch.from_numpy(G[k]).to(DEV) for k in ("x", "t", "y"))
    drop, w = torch.from_numpy(G["drop"]).to(DEV), torch.from_numpy(G["w"]).to(DEV)
    model.eval()
    with torch.no_grad():
        assert _rel(model(x, t, y), G["out_eval"]) < 1.5e-2
    model.train()
    out = model(x, t, y, force_drop_ids=drop)
    assert _rel(out, G["out_drop"]) < 1.5e-2
    model.zero_grad()
    (out * w).sum().backward()
    # 2 x the worst measured on MI355X (round 5: norm 1.3e-3, projection / 4 2.7e-3, element-wise 5.5e-3; the bound was 4e-2)
    _check_grads({n: p.grad for n, p in model.named_parameters()}, G, 1.2e-2, "DiT")


def test_dit_sfron_trajectory_vs_reference_fixture():
    """3 iterations of DiT/forget.py:256-322 (ga, 50 % mask, clip 1.0, AdamW lr 1e-3 x2, EMA 0.9) on the fused HIP runner against
    the losses / gradient norms / parameter updates the reference's functions produced."""
    from sfron import diffusion, step
    G = np.load(os.path.join(GOLD, "dit_gpu.npz"))
    model, sd = _dit_from_fixture(G, 4)
    model.train()
    gm = torch.Generator().manual_seed(int(G["traj_mask_seed"]))
    names = [str(n) for n in G["traj_names"]]
    mask = {"module." + n: (torch.rand(sd[n].shape, generator=gm) < 0.5) for n in names}
    mask["module.pos_embed"] = 0
    runner = step.DiTSFRon(model, diffusion.create_diffusion("", device=DEV), lr=float(G["traj_lr"]), forget_alpha=float(G["traj_forget_alpha"]),
                           grad_clip=1.0, ema_decay=float(G["traj_ema_decay"]), mask=mask, unlearn_loss="ga", forget_class=3)
    p0 = {n: sd[n].clone() for n in names}
    for s in range(3):
        b = {st: {k: torch.from_numpy(G[f"s{s}_{st}_{k}"]).to(DEV) for k in ("x0", "t", "noise", "y", "drop")} for st in ("forget", "remain")}
        for st in b:
            b[st]["drop"] = b[st]["drop"].to(torch.uint8)
        out = runner.step(b["forget"], b["remain"])
        torch.cuda.synchronize()
        fm, rm = out["forget_mse"].mean().item(), out["remain_mse"].mean().item()
        fl = -(out["forget_mse"] + out["forget_vb"]).mean().item()
        rl = (out["remain_mse"] + out["remain_vb"]).mean().item()
        assert abs(fm - G["traj_forget_mse"][s]) < 2e-2 * abs(G["traj_forget_mse"][s]), (s, fm)
        assert abs(rm - G["traj_remain_mse"][s]) < 2e-2 * abs(G["traj_remain_mse"][s]), (s, rm)
        assert abs(fl - G["traj_forget_loss"][s]) < 3e-2 * abs(G["traj_forget_loss"][s]), (s, fl)
        assert abs(rl - G["traj_remain_loss"][s]) < 3e-2 * abs(G["traj_remain_loss"][s]), (s, rl)
        gn = out["stats"][0].item()            # unclipped masked gradient norm of the forget stage (clip_grad_norm_'s return value)
        assert abs(gn - G["traj_gnorm"][s]) < 5e-2 * G["traj_gnorm"][s], (s, gn, G["traj_gnorm"][s])
    runner.guard.poll(block=True)
    eng = model.engine
    # parameter UPDATES after 6 Adam steps, EVERY trainable tensor against the reference's own update (round 4: the fixture now holds
    # the updates themselves; round 3 compared a norm and one seeded projection per tensor and let a tenth of the tensors miss).
    # Adam normalises every coordinate to ~lr, so a coordinate whose gradient is near zero flips sign under bf16 rounding: the
    # cosine bound is per tensor, no tensor is exempt except the K third of qkv.bias (exactly-zero gradient, below).
    U = np.load(os.path.join(GOLD, "dit_gpu_updates.npz"))
    D = GPU_DIT["hidden_size"]
    cs = _update_cosines(U, lambda n: eng.view(eng.params, n).detach().cpu() - p0[n], skip=[n for n in names if n.endswith("attn.qkv.bias")])
    _LAST["dit"] = cs
    worst = sorted(cs.items(), key=lambda kv: kv[1][0])[:5]
    for n, (cos, ratio, rms) in cs.items():
        assert rms > 0.05, (n, rms)                        # every tensor of this fixture moved by a non-negligible amount
        assert cos >= DIT_UPDATE_COS_MIN, (n, cos, worst)
        assert abs(ratio - 1.0) < DIT_UPDATE_NORM_TOL, (n, ratio)
    got_b, want_b = eng.view(eng.params, "blocks.1.attn.qkv.bias").detach().cpu(), torch.from_numpy(G["traj_final_qkv1_bias"])
    p0_b = p0["blocks.1.attn.qkv.bias"]
    for lo in (0, 2 * D):                          # Q and V thirds: the UPDATE, not the value (the value is dominated by the init)
        assert _rel(got_b[lo:lo + D] - p0_b[lo:lo + D], want_b[lo:lo + D] - p0_b[lo:lo + D]) < 0.25
    assert (got_b[D:2 * D] - p0_b[D:2 * D]).abs().max().item() <= 6.5 * float(G["traj_lr"])       # K third: at most lr per Adam step
    assert _rel(eng.view(eng.params, "blocks.0.mlp.fc1.bias"), G["traj_final_fc1_0_bias"]) < 2e-2
    assert _rel(eng.view(runner.ema, "blocks.1.attn.proj.bias"), G["traj_final_ema_proj1_bias"]) < 2e-2


# ------------------------------------------------------------------------------------------------ DDPM Conditional_Model
DDPM_GPU = dict(ch=128, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.0, in_channels=3,
                resolution=16, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1)


def _ddpm_from_fixture(G):
    from oracle import ddpm_ref           # weight generator only
    from sfron import unet
    torch.manual_seed(78)
    sd = ddpm_ref.ConditionalUNet(**DDPM_GPU).state_dict()
    assert list(sd.keys()) == [str(k) for k in G["keys"]]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], G["param_sums"], rtol=1e-9, atol=1e-9)
    model = unet.Conditional_Model(**{k: v for k, v in DDPM_GPU.items() if k not in ("out_ch", "in_channels", "resamp_with_conv", "cond_drop_prob")})
    model.load_state_dict({"module." + k: v for k, v in sd.items()})
    return model, sd


def test_ddpm_unet_forward_backward_vs_reference_fixture():
    G = np.load(os.path.join(GOLD, "ddpm_gpu.npz"))
    model, _ = _ddpm_from_fixture(G)
    x, c, w = (torch.from_numpy(G[k]).to(DEV) for k in ("x", "c", "w"))
    t = torch.from_numpy(G["t"]).float().to(DEV)
    model.eval()
    with torch.no_grad():
        got = model(x, t, c, mode="test", cond_scale=2.0)
    # (1 + s) cond - s null with s = 2: the two branches' bf16 errors (each < 1.5e-2 of its own norm) add with weights 3 and 2 while
    # the difference is no larger than one branch -> bound 3e-2 for the guided output, 1.5e-2 for the single branch below
    assert _rel(got, G["out_test_scale2"]) < 3e-2, _rel(got, G["out_test_scale2"])
    model.train()
    model.zero_grad()
    out = model(x, t, c, mode="train", cond_drop_prob=0.0)
    assert _rel(out, G["out_train_nodrop"]) < 1.5e-2
    (out * w).sum().backward()
    # 2 x the worst measured on MI355X (round 5: norm 3.6e-3, projection / 4 1.7e-2, element-wise 1.6e-2; the bound was 6e-2)
    _check_grads({n: p.grad for n, p in model.named_parameters()}, G, 3.5e-2, "DDPM")


def test_ddpm_sfron_trajectory_vs_reference_fixture():
    from sfron import ddpm
    G = np.load(os.path.join(GOLD, "ddpm_gpu.npz"))
    model, sd = _ddpm_from_fixture(G)
    gm = torch.Generator().manual_seed(8)
    mask = {n: (torch.rand(v.shape, generator=gm) < 0.5) for n, v in sd.items()}
    runner = ddpm.DDPMSFRon(model, lr=1e-3, forget_alpha=10.0, remain_alpha=1.0, grad_clip=1.0, ema_rate=1e-4, mask=mask, unlearn_loss="adaga",
                            lambd=0.5, n_iters=2, decay_forget_alpha=True, use_graphs=False)
    for s in range(2):
        f = dict(x0=G[f"s{s}_fx"], e=G[f"s{s}_fe"], t=G[f"s{s}_ft"], c=G[f"s{s}_fc"], keep_mask=G[f"s{s}_fkeep"])
        r = dict(x0=G[f"s{s}_rx"], e=G[f"s{s}_re"], t=G[f"s{s}_rt"], c=G[f"s{s}_rc"], keep_mask=G[f"s{s}_rkeep"])
        f, r = ({k: torch.from_numpy(v).to(DEV) for k, v in d.items()} for d in (f, r))
        out = runner.step(s, f, r)
        torch.cuda.synchronize()
        assert abs(out["alpha"] - G["traj_alpha"][s]) < 1e-6 * max(1.0, abs(G["traj_alpha"][s]))
        assert abs(out["forget_loss"].item() - G["traj_forget"][s]) < 3e-2 * abs(G["traj_forget"][s]), (s, out["forget_loss"].item())
        assert abs(out["remain_loss"].item() - G["traj_remain"][s]) < 3e-2 * abs(G["traj_remain"][s]), (s, out["remain_loss"].item())
    names = [str(n) for n in G["traj_names"]]
    views = runner.flat.named_views(runner.flat.p)
    # EVERY tensor's update against the reference's own (see the DiT test): cosine + norm ratio, no allowance for misses
    U = np.load(os.path.join(GOLD, "ddpm_gpu_updates.npz"))
    kb = [n for n in names if n.endswith(".k.bias")]
    for n in kb:
        # exactly-zero gradient (a shift of every key leaves the softmax unchanged): fp32 cancellation noise below Adam's eps in
        # the reference, bf16 noise above it here -> Adam turns it into +-lr steps along a direction the function ignores
        assert (views[n].detach().cpu() - sd[n]).abs().max().item() <= 4.5 * 1e-3
    cs = _update_cosines(U, lambda n: views[n].detach().cpu() - sd[n], skip=kb)
    _LAST["ddpm"] = cs
    worst = sorted(cs.items(), key=lambda kv: kv[1][0])[:5]
    for n, (cos, ratio, rms) in cs.items():
        assert rms > 0.05, (n, rms)
        assert cos >= DDPM_UPDATE_COS_MIN, (n, cos, worst)
        assert abs(ratio - 1.0) < DDPM_UPDATE_NORM_TOL, (n, ratio)
    assert _rel(views["conv_out.bias"], G["traj_final_conv_out_bias"]) < 5e-2
    assert _rel(runner.ema_state_dict()["norm_out.weight"], G["traj_final_shadow_norm_out"]) < 1e-3


# ------------------------------------------------------------------------------------------------ LDM UNetModel
SD_GRAD_NORM_TOL, SD_GRAD_REL_TOL = 2.3e-2, 4.6e-2      # 2 x measured on MI355X (round 6): 0.0115 / 0.0230 (were 6e-2 / 6e-2)
SD_TINY = dict(in_channels=4, out_channels=4, model_channels=32, attention_resolutions=(2, 1), num_res_blocks=1, channel_mult=(1, 2), num_heads=2,
               transformer_depth=1, context_dim=24)


def _sd_from_fixture():
    from oracle import sd_ref             # weight generator only (tests/golden/make_golden.py:sd_tiny_weights)
    from sfron import sd_unet
    torch.manual_seed(4321)
    src = sd_ref.UNetModel(**SD_TINY)
    sd_ref.randomize_zero_init(src, std=0.05, seed=4322)
    model = sd_unet.UNetModel(**SD_TINY)
    model.load_state_dict({"model.diffusion_model." + k: v for k, v in src.state_dict().items()})
    return model, src


def test_sd_unet_forward_backward_vs_reference_fixture():
    G = np.load(os.path.join(GOLD, "sd_unet.npz"))
    model, src = _sd_from_fixture()
    model.train()
    x, t, ctx, w = (torch.from_numpy(G[k]).to(DEV) for k in ("x", "t", "ctx", "w"))
    out = model(x, timesteps=t, context=ctx)
    assert _rel(out, G["out"]) < 1.5e-2, _rel(out, G["out"])
    (out * w).sum().backward()
    grads = {n: p.grad for n, p in model.named_parameters()}
    norms = G["grad_norms"]
    gmed = float(np.median(norms))
    worst_norm, worst_rel = (0.0, ""), (0.0, "")
    for (n, p), gn in zip(model.named_parameters(), norms):
        assert torch.isfinite(p.grad).all(), n
        if gn < 2e-3 * gmed:
            assert p.grad.norm().item() < 3e-2 * gmed, n
        else:
            worst_norm = max(worst_norm, (abs(p.grad.norm().item() - gn) / gn, n))
    rels = {k: _rel(grads[k[len("grad/"):]], G[k]) for k in G.files if k.startswith("grad/")}
    worst_rel = max((v, k) for k, v in rels.items())
    print(f"SD tiny UNet vs the reference fixture: worst gradient-norm error {worst_norm[0]:.4f} ({worst_norm[1]}), worst relative L2 error of a stored "
          f"gradient {worst_rel[0]:.4f} ({worst_rel[1]})")
    # bounds = 2 x measured on MI355X (round 6, profiles/r06_sd_parity.txt), as the DiT / DDPM fixtures state theirs
    assert worst_norm[0] < SD_GRAD_NORM_TOL, worst_norm
    assert worst_rel[0] < SD_GRAD_REL_TOL, worst_rel


# bounds = 2 x the worst value measured on MI355X (round 5; tests/debug/print_update_cosines.py prints the per-tensor numbers: "xattn" 35
# tensors, min cosine 0.9743, norm ratios within 1.5 %; "full" 273 tensors, min cosine 0.9435, norm ratios within 6.0 %).  Looser than the
# DiT / DDPM trajectories: the forget-stage loss of this script is an MSE between two eps-predictions of the SAME network (0.0066 here), so
# its gradients are small differences of bf16-rounded activations which Adam normalises to full-size steps.
SD_UPDATE_COS_MIN = {"xattn": 0.949, "full": 0.887}
SD_UPDATE_NORM_TOL = {"xattn": 0.031, "full": 0.12}
# Round 6 (profiles/r06_sd_parity.txt lists every tensor of "full" below 0.985 / beyond 3 %): the tensors of "full" that sit below
# cosine 0.95 (one: middle_block.2.out_layers.0.weight, 0.9435) or beyond 5 % in norm (one: input_blocks.1.0.out_layers.0.bias, 6.0 %) are NOT
# the near-zero-gradient class -- their reference gradients are at the median scale.  They are 1-D tensors (GroupNorm / LayerNorm affine
# parameters, biases): gradients that are sums over (batch x positions) of bf16-rounded products, and this small config has 8 ... 128 positions
# per sample at batch 2 -- few terms, so the rounding noise does not average out.  Every MATRIX (>= 2-D: convolution kernels, Linear weights;
# thousands of terms per element) is within cosine 0.9739 and 3 % of the reference: those carry the tighter bound asked for.
SD_UPDATE_COS_MIN_MATRIX = {"xattn": 0.97, "full": 0.95}
SD_UPDATE_NORM_TOL_MATRIX = {"xattn": 0.02, "full": 0.05}


@pytest.mark.parametrize("method", ["xattn", "full"])
def test_sd_nsfw_removal_trajectory_vs_reference_fixture(method):
    """Two iterations of SD/train-scripts/nsfw_removal.py:108-173 (forget stage on MSE(eps(x_t, c_nude), stopgrad eps(x_t, c_clothes)) with
    the SAME t and noise, Adam step; remain stage on the LDM eps-MSE, Adam step; no mask: the script's mask test is never true) on the HIP
    runner against numbers the REFERENCE UNetModel produced on the same inputs (tests/golden/sd_unet.npz: batches + losses of the
    train_method "xattn" trajectory; sd_unet_updates.npz: EVERY trainable tensor's update for "xattn" and for "full" on the same batches).
    Per tensor: cosine with the reference's update and ratio of the update norms -- no pooled metric, no allowance for misses."""
    from sfron import sd
    G = np.load(os.path.join(GOLD, "sd_unet.npz"))
    U = np.load(os.path.join(GOLD, "sd_unet_updates.npz"))
    model, src = _sd_from_fixture()
    sd0 = {k: v.clone() for k, v in src.state_dict().items()}
    run = sd.SDSFRon(model, lr=1e-3, forget_alpha=1.0, remain_alpha=1.0, train_method=method, mask=None, mask_mode="as_written", use_graphs=False)
    c_f = torch.from_numpy(G["traj_c_f"]).expand(2, -1, -1).contiguous().to(DEV)
    c_p = torch.from_numpy(G["traj_c_p"]).expand(2, -1, -1).contiguous().to(DEV)
    losses = U[method + "::losses"]
    for it in range(2):
        xf = torch.from_numpy(G["traj_xf"][it]).to(DEV)
        forget = dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.from_numpy(G["traj_t_f"][it]).to(DEV), noise=torch.from_numpy(G["traj_noise_f"][it]).to(DEV))
        remain = dict(x=torch.from_numpy(G["traj_xr"][it]).to(DEV), c=c_p, t=torch.from_numpy(G["traj_t_r"][it]).to(DEV),
                      noise=torch.from_numpy(G["traj_noise_r"][it]).to(DEV))
        got = run.step(forget, remain)
        torch.cuda.synchronize()
        assert got["forget_loss"].item() == pytest.approx(losses[it][0], rel=4e-2, abs=1e-5), (it, got["forget_loss"].item(), losses[it][0])
        assert got["remain_loss"].item() == pytest.approx(losses[it][1], rel=3e-2), (it, got["remain_loss"].item(), losses[it][1])
    names = [str(n) for n in U[method + "::names"]]
    view = lambda n: model.view(model.params, n).detach().cpu()
    for n in sd0:                                     # what the optimizer does not own stays bit for bit
        if n not in names:
            assert torch.equal(view(n), sd0[n]), n
    V = {"lr": U["lr"], "scale": U["scale"], "cap": U["cap"], "names": U[method + "::names"]}
    for n in names:
        V["upd::" + n], V["norm::" + n] = U[f"{method}::upd::{n}"], U[f"{method}::norm::{n}"]
    # Tensors whose gradient is EXACTLY zero analytically -- a per-channel constant in front of a GroupNorm with ONE channel per group (the
    # 32-channel levels of this small config: in_layers.2.bias, emb_layers.1.*, the last block's out / skip / ff / proj_out biases) -- hold
    # fp32 cancellation noise in the reference (gradient norms ~1e-8 against a median of ~1e-2: the fixture stores them) and bf16 noise here;
    # Adam normalises either into steps of up to lr along a direction the function ignores.  Compared by size: at most 4 Adam steps of lr.
    gf, gr = U[method + "::gnorm_forget"], U[method + "::gnorm_remain"]
    zero = {n for n, a, b in zip(names, gf, gr) if a < 1e-4 * float(np.median(gf)) and b < 1e-4 * float(np.median(gr))}
    assert len(zero) <= (13 if method == "full" else 0), sorted(zero)
    for n in zero:
        assert (view(n) - sd0[n]).abs().max().item() <= 4.5 * 1e-3, n
    cs = _update_cosines(V, lambda n: view(n) - sd0[n], skip=zero)
    _LAST["sd_" + method] = cs
    worst = sorted(cs.items(), key=lambda kv: kv[1][0])[:5]
    wr = sorted(cs.items(), key=lambda kv: -abs(kv[1][1] - 1.0))[:3]
    print(f"SD {method}: {len(cs)} tensors (+ {len(zero)} with an exactly-zero gradient), min update cosine {worst[0][1][0]:.4f} ({worst[0][0]}), "
          f"worst norm ratio {wr[0][1][1]:.4f} ({wr[0][0]})")
    mats = [n for n in cs if sd0[n].dim() >= 2]          # (the fixture stores the updates flattened: the shape is the parameter's)
    wm = min((cs[n][0], n) for n in mats)
    wmr = max((abs(cs[n][1] - 1.0), n) for n in mats)
    print(f"SD {method}: {len(mats)} matrices: min update cosine {wm[0]:.4f} ({wm[1]}), worst norm error {wmr[0]:.4f} ({wmr[1]})")
    for n, (cos, ratio, rms) in cs.items():
        assert rms > 0.05, (n, rms)
        matrix = sd0[n].dim() >= 2
        assert cos >= (SD_UPDATE_COS_MIN_MATRIX if matrix else SD_UPDATE_COS_MIN)[method], (n, cos, worst)
        assert abs(ratio - 1.0) < (SD_UPDATE_NORM_TOL_MATRIX if matrix else SD_UPDATE_NORM_TOL)[method], (n, ratio, wr)
