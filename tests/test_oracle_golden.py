"""CPU: pin the oracle (oracle/*.py) against golden vectors produced by importing the
reference (tests/golden/make_golden.py).  These run with -m "not gpu"."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_ref as dref
from oracle import dit_ref, sfron_ref, sweep_ref

TINY = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2, num_classes=10)


def tiny_model(seed=1234):
    torch.manual_seed(seed)
    m = dit_ref.DiT(**TINY)
    dit_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_tables_bit_exact(golden_dir):
    g = load(golden_dir, "dit_diffusion.npz")
    tab = dref.DiffusionTables(1000)
    for k in ["betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2"]:
        assert np.array_equal(getattr(tab, k), g["tab_" + k]), k     # fp64, bit-exact
    assert abs(tab.alphas_cumprod[999] - 4.0358e-5) < 1e-8


def test_training_losses_match_reference(golden_dir):
    g = load(golden_dir, "dit_diffusion.npz")
    tab = dref.DiffusionTables(1000)
    x0, noise, t = torch.from_numpy(g["x0"]), torch.from_numpy(g["noise"]), torch.from_numpy(g["t"])
    out = torch.from_numpy(g["model_output"]).clone().requires_grad_(True)
    x_t = dref.q_sample(tab, x0, t, noise)
    assert np.array_equal(x_t.numpy(), g["x_t"])
    terms = dref.training_losses(tab, lambda x, ts, **kw: out, x0, t, {}, noise)
    terms["loss"].mean().backward()
    for k in ("loss", "mse", "vb"):
        np.testing.assert_allclose(terms[k].detach().numpy(), g[k], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out.grad.numpy(), g["dloss_dout"], rtol=1e-5, atol=1e-9)
    # d(loss.mean())/d eps_hat = 2 (eps_hat - eps) / (C*H*W*N) exactly (mse only)
    N, C, H, W = x0.shape
    np.testing.assert_allclose(out.grad[:, :C].numpy(), (2 * (out.detach()[:, :C] - noise) / (C * H * W * N)).numpy(),
                               rtol=1e-6, atol=1e-10)


def test_dit_forward_backward_match_reference(golden_dir):
    g = load(golden_dir, "dit_model.npz")
    m = tiny_model()
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["param_names"])
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["param_sums"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(m.pos_embed.detach().numpy(), g["pos_embed"])
    x, t, y = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["y"])
    m.eval()
    out = m(x, t, y)
    np.testing.assert_allclose(out.detach().numpy(), g["out_eval"], rtol=1e-5, atol=1e-6)
    m.train()
    torch.manual_seed(99)
    np.testing.assert_allclose(m(x, t, y).detach().numpy(), g["out_train_seed99"], rtol=1e-5, atol=1e-6)
    m.eval()
    m.zero_grad()
    (m(x, t, y) * torch.from_numpy(g["w"])).sum().backward()
    for key in g.files:
        if key.startswith("grad::"):
            p = dict(m.named_parameters())[key[6:]]
            np.testing.assert_allclose(p.grad.numpy(), g[key], rtol=2e-4, atol=2e-5, err_msg=key)
    norms = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in m.named_parameters()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-6)


def test_sfron_trajectory_matches_reference(golden_dir):
    g = load(golden_dir, "dit_sfron_traj.npz")
    m = tiny_model()
    gm = torch.Generator().manual_seed(int(g["mask_seed"]))
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in m.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    orc = sfron_ref.DiTSfronOracle(m, dref.DiffusionTables(1000), lr=float(g["lr"]),
                                   forget_alpha=float(g["forget_alpha"]), grad_clip=1.0,
                                   ema_decay=float(g["ema_decay"]), mask=mask, unlearn_loss="ga")
    for step in range(3):
        b = {s: {k: torch.from_numpy(g[f"s{step}_{s}_{k}"]) for k in ("x0", "t", "noise", "y", "drop")}
             for s in ("forget", "remain")}
        r = orc.step(b["forget"], b["remain"])
        assert r["forget_loss"] == pytest.approx(float(g["forget_loss"][step]), rel=2e-5, abs=1e-6)
        assert r["remain_loss"] == pytest.approx(float(g["remain_loss"][step]), rel=2e-5, abs=1e-6)
        assert r["forget_mse"] == pytest.approx(float(g["forget_mse"][step]), rel=2e-5)
        assert r["forget_gnorm"] == pytest.approx(float(g["gnorm"][step]), rel=2e-4)
    names = [n for n, _ in m.named_parameters()]
    assert names == list(g["names"])
    np.testing.assert_allclose([float(p.double().abs().sum()) for _, p in m.named_parameters()],
                               g["final_param_abs"], rtol=2e-5)
    np.testing.assert_allclose(dict(m.named_parameters())["blocks.0.attn.qkv.weight"].detach().numpy(),
                               g["final_qkv0"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(orc.ema["blocks.1.mlp.fc1.bias"].numpy(), g["final_ema_fc1"], rtol=0, atol=3e-5)
    np.testing.assert_allclose([float(orc.ema[n].double().sum()) for n in names], g["final_ema_sums"],
                               rtol=1e-4, atol=1e-4)


def test_ddpm_losses_and_ema(golden_dir):
    g = load(golden_dir, "ddpm_loss.npz")
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.Conv2d(8, 3, 3, padding=1))
    emb = torch.nn.Embedding(10, 3)
    params = list(net.parameters()) + list(emb.parameters())
    np.testing.assert_array_equal(np.concatenate([p.detach().flatten().numpy() for p in params]), g["net_state"])
    x0, e, t, c = (torch.from_numpy(g[k]) for k in ("x0", "e", "t", "c"))
    b = sfron_ref.ddpm_get_betas()
    np.testing.assert_array_equal(b.numpy(), g["betas"])
    fn = lambda x, tf: net(x) * (1 + 0.001 * tf.view(-1, 1, 1, 1)) + emb(c).view(-1, 3, 1, 1)
    per = sfron_ref.ddpm_loss_per_sample(fn, x0, t, e, b)
    np.testing.assert_allclose(per.detach().numpy(), g["per_sample"], rtol=1e-6)
    np.testing.assert_allclose(per.mean(dim=0).detach().numpy(), g["simple"], rtol=1e-6)
    ada = sfron_ref.ddpm_adaptive_loss(per, 0.5)
    np.testing.assert_allclose(ada.detach().numpy(), g["adaptive"], rtol=1e-6)
    (-ada).backward()
    np.testing.assert_allclose(torch.cat([p.grad.flatten() for p in params]).numpy(), g["grad_neg_adaptive"],
                               rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose([sfron_ref.cosine_alpha(10.0, s, 50) for s in range(50)], g["cosine"], rtol=0, atol=0)
    shadow = [p.detach().clone() for p in net.parameters()]
    np.testing.assert_array_equal(np.concatenate([s.flatten().numpy() for s in shadow]), g["ema_before"])
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01)
    sweep_ref.ema_update_ddpm_(shadow, [p.data for p in net.parameters()], 1e-4)
    np.testing.assert_array_equal(np.concatenate([s.flatten().numpy() for s in shadow]), g["ema_after"])


def test_fisher_mask_bit_exact(golden_dir):
    g = load(golden_dir, "fisher_mask.npz")
    for th in g["ths"]:
        for k in ("a", "b"):
            got = sweep_ref.mask_from_fisher(torch.from_numpy(g[f"ff_{k}"]), torch.from_numpy(g[f"rf_{k}"]), float(th))
            assert got.dtype == torch.bool
            assert np.array_equal(got.numpy(), g[f"mask_{k}_{float(th)}"])
    # 0/0 -> (1e-15)/(1e-15) = 1 >= th is True for th <= 1
    z = torch.zeros(4)
    assert sweep_ref.mask_from_fisher(z, z, 1.0).all() and not sweep_ref.mask_from_fisher(z, z, 3.0).any()


# ------------------------------------------------------------------ DDPM (BASELINE config 0: CPU plumbing)
DDPM_TINY = dict(ch=128, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(4,), dropout=0.1, in_channels=3,
                 resolution=8, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1)


def test_ddpm_unet_matches_reference(golden_dir):
    from oracle import ddpm_ref
    g = load(golden_dir, "ddpm_model.npz")
    torch.manual_seed(77)
    m = ddpm_ref.ConditionalUNet(**DDPM_TINY)
    assert list(m.state_dict().keys()) == list(g["keys"])
    full = ddpm_ref.ConditionalUNet()          # cifar10_sfron.yml shape
    assert sum(p.numel() for p in full.parameters()) == int(g["n_params_full"]) == 38_632_323
    x, t, c, w = (torch.from_numpy(g[k]) for k in ("x", "t", "c", "w"))
    m.train()
    torch.manual_seed(5)
    out = m(x, t.float(), c, mode="train", cond_drop_prob=0.5)
    np.testing.assert_allclose(out.detach().numpy(), g["out_train_seed5"], rtol=1e-5, atol=1e-5)
    m.eval()
    np.testing.assert_allclose(m(x, t.float(), c, mode="test", cond_scale=2.0).detach().numpy(), g["out_test_scale2"],
                               rtol=1e-5, atol=2e-5)
    m.zero_grad()
    (m(x, t.float(), c, mode="train", cond_drop_prob=0.0) * w).sum().backward()
    for key in g.files:
        if key.startswith("grad::"):
            np.testing.assert_allclose(dict(m.named_parameters())[key[6:]].grad.numpy(), g[key], rtol=2e-4, atol=2e-5, err_msg=key)


def test_ddpm_sfron_trajectory_matches_reference(golden_dir):
    """BASELINE config 0 in miniature: SFR-on on the DDPM U-Net (adaga, cosine alpha, clip in both stages, EMAHelper),
    oracle vs the trajectory composed from the reference's own functions."""
    from oracle import ddpm_ref
    g = load(golden_dir, "ddpm_model.npz")
    torch.manual_seed(77)
    m = ddpm_ref.ConditionalUNet(**DDPM_TINY)
    gm = torch.Generator().manual_seed(8)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in m.named_parameters()}

    class Wrapped(torch.nn.Module):             # DDPMSfronOracle calls model(x, t_float, c, drop); RNG order as the reference
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, x, tf, c, drop=None):
            return self.net(x, tf, c, mode="train", cond_drop_prob=0.1)
    wm = Wrapped(m)
    orc = sfron_ref.DDPMSfronOracle(wm, sfron_ref.ddpm_get_betas(), lr=1e-3, forget_alpha=10.0, remain_alpha=1.0, grad_clip=1.0,
                                    ema_mu=1e-4, mask={"net." + k: v for k, v in mask.items()}, unlearn_loss="adaga",
                                    lambd=0.5, n_iters=2, decay_forget_alpha=True)
    for step in range(2):
        gt = lambda k: torch.from_numpy(g[f"s{step}_{k}"])
        torch.manual_seed(100 + step)
        r = orc.step(step, dict(x0=gt("fx"), e=gt("fe"), t=gt("ft"), c=gt("fc"), drop=None),
                     dict(x0=gt("rx"), e=gt("re"), t=gt("rt"), c=gt("rc"), drop=None))
        assert r["alpha"] == pytest.approx(float(g["traj_alpha"][step]))
        assert r["forget_loss"] == pytest.approx(float(g["traj_forget"][step]), rel=1e-4)
        assert r["remain_loss"] == pytest.approx(float(g["traj_remain"][step]), rel=1e-4)
    np.testing.assert_allclose(m.conv_in.weight.detach().numpy(), g["final_conv_in"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(orc.shadow["net.conv_out.weight"].numpy(), g["final_shadow_conv_out"], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ sampling path
def test_sampling_restatement_matches_reference(golden_dir):
    g = load(golden_dir, "dit_sampling.npz")
    for name, (T, spec) in {"s250": (1000, "250"), "sddim25": (1000, "ddim25"), "s10": (1000, "10"), "ssec": (300, "10,15,20")}.items():
        assert dref.space_timesteps(T, spec) == list(g["steps_" + name])
    tab = dref.DiffusionTables(1000, "10")
    assert tab.timestep_map == list(g["map10"])
    for k in ["betas", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod"]:
        np.testing.assert_array_equal(getattr(tab, k if k != "betas" else "betas"), g["tab10_" + k])
    z, A = torch.from_numpy(g["z"]), torch.from_numpy(g["A"])
    stub = lambda x, ts, **kw: torch.einsum("oc,nchw->nohw", A, x) * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05
    for clip in (True, False):
        torch.manual_seed(77)
        got = dref.p_sample_loop(tab, stub, z.shape, z, clip_denoised=clip, model_kwargs={})
        np.testing.assert_allclose(got.numpy(), g[f"stub_clip{int(clip)}"], rtol=1e-6, atol=1e-6)
    # the un-clamped loop on a CONTRACTIVE stub (the linear stub's un-clamped trajectory overflows: its fixture entry is NaN / 1e36 and
    # pins nothing beyond "the same overflow"): finite numbers of O(0.1), compared tightly
    assert np.isfinite(g["stub2_clip0"]).all() and np.abs(g["stub2_clip0"]).max() < 10
    s1m = torch.tensor(np.sqrt(1.0 - g["abar1000"]), dtype=torch.float32)
    C = z.shape[1]

    def stub2(x, ts, **kw):
        lin = torch.einsum("oc,nchw->nohw", A, x)
        eps = 0.9 * x / s1m[ts].view(-1, 1, 1, 1) + 0.02 * lin[:, :C]
        return torch.cat([eps, lin[:, C:] * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05], dim=1)
    torch.manual_seed(77)
    got = dref.p_sample_loop(tab, stub2, z.shape, z, clip_denoised=False, model_kwargs={})
    np.testing.assert_allclose(got.numpy(), g["stub2_clip0"], rtol=1e-5, atol=1e-6)
    torch.manual_seed(5)
    one = dref.p_sample(tab, stub, z, torch.tensor([0, 9, 3, 0]), clip_denoised=False, model_kwargs={})
    np.testing.assert_allclose(one["sample"].numpy(), g["one_sample_seed5"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(one["pred_xstart"].numpy(), g["one_pred_xstart"], rtol=1e-6, atol=1e-6)


def test_forward_with_cfg_and_guided_sampling_match_reference(golden_dir):
    g = load(golden_dir, "dit_sampling.npz")
    gm = load(golden_dir, "dit_model.npz")
    m = dit_ref.DiT(**TINY)
    torch.manual_seed(1234)
    m2 = dit_ref.DiT(**TINY)
    dit_ref.randomize_zero_init(m2, std=0.05, seed=1235)
    m.load_state_dict(m2.state_dict())
    np.testing.assert_allclose([float(v.double().sum()) for v in m.state_dict().values()], gm["param_sums"], rtol=1e-12)
    m.eval()
    zz, y = torch.from_numpy(g["cfg_z"]), torch.from_numpy(g["cfg_y"])
    n = zz.shape[0]
    zc, yc = torch.cat([zz, zz], 0), torch.cat([y, torch.tensor([10] * n)], 0)
    with torch.no_grad():
        out = m.forward_with_cfg(zc, torch.tensor([999, 0, 500, 999, 0, 500]), yc, 4.0)
    np.testing.assert_allclose(out.numpy(), g["cfg_forward"], rtol=1e-5, atol=1e-5)
    tab = dref.DiffusionTables(1000, "5")
    assert tab.timestep_map == list(g["map5"])
    smp = dref.p_sample_loop(tab, m.forward_with_cfg, zc.shape, zc, clip_denoised=False, model_kwargs=dict(y=yc, cfg_scale=4.0),
                                      step_noise=torch.from_numpy(g["cfg_step_noise"]))
    np.testing.assert_allclose(smp.numpy(), g["cfg_samples"], rtol=1e-4, atol=1e-4)


def test_ddpm_sampler_restatement_matches_reference(golden_dir):
    g = load(golden_dir, "ddpm_sampler.npz")
    b = sfron_ref.ddpm_get_betas()
    np.testing.assert_array_equal(sfron_ref.ddpm_compute_alpha(b, torch.tensor([0, 999, 500])).flatten().numpy(), g["alpha_t"])
    np.testing.assert_array_equal(sfron_ref.ddpm_compute_alpha(b, torch.tensor([-1])).flatten().numpy(), g["alpha_m1"])
    x, c, A = torch.from_numpy(g["x"]), torch.from_numpy(g["c"]), torch.from_numpy(g["A"])
    model = lambda xt, t, cc, cond_scale=3.0, mode="test": torch.einsum("oc,nchw->nohw", A, xt) * torch.cos(t / 300.0).view(-1, 1, 1, 1) + 0.01 * cc.view(-1, 1, 1, 1) * cond_scale
    for eta in (0.0, 0.5):
        torch.manual_seed(13)
        xs, x0s = sfron_ref.ddpm_generalized_steps_conditional(x, c, list(g["seq"]), model, b, cond_scale=2.0, eta=eta)
        np.testing.assert_allclose(xs[-1].numpy(), g[f"last_eta{eta}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(xs[5].numpy(), g[f"x_mid_eta{eta}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x0s[0].numpy(), g[f"x0_first_eta{eta}"], rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------- SD / LDM
def _sd_fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "sd_unet.npz"))


SD_TINY = dict(in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2],
               num_heads=2, transformer_depth=1, context_dim=24)


def _sd_tiny_model(seed=4321):
    from oracle import sd_ref
    torch.manual_seed(seed)
    m = sd_ref.UNetModel(**SD_TINY)
    sd_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m


def test_sd_unet_parameter_spec_matches_reference_v1(golden_dir):
    """Names, shapes and order of the v1-inference.yaml UNet (859,520,964 parameters in 686 tensors): the oracle class built on
    the meta device against the list the imported reference class produced (SD/ldm/modules/diffusionmodules/openaimodel.py:428-846)."""
    from oracle import sd_ref
    g = _sd_fixture(golden_dir)
    with torch.device("meta"):
        m = sd_ref.UNetModel()
    spec = "\n".join(f"{n} {tuple(p.shape)}" for n, p in m.named_parameters())
    want = bytes(g["v1_param_spec"]).decode()
    assert spec == want
    assert sum(p.numel() for p in m.parameters()) == int(g["v1_param_count"]) == 859_520_964
    assert len(list(m.named_parameters())) == 686
    assert sum(p.numel() for n, p in m.named_parameters() if "attn2" in n) == 43_962_560        # train_method "xattn"


def test_sd_unet_forward_backward_matches_reference(golden_dir):
    from oracle import sd_ref
    g = _sd_fixture(golden_dir)
    m = _sd_tiny_model()
    m.train()
    x, t, ctx, w = (torch.from_numpy(g[k]) for k in ("x", "t", "ctx", "w"))
    np.testing.assert_allclose(sd_ref.timestep_embedding(t, 32).numpy(), g["temb"], rtol=1e-6, atol=1e-6)
    out = m(x, timesteps=t, context=ctx)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    (out * w).sum().backward()
    norms = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(params[k[5:]].grad.numpy(), g[k], rtol=1e-3, atol=2e-6, err_msg=k)


def test_sd_schedule_and_q_sample_match_reference(golden_dir):
    from oracle import sd_ref
    g = _sd_fixture(golden_dir)
    s = sd_ref.LDMSchedule()
    np.testing.assert_array_equal(s.betas.numpy(), g["betas"].astype(np.float32))
    np.testing.assert_array_equal(s.sqrt_alphas_cumprod.numpy(), g["sqrt_ac"])
    np.testing.assert_array_equal(s.sqrt_one_minus_alphas_cumprod.numpy(), g["sqrt_1m_ac"])
    xt = s.q_sample(torch.from_numpy(g["q_x0"]), torch.from_numpy(g["t"]), torch.from_numpy(g["q_noise"]))
    np.testing.assert_array_equal(xt.numpy(), g["q_xt"])


def test_sd_nsfw_removal_trajectory_matches_reference(golden_dir):
    """Two iterations of train-scripts/nsfw_removal.py:108-173 (train_method "xattn"): losses per iteration and the trained /
    untouched weights against the trajectory composed from the imported reference UNet."""
    from oracle import sd_ref
    g = _sd_fixture(golden_dir)
    m = _sd_tiny_model()
    orc = sd_ref.SDSfronOracle(m, sd_ref.LDMSchedule(), lr=1e-3, forget_alpha=1.0, remain_alpha=1.0, train_method="xattn")
    c_f, c_p = torch.from_numpy(g["traj_c_f"]).expand(2, -1, -1), torch.from_numpy(g["traj_c_p"]).expand(2, -1, -1)
    for it in range(2):
        xf = torch.from_numpy(g["traj_xf"][it])
        forget = dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.from_numpy(g["traj_t_f"][it]), noise=torch.from_numpy(g["traj_noise_f"][it]))
        remain = dict(x=torch.from_numpy(g["traj_xr"][it]), c=c_p, t=torch.from_numpy(g["traj_t_r"][it]), noise=torch.from_numpy(g["traj_noise_r"][it]))
        got = orc.step(forget, remain)
        np.testing.assert_allclose([got["forget_loss"], got["remain_loss"]], g["traj_losses"][it], rtol=2e-4)
    params = dict(m.named_parameters())
    k = "input_blocks.1.1.transformer_blocks.0.attn2.to_v.weight"
    np.testing.assert_allclose(params[k].detach().numpy(), g["traj_final/" + k], rtol=1e-3, atol=2e-5)
    np.testing.assert_array_equal(params["time_embed.0.weight"].detach().numpy(), g["traj_untouched/time_embed.0.weight"])


def test_compvis_to_diffusers_export_matches_reference_mapping(golden_dir, tmp_path):
    """SD/train-scripts/convertModels.py:242-301,348-591 as savemodelDiffusers (:1006-1128) calls them: key mapping of the v1 UNet
    (686 tensors) and of a small config, config dict, and the saved file (host-side code: runs without a GPU)."""
    from oracle import sd_ref
    from sfron import export
    g = np.load(os.path.join(golden_dir, "compvis_export.npz"))
    for tag, kw in (("v1", dict()), ("small", dict(model_channels=32, channel_mult=(1, 2, 4), attention_resolutions=(2, 1), num_res_blocks=1,
                                                   num_heads=2, context_dim=24))):
        with torch.device("meta"):
            m = sd_ref.UNetModel(**kw)
        want = dict(line.split(" ") for line in bytes(g[tag + "_map"]).decode().split("\n"))        # new -> old (prefixed)
        sd = {"model.diffusion_model." + n: p for n, p in m.named_parameters()}
        sd["cond_stage_model.transformer.dummy"] = torch.empty(1, device="meta")
        got = export.compvis_unet_to_diffusers({"state_dict": sd}, layers_per_block=kw.get("num_res_blocks", 2))
        assert len(got) == len(want) == len(list(m.named_parameters()))
        ids = {id(v): k for k, v in sd.items()}
        assert {nk: ids[id(v)] for nk, v in got.items()} == want
        cfg = export.unet_diffusers_config(**kw)
        assert repr(sorted(cfg.items())) == bytes(g[tag + "_config"]).decode()
    # the save path: CompVis dict as it is + converted UNet, torch.load-able
    m = sd_ref.UNetModel(**kw)
    sd = {"model.diffusion_model." + n: p.detach() for n, p in m.named_parameters()}
    export.save_model(sd, tmp_path / "compvis-x.pt", tmp_path / "diffusers-x.pt", layers_per_block=1)
    back = torch.load(tmp_path / "diffusers-x.pt")
    assert torch.equal(back["conv_in.weight"], sd["model.diffusion_model.input_blocks.0.0.weight"])
    assert set(torch.load(tmp_path / "compvis-x.pt")) == set(sd)
