"""GPU parity of the sampling path: respaced schedule, p_sample / p_sample_loop and classifier-free
guidance through the C ABI, against the reference's own outputs (tests/golden/dit_sampling.npz) for the stub model and
against the oracle (pinned to the reference by tests/test_oracle_golden.py) for the real DiT."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dit_sampling.npz"))


def test_space_timesteps_and_respaced_tables():
    from sfron import diffusion
    for name, (T, spec) in {"s250": (1000, "250"), "sddim25": (1000, "ddim25"), "s10": (1000, "10"), "ssec": (300, "10,15,20")}.items():
        assert sorted(diffusion.space_timesteps(T, spec)) == list(G["steps_" + name])
    d = diffusion.create_diffusion("10", device=DEV)
    assert d.num_timesteps == 10 and d.timestep_map == list(G["map10"])
    for k in ["betas", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]:
        np.testing.assert_array_equal(d.tables[k], G["tab10_" + k])
    full = diffusion.create_diffusion("", device=DEV)
    assert full.num_timesteps == 1000 and full._identity_map


def _stub():
    A = torch.from_numpy(G["A"]).to(DEV)
    seen = []

    def stub(x, ts, **kw):
        seen.append(ts.clone())
        return torch.einsum("oc,nchw->nohw", A, x) * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05
    return stub, seen


def test_p_sample_loop_stub_model_matches_reference():
    """clip_denoised=True trajectory of the fixture (the un-clamped loop: next test, on a contractive stub)."""
    from sfron import diffusion
    d = diffusion.create_diffusion("10", device=DEV)
    z = torch.from_numpy(G["z"])
    torch.manual_seed(77)                                   # the reference drew one randn_like(x) per step from this CPU stream
    step_noise = [torch.randn_like(z).to(DEV) for _ in range(10)]
    stub, seen = _stub()
    got = d.p_sample_loop(stub, z.shape, z.to(DEV), clip_denoised=True, model_kwargs={}, device=DEV, step_noise=step_noise)
    np.testing.assert_allclose(got.cpu().numpy(), G["stub_clip1"], rtol=2e-5, atol=2e-5)
    assert [int(s[0]) for s in seen] == list(G["map10"])[::-1]          # the model sees ORIGINAL timesteps, last first


def test_p_sample_loop_unclamped_contractive_stub_matches_reference():
    """clip_denoised=False (what DiT/forget.py:114-145 samples with) over all 10 respaced steps, against the REFERENCE's output for a stub
    whose eps-hat keeps pred_xstart bounded (fixture stub2_clip0: finite, O(0.1))."""
    from sfron import diffusion
    d = diffusion.create_diffusion("10", device=DEV)
    z = torch.from_numpy(G["z"])
    A = torch.from_numpy(G["A"]).to(DEV)
    s1m = torch.tensor(np.sqrt(1.0 - G["abar1000"]), dtype=torch.float32, device=DEV)
    C = z.shape[1]

    def stub2(x, ts, **kw):
        lin = torch.einsum("oc,nchw->nohw", A, x)
        eps = 0.9 * x / s1m[ts].view(-1, 1, 1, 1) + 0.02 * lin[:, :C]
        return torch.cat([eps, lin[:, C:] * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05], dim=1)
    torch.manual_seed(77)
    step_noise = [torch.randn_like(z).to(DEV) for _ in range(10)]
    got = d.p_sample_loop(stub2, z.shape, z.to(DEV), clip_denoised=False, model_kwargs={}, device=DEV, step_noise=step_noise)
    assert torch.isfinite(got).all()
    np.testing.assert_allclose(got.cpu().numpy(), G["stub2_clip0"], rtol=2e-4, atol=2e-5)


def test_p_sample_single_step_matches_reference():
    from sfron import diffusion
    d = diffusion.create_diffusion("10", device=DEV)
    z = torch.from_numpy(G["z"])
    torch.manual_seed(5)
    noise = torch.randn_like(z)
    stub, _ = _stub()
    out = d.p_sample(stub, z.to(DEV), torch.tensor([0, 9, 3, 0], device=DEV), clip_denoised=False, model_kwargs={}, noise=noise.to(DEV))
    np.testing.assert_allclose(out["sample"].cpu().numpy(), G["one_sample_seed5"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(out["pred_xstart"].cpu().numpy(), G["one_pred_xstart"], rtol=2e-6, atol=2e-6)
    # t == 0 rows carry no noise: identical whatever the draw
    out2 = d.p_sample(stub, z.to(DEV), torch.tensor([0, 9, 3, 0], device=DEV), clip_denoised=False, model_kwargs={}, noise=(noise * 3).to(DEV))
    assert torch.equal(out2["sample"][0], out["sample"][0]) and torch.equal(out2["sample"][3], out["sample"][3])


def test_guided_sampling_real_dit_vs_oracle():
    """DiT/forget.py:114-145: z doubled, null labels, cfg 4.0, clip_denoised=False -- 5 respaced steps on a small DiT."""
    from oracle import diffusion_ref
    from sfron import diffusion
    from test_gpu_dit import CASES, build_pair, rel_err
    cfg = CASES["hd64"]
    n = 3
    ref, model = build_pair(cfg, 2 * n, seed=4)
    ref.eval(); model.eval()
    g = torch.Generator().manual_seed(9)
    S = cfg["input_size"]
    zz = torch.randn(n, 4, S, S, generator=g)
    y = torch.tensor([1, 9, 4])
    zc, yc = torch.cat([zz, zz], 0), torch.cat([y, torch.tensor([cfg["num_classes"]] * n)], 0)
    tt = torch.tensor([999, 0, 500, 999, 0, 500])
    with torch.no_grad():
        want = ref.forward_with_cfg(zc, tt, yc, 4.0)
    got = model.forward_with_cfg(zc.to(DEV), tt.to(DEV), yc.to(DEV), 4.0)
    assert rel_err(got, want) < 2e-2
    assert torch.equal(got[:n, :3], got[n:, :3])            # both halves carry the guided epsilon
    noises = torch.randn(5, 2 * n, 4, S, S, generator=g)
    tab = diffusion_ref.DiffusionTables(1000, "5")
    want_s = diffusion_ref.p_sample_loop(tab, ref.forward_with_cfg, zc.shape, zc, clip_denoised=False,
                                         model_kwargs=dict(y=yc, cfg_scale=4.0), step_noise=noises)
    d = diffusion.create_diffusion("5", device=DEV)
    got_s = d.p_sample_loop(model.forward_with_cfg, zc.shape, zc.to(DEV), clip_denoised=False,
                            model_kwargs=dict(y=yc.to(DEV), cfg_scale=4.0), device=DEV, step_noise=noises.to(DEV))
    assert torch.isfinite(got_s).all()
    assert rel_err(got_s, want_s) < 3e-2


def test_latent_front_end_sample_kernel_and_loader(tmp_path):
    """vae.encode(x).latent_dist.sample().mul_(0.18215) (DiT/forget.py:265-267) from cached posterior moments: the sample kernel
    against torch (diffusers' DiagonalGaussianDistribution: logvar clamped to [-30, 20]), and the loader's device batches."""
    import numpy as np
    from sfron import _lib, latents
    from sfron._lib import check, ptr, stream_ptr
    g = torch.Generator().manual_seed(0)
    mom = torch.randn(5, 8, 32, 32, generator=g)
    mom[0, 4:] = 25.0; mom[1, 4:] = -40.0                       # clamp branches
    eps = torch.randn(5, 4, 32, 32, generator=g)
    out = torch.empty(5, 4, 32, 32, device=DEV)
    mom_d, eps_d = mom.to(DEV), eps.to(DEV)                      # named: the kernel reads them after this statement returns
    check(_lib.lib().sfron_latent_sample(ptr(mom_d), ptr(eps_d), 5, 4, 1024, 0.18215, ptr(out), stream_ptr()), "latent_sample")
    mean, lv = mom.chunk(2, dim=1)
    want = (mean + torch.exp(0.5 * lv.clamp(-30.0, 20.0)) * eps) * 0.18215
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=2e-6, atol=1e-6)
    rng = np.random.default_rng(1)
    for i, c in enumerate(("n0", "n1", "n2")):
        latents.write_shard(str(tmp_path), c, i, rng.standard_normal((6, 8, 32, 32)).astype(np.float16))
    ld = latents.UnlearnLatentLoader(latents.LatentCache(str(tmp_path)), forget_class=1, global_batch=4, device=DEV)
    f, r = ld.next("forget"), ld.next("remain")
    assert f["x0"].shape == (4, 4, 32, 32) and f["x0"].is_cuda and bool((f["y"] == 1).all()) and bool((r["y"] != 1).all())
    assert f["t"].dtype == torch.int64 and f["drop"].dtype == torch.uint8 and torch.isfinite(r["x0"]).all()
    f2 = ld.next("forget")                                       # the batch staged ahead on the copy stream
    assert not torch.equal(f2["noise"], f["noise"])
