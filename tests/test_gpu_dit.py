"""GPU parity: whole-model DiT forward/backward and the SFR-on iteration vs the oracle (CPU fp32).
The oracle's PatchEmbed / Attention / Mlp restate timm's published behaviour (timm is un-vendored and un-pinned by the reference):
the 4e-2 per-tensor gradient bounds below are against that restatement -- parity unpinned at timm."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    "hd64": dict(input_size=32, patch_size=4, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10),
    "hd72": dict(input_size=16, patch_size=2, in_channels=4, hidden_size=144, depth=3, num_heads=2, num_classes=10),
    # "rl" relabels to (forget_class + 100) % 1000 (DiT/forget.py:275-279): needs > 103 classes
    "hd64_nc200": dict(input_size=32, patch_size=4, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=200),
}


def build_pair(cfg, batch, seed=0):
    from oracle import dit_ref
    from sfron import dit
    torch.manual_seed(seed)
    ref = dit_ref.DiT(**cfg)
    dit_ref.randomize_zero_init(ref, std=0.05, seed=seed + 1)
    model = dit.DiT(batch_size=batch, **cfg)
    assert [n for n, _ in model.named_parameters()] == [n for n, _ in ref.named_parameters()]
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    model.load_state_dict(ref.state_dict())
    return ref, model


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("case", ["hd64", "hd72"])
def test_dit_forward_backward_vs_oracle(case):
    cfg = CASES[case]
    B = 4
    ref, model = build_pair(cfg, B)
    gen = torch.Generator().manual_seed(3)
    S = cfg["input_size"]
    x = torch.randn(B, 4, S, S, generator=gen)
    t = torch.tensor([0, 999, 17, 500])
    y = torch.tensor([1, 9, 4, 4])
    drop = torch.tensor([0, 1, 0, 0])
    w = torch.randn(B, 8, S, S, generator=gen) * 0.1
    ref.train()
    out_ref = ref(x, t, y, force_drop_ids=drop)
    (out_ref * w).sum().backward()
    model.train()
    out = model(x.to(DEV), t.to(DEV), y.to(DEV), force_drop_ids=drop.to(DEV))
    assert out.requires_grad
    assert rel_err(out, out_ref) < 1.5e-2
    model.zero_grad()
    (out * w.to(DEV)).sum().backward()
    worst = {}
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if not q.requires_grad:
            assert p.grad is None
            continue
        e = rel_err(p.grad, q.grad)
        worst[n] = e
        assert e < 4e-2, (n, e)
    # global gradient direction: cosine similarity of the flattened gradients
    gm = torch.cat([p.grad.flatten().cpu() for _, p in model.named_parameters() if p.grad is not None])
    gr = torch.cat([q.grad.flatten() for _, q in ref.named_parameters() if q.grad is not None])
    cos = torch.dot(gm, gr) / (gm.norm() * gr.norm())
    assert cos > 0.9995, cos


TINY = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2, num_classes=10)     # the fixtures' config: 16 tokens


def _tiny_from_fixture(g, batch):
    from oracle import dit_ref            # weight generator only (tests/golden/make_golden.py:tiny_weights)
    from sfron import dit
    torch.manual_seed(1234)
    src = dit_ref.DiT(**TINY)
    dit_ref.randomize_zero_init(src, std=0.05, seed=1235)
    sd = src.state_dict()
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["param_sums"], rtol=1e-9, atol=1e-9)
    model = dit.DiT(batch_size=batch, **TINY)
    model.load_state_dict(sd)
    return model, sd


def test_dit_matches_golden_reference_output(golden_dir):
    """tests/golden/dit_model.npz -- the reference DiT class (DiT/models.py:145-248) at 16 tokens (a patch-8-style short sequence,
    heads of 32): pos_embed table bit for bit, eval forward, and the ten gradient tensors + every gradient norm the reference's
    autograd produced, against the HIP engine DIRECTLY (short-sequence attention kernels, generic GEMM tiles)."""
    from sfron import dit
    g = np.load(os.path.join(golden_dir, "dit_model.npz"))
    pe = dit.get_2d_sincos_pos_embed(64, 4)
    np.testing.assert_array_equal(pe.astype(np.float32)[None], g["pos_embed"])
    model, _ = _tiny_from_fixture(g, 3)
    x, t, y, w = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "y", "w"))
    model.eval()
    out = model(x, t, y)
    assert rel_err(out, torch.from_numpy(g["out_eval"])) < 1.5e-2
    model.zero_grad()
    (out * w).sum().backward()
    norms = g["grad_norms"]
    gmax = norms.max()
    for (n, p), gn in zip(model.named_parameters(), norms):
        if gn < 0:                                   # the reference had no gradient there (frozen pos_embed)
            assert p.grad is None, n
        elif gn < 1e-3 * gmax:
            assert p.grad.norm().item() < 3e-2 * gmax
        else:
            assert abs(p.grad.norm().item() - gn) < 4e-2 * gn, (n, p.grad.norm().item(), gn)
    grads = {n: p.grad for n, p in model.named_parameters()}
    for k in g.files:
        if k.startswith("grad::"):
            n = k[len("grad::"):]
            if n == "blocks.0.attn.qkv.bias":         # K third: exactly zero gradient, cancellation noise on both sides
                D = TINY["hidden_size"]
                for lo in (0, 2 * D):
                    assert rel_err(grads[n][lo:lo + D], torch.from_numpy(g[k])[lo:lo + D]) < 4e-2, n
                continue
            assert rel_err(grads[n], torch.from_numpy(g[k])) < 4e-2, (n, rel_err(grads[n], torch.from_numpy(g[k])))


def test_sfron_trajectory_vs_golden_reference(golden_dir):
    """tests/golden/dit_sfron_traj.npz: 3 iterations of DiT/forget.py:256-322 composed from the reference's own functions (16 tokens),
    against the fused HIP runner directly: per-step losses and gradient norms, final parameter sums, two tensors in full."""
    from sfron import diffusion, step
    g = np.load(os.path.join(golden_dir, "dit_sfron_traj.npz"))
    gm_ = np.load(os.path.join(golden_dir, "dit_model.npz"))
    model, sd = _tiny_from_fixture(gm_, 4)
    model.train()
    gen = torch.Generator().manual_seed(int(g["mask_seed"]))
    names = [str(n) for n in g["names"]]
    mask = {"module." + n: (torch.rand(sd[n].shape, generator=gen) < 0.5) for n in names if n != "pos_embed"}
    mask["module.pos_embed"] = 0
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), lr=float(g["lr"]), forget_alpha=float(g["forget_alpha"]), grad_clip=1.0,
                           ema_decay=float(g["ema_decay"]), mask=mask, unlearn_loss="ga", forget_class=3)
    for s_ in range(3):
        b = {st: {k: torch.from_numpy(g[f"s{s_}_{st}_{k}"]).to(DEV) for k in ("x0", "t", "noise", "y", "drop")} for st in ("forget", "remain")}
        for st in b:
            b[st]["drop"] = b[st]["drop"].to(torch.uint8)
        out = runner.step(b["forget"], b["remain"])
        torch.cuda.synchronize()
        assert out["forget_mse"].mean().item() == pytest.approx(float(g["forget_mse"][s_]), rel=2e-2)
        assert out["remain_mse"].mean().item() == pytest.approx(float(g["remain_mse"][s_]), rel=2e-2)
        assert -(out["forget_mse"] + out["forget_vb"]).mean().item() == pytest.approx(float(g["forget_loss"][s_]), rel=3e-2)
        assert (out["remain_mse"] + out["remain_vb"]).mean().item() == pytest.approx(float(g["remain_loss"][s_]), rel=3e-2)
        assert out["stats"][0].item() == pytest.approx(float(g["gnorm"][s_]), rel=5e-2)
    runner.guard.poll(block=True)
    eng = model.engine
    # Adam moves a coordinate by <= lr per step whatever its gradient: |sum| of a tensor may differ by (#coords whose tiny gradient
    # changes sign) * lr; compare on the scale of the reference's absolute sums
    for n, want, wabs in zip(names, g["final_param_sums"], g["final_param_abs"]):
        if n.endswith("attn.qkv.bias"):
            continue        # K third: exactly-zero gradient; bf16 cancellation noise above Adam's eps becomes +-lr steps (harmless direction)
        got = eng.view(eng.params, n).double().sum().item()
        assert abs(got - want) < 2e-3 * wabs + 1e-6, (n, got, want)
    assert rel_err(eng.view(eng.params, "blocks.0.attn.qkv.weight"), torch.from_numpy(g["final_qkv0"])) < 2e-2
    assert rel_err(eng.view(runner.ema, "blocks.1.mlp.fc1.bias"), torch.from_numpy(g["final_ema_fc1"])) < 2e-2


@pytest.mark.parametrize("B,T,H,hd", [(2, 16, 2, 32), (3, 16, 6, 64), (1, 16, 16, 72), (2, 36, 3, 40), (2, 4, 1, 8)])
def test_short_sequence_attention_vs_torch(B, T, H, hd):
    """T < 64 (the registry's patch-8 models at 256 px: 16 tokens) on the plain-FMA kernels, forward and backward against torch
    autograd in fp32 on the same bf16 inputs."""
    from sfron import ops
    gen = torch.Generator().manual_seed(T + H + hd)
    D = H * hd
    qkv = (torch.randn(B * T, 3 * D, generator=gen) * 1.2).to(torch.bfloat16).to(DEV)
    d_o = (torch.randn(B * T, D, generator=gen) * 0.2).to(torch.bfloat16).to(DEV)
    o, lse = ops.attn_fwd(qkv, B, T, H, hd)
    dqkv = ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hd)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    s_ = (q * hd ** -0.5) @ k.transpose(-2, -1)
    ref = (s_.softmax(-1) @ v).transpose(1, 2).reshape(B * T, D)
    ref.backward(d_o.float())
    assert rel_err(o, ref) < 6e-3
    np.testing.assert_allclose(lse.view(B, H, T).cpu().numpy(), torch.logsumexp(s_, -1).detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert rel_err(dqkv, x.grad) < 1e-2
    assert torch.equal(dqkv, ops.attn_bwd(qkv, o, d_o, lse, B, T, H, hd))


@pytest.mark.parametrize("name", ["DiT-S/8", "DiT-B/8"])
def test_registry_patch8_models_run(name):
    """DiT/models.py:328-370 registers */8 models; at 256 px they see 16 tokens.  Forward + backward against the oracle."""
    from oracle import dit_ref
    from sfron import dit
    B = 2
    torch.manual_seed(3)
    ref = dit_ref.build(name, input_size=32)
    dit_ref.randomize_zero_init(ref, std=0.02, seed=4)
    model = dit.DiT_models[name](input_size=32, num_classes=1000, batch_size=B)
    model.load_state_dict(ref.state_dict())
    assert model.engine.tokens == 16
    gen = torch.Generator().manual_seed(8)
    x, t, y = torch.randn(B, 4, 32, 32, generator=gen), torch.tensor([5, 990]), torch.tensor([207, 3])
    w = torch.randn(B, 8, 32, 32, generator=gen) * 0.05
    ref.eval(); model.eval()
    out_ref = ref(x, t, y)
    (out_ref * w).sum().backward()
    out = model(x.to(DEV), t.to(DEV), y.to(DEV))
    assert rel_err(out, out_ref) < 1.5e-2
    (out * w.to(DEV)).sum().backward()
    gm = torch.cat([p.grad.flatten().cpu() for _, p in model.named_parameters() if p.grad is not None])
    gr = torch.cat([q.grad.flatten() for _, q in ref.named_parameters() if q.grad is not None])
    assert (torch.dot(gm, gr) / (gm.norm() * gr.norm())).item() > 0.9995


@pytest.mark.parametrize("case,loss,micro", [("hd64", "ga", 1), ("hd72", "ga", 1), ("hd64_nc200", "rl", 1), ("hd72", "ga", 2)])
def test_sfron_iterations_vs_oracle(case, loss, micro):
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    cfg = CASES[case]
    B = 4
    ref, model = build_pair(cfg, B, seed=7)
    model.train()
    gm = torch.Generator().manual_seed(5)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=mask, unlearn_loss=loss, forget_class=3)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), micro_batches=micro, **hp)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"])
    for it in range(3):
        f, r = data.synthetic_batch(1, it, "forget", **kw), data.synthetic_batch(1, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                        {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        assert got["forget_mse"].mean().item() == pytest.approx(want["forget_mse"], rel=3e-2)
        assert got["remain_mse"].mean().item() == pytest.approx(want["remain_mse"], rel=3e-2)
        assert got["stats"][0].item() == pytest.approx(want["forget_gnorm"], rel=5e-2)
    # parameter UPDATE (p - p0) agrees in direction and size; EMA follows
    eng = model.engine
    num = den = err = 0.0
    same = tot = 0
    for n, q in ref.named_parameters():
        if not q.requires_grad:
            continue
        du_ref = (q.detach() - p0[n]).flatten()
        du = (eng.view(eng.params, n).cpu() - p0[n]).flatten()
        num += torch.dot(du, du_ref).item()
        den += du_ref.norm().item() ** 2
        err += (du - du_ref).norm().item() ** 2
        big = du_ref.abs() > 0.05 * du_ref.abs().max()          # coordinates with a non-negligible oracle update
        same += int((torch.sign(du[big]) == torch.sign(du_ref[big])).sum())
        tot += int(big.sum())
    assert num / den > 0.9, num / den
    # (an absolute |du - du_ref| bound would be vacuous: Adam moves a weight by at most lr per step whatever the gradient.)
    # What carries information: the share of coordinates that moved the same way and the bulk error of the update vector.
    print(f"{case}/{loss}/micro{micro}: update sign agreement {same / tot:.4f}, bulk relative error {(err / den) ** 0.5:.3f}")
    assert same / tot > 0.995, same / tot                # measured 0.9998
    assert (err / den) ** 0.5 < 0.08, (err / den) ** 0.5   # measured 0.017 .. 0.031
    for n in ("blocks.0.mlp.fc1.weight", "pos_embed", "final_layer.linear.bias"):
        e = eng.view(runner.ema, n).cpu()
        assert torch.allclose(e, orc.ema[n], atol=5e-4), n
    assert runner.opt.step_count == 6


@pytest.mark.parametrize("batch", [1, 5])
def test_dit_odd_batch_sizes_and_label_drop_extremes(batch):
    """Ragged ends of the batch dimension (1 and 5 samples: M = 64 / 320 token rows) with every label dropped and none."""
    cfg = CASES["hd64"]
    ref, model = build_pair(cfg, batch, seed=2)
    gen = torch.Generator().manual_seed(batch)
    S = cfg["input_size"]
    x = torch.randn(batch, 4, S, S, generator=gen)
    t = torch.tensor([0, 999, 1, 998, 500][:batch])
    y = torch.randint(0, 10, (batch,), generator=gen)
    w = torch.randn(batch, 8, S, S, generator=gen) * 0.1
    for drop in (torch.ones(batch, dtype=torch.long), torch.zeros(batch, dtype=torch.long)):
        ref.train(); ref.zero_grad()
        out_ref = ref(x, t, y, force_drop_ids=drop)
        (out_ref * w).sum().backward()
        model.train(); model.zero_grad()
        out = model(x.to(DEV), t.to(DEV), y.to(DEV), force_drop_ids=drop.to(DEV))
        assert rel_err(out, out_ref) < 1.5e-2
        (out * w.to(DEV)).sum().backward()
        gm = torch.cat([p.grad.flatten().cpu() for _, p in model.named_parameters() if p.grad is not None])
        gr = torch.cat([q.grad.flatten() for _, q in ref.named_parameters() if q.grad is not None])
        assert torch.dot(gm, gr) / (gm.norm() * gr.norm()) > 0.9995
        # dropped labels touch only the null-class row of the embedding table (models.py:78-94)
        tab = dict(model.named_parameters())["y_embedder.embedding_table.weight"].grad
        rows = tab.abs().sum(1).cpu() > 0
        assert rows[cfg["num_classes"]] == bool(drop[0]) and int(rows.sum()) <= batch


def test_dit_b4_widths_vs_oracle():
    """BASELINE config 2 widths (DiT-B/4: D 768, 12 heads of 64, patch 4, 64 tokens) at depth 2, batch 8."""
    cfg = dict(input_size=32, patch_size=4, in_channels=4, hidden_size=768, depth=2, num_heads=12, num_classes=1000)
    B = 8
    ref, model = build_pair(cfg, B, seed=6)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, 4, 32, 32, generator=gen)
    t = torch.randint(0, 1000, (B,), generator=gen)
    y = torch.randint(0, 1000, (B,), generator=gen)
    drop = (torch.rand(B, generator=gen) < 0.3).long()
    w = torch.randn(B, 8, 32, 32, generator=gen) * 0.1
    ref.train()
    out_ref = ref(x, t, y, force_drop_ids=drop)
    (out_ref * w).sum().backward()
    model.train()
    out = model(x.to(DEV), t.to(DEV), y.to(DEV), force_drop_ids=drop.to(DEV))
    assert rel_err(out, out_ref) < 1.5e-2
    (out * w.to(DEV)).sum().backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.requires_grad:
            assert rel_err(p.grad, q.grad) < 4e-2, n


def test_checkpoint_format_and_resume(tmp_path):
    """DiT/forget.py:346-353: {"model","ema","opt","args"}; the reference side (oracle DiT + torch.optim.AdamW) loads it, and a
    runner resumed from it continues bit for bit."""
    from oracle import dit_ref
    from sfron import data, diffusion, step
    cfg = CASES["hd64"]
    B = 4
    ref, model = build_pair(cfg, B, seed=11)
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(2, it, "forget", **kw), data.synthetic_batch(2, it, "remain", **kw))
    runner.step(*bat(0))
    path = tmp_path / "0000001.pt"
    torch.save(runner.checkpoint(args={"lr": 2e-4}), path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "ema", "opt", "args"}
    # reference side: the model class and torch's AdamW accept the dicts as they are
    ref.load_state_dict(ck["model"], strict=True)
    ema_ref = dit_ref.DiT(**cfg)
    ema_ref.load_state_dict(ck["ema"], strict=True)
    opt = torch.optim.AdamW(ref.parameters(), lr=2e-4, weight_decay=0)
    opt.load_state_dict(ck["opt"])
    n_state = sum(1 for p in ref.parameters() if p.requires_grad)
    assert len(opt.state_dict()["state"]) == n_state and all(float(s["step"]) == 2.0 for s in opt.state_dict()["state"].values())
    # resume: a fresh runner loaded from the checkpoint takes the same next step
    runner.step(*bat(1))
    want = model.engine.params.clone()
    _, model2 = build_pair(cfg, B, seed=12)                      # different weights, then overwritten by the checkpoint
    runner2 = step.DiTSFRon(model2, diffusion.create_diffusion(""), **hp)
    runner2.load_checkpoint(ck)
    assert runner2.opt.step_count == 2
    runner2.step(*bat(1))
    assert torch.equal(model2.engine.params, want)
    assert torch.equal(runner2.ema, runner.ema)


def test_joint_method_vs_oracle():
    """method "joint" (DiT/forget.py:314-316): loss = remain + forget_alpha * forget, ONE AdamW step per iteration, no mask, no
    clip; three iterations against the oracle restating the same lines."""
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    cfg = CASES["hd64"]
    B = 4
    ref, model = build_pair(cfg, B, seed=6)
    model.train()
    hp = dict(lr=1e-3, forget_alpha=0.5, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3, method="joint")
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    p0 = {n: p.detach().clone() for n, p in ref.named_parameters()}
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"])
    for it in range(3):
        f, r = data.synthetic_batch(5, it, "forget", **kw), data.synthetic_batch(5, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()}, {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        assert got["forget_mse"].mean().item() == pytest.approx(want["forget_mse"], rel=3e-2)
        assert got["remain_mse"].mean().item() == pytest.approx(want["remain_mse"], rel=3e-2)
    runner.guard.poll(block=True)
    assert runner.opt.step_count == 3                       # one optimizer step per iteration
    eng = model.engine
    same = tot = 0
    for n, q in ref.named_parameters():
        if not q.requires_grad or n.endswith("attn.qkv.bias"):
            continue
        du_ref, du = (q.detach() - p0[n]).flatten(), (eng.view(eng.params, n).cpu() - p0[n]).flatten()
        big = du_ref.abs() > 0.05 * du_ref.abs().max()
        same += int((torch.sign(du[big]) == torch.sign(du_ref[big])).sum()); tot += int(big.sum())
    assert same / tot > 0.97, same / tot
    for n in ("blocks.1.mlp.fc2.bias", "final_layer.linear.weight"):
        assert rel_err(eng.view(runner.ema, n), orc.ema[n]) < 2e-2, n
    with pytest.raises(ValueError):
        step.DiTSFRon(model, diffusion.create_diffusion(""), method="sa")


def test_fused_clip_norm_equals_the_pass_over_the_gradient_arena():
    """step.DiTSFRon.fuse_clip_norm (single-process runs): the forget stage's clip norm is assembled from the weight-gradient GEMMs' own masked
    sums of squares + one launch over biases / embedders / final layer + the rank-(batch) adaLN range, instead of a pass over the gradient
    arena.  Same values in another summation order: the norm agrees to 1e-6, the parameters after three iterations to lr * 1e-5."""
    from sfron import data, diffusion, step
    cfg = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=192, depth=2, num_heads=3, num_classes=10)      # widths the 192 x 192 tile takes
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    gm = torch.Generator().manual_seed(77)

    def run(fused, early=True):
        ref, model = build_pair(cfg, B, seed=41)
        mask = {n: (torch.rand(p.shape, generator=torch.Generator().manual_seed(5 + i)) < 0.5) for i, (n, p) in enumerate(ref.named_parameters())
                if p.requires_grad}
        mask["pos_embed"] = 0
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=mask,
                               unlearn_loss="ga", forget_class=3)
        runner.fuse_clip_norm = fused
        runner.opt.early_ada = early            # the adaLN matrix's share of the norm on its own stream, behind sfron_aux_wait_ada (the default)
        assert (model.engine.fused_sumsq_plan() is not None)
        norms = []
        for it in range(3):
            out = runner.step(data.synthetic_batch(9, it, "forget", **kw), data.synthetic_batch(9, it, "remain", **kw))
            norms.append(out["stats"][0].item())
            if fused:
                assert runner._sq_buf is not None
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        return model.engine.params.clone(), norms
    p1, n1 = run(True)
    p2, n2 = run(True, early=False)
    assert n1 == n2 and torch.equal(p1, p2)         # the same launches on another stream: the same bits
    p0, n0 = run(False)
    for a, b in zip(n1, n0):
        assert a > 0 and abs(a - b) <= 1e-6 * b, (n1, n0)
    assert (p1 - p0).abs().max().item() <= 2e-4 * 1e-5
