"""GPU: (1) Fisher accumulation + mask generation vs the oracle; (2) the north-star acceptance check in its
tractable form: 50 SFR-on iterations of a small DiT, HIP path vs CPU oracle, eps-pred MSE within 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(input_size=32, patch_size=4, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10)


def build_pair(batch, seed=0):
    from oracle import dit_ref
    from sfron import dit
    torch.manual_seed(seed)
    ref = dit_ref.DiT(**CFG)
    dit_ref.randomize_zero_init(ref, std=0.05, seed=seed + 1)
    model = dit.DiT(batch_size=batch, **CFG)
    model.load_state_dict(ref.state_dict())
    return ref, model


def test_fisher_and_mask_vs_oracle():
    from oracle import diffusion_ref as dref
    from oracle import sweep_ref
    from sfron import data, diffusion, fisher
    B, n_iters = 4, 3
    ref, model = build_pair(B)
    ref.eval(); model.eval()           # generate_fisher.py leaves the model in eval for the remain loop; no label dropout
    tab = dref.DiffusionTables(1000)
    acc = {s: fisher.FisherAccumulator(model, diffusion.create_diffusion(""), n_iters) for s in ("forget", "remain")}
    F_ref = {s: {n: torch.zeros_like(p) for n, p in ref.named_parameters() if p.requires_grad} for s in ("forget", "remain")}
    kw = dict(global_batch=B, num_classes=10, forget_class=3)
    for s in ("forget", "remain"):
        for it in range(n_iters):
            b = data.synthetic_batch(5, it, s, **kw)
            ref.zero_grad()
            terms = dref.training_losses(tab, lambda x, t, y: ref(x, t, y), b["x0"], b["t"], dict(y=b["y"]), b["noise"])
            terms["loss"].mean().backward()
            for n, p in ref.named_parameters():
                if p.grad is not None:
                    sweep_ref.fisher_accumulate_(F_ref[s][n], p.grad, n_iters)
            bd = {k: v.to(DEV) for k, v in b.items()}
            bd["drop"] = None
            acc[s].accumulate(bd)
    sd_f, sd_r = acc["forget"].state_dict(), acc["remain"].state_dict()
    assert sd_f["module.pos_embed"] == 0
    for n in F_ref["forget"]:
        a, b = sd_f["module." + n], F_ref["forget"][n]
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        assert rel < 8e-2, (n, rel)          # g^2 doubles the bf16 gradient error
    masks = fisher.masks_from_fisher(sd_f, sd_r, 1.0)
    assert masks["module.pos_embed"] == 0
    # the mask kernel is bit-exact on ITS OWN Fisher inputs
    for n in ("blocks.0.mlp.fc1.weight", "final_layer.linear.bias"):
        want = sweep_ref.mask_from_fisher(sd_f["module." + n], sd_r["module." + n], 1.0)
        assert torch.equal(masks["module." + n], want)
    # and agrees with the oracle's mask except where the saliency ratio sits at the threshold
    m_ref = sweep_ref.mask_from_fisher(F_ref["forget"]["blocks.0.mlp.fc1.weight"], F_ref["remain"]["blocks.0.mlp.fc1.weight"], 1.0)
    agree = (masks["module.blocks.0.mlp.fc1.weight"] == m_ref).float().mean().item()
    assert agree > 0.9, agree


def test_fifty_step_eps_mse_within_1e4_of_oracle():
    """BASELINE.json acceptance: eps-pred MSE within 1e-4 of the reference path after 50 steps (small DiT so the CPU
    oracle finishes in seconds; same seeds, same masks, same hyper-parameters)."""
    from oracle import diffusion_ref as dref
    from oracle import sfron_ref
    from sfron import data, diffusion, step
    B = 4
    ref, model = build_pair(B, seed=3)
    model.train()
    gm = torch.Generator().manual_seed(11)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    hp = dict(lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=mask, unlearn_loss="ga", forget_class=3)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), **hp)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
    kw = dict(global_batch=B, num_classes=10, forget_class=3)
    worst = 0.0
    for it in range(50):
        f, r = data.synthetic_batch(9, it, "forget", **kw), data.synthetic_batch(9, it, "remain", **kw)
        want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                        {k: v.long() if k == "drop" else v for k, v in r.items()})
        got = runner.step({k: v.to(DEV) for k, v in f.items()}, {k: v.to(DEV) for k, v in r.items()})
        worst = max(worst, abs(got["remain_mse"].mean().item() - want["remain_mse"]),
                    abs(got["forget_mse"].mean().item() - want["forget_mse"]))
    # eval-mode eps-MSE of both models on a held-out batch after the 50 steps
    ref.eval()
    hb = data.synthetic_batch(10, 0, "remain", **kw)
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
    print(f"max per-step |mse gap| over 50 steps = {worst:.2e}; held-out eps-MSE gap after 50 steps = {final_gap:.2e}")
    assert final_gap < 1e-4, final_gap
    assert worst < 5e-3, worst        # per-step training-batch mse (bf16 forward noise on a loss of O(1))
