"""__graft_entry__.smoke(): one tiny SFR-on iteration on cuda:0 through the HIP path, checked against the oracle."""
import torch


def run():
    from oracle import diffusion_ref as dref
    from oracle import dit_ref, sfron_ref
    from . import data, diffusion, dit, step

    cfg = dict(input_size=32, patch_size=4, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10)
    torch.manual_seed(0)
    ref = dit_ref.DiT(**cfg)
    dit_ref.randomize_zero_init(ref, std=0.05, seed=1)
    model = dit.DiT(batch_size=4, **cfg)
    model.load_state_dict(ref.state_dict())
    model.train()
    gm = torch.Generator().manual_seed(5)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in ref.named_parameters() if p.requires_grad}
    mask["module.pos_embed"] = 0
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), lr=1e-3, forget_alpha=0.5, mask=mask)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), lr=1e-3, forget_alpha=0.5, mask=mask)
    kw = dict(global_batch=4, num_classes=10, forget_class=3)
    f, r = data.synthetic_batch(0, 0, "forget", **kw), data.synthetic_batch(0, 0, "remain", **kw)
    want = orc.step({k: v.long() if k == "drop" else v for k, v in f.items()},
                    {k: v.long() if k == "drop" else v for k, v in r.items()})
    got = runner.step({k: v.cuda() for k, v in f.items()}, {k: v.cuda() for k, v in r.items()})
    torch.cuda.synchronize()
    fm, rm = got["forget_mse"].mean().item(), got["remain_mse"].mean().item()
    assert abs(fm - want["forget_mse"]) < 2e-2 * abs(want["forget_mse"]) + 1e-3, (fm, want["forget_mse"])
    assert abs(rm - want["remain_mse"]) < 2e-2 * abs(want["remain_mse"]) + 1e-3, (rm, want["remain_mse"])
    worst = 0.0
    for name, p in ref.named_parameters():
        d = (model.engine.view(model.engine.params, name).cpu() - p.detach()).abs().max().item()
        worst = max(worst, d)
    assert worst < 3e-3, worst      # two Adam steps of lr 1e-3 each move a weight by <= 2e-3
    print(f"smoke ok: forget_mse {fm:.5f} (oracle {want['forget_mse']:.5f}), remain_mse {rm:.5f} "
          f"(oracle {want['remain_mse']:.5f}), max |dp| {worst:.2e}")
