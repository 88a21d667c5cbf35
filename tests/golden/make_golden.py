#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference
(/root/reference, build container only -- the reference never travels to the GPU box).

Run:  python tests/golden/make_golden.py
Outputs (small .npz files, inputs + expected outputs only -- no reference source text):
  dit_diffusion.npz   tables + training_losses (loss/mse/vb + dL/d model_output) from
                      DiT/diffusion (create_diffusion("")), incl. t=0, t=999, |x0|>0.999 rows
  dit_model.npz       reference DiT class (DiT/models.py) forward + backward on a tiny config.
                      The three timm classes are NOT in /root/reference (un-vendored, un-pinned):
                      the harness supplies oracle.dit_ref.{PatchEmbed,Attention,Mlp} in their
                      place, so this fixture pins everything in models.py EXCEPT those classes.
  dit_gpu.npz         the same reference class at a shape the HIP engine accepts (64 tokens, D 128): forward / backward (a number per
                      gradient tensor: norm + a seeded random projection; ten tensors in full) and a 3-iteration SFR-on
                      trajectory -- consumed by tests/test_gpu_reference_fixtures.py with no oracle in between
  dit_sfron_traj.npz  3 SFR-on iterations composed from reference functions in the order of
                      DiT/forget.py:256-322 (forget.py itself needs torchvision/diffusers/CUDA)
  ddpm_loss.npz       DDPM/functions/losses.py (simple, adaptive), cosine schedule, EMAHelper
  fisher_mask.npz     DiT/generate_mask.py main() run on synthetic Fisher files (0/0, int-0 entries)
  ddpm_model.npz      DDPM/models/diffusion.py Conditional_Model forward/backward (tiny config) + a 2-iteration SFR-on
                      trajectory composed from DDPM/functions/losses.py, models/ema.py in runners/diffusion.py order
  ddpm_gpu.npz        the same class at ch 128 / 16 x 16 / attention over 64 tokens / dropout 0 with explicit keep masks, for the GPU tests
  compvis_export.npz  SD/train-scripts/convertModels.py create_unet_diffusers_config + convert_ldm_unet_checkpoint: the CompVis ->
                      diffusers key mapping of the v1 UNet and of a small config (names only; tensors pass through unchanged)
  sd_unet.npz         SD/ldm/modules/diffusionmodules/openaimodel.py UNetModel (+ attention.py, util.py): parameter spec of the
                      v1-inference.yaml UNet (meta device), forward/backward on a tiny config, the "linear" LDM schedule, and a
                      2-iteration trajectory of the nsfw_removal.py loop body composed from the reference UNet
  sd_unet_updates.npz every trainable tensor's update over that trajectory (train_method "xattn") and over the same batches with
                      train_method "full" (+ its losses): per-tensor cosines / norm ratios on the GPU (test_gpu_reference_fixtures.py)
"""
import argparse
import importlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import dit_ref  # noqa: E402  (stand-ins for the absent timm classes + weight generator)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_ref_dit_diffusion():
    sys.path.insert(0, os.path.join(REF, "DiT"))
    import diffusion as ref_diffusion
    return ref_diffusion


def import_ref_dit_models():
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Attention, vt.Mlp = dit_ref.PatchEmbed, dit_ref.Attention, dit_ref.Mlp
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm.models = timm_models
    timm_models.vision_transformer = vt
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.vision_transformer": vt})
    return _load("ref_dit_models", os.path.join(REF, "DiT", "models.py"))


TINY = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=2, num_classes=10)


def tiny_weights(seed=1234):
    torch.manual_seed(seed)
    m = dit_ref.DiT(**TINY)
    dit_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m.state_dict()


def gen_diffusion(ref_diffusion):
    d = ref_diffusion.create_diffusion(timestep_respacing="")
    g = torch.Generator().manual_seed(7)
    N, C, H = 8, 4, 8
    x0 = torch.randn(N, C, H, H, generator=g) * 0.7
    x0[0, 0, 0, :4] = torch.tensor([-1.0, 1.0, -0.9995, 0.9995])   # discretized-NLL edge branches
    x0[1, 1, 2, :2] = torch.tensor([-1.2, 1.3])
    noise = torch.randn(N, C, H, H, generator=g)
    t = torch.tensor([0, 0, 999, 1, 500, 37, 998, 250])
    out = (torch.randn(N, 2 * C, H, H, generator=g) * 0.8).requires_grad_(True)
    terms = d.training_losses(lambda x, ts, **kw: out, x0, t, model_kwargs={}, noise=noise)
    terms["loss"].mean().backward()
    x_t = d.q_sample(x0, t, noise=noise)
    tabs = {k: getattr(d, k) for k in ["betas", "alphas_cumprod", "sqrt_alphas_cumprod",
                                        "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                                        "sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped",
                                        "posterior_mean_coef1", "posterior_mean_coef2"]}
    np.savez_compressed(os.path.join(HERE, "dit_diffusion.npz"),
                        x0=x0.numpy(), noise=noise.numpy(), t=t.numpy(), model_output=out.detach().numpy(),
                        x_t=x_t.numpy(), loss=terms["loss"].detach().numpy(), mse=terms["mse"].detach().numpy(),
                        vb=terms["vb"].detach().numpy(), dloss_dout=out.grad.numpy(),
                        **{"tab_" + k: v for k, v in tabs.items()})
    return d


def gen_model(ref_models):
    sd = tiny_weights()
    m = ref_models.DiT(**TINY)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 4, 8, 8, generator=g)
    t = torch.tensor([3, 999, 421])
    y = torch.tensor([1, 9, 4])
    m.eval()
    out_eval = m(x, t, y)
    m.train()
    torch.manual_seed(99)          # label-dropout draw inside LabelEmbedder.token_drop (models.py:83)
    out_train = m(x, t, y)
    w = torch.randn(out_eval.shape, generator=g)
    m.zero_grad()
    m.eval()
    (m(x, t, y) * w).sum().backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    pick = ["x_embedder.proj.weight", "t_embedder.mlp.0.weight", "y_embedder.embedding_table.weight",
            "blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias", "blocks.0.adaLN_modulation.1.weight",
            "blocks.1.mlp.fc1.weight", "blocks.1.mlp.fc2.bias", "final_layer.linear.weight",
            "final_layer.adaLN_modulation.1.bias"]
    np.savez_compressed(os.path.join(HERE, "dit_model.npz"),
                        x=x.numpy(), t=t.numpy(), y=y.numpy(), w=w.numpy(), out_eval=out_eval.detach().numpy(),
                        out_train_seed99=out_train.detach().numpy(),
                        pos_embed=m.pos_embed.detach().numpy(),
                        param_names=np.array(list(sd.keys())),
                        param_sums=np.array([float(v.double().sum()) for v in sd.values()]),
                        grad_norms=np.array([float(grads[n].double().norm()) if n in grads else -1.0
                                             for n, _ in m.named_parameters()]),
                        **{"grad::" + n: grads[n].numpy() for n in pick})


def gen_traj(ref_models, ref_diffusion):
    """DiT/forget.py:256-322 composed from reference DiT + reference diffusion + torch AdamW."""
    from collections import OrderedDict
    from copy import deepcopy
    sd = tiny_weights()
    model = ref_models.DiT(**TINY)
    model.load_state_dict(sd)
    ema = deepcopy(model)
    diffusion = ref_diffusion.create_diffusion(timestep_respacing="")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0)
    gm = torch.Generator().manual_seed(5)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in model.named_parameters()
            if p.requires_grad}
    mask["module.pos_embed"] = 0
    model.train()
    forget_alpha, grad_clip, N = 0.5, 1.0, 4
    g = torch.Generator().manual_seed(2024)
    rec = {"forget_loss": [], "remain_loss": [], "forget_mse": [], "remain_mse": [], "gnorm": []}
    inputs = {}
    for step in range(3):
        batch = {}
        for stream in ("forget", "remain"):
            batch[stream] = dict(
                x0=torch.randn(N, 4, 8, 8, generator=g) * 0.8,
                t=torch.randint(0, 1000, (N,), generator=g),
                noise=torch.randn(N, 4, 8, 8, generator=g),
                y=(torch.full((N,), 3) if stream == "forget" else torch.randint(0, 10, (N,), generator=g)),
                drop=(torch.rand(N, generator=g) < 0.25).long())
            for k, v in batch[stream].items():
                inputs[f"s{step}_{stream}_{k}"] = v.numpy()
        # the reference draws drop ids with torch.rand inside forward; to make them an explicit
        # input we call the reference LabelEmbedder through its own force_drop_ids argument.
        def run(b):
            fd = b["drop"]
            def fwd(x, ts, y):
                xx = model.x_embedder(x) + model.pos_embed
                c = model.t_embedder(ts) + model.y_embedder(y, model.training, force_drop_ids=fd)
                for blk in model.blocks:
                    xx = blk(xx, c)
                return model.unpatchify(model.final_layer(xx, c))
            return diffusion.training_losses(fwd, b["x0"], b["t"], dict(y=b["y"]), noise=b["noise"])
        tf = run(batch["forget"])
        ori_forget = -tf["loss"].mean()
        opt.zero_grad()
        (forget_alpha * ori_forget).backward()
        for name, p in model.named_parameters():
            if p.grad is not None:
                p.grad *= mask["module." + name]
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        opt.step()
        tr = run(batch["remain"])
        ori_remain = tr["loss"].mean()
        opt.zero_grad()
        ori_remain.backward()
        opt.step()
        with torch.no_grad():
            ep, mp = OrderedDict(ema.named_parameters()), OrderedDict(model.named_parameters())
            for name, p in mp.items():
                ep[name].mul_(0.9).add_(p.data, alpha=1 - 0.9)
        rec["forget_loss"].append(ori_forget.item()); rec["remain_loss"].append(ori_remain.item())
        rec["forget_mse"].append(float(tf["mse"].mean())); rec["remain_mse"].append(float(tr["mse"].mean()))
        rec["gnorm"].append(float(gn))
    names = [n for n, _ in model.named_parameters()]
    np.savez_compressed(os.path.join(HERE, "dit_sfron_traj.npz"),
                        names=np.array(names),
                        final_param_sums=np.array([float(p.double().sum()) for _, p in model.named_parameters()]),
                        final_param_abs=np.array([float(p.double().abs().sum()) for _, p in model.named_parameters()]),
                        final_ema_sums=np.array([float(p.double().sum()) for _, p in ema.named_parameters()]),
                        final_qkv0=dict(model.named_parameters())["blocks.0.attn.qkv.weight"].detach().numpy(),
                        final_ema_fc1=dict(ema.named_parameters())["blocks.1.mlp.fc1.bias"].detach().numpy(),
                        mask_seed=np.array(5), lr=np.array(1e-3), forget_alpha=np.array(forget_alpha),
                        ema_decay=np.array(0.9),
                        **{k: np.array(v) for k, v in rec.items()}, **inputs)


GPU_DIT = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10)


def gpu_dit_weights(seed=4242):
    torch.manual_seed(seed)
    m = dit_ref.DiT(**GPU_DIT)
    dit_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m.state_dict()


def _proj(t, name):
    """<t, r> with r ~ N(0,1) seeded by the tensor's name: one number that pins a whole tensor (the test regenerates r)"""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return float((t.double().flatten() * torch.randn(t.numel(), generator=g, dtype=torch.float64)).sum())


def _save_updates(fname, names, fin, p0, lr, cap=1 << 30):
    """Every trainable tensor's UPDATE of a trajectory, (p_final - p0) / lr, as int8 in units of 1/20 (an Adam step moves a coordinate by
    at most ~lr: a handful of steps stay inside +-6.35; resolution 0.05 lr) -- enough for a per-tensor cosine against the HIP path's
    update (tests/test_gpu_reference_fixtures.py); the norms are stored exactly.  A tensor with more than `cap` coordinates is stored as a
    strided sample (every tensor still has an entry; the test samples the same way).  Separate file: the older fixtures keep their bits."""
    out = dict(names=np.array(names), lr=np.array(lr), scale=np.array(20.0), cap=np.array(cap))
    for n in names:
        u = ((fin[n].detach() - p0[n]).double() / lr).numpy().ravel()
        out["norm::" + n] = np.array(float(np.linalg.norm(u)))          # of the WHOLE tensor
        stride = -(-u.size // cap)                                       # tensors above `cap` coordinates: every stride-th one
        out["upd::" + n] = np.clip(np.rint(u[::stride] * 20.0), -127, 127).astype(np.int8)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written:", sum(int(np.prod(fin[n].shape)) for n in names), "update coordinates")


def gen_dit_gpu(ref_models, ref_diffusion):
    """The reference DiT class at a shape the HIP engine takes (16 x 16 latents, patch 2 -> 64 tokens, D 128, 2 heads of 64, depth 2):
    forward (eval / train with explicit drop ids), backward, and 3 SFR-on iterations in DiT/forget.py:256-322 order -- so that the
    GPU tests compare the HIP path with numbers the REFERENCE produced, with no oracle in between (timm stand-ins as above)."""
    from collections import OrderedDict
    from copy import deepcopy
    sd = gpu_dit_weights()
    m = ref_models.DiT(**GPU_DIT)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(311)
    N = 4
    x = torch.randn(N, 4, 16, 16, generator=g)
    t = torch.tensor([3, 999, 421, 0])
    y = torch.tensor([1, 9, 4, 3])
    drop = torch.tensor([0, 1, 0, 0])
    w = torch.randn(N, 8, 16, 16, generator=g) * 0.1

    def fwd(model, x_, t_, y_, fd):
        xx = model.x_embedder(x_) + model.pos_embed
        c = model.t_embedder(t_) + model.y_embedder(y_, True, force_drop_ids=fd)
        for blk in model.blocks:
            xx = blk(xx, c)
        return model.unpatchify(model.final_layer(xx, c))
    m.eval()
    out_eval = m(x, t, y)
    out_drop = fwd(m, x, t, y, drop)
    m.zero_grad()
    (fwd(m, x, t, y, drop) * w).sum().backward()
    names = [n for n, p in m.named_parameters() if p.grad is not None]
    grads = {n: dict(m.named_parameters())[n].grad.clone() for n in names}
    pick = ["x_embedder.proj.bias", "t_embedder.mlp.2.bias", "y_embedder.embedding_table.weight", "blocks.0.attn.qkv.bias",
            "blocks.0.attn.proj.bias", "blocks.1.mlp.fc1.bias", "blocks.1.mlp.fc2.bias", "blocks.1.adaLN_modulation.1.bias",
            "final_layer.linear.weight", "final_layer.adaLN_modulation.1.bias"]
    out = dict(x=x.numpy(), t=t.numpy(), y=y.numpy(), drop=drop.numpy(), w=w.numpy(), out_eval=out_eval.detach().numpy(),
               out_drop=out_drop.detach().numpy(), param_names=np.array(list(sd.keys())),
               param_sums=np.array([float(v.double().sum()) for v in sd.values()]),
               grad_names=np.array(names), grad_norms=np.array([float(grads[n].double().norm()) for n in names]),
               grad_proj=np.array([_proj(grads[n], n) for n in names]))
    out.update({"grad::" + n: grads[n].numpy() for n in pick})
    # ---- 3 SFR-on iterations (ga, mask, clip, AdamW x2, EMA 0.9) at batch 4
    model = ref_models.DiT(**GPU_DIT)
    model.load_state_dict(sd)
    ema = deepcopy(model)
    diffusion = ref_diffusion.create_diffusion(timestep_respacing="")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0)
    gm = torch.Generator().manual_seed(5)
    mask = {"module." + n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in model.named_parameters() if p.requires_grad}
    model.train()
    forget_alpha, grad_clip = 0.5, 1.0
    g2 = torch.Generator().manual_seed(2025)
    rec = {"forget_loss": [], "remain_loss": [], "forget_mse": [], "remain_mse": [], "gnorm": []}
    p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    for step in range(3):
        batch = {}
        for stream in ("forget", "remain"):
            batch[stream] = dict(x0=torch.randn(N, 4, 16, 16, generator=g2) * 0.8, t=torch.randint(0, 1000, (N,), generator=g2),
                                 noise=torch.randn(N, 4, 16, 16, generator=g2),
                                 y=(torch.full((N,), 3) if stream == "forget" else torch.randint(0, 10, (N,), generator=g2)),
                                 drop=(torch.rand(N, generator=g2) < 0.25).long())
            for k, v in batch[stream].items():
                out[f"s{step}_{stream}_{k}"] = v.numpy()

        def run(b):
            return diffusion.training_losses(lambda xx, ts, y: fwd(model, xx, ts, y, b["drop"]), b["x0"], b["t"], dict(y=b["y"]),
                                             noise=b["noise"])
        tf = run(batch["forget"])
        ori_forget = -tf["loss"].mean()
        opt.zero_grad()
        (forget_alpha * ori_forget).backward()
        for name, p in model.named_parameters():
            if p.grad is not None:
                p.grad *= mask["module." + name]
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        opt.step()
        tr = run(batch["remain"])
        ori_remain = tr["loss"].mean()
        opt.zero_grad()
        ori_remain.backward()
        opt.step()
        with torch.no_grad():
            ep, mp = OrderedDict(ema.named_parameters()), OrderedDict(model.named_parameters())
            for name, p in mp.items():
                ep[name].mul_(0.9).add_(p.data, alpha=1 - 0.9)
        rec["forget_loss"].append(ori_forget.item()); rec["remain_loss"].append(ori_remain.item())
        rec["forget_mse"].append(float(tf["mse"].mean())); rec["remain_mse"].append(float(tr["mse"].mean()))
        rec["gnorm"].append(float(gn))
    tn = [n for n, p in model.named_parameters() if p.requires_grad]
    fin = dict(model.named_parameters())
    out.update(traj_names=np.array(tn), traj_lr=np.array(1e-3), traj_forget_alpha=np.array(forget_alpha), traj_ema_decay=np.array(0.9),
               traj_mask_seed=np.array(5),
               traj_update_norms=np.array([float((fin[n].detach() - p0[n]).double().norm()) for n in tn]),
               traj_update_proj=np.array([_proj(fin[n].detach() - p0[n], n) for n in tn]),
               traj_final_qkv1_bias=fin["blocks.1.attn.qkv.bias"].detach().numpy(),
               traj_final_fc1_0_bias=fin["blocks.0.mlp.fc1.bias"].detach().numpy(),
               traj_final_ema_proj1_bias=dict(ema.named_parameters())["blocks.1.attn.proj.bias"].detach().numpy(),
               **{"traj_" + k: np.array(v) for k, v in rec.items()})
    np.savez_compressed(os.path.join(HERE, "dit_gpu.npz"), **out)
    _save_updates("dit_gpu_updates.npz", tn, fin, p0, 1e-3)
    print("dit_gpu.npz:", len(out), "entries;", "traj", rec)


def gen_ddpm():
    losses = _load("ref_ddpm_losses", os.path.join(REF, "DDPM", "functions", "losses.py"))
    ema_mod = _load("ref_ddpm_ema", os.path.join(REF, "DDPM", "models", "ema.py"))
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.SiLU(), torch.nn.Conv2d(8, 3, 3, padding=1))
    emb = torch.nn.Embedding(10, 3)
    params = list(net.parameters()) + list(emb.parameters())

    def model(x, tf, c, cond_drop_prob=0.1, mode="train"):
        return net(x) * (1 + 0.001 * tf.view(-1, 1, 1, 1)) + emb(c).view(-1, 3, 1, 1)

    g = torch.Generator().manual_seed(17)
    N = 6
    x0 = torch.rand(N, 3, 8, 8, generator=g) * 2 - 1
    e = torch.randn(N, 3, 8, 8, generator=g)
    t = torch.tensor([0, 999, 5, 500, 250, 750])
    c = torch.tensor([0, 1, 2, 3, 4, 5])
    b = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()   # runners/diffusion.py:36-66,83
    net_state0 = np.concatenate([p.detach().flatten().numpy().copy() for p in params])
    simple = losses.loss_registry_conditional["simple"](model, x0, t, c, e, b)
    per = losses.loss_registry_conditional["simple"](model, x0, t, c, e, b, keepdim=True)
    ada = losses.adaptive_loss(losses.loss_registry_conditional["simple"], model, x0, t, c, e, b, lambd=0.5)
    for p in params:
        p.grad = None
    (-ada).backward()
    g_ada = torch.cat([p.grad.flatten() for p in params])
    cos = np.array([losses.cosine_lr_scheduler(10.0, s, 50) for s in range(50)])
    helper = ema_mod.EMAHelper(mu=1e-4)
    helper.register(net)
    before = {k: v.clone() for k, v in helper.shadow.items()}
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.01)
    helper.update(net)
    np.savez_compressed(os.path.join(HERE, "ddpm_loss.npz"),
                        x0=x0.numpy(), e=e.numpy(), t=t.numpy(), c=c.numpy(), betas=b.numpy(),
                        net_state=net_state0,
                        simple=simple.detach().numpy(), per_sample=per.detach().numpy(),
                        adaptive=ada.detach().numpy(), grad_neg_adaptive=g_ada.numpy(), cosine=cos,
                        ema_before=np.concatenate([v.flatten().numpy() for v in before.values()]),
                        ema_after=np.concatenate([v.flatten().numpy() for v in helper.shadow.values()]))


def gen_mask():
    gm = _load("ref_generate_mask", os.path.join(REF, "DiT", "generate_mask.py"))
    g = torch.Generator().manual_seed(23)
    ff = {"module.a": torch.rand(64, 33, generator=g) ** 4, "module.b": torch.rand(257, generator=g) * 1e-12,
          "module.pos_embed": 0}
    rf = {"module.a": torch.rand(64, 33, generator=g) ** 4, "module.b": torch.rand(257, generator=g) * 1e-12,
          "module.pos_embed": 0}
    ff["module.a"][0, :5] = 0.0
    rf["module.a"][0, :3] = 0.0          # 0/0 and x/0 entries
    rf["module.a"][1, :4] = 0.0
    ff["module.b"][:7] = 0.0
    ths = [0.5, 1.0, 3.0]
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "207"))
        torch.save(ff, os.path.join(d, "207", "forget_fisher.pt"))
        torch.save(rf, os.path.join(d, "207", "remain_fisher.pt"))
        gm.main(argparse.Namespace(mask_path=d, forget_class=[207], thresholds=ths))
        out = {}
        for th in ths:
            mk = torch.load(os.path.join(d, "207", f"fisher_{th}.pt"))
            assert isinstance(mk["module.pos_embed"], int)
            out[f"mask_a_{th}"] = mk["module.a"].numpy()
            out[f"mask_b_{th}"] = mk["module.b"].numpy()
    np.savez_compressed(os.path.join(HERE, "fisher_mask.npz"), ff_a=ff["module.a"].numpy(),
                        rf_a=rf["module.a"].numpy(), ff_b=ff["module.b"].numpy(), rf_b=rf["module.b"].numpy(),
                        ths=np.array(ths), **out)


DDPM_TINY = dict(ch=128, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(4,), dropout=0.1, in_channels=3,
                 resolution=8, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1)   # ch must be 128 (reference quirk)


def ddpm_ref_config(kw):
    ns = argparse.Namespace
    return ns(model=ns(ch=kw["ch"], out_ch=kw["out_ch"], ch_mult=list(kw["ch_mult"]), num_res_blocks=kw["num_res_blocks"],
                       attn_resolutions=list(kw["attn_resolutions"]), dropout=kw["dropout"], in_channels=kw["in_channels"],
                       resamp_with_conv=kw["resamp_with_conv"], type="simple", cond_drop_prob=kw["cond_drop_prob"]),
              data=ns(image_size=kw["resolution"], n_classes=kw["n_classes"]),
              diffusion=ns(num_diffusion_timesteps=1000))


def gen_ddpm_model():
    """DDPM/models/diffusion.py Conditional_Model: forward (train with seeded RNG, test with cond_scale), backward,
    and 2 SFR-on iterations composed from reference functions in the order of DDPM/runners/diffusion.py:1075-1180."""
    from oracle import ddpm_ref
    ref_mod = _load("ref_ddpm_model", os.path.join(REF, "DDPM", "models", "diffusion.py"))
    losses = _load("ref_ddpm_losses2", os.path.join(REF, "DDPM", "functions", "losses.py"))
    ema_mod = _load("ref_ddpm_ema2", os.path.join(REF, "DDPM", "models", "ema.py"))
    torch.manual_seed(77)
    mine = ddpm_ref.ConditionalUNet(**DDPM_TINY)
    sd = mine.state_dict()
    model = ref_mod.Conditional_Model(ddpm_ref_config(DDPM_TINY))
    model.load_state_dict(sd, strict=True)                       # identical key set
    full = ref_mod.Conditional_Model(ddpm_ref_config(dict(DDPM_TINY, ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2,
                                                          attn_resolutions=(16,), resolution=32)))
    n_full = sum(p.numel() for p in full.parameters())
    g = torch.Generator().manual_seed(21)
    N = 4
    x = torch.rand(N, 3, 8, 8, generator=g) * 2 - 1
    t = torch.tensor([0, 999, 17, 500])
    c = torch.tensor([3, 3, 1, 9])
    model.train()
    torch.manual_seed(5)
    out_train = model(x, t.float(), c, mode="train", cond_drop_prob=0.5)
    model.eval()
    out_test = model(x, t.float(), c, mode="test", cond_scale=2.0)
    w = torch.randn(out_test.shape, generator=g)
    model.zero_grad()
    (model(x, t.float(), c, mode="train", cond_drop_prob=0.0) * w).sum().backward()
    pick = ["conv_in.weight", "down.0.block.0.temb_cemb_proj.bias", "down.1.attn.0.q.bias", "mid.attn_1.proj_out.bias",
            "up.0.block.1.nin_shortcut.bias", "classes_emb.weight", "norm_out.weight"]
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if n in pick}
    # ---- 2 SFR-on iterations, adaga, cosine alpha, clip in both stages, EMAHelper (runners/diffusion.py:1075-1180)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=1e-8)
    helper = ema_mod.EMAHelper(mu=1e-4)
    helper.register(model)
    b = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    gm = torch.Generator().manual_seed(8)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in model.named_parameters()}
    rec = {"forget": [], "remain": [], "alpha": []}
    inputs = {}
    n_iters, forget_alpha = 2, 10.0
    for step in range(n_iters):
        alpha = losses.cosine_lr_scheduler(forget_alpha, step, n_iters)
        fx = torch.rand(N, 3, 8, 8, generator=g) * 2 - 1
        fe = torch.randn(N, 3, 8, 8, generator=g)
        ft = torch.randint(0, 1000, (N // 2 + 1,), generator=g)
        ft = torch.cat([ft, 1000 - ft - 1], dim=0)[:N]
        fc = torch.zeros(N, dtype=torch.long)
        rx = torch.rand(N, 3, 8, 8, generator=g) * 2 - 1
        re_ = torch.randn(N, 3, 8, 8, generator=g)
        rt = torch.randint(0, 1000, (N // 2 + 1,), generator=g)
        rt = torch.cat([rt, 1000 - rt - 1], dim=0)[:N]
        rc = torch.randint(1, 10, (N,), generator=g)
        for k, v in dict(fx=fx, fe=fe, ft=ft, fc=fc, rx=rx, re=re_, rt=rt, rc=rc).items():
            inputs[f"s{step}_{k}"] = v.numpy()
        torch.manual_seed(100 + step)
        ori_forget = -losses.adaptive_loss(losses.loss_registry_conditional["simple"], model, fx, ft, fc, fe, b, lambd=0.5)
        opt.zero_grad()
        (alpha * ori_forget).backward()
        for n, p in model.named_parameters():
            if p.grad is not None:
                p.grad *= mask[n]
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        ori_remain = losses.loss_registry_conditional["simple"](model, rx, rt, rc, re_, b)
        opt.zero_grad()
        (1.0 * ori_remain).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        helper.update(model)
        rec["forget"].append(ori_forget.item()); rec["remain"].append(ori_remain.item()); rec["alpha"].append(alpha)
    np.savez_compressed(os.path.join(HERE, "ddpm_model.npz"),
                        x=x.numpy(), t=t.numpy(), c=c.numpy(), w=w.numpy(), n_params_full=np.array(n_full),
                        keys=np.array(list(sd.keys())), out_train_seed5=out_train.detach().numpy(),
                        out_test_scale2=out_test.detach().numpy(),
                        traj_forget=np.array(rec["forget"]), traj_remain=np.array(rec["remain"]), traj_alpha=np.array(rec["alpha"]),
                        final_conv_in=model.conv_in.weight.detach().numpy(),
                        final_shadow_conv_out=helper.shadow["conv_out.weight"].numpy(),
                        **{"grad::" + n: grads[n].numpy() for n in pick}, **inputs)


DDPM_GPU = dict(ch=128, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), dropout=0.0, in_channels=3,
                resolution=16, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1)


def gen_ddpm_gpu():
    """Conditional_Model at a size the HIP convolution tiles take (ch 128, 16 x 16 images, attention at 8 x 8 = 64 tokens, dropout 0
    so that the only random draw is the classifier-free keep mask, stored explicitly): test-mode forward, gradients of every tensor
    (norm + seeded projection; seven in full) and 2 SFR-on iterations in DDPM/runners/diffusion.py:1075-1180 order -- for
    tests/test_gpu_reference_fixtures.py (HIP vs these numbers, no oracle in between)."""
    from oracle import ddpm_ref
    ref_mod = _load("ref_ddpm_model_g", os.path.join(REF, "DDPM", "models", "diffusion.py"))
    losses = _load("ref_ddpm_losses_g", os.path.join(REF, "DDPM", "functions", "losses.py"))
    ema_mod = _load("ref_ddpm_ema_g", os.path.join(REF, "DDPM", "models", "ema.py"))
    torch.manual_seed(78)
    sd = ddpm_ref.ConditionalUNet(**DDPM_GPU).state_dict()
    model = ref_mod.Conditional_Model(ddpm_ref_config(DDPM_GPU))
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(22)
    N, S = 4, 16
    x = torch.rand(N, 3, S, S, generator=g) * 2 - 1
    t = torch.tensor([0, 999, 17, 500])
    c = torch.tensor([3, 3, 1, 9])
    w = torch.randn(N, 3, S, S, generator=g) * 0.1
    model.eval()
    out_test = model(x, t.float(), c, mode="test", cond_scale=2.0)
    model.train()
    model.zero_grad()
    out_train = model(x, t.float(), c, mode="train", cond_drop_prob=0.0)
    (out_train * w).sum().backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    grads = {n: dict(model.named_parameters())[n].grad.clone() for n in names}
    pick = ["conv_in.weight", "down.0.block.0.temb_cemb_proj.bias", "down.1.attn.0.q.bias", "mid.attn_1.proj_out.bias",
            "up.0.block.1.nin_shortcut.bias", "classes_emb.weight", "norm_out.weight"]
    out = dict(x=x.numpy(), t=t.numpy(), c=c.numpy(), w=w.numpy(), keys=np.array(list(sd.keys())),
               param_sums=np.array([float(v.double().sum()) for v in sd.values()]),
               out_test_scale2=out_test.detach().numpy(), out_train_nodrop=out_train.detach().numpy(),
               grad_names=np.array(names), grad_norms=np.array([float(grads[n].double().norm()) for n in names]),
               grad_proj=np.array([_proj(grads[n], n) for n in names]))
    out.update({"grad::" + n: grads[n].numpy() for n in pick})
    # ---- 2 SFR-on iterations: adaga (lambd 0.5), cosine alpha from 10, mask, clip 1.0 in both stages, Adam 1e-3, EMAHelper(1e-4)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=1e-8)
    helper = ema_mod.EMAHelper(mu=1e-4)
    helper.register(model)
    b = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    gm = torch.Generator().manual_seed(8)
    mask = {n: (torch.rand(p.shape, generator=gm) < 0.5) for n, p in model.named_parameters()}
    p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    rec = {"forget": [], "remain": [], "alpha": []}
    n_iters, forget_alpha = 2, 10.0
    for step in range(n_iters):
        alpha = losses.cosine_lr_scheduler(forget_alpha, step, n_iters)
        fx = torch.rand(N, 3, S, S, generator=g) * 2 - 1
        fe = torch.randn(N, 3, S, S, generator=g)
        ft = torch.randint(0, 1000, (N // 2 + 1,), generator=g)
        ft = torch.cat([ft, 1000 - ft - 1], dim=0)[:N]
        fc = torch.zeros(N, dtype=torch.long)
        rx = torch.rand(N, 3, S, S, generator=g) * 2 - 1
        re_ = torch.randn(N, 3, S, S, generator=g)
        rt = torch.randint(0, 1000, (N // 2 + 1,), generator=g)
        rt = torch.cat([rt, 1000 - rt - 1], dim=0)[:N]
        rc = torch.randint(1, 10, (N,), generator=g)
        # the keep masks the reference's prob_mask_like((b,), 0.9) will draw from these seeds (the only RNG use with dropout 0)
        torch.manual_seed(200 + step)
        fkeep = torch.zeros((N,)).float().uniform_(0, 1) < 0.9
        torch.manual_seed(300 + step)
        rkeep = torch.zeros((N,)).float().uniform_(0, 1) < 0.9
        for k, v in dict(fx=fx, fe=fe, ft=ft, fc=fc, rx=rx, re=re_, rt=rt, rc=rc, fkeep=fkeep, rkeep=rkeep).items():
            out[f"s{step}_{k}"] = v.numpy()
        torch.manual_seed(200 + step)
        ori_forget = -losses.adaptive_loss(losses.loss_registry_conditional["simple"], model, fx, ft, fc, fe, b, lambd=0.5)
        opt.zero_grad()
        (alpha * ori_forget).backward()
        for n, p in model.named_parameters():
            if p.grad is not None:
                p.grad *= mask[n]
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        torch.manual_seed(300 + step)
        ori_remain = losses.loss_registry_conditional["simple"](model, rx, rt, rc, re_, b)
        opt.zero_grad()
        (1.0 * ori_remain).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        helper.update(model)
        rec["forget"].append(ori_forget.item()); rec["remain"].append(ori_remain.item()); rec["alpha"].append(alpha)
    fin = dict(model.named_parameters())
    tn = list(fin)
    out.update(traj_forget=np.array(rec["forget"]), traj_remain=np.array(rec["remain"]), traj_alpha=np.array(rec["alpha"]),
               traj_names=np.array(tn), traj_update_norms=np.array([float((fin[n].detach() - p0[n]).double().norm()) for n in tn]),
               traj_update_proj=np.array([_proj(fin[n].detach() - p0[n], n) for n in tn]),
               traj_final_conv_out_bias=model.conv_out.bias.detach().numpy(),
               traj_final_shadow_norm_out=helper.shadow["norm_out.weight"].numpy())
    np.savez_compressed(os.path.join(HERE, "ddpm_gpu.npz"), **out)
    _save_updates("ddpm_gpu_updates.npz", tn, fin, p0, 1e-3, cap=32768)
    print("ddpm_gpu.npz:", len(out), "entries; traj", rec)


def gen_sampling(ref_models, ref_diffusion):
    """Sampling path: space_timesteps / respaced tables, p_sample_loop on a stub model (clip on and off), and
    the DiT's forward_with_cfg driven through p_sample_loop as DiT/forget.py:114-145 does (clip_denoised=False, cfg 4.0)."""
    from diffusion.respace import space_timesteps
    out = {}
    for name, (T, spec) in {"s250": (1000, "250"), "sddim25": (1000, "ddim25"), "s10": (1000, "10"), "ssec": (300, "10,15,20")}.items():
        out["steps_" + name] = np.array(sorted(space_timesteps(T, spec)))
    d = ref_diffusion.create_diffusion(timestep_respacing="10")
    out["map10"] = np.array(d.timestep_map)
    for k in ["betas", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod"]:
        out["tab10_" + k] = getattr(d, k)
    g = torch.Generator().manual_seed(31)
    N, C, H = 4, 4, 8
    z = torch.randn(N, C, H, H, generator=g)
    A = torch.randn(2 * C, C, generator=g) * 0.4

    def stub(x, ts, **kw):                       # sees ORIGINAL timesteps through the wrapped model
        return torch.einsum("oc,nchw->nohw", A, x) * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05
    out.update(z=z.numpy(), A=A.numpy())
    for clip in (True, False):
        torch.manual_seed(77)
        out[f"stub_clip{int(clip)}"] = d.p_sample_loop(stub, z.shape, z, clip_denoised=clip, model_kwargs={}, device="cpu").numpy()
    # a CONTRACTIVE stub for the un-clamped loop (the linear stub above grows to ~1e36 without the clamp -- NaN == NaN pins nothing):
    # eps-hat = 0.9 x / sqrt(1 - abar_t) + a small channel mix, i.e. close to the exact eps of data concentrated near 0, so that
    # pred_xstart = sqrt(1/abar) x - sqrt(1/abar - 1) eps-hat stays O(|x|) at every step; the variance half as before
    abar1000 = np.cumprod(1.0 - np.linspace(1e-4, 2e-2, 1000, dtype=np.float64))
    s1m = torch.tensor(np.sqrt(1.0 - abar1000), dtype=torch.float32)
    out["abar1000"] = abar1000

    def stub2(x, ts, **kw):
        lin = torch.einsum("oc,nchw->nohw", A, x)
        eps = 0.9 * x / s1m[ts].view(-1, 1, 1, 1) + 0.02 * lin[:, :C]
        return torch.cat([eps, lin[:, C:] * torch.cos(ts.float() / 300.0).view(-1, 1, 1, 1) + 0.05], dim=1)
    torch.manual_seed(77)
    out["stub2_clip0"] = d.p_sample_loop(stub2, z.shape, z, clip_denoised=False, model_kwargs={}, device="cpu").numpy()
    assert np.isfinite(out["stub2_clip0"]).all() and np.abs(out["stub2_clip0"]).max() < 1e3, np.abs(out["stub2_clip0"]).max()
    one = d.p_sample(stub, z, torch.tensor([0, 9, 3, 0]), clip_denoised=False, model_kwargs={})
    torch.manual_seed(5)
    one = d.p_sample(stub, z, torch.tensor([0, 9, 3, 0]), clip_denoised=False, model_kwargs={})
    out["one_sample_seed5"] = one["sample"].numpy(); out["one_pred_xstart"] = one["pred_xstart"].numpy()
    # the real model + classifier-free guidance, 5 steps
    m = ref_models.DiT(**TINY)
    m.load_state_dict(tiny_weights())
    m.eval()
    d5 = ref_diffusion.create_diffusion(timestep_respacing="5")
    n = 3
    zz = torch.randn(n, 4, 8, 8, generator=g)
    y = torch.tensor([1, 9, 4])
    zc, yc = torch.cat([zz, zz], 0), torch.cat([y, torch.tensor([10] * n)], 0)
    noises = torch.randn(5, 2 * n, 4, 8, 8, generator=g)
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it)          # explicit per-step noise (same tensors are stored in the fixture)
    try:
        with torch.no_grad():
            smp = d5.p_sample_loop(m.forward_with_cfg, zc.shape, zc, clip_denoised=False, model_kwargs=dict(y=yc, cfg_scale=4.0), device="cpu")
            cfg_out = m.forward_with_cfg(zc, torch.tensor([999, 0, 500, 999, 0, 500]), yc, 4.0)
    finally:
        torch.randn_like = orig
    out.update(cfg_z=zz.numpy(), cfg_y=y.numpy(), cfg_step_noise=noises.numpy(), cfg_samples=smp.numpy(), cfg_forward=cfg_out.numpy(),
               map5=np.array(d5.timestep_map))
    np.savez_compressed(os.path.join(HERE, "dit_sampling.npz"), **out)


def gen_ddpm_sampler():
    """DDPM/functions/denoising.py generalized_steps_conditional on a stub model (eta 0 and 0.5, 10 of 1000 steps)."""
    den = _load("ref_ddpm_denoising", os.path.join(REF, "DDPM", "functions", "denoising.py"))
    b = torch.from_numpy(np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)).float()
    g = torch.Generator().manual_seed(41)
    x = torch.randn(3, 3, 8, 8, generator=g)
    c = torch.tensor([1, 5, 9])
    A = torch.randn(3, 3, generator=g) * 0.4
    model = lambda xt, t, cc, cond_scale=3.0, mode="test": torch.einsum("oc,nchw->nohw", A, xt) * torch.cos(t / 300.0).view(-1, 1, 1, 1) + 0.01 * cc.view(-1, 1, 1, 1) * cond_scale
    seq = list(range(0, 1000, 100))
    out = dict(x=x.numpy(), c=c.numpy(), A=A.numpy(), seq=np.array(seq))
    for eta in (0.0, 0.5):
        torch.manual_seed(13)
        xs, x0s = den.generalized_steps_conditional(x, c, seq, model, b, cond_scale=2.0, eta=eta)
        out[f"last_eta{eta}"] = xs[-1].numpy(); out[f"x0_first_eta{eta}"] = x0s[0].numpy(); out[f"x_mid_eta{eta}"] = xs[5].numpy()
    out["alpha_t"] = den.compute_alpha(b, torch.tensor([0, 999, 500])).flatten().numpy()
    out["alpha_m1"] = den.compute_alpha(b, torch.tensor([-1])).flatten().numpy()
    np.savez_compressed(os.path.join(HERE, "ddpm_sampler.npz"), **out)


# ---------------------------------------------------------------------------------------------------------------- SD / LDM
SD_TINY = dict(image_size=8, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[2, 1], num_res_blocks=1,
               channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=24,
               use_checkpoint=False, legacy=False)
SD_V1 = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
             channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
             use_checkpoint=True, legacy=False)


def import_ref_sd():
    """SD/ldm/modules import with a harness stand-in for the NAME omegaconf.listconfig.ListConfig (a type check at
    openaimodel.py:497-500; omegaconf itself is absent here).  The stand-in is harness code, not reference code."""
    oc, lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass
    lc.ListConfig = ListConfig
    oc.listconfig = lc
    sys.modules.update({"omegaconf": oc, "omegaconf.listconfig": lc})
    sys.path.insert(0, os.path.join(REF, "SD"))
    om = importlib.import_module("ldm.modules.diffusionmodules.openaimodel")
    ut = importlib.import_module("ldm.modules.diffusionmodules.util")
    return om, ut


def sd_tiny_weights(seed=4321):
    from oracle import sd_ref
    torch.manual_seed(seed)
    m = sd_ref.UNetModel(**{k: v for k, v in SD_TINY.items() if k not in ("image_size", "use_spatial_transformer", "use_checkpoint", "legacy")})
    sd_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m.state_dict()


def gen_sd():
    from oracle import sd_ref
    om, ut = import_ref_sd()
    out = {}
    # (1) key / shape list of the v1-inference.yaml UNet, on the meta device (859,520,964 parameters, 686 tensors)
    with torch.device("meta"):
        full = om.UNetModel(**SD_V1)
    spec = "\n".join(f"{n} {tuple(p.shape)}" for n, p in full.named_parameters())
    out["v1_param_spec"] = np.frombuffer(spec.encode(), dtype=np.uint8)
    out["v1_param_count"] = np.array(sum(p.numel() for p in full.parameters()), dtype=np.int64)
    # (2) tiny config: forward + backward of the reference class on weights generated through the oracle class (same keys)
    ref = om.UNetModel(**SD_TINY)
    sd = sd_tiny_weights()
    assert list(ref.state_dict().keys()) == list(sd.keys())
    ref.load_state_dict(sd)
    ref.train()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(3, 4, 8, 8, generator=g)
    t = torch.tensor([0, 500, 999])
    ctx = torch.randn(3, 5, 24, generator=g)
    w = torch.randn(3, 4, 8, 8, generator=g) * 0.1
    o = ref(x, timesteps=t, context=ctx)
    (o * w).sum().backward()
    out.update(x=x.numpy(), t=t.numpy(), ctx=ctx.numpy(), w=w.numpy(), out=o.detach().numpy())
    names = [n for n, _ in ref.named_parameters()]
    out["grad_norms"] = np.array([p.grad.norm().item() for _, p in ref.named_parameters()], dtype=np.float64)
    for n in ("time_embed.0.weight", "input_blocks.0.0.weight", "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight",
              "input_blocks.2.0.op.weight", "middle_block.1.transformer_blocks.0.ff.net.0.proj.weight", "output_blocks.1.2.conv.weight",
              "output_blocks.3.0.skip_connection.weight", "out.2.weight", "out.0.bias"):
        assert n in names, n
        out["grad/" + n] = dict(ref.named_parameters())[n].grad.numpy().copy()
    out["temb"] = ut.timestep_embedding(t, 32).numpy()
    # (3) schedule: make_beta_schedule("linear", 1000, 0.00085, 0.012) and the fp32 tables register_schedule derives from it
    betas = ut.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    out.update(betas=betas, sqrt_ac=torch.tensor(np.sqrt(ac), dtype=torch.float32).numpy(),
               sqrt_1m_ac=torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32).numpy())
    xs = torch.randn(3, 4, 8, 8, generator=g)
    nz = torch.randn(3, 4, 8, 8, generator=g)
    sa, sb = torch.tensor(np.sqrt(ac), dtype=torch.float32), torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
    out.update(q_x0=xs.numpy(), q_noise=nz.numpy(),
               q_xt=(ut.extract_into_tensor(sa, t, xs.shape) * xs + ut.extract_into_tensor(sb, t, xs.shape) * nz).numpy())
    # (4) two iterations of the loop body of train-scripts/nsfw_removal.py:108-173 composed from the reference UNet and the
    #     schedule above in that order (the script itself needs pytorch_lightning / diffusers): train_method "xattn", Adam lr 1e-3
    ref2 = om.UNetModel(**SD_TINY)
    ref2.load_state_dict(sd)
    ref2.train()
    params = [p for n, p in ref2.named_parameters() if "attn2" in n]
    opt = torch.optim.Adam(params, lr=1e-3)
    g2 = torch.Generator().manual_seed(23)
    c_f, c_p = torch.randn(1, 5, 24, generator=g2).expand(2, -1, -1), torch.randn(1, 5, 24, generator=g2).expand(2, -1, -1)
    traj = []
    batches = []
    for it in range(2):
        xf = torch.randn(2, 4, 8, 8, generator=g2); xr = torch.randn(2, 4, 8, 8, generator=g2)
        tt = torch.randint(0, 1000, (2,), generator=g2); nf = torch.randn(2, 4, 8, 8, generator=g2)
        tr = torch.randint(0, 1000, (2,), generator=g2); nr = torch.randn(2, 4, 8, 8, generator=g2)
        batches.append((xf, xr, tt, nf, tr, nr))
        q = lambda x0, t_, n_: ut.extract_into_tensor(sa, t_, x0.shape) * x0 + ut.extract_into_tensor(sb, t_, x0.shape) * n_
        opt.zero_grad()
        f_out = ref2(q(xf, tt, nf), timesteps=tt, context=c_f)
        p_out = ref2(q(xf, tt, nf), timesteps=tt, context=c_p).detach()        # pseudo branch: same images, other prompt
        lf = torch.nn.MSELoss()(f_out, p_out)
        (1.0 * lf).backward()
        opt.step()
        opt.zero_grad()
        r_out = ref2(q(xr, tr, nr), timesteps=tr, context=c_p)
        lr_ = torch.nn.functional.mse_loss(nr, r_out, reduction="none").mean([1, 2, 3]).mean()
        (1.0 * lr_).backward()
        opt.step()
        traj.append((lf.item(), lr_.item()))
    out["traj_losses"] = np.array(traj, dtype=np.float64)
    out["traj_c_f"], out["traj_c_p"] = c_f[:1].numpy().copy(), c_p[:1].numpy().copy()
    for k, vals in zip(("xf", "xr", "t_f", "noise_f", "t_r", "noise_r"), zip(*batches)):
        out["traj_" + k] = np.stack([v.numpy() for v in vals])
    k = "input_blocks.1.1.transformer_blocks.0.attn2.to_v.weight"
    out["traj_final/" + k] = dict(ref2.named_parameters())[k].detach().numpy().copy()
    out["traj_untouched/time_embed.0.weight"] = dict(ref2.named_parameters())["time_embed.0.weight"].detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "sd_unet.npz"), **out)
    print("sd_unet.npz:", {k: getattr(v, "shape", None) for k, v in list(out.items())[:6]}, "v1 params", int(out["v1_param_count"]))



def gen_sd_updates():
    """sd_unet_updates.npz: every trainable tensor's UPDATE over the 2-iteration nsfw_removal.py:108-173 trajectory of the reference
    UNetModel -- (a) train_method "xattn" (the attn2 tensors): the SAME trajectory sd_unet.npz holds (same seeds, recomputed here and
    checked against that file's losses; sd_unet.npz itself is not rewritten), (b) train_method "full" (every parameter,
    nsfw_removal.py:67-77) on the same batches, with its losses.  Separate file: the older fixture keeps its bits."""
    om, ut = import_ref_sd()
    sd = sd_tiny_weights()
    old = np.load(os.path.join(HERE, "sd_unet.npz"))
    betas = ut.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    sa, sb = torch.tensor(np.sqrt(ac), dtype=torch.float32), torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
    c_f, c_p = torch.from_numpy(old["traj_c_f"]).expand(2, -1, -1), torch.from_numpy(old["traj_c_p"]).expand(2, -1, -1)
    q = lambda x0, t_, n_: ut.extract_into_tensor(sa, t_, x0.shape) * x0 + ut.extract_into_tensor(sb, t_, x0.shape) * n_
    out = {}
    for method in ("xattn", "full"):
        ref = om.UNetModel(**SD_TINY)
        ref.load_state_dict(sd)
        ref.train()
        named = [(n, p) for n, p in ref.named_parameters() if method == "full" or "attn2" in n]      # nsfw_removal.py:67-77
        p0 = {n: p.detach().clone() for n, p in named}
        opt = torch.optim.Adam([p for _, p in named], lr=1e-3)
        traj = []
        gn = {}
        for it in range(2):
            xf, xr = torch.from_numpy(old["traj_xf"][it]), torch.from_numpy(old["traj_xr"][it])
            tt, nf = torch.from_numpy(old["traj_t_f"][it]), torch.from_numpy(old["traj_noise_f"][it])
            tr, nr = torch.from_numpy(old["traj_t_r"][it]), torch.from_numpy(old["traj_noise_r"][it])
            opt.zero_grad()
            f_out = ref(q(xf, tt, nf), timesteps=tt, context=c_f)
            p_out = ref(q(xf, tt, nf), timesteps=tt, context=c_p).detach()
            lf = torch.nn.MSELoss()(f_out, p_out)
            (1.0 * lf).backward()
            if it == 0:
                gn["forget"] = np.array([p.grad.double().norm().item() for _, p in named])
            opt.step()
            opt.zero_grad()
            r_out = ref(q(xr, tr, nr), timesteps=tr, context=c_p)
            lr_ = torch.nn.functional.mse_loss(nr, r_out, reduction="none").mean([1, 2, 3]).mean()
            (1.0 * lr_).backward()
            if it == 0:
                gn["remain"] = np.array([p.grad.double().norm().item() for _, p in named])
            opt.step()
            traj.append((lf.item(), lr_.item()))
        traj = np.array(traj, dtype=np.float64)
        if method == "xattn":
            assert np.allclose(traj, old["traj_losses"], rtol=1e-5, atol=0), (traj, old["traj_losses"])
            k = "input_blocks.1.1.transformer_blocks.0.attn2.to_v.weight"
            assert np.allclose(dict(named)[k].detach().numpy(), old["traj_final/" + k], rtol=0, atol=2e-6)
        out[method] = (traj, [n for n, _ in named], {n: p for n, p in named}, p0, gn)
    blob = dict(lr=np.array(1e-3), scale=np.array(20.0), cap=np.array(32768))
    for method, (traj, names, fin, p0, gn) in out.items():
        blob[method + "::losses"] = traj
        blob[method + "::names"] = np.array(names)
        # the reference's own gradient norm per tensor in the first forget / remain stage: a tensor whose gradient is EXACTLY zero
        # analytically (a per-channel constant in front of a GroupNorm with one channel per group: SD_TINY's 32-channel levels) holds
        # fp32 cancellation noise there, and its "update" is Adam's normalisation of that noise -- the GPU test compares those by size
        blob[method + "::gnorm_forget"], blob[method + "::gnorm_remain"] = gn["forget"], gn["remain"]
        for n in names:
            u = ((fin[n].detach() - p0[n]).double() / 1e-3).numpy().ravel()
            blob[f"{method}::norm::{n}"] = np.array(float(np.linalg.norm(u)))
            stride = -(-u.size // 32768)
            blob[f"{method}::upd::{n}"] = np.clip(np.rint(u[::stride] * 20.0), -127, 127).astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "sd_unet_updates.npz"), **blob)
    print("sd_unet_updates.npz written:", {m: (len(v[1]), v[0].tolist()) for m, v in out.items()})
    g = out["full"][4]
    med = np.median(g["remain"])
    print("full: tensors with a remain-stage gradient norm below 1e-4 x the median:",
          [(n, float(a), float(b)) for n, a, b in zip(out["full"][1], g["forget"], g["remain"]) if b < 1e-4 * med][:40])


def import_ref_convert():
    """SD/train-scripts/convertModels.py needs diffusers / transformers / omegaconf only for NAMES imported at module top (the
    UNet key conversion itself is plain dict manipulation): the harness supplies empty stand-in classes."""
    def stub(name, names):
        m = types.ModuleType(name)
        for n in names:
            setattr(m, n, type(n, (), {}))
        sys.modules[name] = m
    stub("omegaconf", ["OmegaConf"])
    stub("diffusers", ["AutoencoderKL", "DDIMScheduler", "DPMSolverMultistepScheduler", "EulerAncestralDiscreteScheduler", "EulerDiscreteScheduler",
                       "HeunDiscreteScheduler", "LDMTextToImagePipeline", "LMSDiscreteScheduler", "PNDMScheduler", "StableDiffusionPipeline",
                       "UNet2DConditionModel"])
    for n in ("diffusers.pipelines", "diffusers.pipelines.latent_diffusion", "diffusers.pipelines.paint_by_example", "diffusers.pipelines.stable_diffusion"):
        stub(n, ["StableDiffusionSafetyChecker", "PaintByExampleImageEncoder", "PaintByExamplePipeline"])
    stub("diffusers.pipelines.latent_diffusion.pipeline_latent_diffusion", ["LDMBertConfig", "LDMBertModel"])
    stub("transformers", ["AutoFeatureExtractor", "BertTokenizerFast", "CLIPImageProcessor", "CLIPTextModel", "CLIPTextModelWithProjection",
                          "CLIPTokenizer", "CLIPVisionConfig"])
    return _load("ref_convert_models", os.path.join(REF, "SD", "train-scripts", "convertModels.py"))


class _Cfg(dict):
    """attribute + item + `in` access, as the OmegaConf nodes the reference reads"""
    __getattr__ = dict.__getitem__


def gen_compvis_export():
    from oracle import sd_ref
    conv = import_ref_convert()
    out = {}
    for tag, kw in (("v1", dict()), ("small", dict(model_channels=32, channel_mult=(1, 2, 4), attention_resolutions=(2, 1), num_res_blocks=1,
                                                   num_heads=2, context_dim=24))):
        with torch.device("meta"):
            m = sd_ref.UNetModel(**kw)
        full = dict(model_channels=320, channel_mult=(1, 2, 4, 4), attention_resolutions=(4, 2, 1), num_res_blocks=2, in_channels=4,
                    out_channels=4, context_dim=768, num_heads=8)
        full.update(kw)
        cfg = _Cfg(model=_Cfg(params=_Cfg(unet_config=_Cfg(params=_Cfg(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in full.items()})),
                                          first_stage_config=_Cfg(params=_Cfg(ddconfig=_Cfg(ch_mult=[1, 2, 4, 4]))))))
        ucfg = conv.create_unet_diffusers_config(cfg, image_size=512)
        ck = {"model.diffusion_model." + n: p for n, p in m.named_parameters()}
        ck["first_stage_model.encoder.conv_in.weight"] = torch.empty(1, device="meta")      # non-UNet entries are ignored
        ids = {id(v): k for k, v in ck.items()}
        new = conv.convert_ldm_unet_checkpoint(dict(ck), ucfg)
        lines = [f"{nk} {ids[id(v)]}" for nk, v in new.items()]
        out[tag + "_map"] = np.frombuffer("\n".join(lines).encode(), dtype=np.uint8)
        out[tag + "_config"] = np.frombuffer(repr(sorted(ucfg.items())).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "compvis_export.npz"), **out)
    print("compvis_export.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None, help="regenerate just these fixtures (e.g. dit_gpu sampling); default: all")
    only = ap.parse_args().only
    want = lambda k: only is None or k in only
    torch.set_num_threads(4)
    ref_diffusion = import_ref_dit_diffusion()
    ref_models = import_ref_dit_models()
    if want("diffusion"):
        gen_diffusion(ref_diffusion)
    if want("model"):
        gen_model(ref_models)
    if want("traj"):
        gen_traj(ref_models, ref_diffusion)
    if want("dit_gpu"):
        gen_dit_gpu(ref_models, ref_diffusion)
    if want("ddpm"):
        gen_ddpm()
    if want("mask"):
        gen_mask()
    if want("ddpm_model"):
        gen_ddpm_model()
    if want("ddpm_gpu"):
        gen_ddpm_gpu()
    if want("sampling"):
        gen_sampling(ref_models, ref_diffusion)
    if want("ddpm_sampler"):
        gen_ddpm_sampler()
    if want("sd"):
        gen_sd()
    if want("sd_updates"):
        gen_sd_updates()
    if want("compvis_export"):
        gen_compvis_export()
    print("golden vectors written to", HERE)
