"""CPU, world_size 2 over gloo: the data-parallel rule of the SFR-on step.

(1) synthetic_batch shards tile the global batch for any world size;
(2) per-rank gradients of grad_scale = alpha / GLOBAL_batch, SUM-all-reduced in buckets (sfron.dp), equal the
    single-process gradient of the global batch (oracle model, fp32) -- the rule step.DiTSFRon applies on GPUs;
(3) the whole oracle SFR-on iteration run on 2 ranks with all-reduced grads matches the 1-rank iteration."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TINY = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=32, depth=2, num_heads=2, num_classes=10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(seed=0):
    from oracle import dit_ref
    torch.manual_seed(seed)
    m = dit_ref.DiT(**TINY)
    dit_ref.randomize_zero_init(m, std=0.05, seed=seed + 1)
    return m.train()


def _grads_global(alpha, gb):
    from oracle import diffusion_ref as dref
    from sfron import data
    m = _model()
    b = data.synthetic_batch(3, 0, "remain", gb, input_size=8, num_classes=10, forget_class=3)
    terms = dref.training_losses(dref.DiffusionTables(1000), lambda x, t, y: m(x, t, y, force_drop_ids=b["drop"].long()),
                                 b["x0"], b["t"], dict(y=b["y"]), b["noise"])
    (alpha * terms["loss"].mean()).backward()
    return torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])


def _worker(rank, world, port, gb, alpha, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import diffusion_ref as dref
        from sfron import data, dp
        m = _model()
        b = data.synthetic_batch(3, 0, "remain", gb, rank, world, input_size=8, num_classes=10, forget_class=3)
        terms = dref.training_losses(dref.DiffusionTables(1000), lambda x, t, y: m(x, t, y, force_drop_ids=b["drop"].long()),
                                     b["x0"], b["t"], dict(y=b["y"]), b["noise"])
        # loss.hip semantics: d(grad_scale * sum_i loss_i), grad_scale = alpha / GLOBAL batch
        (alpha / gb * terms["loss"].sum()).backward()
        flat = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        dp.allreduce_flat_(flat, bucket_elems=1000)          # several buckets + a ragged tail
        if rank == 0:
            out_q.put(flat)
    finally:
        dist.destroy_process_group()


def test_shards_tile_the_global_batch():
    from sfron import data
    g = data.synthetic_batch(9, 4, "forget", 8, input_size=8)
    for world in (2, 4, 8):
        parts = [data.synthetic_batch(9, 4, "forget", 8, r, world, input_size=8) for r in range(world)]
        for k in g:
            assert torch.equal(torch.cat([p[k] for p in parts]), g[k]), k
    r = data.synthetic_batch(9, 4, "remain", 64, num_classes=10, forget_class=3, input_size=8)
    assert (r["y"] != 3).all() and r["y"].min() >= 0 and r["y"].max() <= 9
    assert (data.synthetic_batch(9, 4, "forget", 8, forget_class=3, input_size=8)["y"] == 3).all()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_global_batch_gradient():
    gb, alpha = 4, 0.7
    want = _grads_global(alpha, gb)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gb, alpha, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-6), (got - want).abs().max()


# ------------------------------------------------------------------------------------------------ more of the N > 1 rule (round 3)
def _spawn(target, args, n_out=1, world=2, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got = [_un(q.get(timeout=timeout)) for _ in range(n_out)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _np(x):
    """tensors cross the result queue BY VALUE (numpy): a torch tensor travels as a shared-memory handle that dies with the worker"""
    if torch.is_tensor(x):
        return ("__t__", x.detach().numpy().copy())
    if isinstance(x, (tuple, list)):
        return type(x)(_np(v) for v in x)
    return x


def _un(x):
    if isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], str) and x[0] == "__t__":
        return torch.from_numpy(x[1])
    if isinstance(x, (tuple, list)):
        return type(x)(_un(v) for v in x)
    return x


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _bf16_worker(rank, world, port, out_q):
    _init(rank, world, port)
    try:
        from sfron import dp
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(10_007, generator=g) * torch.logspace(-6, 0, 10_007)       # six decades of magnitudes
        exact = flat.clone()
        dp.allreduce_flat_(exact, bucket_elems=4096)
        low = flat.clone()
        scratch = torch.empty(4096, dtype=torch.bfloat16)
        dp.allreduce_flat_(low, bucket_elems=4096, transport="bf16", scratch=scratch)
        one = flat.clone()
        dp.allreduce_(one, transport="bf16")                                            # single range, scratch allocated inside
        if rank == 0:
            out_q.put(_np((flat, exact, low, one)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bf16_gradient_transport_is_bounded_against_the_fp32_exchange():
    """grad_transport="bf16" (step.DiTSFRon): each rank's addend is rounded to bf16 (2^-9 relative), the sum is formed by the backend in
    bf16 and widened back: element-wise within 2^-8 * (|a| + |b|) + one rounding of the sum, rel-L2 of the whole arena < 6e-3."""
    (mine, exact, low, one), = _spawn(_bf16_worker, ())
    other = exact - mine
    bound = 2.0 ** -8 * (mine.abs() + other.abs()) + 2.0 ** -8 * exact.abs() + 1e-30
    assert ((low - exact).abs() <= bound).all()
    assert ((low - exact).norm() / exact.norm()).item() < 6e-3
    assert torch.equal(low, one)                     # bucket boundaries do not change an element-wise rounding
    assert low.dtype == torch.float32


def _ranges_worker(rank, world, port, out_q):
    _init(rank, world, port)
    try:
        from sfron import dp
        flat = torch.arange(64, dtype=torch.float32) * (rank + 1)
        dp.allreduce_ranges_(flat, [(8, 16), (40, 56)], bucket_elems=5)
        if rank == 1:
            out_q.put(_np(flat))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_allreduce_ranges_touches_only_the_trainable_ranges():
    """SD train_method "xattn" (sd.SDSFRon): only the arena ranges the optimizer owns are exchanged; everything else keeps this
    rank's value (and never grows by a factor of world per exchange)."""
    (got,) = _spawn(_ranges_worker, ())
    base = torch.arange(64, dtype=torch.float32)
    want = base * 2
    for lo, hi in ((8, 16), (40, 56)):
        want[lo:hi] = base[lo:hi] * 3
    assert torch.equal(got, want)


DDPM_TINY = dict(ch=128, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(4,), dropout=0.0, in_channels=3,
                 resolution=8, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1)


def _ddpm_batch(gb, seed=7):
    g = torch.Generator().manual_seed(seed)
    return dict(x0=torch.rand(gb, 3, 8, 8, generator=g) * 2 - 1, e=torch.randn(gb, 3, 8, 8, generator=g),
                t=torch.randint(0, 1000, (gb,), generator=g), c=torch.zeros(gb, dtype=torch.int64),
                keep=torch.rand(gb, generator=g) < 0.9)


def _ddpm_model():
    from oracle import ddpm_ref
    torch.manual_seed(11)
    return ddpm_ref.ConditionalUNet(**DDPM_TINY).train()


def _ddpm_adaga_worker(rank, world, port, gb, alpha, lambd, out_q):
    _init(rank, world, port)
    try:
        from oracle import sfron_ref
        from sfron import dp
        m = _ddpm_model()
        b = dp.shard(_ddpm_batch(gb), rank, world)
        betas = sfron_ref.ddpm_get_betas()
        per = sfron_ref.ddpm_loss_per_sample(lambda x, tf: m(x, tf, b["c"], keep_mask=b["keep"]), b["x0"], b["t"], b["e"], betas)
        # the rule of sfron.ddpm._Reduce on N ranks (csrc/loss.hip k_ddpm_loss_coef): w_i = 1 / (l_i^lambd + 1e-8) detached, ONE scalar
        # all-reduce of sum_i w_i, this rank's loss share = sum_local(w_i l_i) / W_global -- losses.py:49-69 normalises over the GLOBAL batch
        w = 1.0 / (per.detach().pow(lambd) + 1e-8)
        wsum = w.sum()
        dist.all_reduce(wsum)
        (alpha * -((w * per).sum() / wsum)).backward()
        flat = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        dp.allreduce_flat_(flat, bucket_elems=1 << 16)
        if rank == 0:
            out_q.put(_np(flat))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddpm_adaga_two_ranks_equal_the_global_batch():
    """DDPM/functions/losses.py:49-69 normalises the adaptive weights over the whole batch: on N ranks that is one scalar all-reduce
    of sum_i w_i before the backward pass (sfron.ddpm._Reduce); the SUM-all-reduced gradient then equals the 1-rank gradient."""
    from oracle import sfron_ref
    gb, alpha, lambd = 4, 3.0, 0.5
    m = _ddpm_model()
    b = _ddpm_batch(gb)
    per = sfron_ref.ddpm_loss_per_sample(lambda x, tf: m(x, tf, b["c"], keep_mask=b["keep"]), b["x0"], b["t"], b["e"],
                                         sfron_ref.ddpm_get_betas())
    (alpha * -sfron_ref.ddpm_adaptive_loss(per, lambd)).backward()
    want = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    (got,) = _spawn(_ddpm_adaga_worker, (gb, alpha, lambd), timeout=500)
    assert torch.allclose(got, want, rtol=2e-4, atol=1e-6 * want.abs().max().item()), ((got - want).norm() / want.norm()).item()


SD_TINY = dict(in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2],
               num_heads=2, transformer_depth=1, context_dim=24)


def _sd_model():
    from oracle import sd_ref
    torch.manual_seed(21)
    m = sd_ref.UNetModel(**SD_TINY)
    sd_ref.randomize_zero_init(m, std=0.05, seed=3)
    return m.train()


def _sd_batch(gb, seed=5):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(x_f=r(gb, 4, 8, 8), x_p=r(gb, 4, 8, 8), c_f=r(gb, 5, 24), c_p=r(gb, 5, 24), t=torch.randint(0, 1000, (gb,), generator=g),
                noise=r(gb, 4, 8, 8))


def _sd_grads(m, s, b, fa, world):
    """forget stage of nsfw_removal.py:134-156 on one shard: d(fa * mean over the GLOBAL batch of (f_out - sg p_out)^2), i.e. this
    rank's mean divided by world -- the scaling of sfron.sd.SDSFRon._d_loss."""
    from torch.nn import functional as F
    f_out = m(s.q_sample(b["x_f"], b["t"], b["noise"]), b["t"], context=b["c_f"])
    p_out = m(s.q_sample(b["x_p"], b["t"], b["noise"]), b["t"], context=b["c_p"]).detach()
    (fa * F.mse_loss(f_out, p_out) / world).backward()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


def _sd_worker(rank, world, port, gb, fa, out_q):
    _init(rank, world, port)
    try:
        from oracle import sd_ref
        from sfron import dp
        m = _sd_model()
        g = _sd_grads(m, sd_ref.LDMSchedule(), dp.shard(_sd_batch(gb), rank, world), fa, world)
        names = sorted(g)
        flat = torch.cat([g[n].flatten() for n in names])
        # train_method "xattn": only the cross-attention ranges travel
        off, ranges = 0, []
        for n in names:
            k = g[n].numel()
            if "attn2" in n:
                ranges.append((off, off + k))
            off += k
        local = flat.clone()
        dp.allreduce_ranges_(flat, ranges, bucket_elems=1 << 12)
        if rank == 0:
            out_q.put(_np((names, [tuple(g[n].shape) for n in names], flat, local, ranges)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sd_forget_stage_two_ranks_xattn_ranges_equal_the_global_batch():
    from oracle import sd_ref
    gb, fa = 4, 2.0
    want = _sd_grads(_sd_model(), sd_ref.LDMSchedule(), _sd_batch(gb), fa, 1)
    (names, shapes, flat, local, ranges), = _spawn(_sd_worker, (gb, fa), timeout=500)
    off = 0
    seen = 0
    for n, shp in zip(names, shapes):
        k = 1
        for d in shp:
            k *= d
        got = flat[off:off + k].view(shp)
        if "attn2" in n:
            seen += 1
            assert torch.allclose(got, want[n], rtol=2e-4, atol=1e-6 + 1e-5 * want[n].abs().max().item()), n
        else:
            assert torch.equal(got, local[off:off + k].view(shp)), n          # frozen under xattn: not exchanged
        off += k
    assert seen >= 8 and len(ranges) == seen
