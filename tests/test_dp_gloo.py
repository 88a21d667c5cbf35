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
