"""GPU: the overlapped gradient exchange of the data-parallel path (per-block events on the weight-gradient stream, late-bias
staging + scatter, communication stream) at world size 1 over RCCL: it must reproduce the plain path bit for bit.  (More ranks
cannot be started on the one-GPU test box; the collective sequence is rank-independent by construction, and the CPU gloo tests
cover the sharding / reduction arithmetic.)"""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_overlapped_allreduce_path_matches_plain_path():
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    cfg = CASES["hd72"]
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(5, it, "forget", **kw), data.synthetic_batch(5, it, "remain", **kw))
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)

    def run(overlap, single_process_shortcuts=False):
        _, model = build_pair(cfg, B, seed=21)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), overlap_allreduce=overlap, **hp)
        assert runner._overlap_enabled() == overlap
        if not single_process_shortcuts:
            # the data-parallel machinery is compared with the plain GEMM + flat-sweep pass: the single-process shortcuts (adaLN
            # gradient as a rank-(batch) sweep -- another summation order --, block sweeps beside the forward pass) are checked below
            runner.factored_ada = runner.sweep_beside_forward = False
        for it in range(2):
            out = runner.step(*bat(it))
        torch.cuda.synchronize()
        return model.engine.params.clone(), model.engine.grads.clone(), out["stats"].clone()

    p0, g0, s0 = run(False)
    p2, _, s2 = run(False, single_process_shortcuts=True)
    # the shortcuts change summation orders only: the same parameters to fp32 rounding after two iterations (4 optimizer steps)
    assert torch.allclose(s2, s0, rtol=1e-5) and (p2 - p0).abs().max().item() <= 2e-4 * 4 * 1e-3 + 1e-7, (p2 - p0).abs().max().item()
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        p1, g1, s1 = run(True)
        # the run-time check bench.py uses before it turns the overlap on: one pass each way from the same state
        _, model = build_pair(cfg, B, seed=21)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
        before = model.engine.params.clone()
        assert runner.verify_overlap(bat(0)[0]) is True
        assert torch.equal(before, model.engine.params) and runner.overlap is False
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.equal(g0, g1), "late-bias staging + scatter must leave the gradient arena as the plain backward does"
    assert torch.equal(p0, p1) and torch.equal(s0, s1)


def test_two_rank_overlapped_exchange_matches_single_process(tmp_path):
    """Two ranks (two processes sharing cuda:0, gloo transport -- RCCL refuses two ranks on one device) run the overlapped
    exchange on their halves of the global batch; gradients norms, parameters and EMA match the single-process run on the whole
    batch and the replicas stay identical (tools/rehearse_dp2.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SFRON_REHEARSE_REF=str(tmp_path / "dp2_ref.pt"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rehearse_dp2.py"), "--single"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "tools", "rehearse_dp2.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("PASS=True") == 2


def test_pipelined_synchronous_exchange_matches_plain_path():
    """The synchronous fallback exchange of a data-parallel run (what runs when verify_overlap votes no): buckets all-reduced on a
    communication stream, the norm pre-pass / the remain-stage AdamW + EMA of bucket i on the compute stream as soon as bucket i is
    reduced.  At world size 1 over RCCL (force_dp: the collectives are identities) it must reproduce the plain pass: identical
    gradients, parameters to fp32 rounding (the clip norm is summed per bucket instead of per arena chunk)."""
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    cfg = CASES["hd72"]
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(6, it, "forget", **kw), data.synthetic_batch(6, it, "remain", **kw))
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)

    def run(force, transport="fp32"):
        _, model = build_pair(cfg, B, seed=23)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), grad_transport=transport, **hp)
        runner.factored_ada = runner.sweep_beside_forward = False
        runner.force_dp = force
        runner.bucket_elems = 50_000                      # several buckets in front of the blocks
        for it in range(2):
            out = runner.step(*bat(it))
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        return model.engine.params.clone(), model.engine.grads.clone(), out["stats"].clone(), runner.ema.clone()

    p0, g0, s0, e0 = run(False)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        p1, g1, s1, e1 = run(True)
        assert len(step.DiTSFRon._dp_buckets.__doc__) > 0
    finally:
        if created:
            dist.destroy_process_group()
    assert torch.allclose(s1, s0, rtol=1e-5)
    assert (p1 - p0).abs().max().item() <= 1e-6 and (e1 - e0).abs().max().item() <= 1e-6
    assert (g1 - g0).abs().max().item() <= 1e-6 * g0.abs().max().item()


def test_fp8_under_the_pipelined_synchronous_exchange_keeps_the_shadow_fresh():
    """fp8 + data parallel + synchronous exchange with the adaLN gradient inside the arena (factored_ada off) took the
    fused re-quantising sweep although the pipelined sweep consumes buckets: an UnboundLocalError in the remain stage and a stale
    e4m3 shadow after the forget stage.  With force_dp at world size 1 (collectives are identities) the run must equal the
    single-process fp8 run in its non-fused form, and the shadow must be the e4m3 image of the masters."""
    from sfron import data, diffusion, step
    from test_gpu_dit import build_pair
    cfg = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10)
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(8, it, "forget", **kw), data.synthetic_batch(8, it, "remain", **kw))
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)

    def run(force):
        _, model = build_pair(cfg, B, seed=31)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), fp8=True, **hp)
        runner.factored_ada = runner.sweep_beside_forward = False
        runner.force_dp = force
        for it in range(2):
            out = runner.step(*bat(it))
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        eng = model.engine
        # the shadow must be the e4m3 image of the CURRENT masters under the scales in use (a stale one -- the bug -- is the image of the
        # masters one optimizer step earlier): every block tensor, bit for bit (+0 == -0)
        from oracle import fp8_ref
        tab, sc = eng.fp8["table"].cpu().tolist(), eng.fp8["scales"].cpu().tolist()
        for (off, n), s in zip(tab, sc):
            want8 = fp8_ref.e4m3_bytes(eng.params[off:off + n].cpu(), float(s))
            got8 = eng.fp8["w8"][off:off + n].cpu()
            assert ((got8 == want8) | (((got8 & 0x7F) == 0) & ((want8 & 0x7F) == 0))).all(), (off, n)
        return eng.params.clone(), eng.fp8["scales"].clone(), out["stats"].clone()

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        p1, sc1, s1 = run(True)
    finally:
        if created:
            dist.destroy_process_group()
    p0, sc0, s0 = run(False)
    assert torch.allclose(s1, s0, rtol=1e-5)
    assert (p1 - p0).abs().max().item() <= 2e-6
    assert torch.equal(sc1, sc0)


@pytest.mark.parametrize("fp8", [False, True])
def test_remain_sweep_beside_the_next_step_gives_the_same_state(fp8):
    """sweep_across_steps: the remain-stage AdamW + EMA of the block ranges runs on the sweep stream beside the NEXT step's forget
    forward pass (which waits block by block); step() returns with it in flight.  Same kernels on the same operands: parameters,
    moments, EMA and the bf16 shadow after four steps equal the in-step form bit for bit; the runner's accessors (checkpoint) order
    themselves behind the sweep."""
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    # fp8: a width the e4m3 tile takes (hidden 128, 64 tokens, batch 4 -> M = 256: tests/test_gpu_fp8.py CFG)
    cfg = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=10) if fp8 else CASES["hd72"]
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(6, it, "forget", **kw), data.synthetic_batch(6, it, "remain", **kw))
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)

    def run(across, defer=False, ada_side=False):
        _, model = build_pair(cfg, B, seed=29)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), fp8=fp8, **hp)     # fp8: the sweeps also rewrite the e4m3 shadow
        assert runner.sweep_across_steps is False          # opt-in
        runner.sweep_across_steps = across
        # round 6: the second-stream launches of a beside-forward sweep issued by the forward pass itself, between its conditioning prologue
        # and block 0 (sfron_dit_forward_phase); the sweep left across the step boundary is handed to the next step's pass or to drain_sweep
        runner.defer_sweep_launch = defer
        # round 6 (late): the adaLN matrix's sweep on the sweep stream in front of the block ranges, the next pass's prologue beside it and its
        # adaLN product behind an event (sfron_dit_forward_phase 3 / 4); one head block, so that a range does go to the second stream at depth 2
        runner.ada_side = ada_side
        if ada_side:
            runner.sweep_beside_head = 1
        for it in range(4):
            runner.step(*bat(it))
        if ada_side:
            assert model.engine._shared.get("ada_done") is not None or not across
        if across:
            assert runner._ready_owner is not None and model.engine._sweep_pending is not None
        ck = runner.checkpoint()                           # drains the sweep before it reads
        assert getattr(model.engine, "_sweep_pending", None) is None
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        eng = model.engine
        return (eng.params.clone(), runner.opt.m.clone(), runner.opt.v.clone(), runner.ema.clone(), eng.params_bf16.clone(), ck)

    a = run(False)
    for b in (run(True), run(True, defer=True), run(False, defer=True), run(False, ada_side=True), run(True, ada_side=True),
              run(True, defer=True, ada_side=True)):
        for x, y in zip(a[:5], b[:5]):
            assert torch.equal(x, y)
        for k in a[5]["model"]:
            assert torch.equal(a[5]["model"][k], b[5]["model"][k])
        for k in a[5]["ema"]:
            assert torch.equal(a[5]["ema"][k], b[5]["ema"][k])


def test_batch_size_change_while_the_remain_sweep_is_in_flight():
    """with sweep_across_steps the runner keeps the per-block events of the sweep it left in flight; a batch-size
    change between two steps replaces the engine (the old one is drained and closed, its events destroyed).  The next step must not
    hand the stale handles to hipStreamWaitEvent: it runs on the new engine, and the state equals a run without the cross-step
    sweep bit for bit (same kernels on the same operands)."""
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    cfg = CASES["hd72"]
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)
    sizes = [4, 4, 2, 2, 4]

    def bat(it, B):
        kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
        return data.synthetic_batch(8, it, "forget", **kw), data.synthetic_batch(8, it, "remain", **kw)

    def run(across, user_resize=False):
        _, model = build_pair(cfg, sizes[0], seed=31)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), **hp)
        runner.sweep_across_steps = across
        for it, B in enumerate(sizes):
            if user_resize and it > 0 and B != sizes[it - 1]:
                model.set_batch_size(B)                    # the user resizes between two steps (step() would do the same)
            runner.step(*bat(it, B))
        runner.sync_sweep()
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        eng = model.engine
        return eng.params.clone(), runner.opt.m.clone(), runner.opt.v.clone(), runner.ema.clone(), eng.params_bf16.clone()

    ref = run(False)
    for got in (run(True), run(True, user_resize=True)):
        for x, y in zip(ref, got):
            assert torch.equal(x, y)


def test_guard_rejects_non_int64_labels_and_length_mismatch():
    """k_guard_inputs reads y and t as int64 vectors of one length; anything else raises instead of being
    reinterpreted."""
    from sfron import guard
    from sfron._lib import SfronError
    g = guard.StepGuard(torch.device(DEV))
    y = torch.zeros(4, dtype=torch.int64, device=DEV)
    t = torch.zeros(4, dtype=torch.int64, device=DEV)
    ys, ts = g.check_inputs(y, t, 10, 1000)
    assert ys.dtype == torch.int64 and torch.equal(ys, y) and torch.equal(ts, t)
    with pytest.raises(SfronError):
        g.check_inputs(y.to(torch.int32), t, 10, 1000)
    with pytest.raises(SfronError):
        g.check_inputs(y, t.to(torch.int32), 10, 1000)
    with pytest.raises(SfronError):
        g.check_inputs(y, t[:3], 10, 1000)


def test_data_parallel_rank_runs_the_single_process_kernel_sequence():
    """an N > 1 rank must not be slower than the N = 1 rank before a byte crosses the links.  With gradients exchanged
    (force_dp at world size 1 over RCCL: every collective is an identity) the step keeps the single-process shortcuts -- the adaLN
    gradient as its two all-GATHERED factors swept as a rank-(world x batch) product, the block sweeps beside the next forward pass,
    the remain-stage sweep across the step boundary -- so the state after three steps equals the plain single-process run BIT FOR BIT
    (the same kernels on the same operands), for the overlapped exchange and for the synchronous one."""
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    cfg = CASES["hd72"]
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    bat = lambda it: (data.synthetic_batch(12, it, "forget", **kw), data.synthetic_batch(12, it, "remain", **kw))
    hp = dict(lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None, unlearn_loss="ga", forget_class=3)

    def run(force, overlap):
        _, model = build_pair(cfg, B, seed=33)
        runner = step.DiTSFRon(model, diffusion.create_diffusion(""), overlap_allreduce=overlap, grad_transport="auto", **hp)
        assert runner.grad_transport == "fp32"             # the rule: bf16 from four ranks on
        assert runner.factored_ada and runner.sweep_beside_forward
        runner.sweep_across_steps = True
        runner.force_dp = force
        # the one single-process shortcut a data-parallel rank cannot keep: the clip norm assembled from the weight-gradient GEMMs' own sums
        # (step.fuse_clip_norm) -- the norm must be that of the REDUCED gradient.  Same values, another summation order: off for the
        # bit-for-bit comparison (this width does not take it anyway).
        runner.fuse_clip_norm = False
        for it in range(3):
            runner.step(*bat(it))
        if force:
            assert runner._ada_all is not None             # the factors were gathered, not reduced
        runner.sync_sweep()
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        eng = model.engine
        return eng.params.clone(), runner.opt.m.clone(), runner.opt.v.clone(), runner.ema.clone(), eng.params_bf16.clone()

    ref = run(False, False)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        for overlap in (True, False):
            got = run(True, overlap)
            for a, b in zip(ref, got):
                assert torch.equal(a, b), f"overlap={overlap}"
    finally:
        if created:
            dist.destroy_process_group()


def test_sibling_engine_orders_itself_behind_a_sweep_in_flight():
    """ADVICE r3 (engine.py): the pending-sweep stream belongs to the shared parameter arenas.  A sibling() engine over the same
    parameters (micro-batch chain, joint method) sees it and its forward pass drains it first -- the output equals the one computed
    after an explicit synchronisation, bit for bit."""
    from sfron import data, diffusion, step
    from test_gpu_dit import CASES, build_pair
    cfg = CASES["hd72"]
    B = 4
    kw = dict(global_batch=B, num_classes=cfg["num_classes"], forget_class=3, input_size=cfg["input_size"], device=DEV)
    _, model = build_pair(cfg, B, seed=41)
    runner = step.DiTSFRon(model, diffusion.create_diffusion(""), lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99, mask=None)
    runner.sweep_across_steps = True
    runner.step(data.synthetic_batch(14, 0, "forget", **kw), data.synthetic_batch(14, 0, "remain", **kw))
    eng = model.engine
    sib = eng.sibling(B)
    assert eng._sweep_pending is not None and sib._sweep_pending is eng._sweep_pending
    b = data.synthetic_batch(14, 1, "remain", **kw)
    out = sib.forward(b["x0"], b["t"], b["y"], b["drop"]).clone()          # drains the sweep first
    assert eng._sweep_pending is None and sib._sweep_pending is None
    torch.cuda.synchronize()
    assert torch.equal(out, sib.forward(b["x0"], b["t"], b["y"], b["drop"]))
    sib.close()
