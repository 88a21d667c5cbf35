#!/usr/bin/env python3
"""Rehearsal of the multi-rank data-parallel path on ONE GPU: two processes share cuda:0 and talk over gloo (RCCL refuses two
ranks on one device).  Checks that the overlapped gradient exchange (per-block all-reduce on a communication stream, late-bias
staging, all-gathered adaLN factors) over a sharded global batch reproduces the single-process run on the whole batch.
  python tools/rehearse_dp2.py --single                      # writes gpurun_out/dp2_ref.pt
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/rehearse_dp2.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.distributed as dist

DEV = "cuda:0"
CFG = dict(input_size=16, patch_size=2, in_channels=4, hidden_size=144, depth=3, num_heads=2, num_classes=10)
GB, STEPS = 8, 3
REF = os.environ.get("SFRON_REHEARSE_REF") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dp2_ref.pt")


def run(world, rank):
    from sfron import data, diffusion, dit, step
    torch.manual_seed(3)
    model = dit.DiT(batch_size=GB // world, **CFG)
    dit.randomize_zero_init(model, std=0.05, seed=4)
    gm = torch.Generator().manual_seed(5)
    runner = step.DiTSFRon(model, diffusion.create_diffusion("", device=DEV), lr=2e-4, forget_alpha=0.3, grad_clip=1.0, ema_decay=0.99,
                           mask=None, unlearn_loss="ga", forget_class=3, overlap_allreduce=world > 1)
    runner.mask_arena = (torch.rand(model.engine.n_trainable, generator=gm) < 0.5).to(torch.uint8).to(DEV)
    runner.opt.mask = runner.mask_arena
    kw = dict(num_classes=CFG["num_classes"], forget_class=3, input_size=CFG["input_size"], device=DEV)
    norms = []
    for it in range(STEPS):
        out = runner.step(data.synthetic_batch(9, it, "forget", GB, rank, world, **kw), data.synthetic_batch(9, it, "remain", GB, rank, world, **kw))
        norms.append(out["stats"][0].item())
    torch.cuda.synchronize()
    return model.engine.params.clone().cpu(), runner.ema.clone().cpu(), norms, runner._overlap_enabled()


if "--single" in sys.argv:
    p, e, norms, ov = run(1, 0)
    os.makedirs(os.path.dirname(REF), exist_ok=True)
    torch.save({"p": p, "e": e, "norms": norms}, REF)
    print("single-process reference written:", norms)
else:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    p, e, norms, ov = run(world, rank)
    ref = torch.load(REF)
    dp_, de = (p - ref["p"]).abs().max().item(), (e - ref["e"]).abs().max().item()
    rel = ((p - ref["p"]).norm() / (ref["p"] - ref["p"].mean()).norm()).item()
    print(f"rank {rank}/{world}: overlap path {ov}; grad norms {['%.5f' % n for n in norms]} vs single {['%.5f' % n for n in ref['norms']]}; "
          f"max |dp| {dp_:.3e}, max |d ema| {de:.3e}, rel {rel:.3e}", flush=True)
    ok = ov and all(abs(a - b) <= 2e-3 * abs(b) for a, b in zip(norms, ref["norms"])) and dp_ < 3 * 2e-4 * STEPS
    allp = [torch.zeros_like(p) for _ in range(world)]
    dist.all_gather(allp, p)
    same = all(torch.equal(allp[0], q) for q in allp)
    print(f"rank {rank}: replicas identical across ranks: {same}; PASS={ok and same}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok and same else 1)
