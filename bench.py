#!/usr/bin/env python3
"""bench.py -- SFR-on unlearning steps/sec, DiT-XL/2 256 px, batch 32 per GPU (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one full SFR-on iteration (DiT/forget.py:256-322): forget fwd/bwd -> mask -> clip -> AdamW,
remain fwd/bwd -> AdamW, EMA; synthetic latents / labels / timesteps / noise already resident in HBM.
Weak scaling: every rank runs batch 32 (global batch 32*N), gradients SUM-all-reduced over RCCL; "value" counts the batch-32 steps
of ALL ranks per second (N x the iteration rate), "ms_per_step" is the wall time of one synchronous iteration.
Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     -- the kernel with the largest share of GPU time (profiles/r06_kernel_table.md): the weight-gradient GEMM
                    (one kernel for the four products dW = dY^T X of a block), timed live with HIP event pairs recorded on the
                    weight-gradient stream it is launched on (all four GEMMs of every 9th block of every backward pass inside
                    the timed region); "others" holds the same measurement for the fc1 forward GEMM (main stream) and an "hbm"
                    entry for the parameter sweep k_masked_clip_adam
  "cpu_baseline" -- the oracle (plain PyTorch fp32, CPU) timed on a bounded sample on this box's host cores.
With N > 1 the overlapped gradient exchange is switched on only after DiTSFRon.verify_overlap() has reproduced the synchronous
exchange on the ranks of this very run ("dp_overlap" in the JSON says which path was timed).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="DiT-XL/2")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro-batches", type=int, default=1, help="independent half-batch chains per pass (1 or 2)")
    ap.add_argument("--cpu-batch", type=int, default=8, help="batch of the bounded CPU-baseline sample")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: keep the synchronous bucketed all-reduce")
    ap.add_argument("--no-sweep-beside", action="store_true", help="A-B knob: forget-stage AdamW of the blocks on the main stream")
    ap.add_argument("--no-factored-ada", action="store_true", help="A-B knob: form the adaLN weight gradient by a GEMM + flat sweep")
    ap.add_argument("--no-sweep-across-steps", action="store_true",
                    help="A-B knob: the remain-stage AdamW + EMA of the blocks finishes inside its step instead of beside the next step's forward pass")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE config 5: the four block GEMMs of every FORWARD pass on the fp8 (e4m3) matrix core, e4m3 weight shadow "
                         "re-quantised after each optimizer step; backward GEMMs stay bf16")
    ap.add_argument("--grad-transport", default="auto", choices=("auto", "fp32", "bf16"),
                    help="N > 1: precision of the gradient exchange; auto = bf16 from four ranks on, the exact fp32 sum below that")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the bounded runs of BASELINE configs 1, 2, 4, 5 behind the headline region (roofline.others.configs)")
    ap.add_argument("--check", action="store_true",
                    help="N-rank == 1-rank parity: after the run every rank also computes the gradient of the WHOLE global batch "
                         "of step 0 by itself and compares it with the all-reduced gradient of the sharded run")
    ap.add_argument("--no-dp-evidence", action="store_true",
                    help="N > 1: skip the default self-verification block of the JSON line (device identities, linearity check of the "
                         "exchange, replica checksums, all-reduce timing)")
    return ap.parse_args()


def host_cores():
    """Threads for the CPU baseline: the box's CPU share (affinity, cgroup quota), at most 16 per GPU."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(model_name, latent, batch_full, cpu_batch):
    """One SFR-on iteration of the ORACLE (CPU fp32 eager) at a reduced batch; cost is linear in batch to within
    the optimizer sweep, which is included unscaled (so the scaled value slightly over-estimates CPU speed... it
    is a baseline, not a target)."""
    from oracle import diffusion_ref as dref
    from oracle import dit_ref, sfron_ref
    from sfron import data
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: building {model_name} oracle on {cores} host threads ...", file=sys.stderr, flush=True)
    torch.manual_seed(0)
    t0 = time.time()
    ref = dit_ref.build(model_name, input_size=latent)
    dit_ref.randomize_zero_init(ref)
    orc = sfron_ref.DiTSfronOracle(ref, dref.DiffusionTables(1000), lr=1e-4, forget_alpha=1e-3, mask=None)
    build_s = time.time() - t0
    kw = dict(global_batch=cpu_batch, input_size=latent)

    def batch(i, s):
        b = data.synthetic_batch(0, i, s, **kw)
        b["drop"] = b["drop"].long()
        return b
    # steady-state steps -- discard the first 2 (allocations, thread pools, the optimizer's lazily created state:
    # AdamW allocates its moments in the first step), time the next 2, report mean and min
    discard, timed = 2, 2
    print(f"[bench] cpu_baseline: built in {build_s:.1f} s; {discard} discarded + {timed} timed iterations ...", file=sys.stderr, flush=True)
    for i in range(discard):
        orc.step(batch(i, "forget"), batch(i, "remain"))
    dts = []
    for i in range(discard, discard + timed):
        t0 = time.time()
        orc.step(batch(i, "forget"), batch(i, "remain"))
        dts.append(time.time() - t0)
    dt = sum(dts) / len(dts)
    steps_per_s_full = (1.0 / dt) * (cpu_batch / batch_full)
    return {"value": steps_per_s_full, "unit": "steps/s", "cores": cores, "kind": "port",
            "iteration_s": {"mean": dt, "min": min(dts), "all": dts, "discarded": discard},
            "sample": f"{timed} timed SFR-on iterations of {model_name} (oracle, CPU fp32 eager, {cores} threads) at batch {cpu_batch} after "
                      f"{discard} discarded ones: mean {dt:.2f} s, min {min(dts):.2f} s per iteration; value = (1/mean) * {cpu_batch}/{batch_full} "
                      f"(linear in batch); model build {build_s:.1f} s not counted"}


def _dit_leg(model_name, batch, latent, dev, fp8, steps, warmup):
    """One more DiT configuration through the same runner as the headline (BASELINE configs 2 and 5): ms / step, TFLOP/s, fraction of peak."""
    from sfron import data, diffusion, dit, step
    model = dit.DiT_models[model_name](input_size=latent, num_classes=1000, batch_size=batch, device=dev)
    torch.manual_seed(1234)
    model.initialize_weights()
    dit.randomize_zero_init(model, std=0.02, seed=1)
    model.train()
    eng = model.engine
    gm = torch.Generator().manual_seed(0)
    runner = step.DiTSFRon(model, diffusion.create_diffusion("", device=dev), lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999,
                           mask=None, unlearn_loss="ga", forget_class=207, fp8=fp8)
    runner.mask_arena = runner.opt.mask = (torch.rand(eng.n_trainable, generator=gm) < 0.5).to(torch.uint8).to(dev)
    runner.sweep_across_steps = True
    bt = [(data.synthetic_batch(0, i, "forget", batch, 0, 1, input_size=latent, device=dev),
           data.synthetic_batch(0, i, "remain", batch, 0, 1, input_size=latent, device=dev)) for i in range(2)]
    for i in range(warmup):
        runner.step(*bt[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = runner.step(*bt[i % 2])
    runner.sync_sweep()              # the last step's block sweep is launched by the NEXT forward pass or by this call: it belongs to the timed steps
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    runner.guard.poll(block=True)
    c = eng.cfg
    T, D, L, F = eng.tokens, c.hidden, c.depth, c.mlp_hidden
    flops = 6.0 * batch * L * (2 * T * D * (3 * D + D + 2 * F) + 4 * T * T * D + 2 * D * 6 * D)
    ok = bool(torch.isfinite(out["remain_mse"]).all().item() and torch.isfinite(out["forget_mse"]).all().item())
    runner.sync_sweep()
    rng = eng.fp8_activation_range() if fp8 else None
    eng.close()
    return {"ms_per_step": ms, "steps_per_s": 1e3 / ms, "steps": steps, "batch": batch, "tflops": flops / ms / 1e9,
            "frac_of_bf16_mfma_peak": flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, "finite_losses": ok,
            **({"fp8_activation_range": rng} if fp8 else {})}


def _ddpm_leg(dev, steps=50, batch=64):
    """BASELINE config 1 as worded: DDPM CIFAR-10 class-forget, 50 SFR-on steps at batch 64 (cifar10_sfron.yml U-Net, adaga, cosine alpha,
    clip in both stages, EMA; DDPM/runners/diffusion.py:1075-1180), stage graphs replayed."""
    from sfron import ddpm, unet
    torch.manual_seed(1234)
    model = unet.Conditional_Model(unet.config_namespace())
    run = ddpm.DDPMSFRon(model, lr=1e-4, forget_alpha=10.0, grad_clip=1.0, ema_rate=1e-4, mask=None, unlearn_loss="adaga", lambd=0.5,
                         n_iters=steps, use_graphs=True)
    g = torch.Generator().manual_seed(1)

    def synth(stream):
        t = torch.randint(0, 1000, (batch // 2 + 1,), generator=g)
        c = torch.zeros(batch, dtype=torch.int64) if stream == "forget" else torch.randint(1, 10, (batch,), generator=g)
        d = dict(x0=torch.rand(batch, 3, 32, 32, generator=g) * 2 - 1, e=torch.randn(batch, 3, 32, 32, generator=g),
                 t=torch.cat([t, 1000 - t - 1])[:batch], c=c)                   # antithetic timesteps (runners/diffusion.py:1091-1094)
        return {k: v.to(dev) for k, v in d.items()}
    bt = [(synth("forget"), synth("remain")) for _ in range(2)]
    unet.PRODUCT_FLOPS = [0.0]
    run.step(0, *bt[0])                      # eager: the tape's Python runs once -> the products of one iteration are counted
    flops, unet.PRODUCT_FLOPS = unet.PRODUCT_FLOPS[0], None
    for i in range(2):                       # capture + first replay
        run.step(i, *bt[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = run.step(i, *bt[i % 2])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": ms, "steps_per_s": 1e3 / ms, "steps": steps, "batch": batch, "product_tflop_per_step": flops / 1e12,
            "tflops": flops / ms / 1e9, "frac_of_bf16_mfma_peak": flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS,
            "finite_losses": bool(torch.isfinite(out["forget_loss"]).item() and torch.isfinite(out["remain_loss"]).item())}


def _sd_leg(dev, cases=(("xattn", 2), ("full", 2), ("full", 8)), steps=8):
    """BASELINE config 4: SD v1 UNet (859.5 M parameters, 64 x 64 latents, 77-token context) SFR-on iterations of
    SD/train-scripts/nsfw_removal.py:108-173: train_method xattn (the "cross-attn path" BASELINE.json names: only the attn2 layers
    train, nsfw_removal.py:66-77) at the README's batch 2, and train_method full at batch 2 and the script's default batch 8."""
    from sfron import sd, sd_unet, unet
    torch.manual_seed(0)
    model = sd_unet.UNetModel()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.parameters():
            if not bool(p.any()):
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    model.sync_bf16()
    gd = torch.Generator(device=dev).manual_seed(2)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=gd)
    res = {}
    for method, B in cases:
        run = sd.SDSFRon(model, lr=1e-5, train_method=method, use_graphs=True)
        c_f, c_p = rn(1, 77, 768).expand(B, -1, -1).contiguous(), rn(1, 77, 768).expand(B, -1, -1).contiguous()

        def batch():
            xf = rn(B, 4, 64, 64)
            return (dict(x_f=xf, x_p=xf, c_f=c_f, c_p=c_p, t=torch.randint(0, 1000, (B,), device=dev, generator=gd), noise=rn(B, 4, 64, 64)),
                    dict(x=rn(B, 4, 64, 64), c=c_p, t=torch.randint(0, 1000, (B,), device=dev, generator=gd), noise=rn(B, 4, 64, 64)))
        bt = [batch() for _ in range(2)]
        unet.PRODUCT_FLOPS = [0.0]
        run.step(*bt[0])                     # eager (counts the products of one iteration)
        flops, unet.PRODUCT_FLOPS = unet.PRODUCT_FLOPS[0], None
        for i in range(2):
            run.step(*bt[i % 2])             # capture + first replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            out = run.step(*bt[i % 2])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[f"batch{B}" if method == "full" else f"{method}_batch{B}"] = {"train_method": method, "ms_per_iteration": ms, "iterations_per_s": 1e3 / ms, "steps": steps, "product_tflop_per_iteration": flops / 1e12,
                            "tflops": flops / ms / 1e9, "frac_of_bf16_mfma_peak": flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                            "finite_losses": bool(all(torch.isfinite(v).all().item() for v in out.values() if torch.is_tensor(v)))}
        del run
        gc.collect()                         # the runner's stage graphs die by the collector (closure cycles): now, not inside the next capture
        torch.cuda.empty_cache()
    return res


def other_configs(dev, latent, budget_s=75.0):
    """BASELINE configs 1, 2, 4, 5 behind the headline region, each through its own runner on synthetic inputs, bounded in time; a leg that
    fails or no longer fits the budget reports why instead of a number (the headline line is never at risk)."""
    t_start = time.perf_counter()
    out = {"note": "measured after the headline's timed region in the same process; product FLOPs of the U-Net legs are counted from the "
                   "launches of one iteration (zero-dilated / channel-padded operands included)"}
    legs = [("config2_dit_b4_bs32", lambda: _dit_leg("DiT-B/4", 32, latent, dev, False, 20, 4)),
            ("config5_dit_xl2_fp8_bs32", lambda: _dit_leg("DiT-XL/2", 32, latent, dev, True, 20, 6)),
            ("config1_ddpm_cifar10_bs64_50steps", lambda: _ddpm_leg(dev)),
            ("config4_sd_v1_unet", lambda: _sd_leg(dev))]
    if os.environ.get("SFRON_BENCH_LEGS"):                  # diagnosis knob (tools only): which legs, in which order ("config5,config2")
        legs = [lg for key in os.environ["SFRON_BENCH_LEGS"].split(",") for lg in legs if lg[0].startswith(key)]
    for name, fn in legs:
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": f"time budget of {budget_s:.0f} s used up by the legs before it"}
            continue
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            out[name]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as e:      # noqa: BLE001 -- report, keep the headline
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def dp_evidence(runner, model, diff, batch0, dev, rank, world, strict_tol=None):
    """What a reader of the JSON line needs to see that N ranks on N DEVICES exchanged gradients correctly, collected on
    every multi-rank run by default, AFTER the timed region:
      devices            every rank's (host, device UUID / PCI bus id), all-gathered; `distinct_devices` must equal world_size
      linearity          the exchange is a SUM: a strided sample of this rank's LOCAL gradient (plain backward pass, no exchange) is
                         all-gathered, summed over ranks in fp64 and compared with the same sample of the EXCHANGED gradient the runner's
                         own pass produces (overlapped or synchronous, whichever the timed region used; the adaLN range goes through the
                         all-gathered factors, so a wrong rank order in all_gather_into_tensor shows here)
      replicas_identical checksums of the parameter arena after the timed steps, all-gathered: every rank applied the same update
      allreduce_ms       one gradient stream's exchange (the flat arena through dp.allreduce_flat_ with the run's transport), timed alone
    Collective: every rank calls it."""
    import socket

    import torch.distributed as dist
    from sfron import dp
    eng = model.engine
    nt = eng.n_trainable
    props = torch.cuda.get_device_properties(dev)
    ident = {"rank": rank, "host": socket.gethostname(), "device_index": dev.index,
             "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None), "name": props.name}
    idents = [None] * world
    dist.all_gather_object(idents, ident, group=runner.pg)
    keys = sorted({(d["host"], d["uuid"] or d["pci_bus_id"] or d["device_index"]) for d in idents})
    # ---- linearity of the exchange on the first forget batch
    f0 = batch0
    b, y = runner._checked(f0, f0["y"])
    n_global = b["x0"].shape[0] * world
    x_t = diff.q_sample(b["x0"], b["t"], b["noise"])
    out = eng.forward(x_t, b["t"], y, b.get("drop"))
    _, _, d_out = diff.loss_fwd_bwd(out, b["x0"], b["t"], b["noise"], -runner.forget_alpha / n_global)
    eng.backward(d_out, y, b.get("drop"))                      # LOCAL gradient of this rank's shard, no exchange
    K = 1 << 20
    idx = torch.arange(0, nt, max(1, nt // K), device=dev)[:K]
    local = eng.grads[:nt][idx].clone()
    allg = torch.empty(world * local.numel(), dtype=torch.float32, device=dev)       # flat: gloo checks the chunk shape against the input's
    dist.all_gather_into_tensor(allg, local, group=runner.pg)
    want = allg.view(world, -1).double().sum(0)
    runner._pass(f0, f0["y"], -runner.forget_alpha)            # the runner's own exchange (adaLN product formed from gathered factors)
    got = eng.grads[:nt][idx].double()
    rel = ((got - want).norm() / (want.norm() + 1e-300)).float()
    dist.all_reduce(rel, op=dist.ReduceOp.MAX, group=runner.pg)
    tol = strict_tol if strict_tol is not None else (1e-2 if runner.grad_transport == "bf16" else 1e-5)
    # ---- replicas: the parameter arenas after the timed steps
    runner.sync_sweep()
    bits = eng.params.view(torch.int32).to(torch.int64)
    chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
    chks = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(chks, chk, group=runner.pg)
    chks = chks.view(world, 2)
    same = bool((chks == chks[0:1]).all().item())
    # ---- one stream's exchange alone
    g = eng.grads[:nt].clone()
    ms = []
    for _ in range(3):
        dist.barrier(group=runner.pg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dp.allreduce_flat_(g, runner.bucket_elems, runner.pg, transport=runner.grad_transport,
                           scratch=runner._scratch(min(nt, runner.bucket_elems)))
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    tt = torch.tensor([min(ms[1:])], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=runner.pg)
    payload = nt * (2 if runner.grad_transport == "bf16" else 4)
    return {"world_size": world, "backend": dist.get_backend(runner.pg), "devices": sorted(idents, key=lambda d: d["rank"]),
            "distinct_devices": len(keys),
            "linearity": {"what": f"sum over {world} ranks of a {idx.numel()}-element strided sample of the local gradients vs the same sample of the "
                                  "exchanged gradient (forget stage, batch 0)", "rel_l2": rel.item(), "tol": tol, "ok": bool(rel.item() < tol)},
            "replicas_identical": same,
            "allreduce_ms_per_stream": tt.item(), "allreduce_payload_bytes": payload,
            "allreduce_busbw_GBps": (2.0 * (world - 1) / world * payload / (tt.item() * 1e-3) / 1e9) if tt.item() > 0 else None,
            "ok": bool(rel.item() < tol) and same and len(keys) == (world if dist.get_backend(runner.pg) == "nccl" else len(keys))}


def launch_ranks(n):
    """Run this script as n ranks under torch.distributed.run (child process, rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] WORLD_SIZE unset and --gpus {n}: starting {' '.join(cmd[1:8])} ... as a child job", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
    return subprocess.run(cmd, env=env).returncode          # stdout / stderr are inherited: rank 0's JSON line passes through


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks as a CHILD job (one process per GPU under torch.distributed.run) and
        # relay its output and exit code.  Nothing in this process has touched the GPU yet (importing torch does not), and it
        # never will: a process that initialised HIP must not exec another program on this pool.
        raise SystemExit(launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    wd = None
    if world > 1:
        from sfron import dp as _dp
        wd = _dp.Watchdog(rank)
        wd.phase("rendezvous + model build", float(os.environ.get("SFRON_BENCH_SETUP_BUDGET_S", "600")))
    # rehearsal on a one-GPU box (tools / tests only): SFRON_BENCH_BACKEND=gloo lets N ranks share the visible devices -- RCCL refuses
    # two ranks on one device; the driver's runs use nccl (= RCCL) with one device per rank
    backend = os.environ.get("SFRON_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from sfron import data, diffusion, dit, step

    latent = args.image_size // 8
    model = dit.DiT_models[args.model](input_size=latent, num_classes=1000, batch_size=args.batch, device=dev)
    torch.manual_seed(1234)          # identical replicas on every rank
    model.initialize_weights()
    dit.randomize_zero_init(model, std=0.02, seed=1)
    model.train()
    eng = model.engine
    gm = torch.Generator().manual_seed(0)
    mask_arena = (torch.rand(eng.n_trainable, generator=gm) < 0.5).to(torch.uint8).to(dev)     # 50 % synthetic saliency mask
    diff = diffusion.create_diffusion("", device=dev)
    runner = step.DiTSFRon(model, diff, lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=None,
                           unlearn_loss="ga", forget_class=207, micro_batches=args.micro_batches, fp8=args.fp8,
                           grad_transport=args.grad_transport)
    runner.mask_arena = mask_arena
    runner.opt.mask = mask_arena
    if args.no_factored_ada:
        runner.factored_ada = False
    if args.no_sweep_beside:
        runner.sweep_beside_forward = False
    # steps run back to back here, as in the reference's training loop (forget.py:256-336 logs every N steps): the remain-stage sweep of
    # the blocks overlaps the next step's forward pass; the timed region ends with torch.cuda.synchronize(), i.e. behind the last sweep
    runner.sweep_across_steps = not args.no_sweep_across_steps and not args.no_sweep_beside
    if os.environ.get("SFRON_BENCH_LOADER_WAVES"):          # A-B knob (tools only): form of the three-slot GEMM tiles, 0 or 4
        from sfron import _lib
        _lib.lib().sfron_gemm_loader_waves(int(os.environ["SFRON_BENCH_LOADER_WAVES"]))
    if os.environ.get("SFRON_BENCH_ATTN_FWD"):              # A-B knob (tools only): 4 = the four-wave attention forward kernel
        from sfron import _lib
        _lib.lib().sfron_attn_fwd_form(int(os.environ["SFRON_BENCH_ATTN_FWD"]))
    if os.environ.get("SFRON_BENCH_EARLY_ADA"):             # A-B knob (tools only): 0 = the adaLN matrix's share of the clip norm on the caller's stream
        runner.opt.early_ada = os.environ["SFRON_BENCH_EARLY_ADA"] != "0"
    if os.environ.get("SFRON_BENCH_ZERO_WS"):               # timing experiments with builds that skip a store: the workspace starts as finite numbers
        model.engine.workspace.zero_()
    if os.environ.get("SFRON_BENCH_SWEEP_PRIORITY"):        # A-B knob (tools only): torch stream priority of the beside-forward sweep stream
        from sfron import streams as _st
        _st.PRIORITY["sweep"] = int(os.environ["SFRON_BENCH_SWEEP_PRIORITY"])
    if os.environ.get("SFRON_BENCH_DEFER_SWEEP"):           # A-B knob (tools only): 0 = start the beside-forward sweep before the pass's prologue (round 5)
        runner.defer_sweep_launch = os.environ["SFRON_BENCH_DEFER_SWEEP"] != "0"
    if os.environ.get("SFRON_BENCH_ADA_SIDE"):              # A-B knob (tools only): the adaLN sweep on the sweep stream, the next pass's prologue beside it
        runner.ada_side = os.environ["SFRON_BENCH_ADA_SIDE"] != "0"
    if os.environ.get("SFRON_BENCH_SWEEP_BESIDE"):          # tuning knob (tools only): "workgroups,head"
        runner.sweep_beside_wg, runner.sweep_beside_head = (int(v) for v in os.environ["SFRON_BENCH_SWEEP_BESIDE"].split(","))

    pool = 4
    gb = args.batch * world
    batches = [(data.synthetic_batch(0, i, "forget", gb, rank, world, input_size=latent, device=dev),
                data.synthetic_batch(0, i, "remain", gb, rank, world, input_size=latent, device=dev)) for i in range(pool)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_budget = float(os.environ.get("SFRON_BENCH_STEP_BUDGET_S", "30"))     # generous: a healthy step takes < 0.2 s
    dp_overlap = False
    if world > 1 and not args.no_overlap and args.micro_batches == 1:
        wd.phase("verify_overlap", 120 + 4 * step_budget)
        # collective: every rank runs both exchange paths on the first batch (on the exact fp32 transport) and the verdict is MIN-all-reduced, so
        # the ranks agree on the path; a rank that fails inside the passes re-raises (it cannot rejoin its peers: the launcher tears the job
        # down, the peers' watchdog ends them with exit 124)
        dp_overlap = runner.verify_overlap(batches[0][0])
        runner.overlap = dp_overlap
        if rank == 0:
            print(f"[bench] overlapped gradient exchange vs synchronous all-reduce on {world} ranks: "
                  f"{'match -> overlap ON' if dp_overlap else 'MISMATCH -> synchronous path'}", file=sys.stderr, flush=True)
    if wd:
        wd.phase("warm-up", 120 + step_budget * args.warmup)
    for i in range(args.warmup):
        runner.step(*batches[i % pool])
    eng = model.engine
    eng.probe_enable(2 * args.steps + 4)
    eng.wgrad_probe_enable(32 * args.steps + 32)
    runner.opt.timed = []                       # (start, end) event pairs around every k_masked_clip_adam launch
    sync()
    if wd:
        wd.phase("timed region", 60 + step_budget * args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = runner.step(*batches[i % pool])
    runner.sync_sweep()             # the last step's remain-stage block sweep (left to the next forward pass to launch) belongs to the timed region
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    runner.guard.poll(block=True)               # NaN / Inf loss or gradient norm, bad labels: raises (non-zero exit)
    loss_ok = bool(torch.isfinite(out["remain_mse"]).all().item() and torch.isfinite(out["forget_mse"]).all().item())
    n_probe, probe_ms = eng.probe_read()
    n_wp, wp_ms = eng.wgrad_probe_read()
    sched_across = bool(runner.sweep_across_steps and runner.sweep_beside_forward)      # as the timed region ran (data-parallel ranks too)
    sweep_where = "inside the timed region"
    if not any(whole for _, _, whole in runner.opt.timed) and world == 1:
        # the remain-stage sweep of the blocks ran split across the step boundary (config.schedule): for the `others.hbm` roofline time
        # it WHOLE -- one launch sequence on one stream -- on two extra steps after the timed region
        runner.sync_sweep()
        runner.sweep_across_steps = False
        runner.opt.timed = []
        for i in range(2):
            runner.step(*batches[i % pool])
        torch.cuda.synchronize()
        runner.guard.poll(block=True)
        sweep_where = "two extra steps after the timed region, sweep inside its step (in the timed region it runs beside the next forward pass)"
    sweep_ms = [a.elapsed_time(b) for a, b, whole in runner.opt.timed if whole]      # the remain-stage sweeps (the forget stage's block
                                                                                        # ranges run beside the next forward pass)
    cfg = eng.cfg
    M = args.batch * eng.tokens
    fc1_flops = 2.0 * M * cfg.mlp_hidden * cfg.hidden
    avg_ms = probe_ms / max(1, n_probe)
    achieved = fc1_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    # algorithmic FLOPs of the whole step: 6 x forward
    T, D, L, F = eng.tokens, cfg.hidden, cfg.depth, cfg.mlp_hidden
    fwd_per_sample = L * (2 * T * D * (3 * D + D + 2 * F) + 4 * T * T * D + 2 * D * 6 * D)
    step_flops = 6.0 * fwd_per_sample * args.batch
    ms_per_step = elapsed / args.steps * 1e3

    # dominant kernel by GPU time (profiles/r06_kernel_table.md): the weight-gradient GEMM k_gemm_pipe<4,2,3,6,true,true,1,2,2>, one
    # kernel name for the four shapes dW = dY^T X of a block (qkv 65.2, proj 21.7, fc1 87.0, fc2 87.0 GFLOP): the probe brackets all
    # four, so the mean launch does their mean
    wg_flops = 0.25 * (2.0 * M * 3 * D * D + 2.0 * M * D * D + 2 * 2.0 * M * F * D)
    wg_ms = wp_ms / max(1, n_wp)
    wg_ach = wg_flops / (wg_ms * 1e-3) / 1e12 if wg_ms > 0 else 0.0
    # parameter sweep, remain stage: 38 B/param (g, p, m, v, EMA in; p, m, v, EMA, bf16 out) over the arena, minus the 4 B/param
    # gradient read of the adaLN matrix, whose gradient the sweep forms from its two factors (k_adam_lowrank)
    nt = eng.n_trainable
    n_ada = (6 * L + 2) * D * D if runner.factored_ada else 0
    sweep_bytes = 38.0 * nt - 4.0 * n_ada
    sw_ms = sum(sweep_ms) / max(1, len(sweep_ms))
    sw_ach = sweep_bytes / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else None    # None: no whole-arena launch to time (config 5 sweeps tensor by tensor)

    # HBM traffic of the probed kernels: PMC counters need their own rocprofv3 passes (FETCH_SIZE, WRITE_SIZE), so the numbers are
    # measured offline on this same command (tools/pmc_traffic.py) and committed under profiles/ TOGETHER WITH the hash of the HIP
    # sources they were measured on: a figure whose hash differs from the tree's is not reported (traffic: null)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_sha import csrc_sha
    tree_sha = csrc_sha()

    def committed_traffic(fname, key="traffic_bytes_per_launch"):
        tpath = os.path.join(ROOT, "profiles", fname)
        if args.model == "DiT-XL/2" and args.batch == 32 and not args.fp8 and os.path.isfile(tpath):
            j = json.load(open(tpath))
            if j.get("csrc_sha") == tree_sha:
                return j.get(key)
        return None
    traffic = committed_traffic("r06_wgrad_traffic.json")

    def sane_sweep_traffic():
        """the committed PMC figure of all sweeps of a step, or None with the reason when it cannot be true: scaled to the remain-stage
        sweep timed above it must not imply more than the HBM peak (round 3's figure did: its tool divided by the wrong step count)"""
        tr, alg = committed_traffic("r06_sweep_traffic.json", "traffic_bytes_per_step"), committed_traffic("r06_sweep_traffic.json", "algorithmic_bytes_per_step")
        if tr is None or alg is None or sw_ach is None:
            return tr, alg, None
        implied = tr / alg * sw_ach
        if implied > HBM_PEAK_GBS:
            return None, alg, f"refused: {tr / 1e9:.1f} GB per step at the measured sweep rate would be {implied:.0f} GB/s > the {HBM_PEAK_GBS:.0f} GB/s peak"
        return tr, alg, None
    sweep_traffic, sweep_alg, sweep_traffic_note = sane_sweep_traffic()

    dp_ev = None
    if world > 1 and not args.no_dp_evidence and args.micro_batches == 1:
        wd.phase("dp evidence", 180 + 8 * step_budget)
        try:
            dp_ev = dp_evidence(runner, model, diff, batches[0][0], dev, rank, world)
        except Exception as e:      # noqa: BLE001 -- the timed result above stands; a one-sided failure leaves the peers to the watchdog
            dp_ev = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:300]}"}
            print(f"[bench] rank {rank}: dp evidence failed: {dp_ev['error']}", file=sys.stderr, flush=True)
    check_res = None
    if args.check and world > 1:
        wd.phase("check", 180 + 8 * step_budget)
        # N-rank == 1-rank parity at equal global batch: the all-reduced gradient of the sharded forget pass of step 0 against
        # the gradient this rank computes by itself on the WHOLE global batch (same loss scale alpha / global_batch)
        f0 = batches[0][0]
        runner._pass(f0, f0["y"], -runner.forget_alpha)
        g_dp = eng.grads[:nt].clone()
        full = data.synthetic_batch(0, 0, "forget", gb, 0, 1, input_size=latent, device=dev)
        model.set_batch_size(gb)
        e2 = model.engine
        x_t = diff.q_sample(full["x0"], full["t"], full["noise"])
        o2 = e2.forward(x_t, full["t"], full["y"], full["drop"])
        _, _, d_o2 = diff.loss_fwd_bwd(o2, full["x0"], full["t"], full["noise"], -runner.forget_alpha / gb)
        e2.backward(d_o2, full["y"], full["drop"])
        rel = ((e2.grads[:nt] - g_dp).norm() / (g_dp.norm() + 1e-30))
        dist.all_reduce(rel, op=dist.ReduceOp.MAX)
        check_res = {"what": f"all-reduced gradient of {world} shards of {args.batch} vs one rank on the global batch of {gb}",
                     "rel_l2": rel.item(), "ok": bool(rel.item() < 5e-3)}
        model.set_batch_size(args.batch)

    if rank == 0:
        res = {
            "metric": f"SFR-on unlearning steps/sec, {args.model} {args.image_size}px bs{args.batch}/GPU",   # BASELINE.json's metric at the defaults
            # whole-job aggregate: every rank runs batch-32 steps (weak scaling), so the job does world * steps of them in `elapsed`
            # (one synchronous iteration over the global batch of 32 * world samples takes ms_per_step)
            "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8_e4m3 forward GEMMs (weights + activations), bf16 backward GEMMs, fp32 accumulate" if args.fp8 else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model} {args.image_size}px SFR-on step (forget+remain fwd/bwd, masked clipped AdamW x2, EMA), "
                                   f"batch {args.batch}/GPU, random-init weights (zero-init tensors re-drawn N(0,0.02)), 50% synthetic mask"
                                   + (" -- BASELINE config 5 (fp8 forward)" if args.fp8 else ""),
                       "global_batch": gb, "tokens": T, "parallelism": f"dp{world}",
                       "schedule": {"forget_sweep_beside_remain_forward": bool(runner.sweep_beside_forward),
                                    "remain_sweep_beside_next_forget_forward": sched_across,
                                    "note": "every sweep is inside the timed region (it ends with torch.cuda.synchronize())"}},
            "finite_losses": loss_ok,
            # config 5: fraction of the e4m3 range the activations reached in this run (static activation scales, saturating conversion)
            "fp8_activation_range": model.engine.fp8_activation_range() if args.fp8 else None,
            "step_tflops_per_gpu": step_flops / (ms_per_step * 1e-3) / 1e12,
            "step_frac_of_bf16_mfma_peak": step_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "dp_overlap": dp_overlap, "grad_transport": runner.grad_transport if world > 1 else None, "check": check_res,
            "world_size": world, "dp": dp_ev,
            "roofline": {"bound": "mfma",
                         "kernel": "k_gemm_pipe<4,2,3,6,true,true,1,2,2,false,4> = 192x192 tile, three LDS slots, eight multiplying waves + four loader waves "
                                   "that issue the LDS-DMA: the weight "
                                   f"gradients dW = dY^T X of a block (qkv [{3 * D}x{D}], proj [{D}x{D}], fc1 [{F}x{D}], fc2 [{D}x{F}], contraction over "
                                   f"{M} token rows; mean {wg_flops / 1e9:.1f} GFLOP per launch); largest share of GPU time (profiles/r06_kernel_table.md); "
                                   "it runs on the weight-gradient stream BESIDE the dgrad chain, so its duration is shared-CU time",
                         "achieved": wg_ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": wg_ach / MFMA_BF16_PEAK_TFLOPS,
                         "traffic": traffic, "flops_per_launch": wg_flops, "avg_launch_ms": wg_ms, "launches_timed": n_wp,
                         "others": {
                             "fwd_fc1_gelu": {"bound": "mfma", "kernel": f"k_gemm_pipe<4,2,4,6,false,false,7,1,2>: Mlp.fc1 + GELU-tanh [{M}x{D}]x[{D}x{F}] (second output: GELU' as one byte per element), block 0 of "
                                              "every forward pass (main stream, nothing beside it; not launched by the fp8 path)", "achieved": achieved if n_probe else None,
                                              "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (achieved / MFMA_BF16_PEAK_TFLOPS) if n_probe else None,
                                              "traffic": committed_traffic("r06_fc1_traffic.json") if n_probe else None,
                                              "flops_per_launch": fc1_flops, "avg_launch_ms": avg_ms, "launches_timed": n_probe},
                             "hbm": {"bound": "hbm", "kernel": "remain-stage parameter sweep: k_masked_clip_adam (AdamW + EMA + bf16 shadow, 38 B/param) over the flat arenas "
                                     "+ k_adam_lowrank over the adaLN matrix (gradient formed from its two factors: 34 B/param)", "achieved": sw_ach, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": (sw_ach / HBM_PEAK_GBS) if sw_ach is not None else None, "traffic": None,
                                     "traffic_all_sweeps_per_step": sweep_traffic, "algorithmic_all_sweeps_per_step": sweep_alg,
                                     "traffic_note": sweep_traffic_note,
                                     "bytes_per_launch": sweep_bytes, "avg_launch_ms": sw_ms, "launches_timed": len(sweep_ms),
                                     "measured": sweep_where}}},
        }
        if world == 1 and not args.no_configs and args.model == "DiT-XL/2" and not args.fp8:
            # the other BASELINE configurations, on the same box right after the headline: free the headline's arenas first
            import gc
            runner.sync_sweep()
            eng.close()
            del runner, model, eng, batches, out, mask_arena
            gc.collect()
            torch.cuda.empty_cache()
            res["roofline"]["others"]["configs"] = other_configs(dev, latent)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.model, latent, args.batch, args.cpu_batch)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        wd.stop()
    if check_res is not None and not check_res["ok"]:
        raise SystemExit(f"--check failed: {check_res}")


if __name__ == "__main__":
    main()
