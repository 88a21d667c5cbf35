"""The SFR-on iteration for DiT class-forgetting, fast path (no autograd, no host syncs).

One call = one iteration of /root/reference/DiT/forget.py:256-322 (method "ron"):
  forget: q_sample -> DiT fwd -> fused loss fwd/bwd (alpha * -loss.mean() for "ga") -> DiT bwd ->
          [DP: all-reduce grads] -> mask -> clip_grad_norm_(grad_clip) -> AdamW.step
  remain: the same with +loss.mean(), no mask, no clip -> AdamW.step (same optimizer state)
  EMA over all named parameters (decay 0.9999), fused into the second Adam sweep.
Data-parallel: every rank holds full replicas (weights, Adam state, EMA, mask) and a shard of each
minibatch; gradients are scaled by 1/global_batch in the loss kernel and SUM-all-reduced over RCCL.
"""
import torch
import torch.distributed as dist

from . import _lib, dp, guard, ops, streams, sweep


def build_mask_arena(engine, mask):
    """Reference mask dict (name -> bool tensor, keys optionally 'module.'-prefixed, python int 0 for never-grad
    params; DiT/forget.py:235-237,289-292) -> uint8 arena over the trainable range, loaded ONCE."""
    arena = torch.ones(engine.n_trainable, dtype=torch.uint8, device=engine.device)
    for name, (off, shape, trainable) in engine.index.items():
        if not trainable:
            continue
        m = mask.get(name, mask.get("module." + name))
        if m is None:
            raise KeyError(f"saliency mask has no entry for {name}")
        n = 1
        for s in shape:
            n *= s
        if isinstance(m, int):
            arena[off:off + n] = 1 if m else 0
        else:
            if tuple(m.shape) != tuple(shape):
                raise ValueError(f"mask[{name}] has shape {tuple(m.shape)}, expected {tuple(shape)}")
            arena[off:off + n] = m.reshape(-1).to(device=engine.device, dtype=torch.uint8)
    return arena


class DiTSFRon:
    def __init__(self, model, diffusion, lr=1e-4, forget_alpha=1e-3, grad_clip=1.0, ema_decay=0.9999, mask=None,
                 unlearn_loss="ga", forget_class=0, process_group=None, bucket_bytes=256 << 20, micro_batches=1,
                 overlap_allreduce=False, grad_transport="fp32", method="ron", fp8=False):
        """micro_batches = 2: each forward/backward pass runs as TWO independent half-batch chains on two HIP streams
        (own workspace, own gradient arena, own side stream); the latency-bound kernels of one chain (attention,
        LayerNorm / gate backward) then run under the GEMMs of the other.  The optimizer sweep sums the two arenas."""
        if unlearn_loss not in ("ga", "rl"):
            raise ValueError(f"unsupported unlearn_loss {unlearn_loss!r} (DiT/forget.py defines only 'ga' and 'rl')")
        # method "ron" = the SFR-on iteration (two optimizer steps); "joint" = DiT/forget.py:314-316: ONE step on
        # remain_loss + forget_alpha * forget_loss, no mask, no clip
        if method not in ("ron", "joint"):
            raise ValueError(f"unsupported method {method!r} (DiT/forget.py:283-320 defines 'ron' and 'joint')")
        if method == "joint" and micro_batches != 1:
            raise ValueError("method 'joint' runs single-chain passes (micro_batches = 1)")
        self.method = method
        self._joint = None
        # fp8 (BASELINE config 5): forward GEMMs of the blocks on the fp8 matrix core (engine.enable_fp8); the e4m3 weight shadow
        # is refreshed after every optimizer step
        # the adaLN weight gradient as two factors + a rank-(batch) sweep (see _pass); the engine's (6L+2)D must be a multiple of 8
        self.factored_ada = ((6 * model.engine.cfg.depth + 2) * model.engine.cfg.hidden) % 8 == 0
        self.force_dp = False           # tests: run the world > 1 exchange code at world size 1 (the collectives are identities)
        self._pipeline = None
        self.sweep_beside_forward = True
        self.sweep_beside_wg, self.sweep_beside_head = 256, 2
        # opt-in: the remain-stage sweep of the block ranges runs beside the NEXT step's forget forward pass; step() then returns with it in
        # flight -- read parameters / optimizer state through this runner (state_dict / checkpoint / sync()), not from the raw arenas
        self.sweep_across_steps = False
        # True: a beside-forward sweep is launched BEHIND that pass's conditioning prologue (engine.forward(between=...)).  Measured neutral
        # without a profiler (profiles/r06_ab_log.txt) and it puts 26 Python-issued launches on the critical path: off by default.
        self.defer_sweep_launch = False
        # True: a beside-forward sweep also takes the adaLN matrix (a third of the parameters) to the sweep stream, in front of the block ranges,
        # at full grid; the head ranges, the next pass's noising and its conditioning prologue (~20 small launches) run beside it and the pass
        # waits for it in front of its adaLN product (sfron_dit_forward_phase 3 / 4).  Same kernels on the same operands: bit-identical.
        # One box, alternating, six pairs: -0.17 ... -0.32 ms per step (profiles/r06_ab_log.txt).
        self.ada_side = True
        # single-process runs: the forget stage's clip norm (forget.py:293-298) is taken where the gradients are produced -- the block
        # weight-gradient GEMMs leave the masked sums of squares of their tiles (engine.arm_sumsq), one small launch covers biases /
        # embedders / final layer, the rank-(batch) adaLN range is summed from its factors: no pass over the 1.8 GB block range of the arena
        self.fuse_clip_norm = True
        self._sq_buf = None
        self._ready_owner = None               # (engine, block events) of that sweep, consumed by the next step()'s first forward pass
        self.fp8 = bool(fp8)
        if self.fp8 and model.engine.fp8 is None:
            model.engine.enable_fp8()
        if unlearn_loss == "rl" and (forget_class + 100) % 1000 >= model.num_classes:
            # DiT/forget.py:275-279 hard-codes (forget_class + 100) % 1000; with fewer classes the reference's nn.Embedding
            # raises an IndexError -- so do we, up front
            raise ValueError(f"'rl' relabels to class {(forget_class + 100) % 1000}, outside num_classes = {model.num_classes}")
        self.model, self.diffusion = model, diffusion
        self.forget_alpha, self.grad_clip, self.ema_decay = forget_alpha, grad_clip, ema_decay
        self.unlearn_loss, self.forget_class = unlearn_loss, forget_class
        self.lr = lr
        self.pg = process_group
        self.world = dp.world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        if micro_batches not in (1, 2):
            raise ValueError("micro_batches must be 1 or 2")
        self.micro = micro_batches
        # overlap_allreduce: exchange each block's gradient range while the backward pass of the earlier blocks is still
        # running.  Off unless asked for: bench.py turns it on at N > 1 only after verify_overlap() has shown, on the ranks
        # of that very run, that it reproduces the synchronous bucketed exchange.
        self.overlap = bool(overlap_allreduce)
        # grad_transport "bf16": gradient ranges cross the xGMI links as bf16 (half the bytes; dp.allreduce_); "fp32" = exact sum
        # "auto": by rule -- bf16 from four ranks on (xGMI is point-to-point: a ring all-reduce is per-link bound and the payload grows
        # as 2 (N - 1) / N of the arena; at N >= 4 that is >= 1.5 x 2 x 1.8 GB per step in fp32), the exact fp32 sum below that
        if grad_transport not in ("fp32", "bf16", "auto"):
            raise ValueError("grad_transport must be 'fp32', 'bf16' or 'auto'")
        self.grad_transport = ("bf16" if dp.world_size(process_group) >= 4 else "fp32") if grad_transport == "auto" else grad_transport
        self._tx_scratch = None
        self._chains = None
        self._comm = None
        self._ada_all = None
        self.guard = guard.StepGuard(model.engine.device)
        self.iteration = 0
        self._bind(mask)

    def _bind(self, mask):
        eng = self.model.engine
        nt = eng.n_trainable
        self.mask_arena = build_mask_arena(eng, mask) if mask is not None else None
        self.opt = sweep.FlatAdam(eng.params[:nt], eng.grads[:nt], lr=self.lr, weight_decay=0.0, adamw=True,
                                  mask=self.mask_arena, w_bf16=eng.params_bf16[:nt])          # forget.py:199
        self.ema = eng.params.clone()                                                      # forget.py:190,230

    def _dp_active(self):
        """gradients must be exchanged: more than one rank, or a test forcing the data-parallel code path at world size 1"""
        return (self.world > 1 or self.force_dp) and dist.is_available() and dist.is_initialized()

    def _scratch(self, n):
        """bf16 staging buffer of the gradient transport (None for fp32), at least n elements, one per stream (the overlapped
        exchange stages block ranges on the communication stream while the tail ranges go through the compute stream)."""
        if self.grad_transport == "fp32" or self.world == 1:
            return None
        if self._tx_scratch is None:
            self._tx_scratch = {}
        dev = self.model.engine.device
        key = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        buf = self._tx_scratch.get(key)
        if buf is None or buf.numel() < n:
            buf = self._tx_scratch[key] = torch.empty(n, dtype=torch.bfloat16, device=dev)
        return buf

    def _ar(self, t):
        dp.allreduce_(t, self.pg, self.grad_transport if self.world > 1 else "fp32", self._scratch(t.numel()))

    def _dp_buckets(self):
        """Exchange buckets of the synchronous path in arena order: everything in front of the blocks in pieces of at most
        bucket_elems (the adaLN matrix alone is 892 MB at DiT-XL/2), one bucket per block (64 MB), the tail."""
        eng = self.model.engine
        lay, nt = eng.layout, eng.n_trainable
        b_lo, stride, L = lay["blocks"], lay["blk_stride"], eng.cfg.depth
        out = [(s, min(s + self.bucket_elems, b_lo)) for s in range(0, b_lo, self.bucket_elems)]
        out += [(b_lo + l * stride, b_lo + (l + 1) * stride) for l in range(L)]
        if b_lo + L * stride < nt:
            out.append((b_lo + L * stride, nt))
        return out

    def _exchange_async(self):
        """The synchronous exchange (no per-block events from the backward pass) moved OFF the compute stream: the buckets are
        all-reduced one after another on the communication stream, each followed by an event; the optimizer sweep consumes
        range i when event i has fired (FlatAdam.step(pipeline=...)), so the sweep of the early buckets and -- in the remain
        stage -- their AdamW + EMA run while the later buckets are still on the links.  Same collectives in the same order on
        every rank."""
        if self._comm is None:
            self._comm = streams.get("comm", beside=[torch.cuda.current_stream()] + self.model.engine.side_streams())
        g = self.model.engine.grads
        cur = torch.cuda.current_stream()
        self._comm.wait_stream(cur)
        out = []
        with torch.cuda.stream(self._comm):
            for lo, hi in self._dp_buckets():
                self._ar(g[lo:hi])
                ev = torch.cuda.Event()
                ev.record(self._comm)
                out.append((lo, hi, ev))
        return out

    def _allreduce_grads(self):
        g = self.model.engine.grads[:self.model.engine.n_trainable]
        dp.allreduce_flat_(g, self.bucket_elems, self.pg, transport=self.grad_transport, scratch=self._scratch(min(g.numel(), self.bucket_elems)))

    # ------------------------------------------------------------------ all-reduce overlapped with the backward pass
    def _overlap_enabled(self):
        """The overlapped exchange runs when asked for (``overlap_allreduce`` / ``self.overlap``), single-chain passes only;
        it needs an initialised process group (world size 1 is allowed: the tests run the same code path over RCCL)."""
        return self.overlap and self.micro == 1 and dist.is_available() and dist.is_initialized()

    def verify_overlap(self, batch, y=None, rtol=2e-3):
        """Run one forget-stage forward/backward twice from the same state -- synchronous bucketed all-reduce, then the
        overlapped exchange -- and compare the reduced gradient arenas (they differ by summation order only).  Collective:
        every rank calls it; returns the same verdict on every rank.  Leaves parameters and optimizer state untouched."""
        if not (dist.is_available() and dist.is_initialized()) or self.micro != 1:
            return False
        eng = self.model.engine
        nt = eng.n_trainable
        y = batch["y"] if y is None else y
        keep, keep_tx = self.overlap, self.grad_transport
        ok = torch.zeros((), dtype=torch.float32, device=eng.device)
        try:
            # compare the two exchange PATHS on the exact fp32 sum: with the bf16 transport (the rule from four ranks on) the synchronous pass
            # would sum the adaLN third of the arena in bf16 while the overlapped pass forms it as an fp32 product of gathered factors -- a
            # 1-2e-3 difference of the transports, not of the paths, right at rtol
            self.grad_transport = "fp32"
            self.overlap = False
            self._pass(batch, y, -self.forget_alpha)
            g_sync = eng.grads[:nt].clone()
            self.overlap = True
            self._pass(batch, y, -self.forget_alpha)
            diff = (eng.grads[:nt] - g_sync).norm() / (g_sync.norm() + 1e-30)
            ok = (torch.isfinite(diff) & (diff < rtol)).to(torch.float32)
        except Exception as e:
            # The two passes issue collectives themselves (bucketed all-reduces, the per-block exchange): a rank that fails part-way
            # cannot rejoin its peers at the vote below -- its MIN all-reduce would pair with a gradient bucket on the other ranks.  So
            # a failure here ends THIS rank non-zero (the launcher tears the job down; the peers' watchdog, dp.Watchdog, ends them
            # with exit 124 if they sit in a collective).  Only the message is added here.
            import sys
            print(f"[sfron] verify_overlap failed on this rank ({type(e).__name__}: {e}); leaving the job", file=sys.stderr, flush=True)
            raise
        finally:
            self.overlap, self.grad_transport = keep, keep_tx
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(ok.item())

    def _backward_allreduce_overlapped(self, d_out, y, drop, factored_ada=False):
        """Backward + gradient exchange.  The library records one event per block on its weight-gradient stream when that
        block's arena range is final (all but proj.bias / fc2.bias, which it parks in ``late_bias``); a communication stream
        waits for each event and SUM-all-reduces the block's range (64 MB for DiT-XL/2) while the backward pass of the earlier
        blocks is still running.  What is only known at the end -- adaLN / embedder / final-layer gradients and the late
        biases -- is exchanged after the pass, then the reduced biases are scattered into the arena.  Every rank issues the
        same collectives in the same order."""
        eng = self.model.engine
        evs = eng.dp_setup()
        if self._comm is None:
            self._comm = streams.get("comm", beside=[torch.cuda.current_stream()] + self.model.engine.side_streams())
        main = torch.cuda.current_stream()
        eng.backward_dp(d_out, y, drop)
        for l in reversed(range(len(evs))):
            lo, hi = eng.block_ranges[l]
            self._comm.wait_event(evs[l])
            with torch.cuda.stream(self._comm):
                self._ar(eng.grads[lo:hi])
        nt = eng.n_trainable
        lay = eng.layout
        b_lo, b_hi = eng.block_ranges[0][0], eng.block_ranges[-1][1]
        # adaLN_modulation weight gradient (a third of the arena, final only in the tail of the pass): instead of all-reducing
        # 892 MB, all-gather its two bf16 factors (dmod [B][(6L+2)D], silu(c) [B][D]: 12.5 MB per rank) and form the product
        # over the GLOBAL batch with one GEMM -- the same sum, taken in one place
        B, NM, D = eng.cfg.batch, eng.ada_dmod.shape[1], eng.cfg.hidden
        key = (self.world, B, NM, D)
        if self._ada_all is None or self._ada_all[0] != key:       # re-made when the per-GPU batch changes (set_batch_size)
            self._ada_all = (key, torch.empty(self.world * B, NM, dtype=torch.bfloat16, device=eng.device),
                             torch.empty(self.world * B, D, dtype=torch.bfloat16, device=eng.device))
        _, dmod_all, sc_all = self._ada_all
        dist.all_gather_into_tensor(dmod_all, eng.ada_dmod, group=self.pg)
        dist.all_gather_into_tensor(sc_all, eng.ada_sc, group=self.pg)
        if factored_ada:
            # ... and not even the product is formed: the optimizer sweep takes the gathered factors as a rank-(world x batch) gradient
            # (csrc/sweep.hip k_adam_lowrank / the EPI_SUMSQ norm pre-pass), exactly as the single-process step does with its own batch
            self.opt.lowrank = dict(lo=lay["ada_w"], NM=NM, D=D, dmod=dmod_all, sc=sc_all, R=self.world * B)
        else:
            ada_w = eng.grads[lay["ada_w"]:lay["ada_w"] + NM * D].view(NM, D)
            ops.gemm(dmod_all, sc_all, NM, D, self.world * B, a_t=True, b_t=True, epilogue=_lib.EPI_F32, c_f32=ada_w)
        # everything else outside the blocks (embedders, label table, adaLN bias, final layer) and the late biases
        self._ar(eng.grads[:lay["ada_w"]])
        self._ar(eng.grads[lay["ada_b"]:b_lo])
        if b_hi < nt:
            self._ar(eng.grads[b_hi:nt])
        self._ar(eng.late_bias.view(-1))
        main.wait_stream(self._comm)             # every block range is reduced: now the reduced late biases may land in it
        eng.scatter_late_bias()

    # ------------------------------------------------------------------ two-chain (micro-batch) pass
    def _setup_chains(self, n_local):
        half = n_local // 2
        self.model.set_batch_size(half)
        e0 = self.model.engine
        e1 = e0.sibling(half)
        self._chains = (e0, e1, streams.get("chain0"), streams.get("chain1"))
        self.opt.g = e0.grads[:e0.n_trainable]
        self.opt.g2 = e1.grads[:e1.n_trainable]

    def _pass2(self, batch, y, sign_alpha):
        n_local = batch["x0"].shape[0]
        if n_local % 2:
            raise ValueError("micro_batches=2 needs an even per-GPU batch")
        if self._chains is None or self._chains[0].cfg.batch != n_local // 2:
            self._setup_chains(n_local)
        e0, e1, s0, s1 = self._chains
        diff = self.diffusion
        n_global = n_local * self.world
        half = n_local // 2
        cur = torch.cuda.current_stream()
        outs = []
        for eng, st, sl in ((e0, s0, slice(0, half)), (e1, s1, slice(half, n_local))):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                x0, t, noise, yy = batch["x0"][sl], batch["t"][sl], batch["noise"][sl], y[sl]
                drop = batch["drop"][sl] if batch.get("drop") is not None else None
                x_t = diff.q_sample(x0, t, noise)
                out = eng.forward(x_t, t, yy, drop)
                mse, vb, d_out = diff.loss_fwd_bwd(out, x0, t, noise, sign_alpha / n_global)
                eng.backward(d_out, yy, drop)
                outs.append((mse, vb))
        cur.wait_stream(s0)
        cur.wait_stream(s1)
        if self.world > 1:                       # one arena to exchange: fold the second chain's gradients in
            nt = e0.n_trainable
            e0.grads[:nt].add_(e1.grads[:nt])
            self.opt.g2 = None
            dp.allreduce_flat_(e0.grads[:nt], self.bucket_elems, self.pg, transport=self.grad_transport,
                               scratch=self._scratch(min(nt, self.bucket_elems)))
        else:
            self.opt.g2 = e1.grads[:e1.n_trainable]
        return torch.cat([outs[0][0], outs[1][0]]), torch.cat([outs[0][1], outs[1][1]])

    def _checked(self, batch, y):
        """Inputs of one pass with the label-dropout draw filled in and out-of-range indices made harmless + flagged:
        the reference's nn.Embedding / table gather raise an IndexError on a bad label or timestep (DiT/models.py:89-93,
        gaussian_diffusion.py:861-873); here the kernels see clamped indices (never an out-of-bounds access) and the guard
        raises SfronError at its next poll."""
        b = dict(batch)
        if b.get("drop") is None:
            # the reference trains both stages under model.train(): LabelEmbedder.token_drop with p = 0.1 (models.py:78-87)
            b["drop"] = self.model._draw_drop(b["x0"].shape[0])
        y_safe, b["t"] = self.guard.check_inputs(y, b["t"], self.model.num_classes, self.diffusion.num_timesteps)
        return b, y_safe

    def _pass(self, batch, y, sign_alpha, factored_ada=False, block_ready=None, async_exchange=False, between=None, ada_ready=None):
        """factored_ada (single process, single chain -- what step() uses): the backward pass leaves the adaLN_modulation weight
        gradient as its two bf16 factors and the next optimizer sweep forms the rank-(batch) product itself (engine.
        backward_factored_ada, csrc/sweep.hip k_adam_lowrank): 892 MB less to write and 892 MB (forget stage: twice) less to read
        per pass at DiT-XL/2.  Everything else (tests, verify_overlap, bench --check, data-parallel runs) gets the full arena."""
        batch, y = self._checked(batch, y)
        if self.micro == 2:
            return self._pass2(batch, y, sign_alpha)
        eng, diff = self.model.engine, self.diffusion
        n_global = batch["x0"].shape[0] * self.world
        x_t = diff.q_sample(batch["x0"], batch["t"], batch["noise"])
        out = eng.forward(x_t, batch["t"], y, batch.get("drop"), block_ready=block_ready, between=between if block_ready is not None else None,
                          ada_ready=ada_ready if block_ready is not None else None)
        mse, vb, d_out = diff.loss_fwd_bwd(out, batch["x0"], batch["t"], batch["noise"], sign_alpha / n_global)
        if self._overlap_enabled():
            self._backward_allreduce_overlapped(d_out, y, batch.get("drop"), factored_ada=factored_ada and self.factored_ada)
        elif factored_ada and not self._dp_active() and self.factored_ada:
            self.opt.lowrank = eng.backward_factored_ada(d_out, y, batch.get("drop"))
        elif factored_ada and self.factored_ada and not async_exchange:
            # data parallel, synchronous exchange: the factors are all-gathered (12.5 MB + 72 KB per rank at DiT-XL/2 instead of 892 MB
            # all-reduced), every other range of the arena is all-reduced in buckets
            lr = eng.backward_factored_ada(d_out, y, batch.get("drop"))
            B, NM, D = lr["R"], lr["NM"], lr["D"]
            key = (self.world, B, NM, D)
            if self._ada_all is None or self._ada_all[0] != key:
                self._ada_all = (key, torch.empty(self.world * B, NM, dtype=torch.bfloat16, device=eng.device),
                                 torch.empty(self.world * B, D, dtype=torch.bfloat16, device=eng.device))
            _, dmod_all, sc_all = self._ada_all
            dist.all_gather_into_tensor(dmod_all, lr["dmod"], group=self.pg)
            dist.all_gather_into_tensor(sc_all, lr["sc"], group=self.pg)
            self.opt.lowrank = dict(lo=lr["lo"], NM=NM, D=D, dmod=dmod_all, sc=sc_all, R=self.world * B)
            nt, lo, hi = eng.n_trainable, lr["lo"], lr["lo"] + NM * D
            dp.allreduce_ranges_(eng.grads, [(0, lo), (hi, nt)], self.bucket_elems, self.pg, transport=self.grad_transport if self.world > 1 else "fp32",
                                 scratch=self._scratch(self.bucket_elems))
        else:
            eng.backward(d_out, y, batch.get("drop"))
            if async_exchange and self._dp_active():
                self._pipeline = self._exchange_async()          # consumed by the optimizer step that follows in step()
            else:
                self._allreduce_grads()
        return mse, vb

    def _fp8_before_sweep(self, fused_q):
        """config 5 with re-quantising sweeps: every `refresh_every` sweeps the per-tensor scales are re-derived from the current
        masters' amax (weights move by at most lr per sweep; the scale keeps 2x headroom and the conversion saturates)"""
        if not fused_q:
            return
        f8 = self.model.engine.fp8
        if f8["sweeps"] % f8["refresh_every"] == 0 and f8["sweeps"] > 0:
            self.model.engine.fp8_refresh_scales()
        f8["sweeps"] += 1

    def _step_joint(self, forget, remain, y_f, sign):
        """DiT/forget.py:314-320 with method "joint": loss = remain_loss + forget_alpha * forget_loss, one backward through BOTH
        graphs, one AdamW step (no mask, no clip), EMA.  Both forward passes see the same weights; their activations live in two
        workspaces (a sibling engine over the same parameters) and their gradients in two arenas that the sweep adds."""
        n = forget["x0"].shape[0]
        if remain["x0"].shape[0] != n:
            raise ValueError("method 'joint' needs equal forget / remain batch sizes per rank")
        self.model.set_batch_size(n)
        e0 = self.model.engine
        if self._joint is None or self._joint[0] is not e0:
            self._joint = (e0, e0.sibling(n))
        e1 = self._joint[1]
        diff = self.diffusion
        n_global = n * self.world
        res = []
        for eng, b, y, scale in ((e0, forget, y_f, sign * self.forget_alpha), (e1, remain, remain["y"], 1.0)):
            b, y = self._checked(b, y)
            x_t = diff.q_sample(b["x0"], b["t"], b["noise"])
            out = eng.forward(x_t, b["t"], y, b.get("drop"))
            mse, vb, d_out = diff.loss_fwd_bwd(out, b["x0"], b["t"], b["noise"], scale / n_global)
            eng.backward(d_out, y, b.get("drop"))
            res.append((mse, vb))
        nt = e0.n_trainable
        if self.world > 1:
            e0.grads[:nt].add_(e1.grads[:nt])
            self.opt.g, self.opt.g2 = e0.grads[:nt], None
            self._allreduce_grads()
        else:
            self.opt.g, self.opt.g2 = e0.grads[:nt], e1.grads[:nt]
        self.opt.step(max_norm=None, use_mask=False, ema=self.ema[:nt], ema_decay=self.ema_decay, ema_mode=1)      # forget.py:320,322
        self.opt.g2 = None
        if self.fp8:
            e0.fp8_requantize()
        if e0.n_total > nt:
            sweep.ema_update(self.ema[nt:], e0.params[nt:], self.ema_decay, mode=1)
        (mse_f, vb_f), (mse_r, vb_r) = res
        self.guard.check_finite((mse_f, vb_f, mse_r, vb_r), None)
        self.iteration += 1
        self.guard.publish(self.iteration)
        return {"forget_mse": mse_f, "forget_vb": vb_f, "remain_mse": mse_r, "remain_vb": vb_r, "forget_sign": sign, "stats": self.opt.stats}

    def step(self, forget, remain):
        """forget / remain: dicts of GPU tensors x0 [N,4,S,S] fp32, y [N] int64, t [N] int64, noise, drop [N] uint8
        (this rank's shard).  Returns per-sample mse / vb tensors (device; no host sync here)."""
        self.guard.poll()                      # raises on what an EARLIER step found (no synchronisation)
        if self.micro == 1:
            self.model.set_batch_size(forget["x0"].shape[0])
        eng = self.model.engine
        if self.unlearn_loss == "ga":
            y_f, sign = forget["y"], -1.0                                                   # forget.py:269-272
        else:
            y_f, sign = torch.full_like(forget["y"], (self.forget_class + 100) % 1000), 1.0  # forget.py:274-282
        if self.method == "joint":
            return self._step_joint(forget, remain, y_f, sign)
        # synchronous data-parallel exchange: pipelined bucket by bucket into the sweep (FlatAdam.step(pipeline=...)) when the adaLN
        # gradient is exchanged as part of the arena; with the factored form (the default) the pass exchanges everything itself
        dp_sync = self._dp_active() and not self._overlap_enabled() and self.micro == 1 and not self.factored_ada
        ready = None
        if self._ready_owner is not None:
            # the events of a sweep left in flight belong to the engine that armed it.  A batch-size change (set_batch_size above, or
            # by the user between two steps) replaced that engine: it was drained and closed -- its events are destroyed -- so there
            # is nothing to wait for and the stale handles must not reach hipStreamWaitEvent
            owner, handles = self._ready_owner
            self._ready_owner = None
            if owner is eng and getattr(eng, "_sweep_pending", None) is not None:
                ready = handles
        between_f = None
        ada_f = eng._shared.pop("ada_done", None)              # the adaLN sweep the last step() left on the sweep stream (ada_side)
        if ready is None:
            ada_f = None                       # nothing in flight that this pass may run beside (a drain orders it behind the whole stream)
        if ready is not None:
            eng._sweep_pending = None          # this forward pass waits for the sweep block by block; its backward pass starts behind all of it
            between_f = eng._shared.pop("deferred", None)      # ... and launches it itself, behind its conditioning prologue (see below)
        sq_plan = None
        if (self.fuse_clip_norm and self.grad_clip is not None and self.micro == 1 and self.factored_ada and not self._dp_active()):
            sq_plan = eng.fused_sumsq_plan()
        if sq_plan is not None:
            need = sq_plan["n_gemm"] + sq_plan["n_ranges"] + ((6 * eng.cfg.depth + 2) * eng.cfg.hidden) // 8
            if self._sq_buf is None or self._sq_buf.numel() < need:
                self._sq_buf = torch.empty(need, dtype=torch.float64, device=eng.device)
            eng.arm_sumsq(self.opt.mask, self._sq_buf[:sq_plan["n_gemm"]])        # consumed by the forget pass's backward
        try:
            mse_f, vb_f = self._pass(forget, y_f, sign * self.forget_alpha, factored_ada=True, block_ready=ready, async_exchange=dp_sync,
                                     between=between_f, ada_ready=ada_f)
        except BaseException:
            if sq_plan is not None:
                eng.disarm_sumsq()           # the one-shot was not consumed: no later backward may write through it
            raise
        if sq_plan is not None and self.opt.lowrank is not None:
            self.opt.fused_sumsq = dict(partials=self._sq_buf[:sq_plan["n_gemm"]], buffer=self._sq_buf, ranges=sq_plan["ranges"],
                                        n_ranges=sq_plan["n_ranges"])
        # The forget-stage AdamW of the 28 block ranges runs on a second stream, on a bounded grid, BESIDE the remain forward pass,
        # which waits for block l's event when it reaches block l; embedders / adaLN / final layer (needed at once) stay on this
        # stream.  Measured (tools/bench_sweep_beside.py): a full-grid sweep beside the GEMM chain gains nothing, one capped at
        # 256 workgroups hides ~0.8 ms of its 2.6 ms.  Single-chain, single-process, bf16 passes only.
        split = None
        # config 5: the sweeps write the e4m3 shadow themselves -- except under the pipelined synchronous exchange (dp_sync), whose
        # sweep consumes buckets, not block ranges: there fp8_requantize() follows each sweep
        fused_q = self.fp8 and self.micro == 1 and not dp_sync
        quant = None
        if fused_q:
            f8 = self.model.engine.fp8
            quant = dict(tensors=f8["tensors"], w8=f8["w8"], scales=f8["scales"])
        # (data-parallel runs too: the gradients a sweep reads are final -- the pass has waited for its exchange -- so an N-rank step is
        # the single-process kernel sequence plus the collectives)
        if self.micro == 1 and dp_sync is False and (self.sweep_beside_forward or fused_q):
            bs = self.model.engine.block_sweep_setup()
            # defer: the ranges that go to the second stream are LAUNCHED by the forward pass they run beside, between its conditioning prologue
            # (ten small dependent launches in front of block 0) and block 0 -- beside a bandwidth-heavy sweep each boundary of that chain
            # costs 60-100 us instead of ~5 (profiles/r06_stage_boundary.txt: ~0.5 ms per pass); the blocks wait for their events as before
            split = dict(ranges=bs["ranges"], stream=bs["stream"] if self.sweep_beside_forward else None, events=bs["events"],
                         max_workgroups=self.sweep_beside_wg, head=self.sweep_beside_head, quant=quant, defer=self.defer_sweep_launch,
                         ada_side=self.ada_side)
        self._fp8_before_sweep(fused_q)
        pipe, self._pipeline = self._pipeline, None
        self.opt.step(max_norm=self.grad_clip, use_mask=True, split=split, pipeline=pipe)   # forget.py:289-299
        if self.fp8 and not fused_q:
            self.model.engine.fp8_requantize()
        beside = split is not None and split["stream"] is not None
        mse_r, vb_r = self._pass(remain, remain["y"], 1.0, factored_ada=True, block_ready=bs["handles"] if beside else None,
                                 async_exchange=dp_sync, between=self.opt.take_deferred(),
                                 ada_ready=split.get("ada_done") if beside else None)
        pipe, self._pipeline = self._pipeline, None
        nt = eng.n_trainable
        self._fp8_before_sweep(fused_q)
        # The remain-stage sweep of the block ranges goes the same way, beside the forget forward pass of the NEXT step() (same stream,
        # same events: the remain forward pass above has consumed them).  step() then returns with that sweep in flight: the next
        # step's forward pass waits block by block, every other reader of the state drains it first (engine.drain_sweep).
        across = beside and self.sweep_across_steps
        if across:
            split_r = dict(ranges=bs["ranges"], stream=bs["stream"], events=bs["events"], max_workgroups=self.sweep_beside_wg,
                           head=self.sweep_beside_head, quant=quant, defer=self.defer_sweep_launch, ada_side=self.ada_side)
        else:
            split_r = dict(ranges=bs["ranges"], stream=None, quant=quant) if fused_q else None
        self.opt.step(max_norm=None, use_mask=False, ema=self.ema[:nt], ema_decay=self.ema_decay, ema_mode=1,   # :320,322
                      split=split_r, pipeline=pipe)
        if across:
            self._ready_owner, eng._sweep_pending = (eng, bs["handles"]), bs["stream"]
            if split_r.get("ada_done") is not None:
                eng._shared["ada_done"] = split_r["ada_done"]
            fn = self.opt.take_deferred()
            if fn is not None:
                eng._shared["deferred"] = fn       # launched by the next step's forward pass (or by drain_sweep, whichever comes first)
        if self.fp8 and not fused_q:
            self.model.engine.fp8_requantize()
        if eng.n_total > nt:
            sweep.ema_update(self.ema[nt:], eng.params[nt:], self.ema_decay, mode=1)         # frozen pos_embed (:60-62)
        # fail loud: a NaN / Inf loss or gradient norm would otherwise poison every weight through
        # the clip coefficient; counted on the device here, raised by the next poll()
        self.guard.check_finite((mse_f, vb_f, mse_r, vb_r), self.opt.stats)
        self.iteration += 1
        self.guard.publish(self.iteration)
        return {"forget_mse": mse_f, "forget_vb": vb_f, "remain_mse": mse_r, "remain_vb": vb_r,
                "forget_sign": sign, "stats": self.opt.stats}

    def ema_state_dict(self):
        self.sync_sweep()
        eng = self.model.engine
        return {name: eng.view(self.ema, name).clone() for name in eng.index}

    def sync_sweep(self):
        """Order the current stream behind a block sweep that step() left in flight (sweep_across_steps)."""
        self.model.engine.drain_sweep()
        self._ready_owner = None

    # ------------------------------------------------------------------ checkpoint (DiT/forget.py:346-353 format)
    def opt_state_dict(self):
        """The shared AdamW state in torch.optim's state_dict layout over ``model.parameters()`` (forget.py:199): entries
        only for parameters that received gradients (pos_embed has none), so the reference's
        ``torch.optim.AdamW(model.parameters(), ...).load_state_dict(ckpt["opt"])`` accepts it."""
        self.sync_sweep()
        eng = self.model.engine
        names = [n for n, _ in self.model.named_parameters()]
        state = {}
        for i, n in enumerate(names):
            off, shape, trainable = eng.index[n]
            if trainable and self.opt.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.opt.step_count)),
                            "exp_avg": eng.view(self.opt.m, n).clone(), "exp_avg_sq": eng.view(self.opt.v, n).clone()}
        group = {"lr": self.lr, "betas": tuple(self.opt.betas), "eps": self.opt.eps, "weight_decay": self.opt.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def checkpoint(self, args=None, data_parallel_prefix=False):
        """{"model", "ema", "opt", "args"} as DiT/forget.py:346-353 saves it (pass the dict to torch.save).  The reference
        saves ``model.state_dict()`` of its nn.DataParallel wrapper (forget.py:193,347), whose keys carry a "module."
        prefix; ``data_parallel_prefix=True`` writes them that way ("ema" is the unwrapped copy there too: no prefix)."""
        pre = "module." if data_parallel_prefix else ""
        self.sync_sweep()
        return {"model": {pre + k: v.clone() for k, v in self.model.state_dict().items()}, "ema": self.ema_state_dict(),
                "opt": self.opt_state_dict(), "args": args}

    def load_checkpoint(self, ckpt):
        """Resume from a checkpoint in that format: ours, or one written by the reference, whose "model" keys carry the
        nn.DataParallel "module." prefix (DiT.load_state_dict strips it)."""
        self.sync_sweep()
        eng = self.model.engine
        self.model.load_state_dict(ckpt["model"])
        for name, v in ckpt["ema"].items():
            name = name[len("module."):] if name.startswith("module.") else name
            eng.view(self.ema, name).copy_(v.to(self.ema.device))
        names = [n for n, _ in self.model.named_parameters()]
        self.opt.m.zero_(); self.opt.v.zero_()
        steps = set()
        for i, st in ckpt["opt"]["state"].items():
            n = names[int(i)]
            eng.view(self.opt.m, n).copy_(st["exp_avg"].to(self.ema.device))
            eng.view(self.opt.v, n).copy_(st["exp_avg_sq"].to(self.ema.device))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not a state this optimizer can hold")
        self.opt.step_count = steps.pop() if steps else 0
        eng.sync_bf16()
