"""Host side of the parameter sweep (csrc/sweep.hip): mask -> clip -> Adam(W) -> EMA on flat arenas.

Mirrors what the reference does per tensor in DiT/forget.py:289-299,320,322 and
DDPM/runners/diffusion.py:1126-1138,1169-1180, as two streaming launches per stage.
"""
import ctypes
import math

import torch

from . import _lib, streams
from ._lib import check, ptr, stream_ptr


class FlatAdam:
    """Adam/AdamW state over one flat fp32 parameter arena, stepped by the HIP sweep kernel.

    One optimizer, several ``step()`` per iteration on shared state (DiT/forget.py:199,299,320):
    ``step_count`` advances once per call, exactly like torch.optim's per-parameter ``step``.
    """

    def __init__(self, params, grads, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=True,
                 mask=None, w_bf16=None, ranges=None):
        assert params.dtype == torch.float32 and params.dim() == 1 and params.is_cuda
        assert grads.shape == params.shape
        if weight_decay != 0.0 and not adamw:
            raise NotImplementedError("coupled (L2) weight decay is not used by the reference configs (wd=0)")
        self.p, self.g = params, grads
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.mask = mask            # uint8/bool [n] or None, device resident (loaded once, not per step)
        self.w_bf16 = w_bf16        # bf16 [n] shadow or None
        self.step_count = 0
        # optional [(lo, hi)] element ranges (multiples of 8) that hold every parameter the optimizer owns: the sweep then streams
        # only those (SD train_method "xattn": 44 M of 860 M parameters); unclipped steps only
        self.ranges = ranges
        self.g2 = None              # optional second gradient arena (micro-batch chains), summed inside the kernels
        # optional rank-R gradient of one weight matrix inside the arena (set per step by the caller, cleared by step()):
        # dict(lo=element offset of W [NM][D], NM=, D=, dmod=bf16 [R][NM], sc=bf16 [R][D], R=) -- the gradient arena is NOT read
        # over [lo, lo + NM * D): the sweep forms dmod^T sc itself (csrc/sweep.hip k_adam_lowrank)
        self.lowrank = None
        # the clip norm's share of the adaLN matrix does not depend on the end of the backward pass: it starts early, on a stream of its own
        # (_early_stream).  NOT the remain-stage sweep of that matrix: started there it stretches the embedders' backward (25 small dependent
        # launches under a bandwidth-saturating kernel) and the step measures 1.8 ms slower (profiles/r05_ab_log.txt)
        self.early_ada, self._ada_stream = True, None
        # optional (set per step by the caller, cleared by step()): the masked sums of squares of PART of the arena were already written by
        # the kernels that produced those gradients -- dict(partials=fp64 tensor [n_gemm] (filled on this stream before step() is called),
        # ranges=int64 device table [n_ranges][2] of the element ranges they do NOT cover, n_ranges=) -- engine.fused_sumsq_plan().  With
        # `lowrank` set as well the norm pre-pass then reads no gradient of the covered matrices at all.
        self.fused_sumsq = None
        self.timed = None           # bench.py: a list that receives a (start, end) torch event pair per sweep launch
        L = _lib.lib()
        self._partials = torch.empty(L.sfron_sweep_partials_len(), dtype=torch.float64, device=params.device)
        self.stats = torch.zeros(4, dtype=torch.float32, device=params.device)

    def take_deferred(self):
        """The side-stream part of the last step(split=dict(defer=True, ...)), as a callable to run once (None if nothing was deferred)."""
        fn, self.deferred = getattr(self, "deferred", None), None
        return fn

    def _segments(self, exclude=None):
        """[(lo, hi, lowrank?)] covering the arena (minus ``exclude`` = (lo, hi), which must not cut the low-rank matrix): one flat
        segment, or flat / rank-R / flat around the low-rank matrix."""
        n = self.p.numel()
        cuts = [(0, n)]
        if exclude is not None:
            cuts = [(0, exclude[0]), (exclude[1], n)]
        lr = self.lowrank
        out = []
        for a, b in cuts:
            if lr is not None and a <= lr["lo"] and lr["lo"] + lr["NM"] * lr["D"] <= b:
                lo, hi = lr["lo"], lr["lo"] + lr["NM"] * lr["D"]
                assert lo % 8 == 0 and hi % 8 == 0 and self.g2 is None
                out += [(a, lo, False), (lo, hi, True), (hi, b, False)]
            else:
                assert lr is None or b <= lr["lo"] or a >= lr["lo"] + lr["NM"] * lr["D"]
                out.append((a, b, False))
        return [sg for sg in out if sg[1] > sg[0]]

    def _early_stream(self, q):
        """The stream for work on the adaLN matrix that may start before the backward pass has ended (``q`` = the low-rank description with the
        engine's ``wait`` hook, single-process passes only), ordered behind that point; None = no such hook: the current stream."""
        wait = None if q is None else (q.get("wait_factors") or q.get("wait"))      # the norm reads the two factors only
        if not self.early_ada or wait is None:
            return None
        if self._ada_stream is None:
            import torch as _t
            # process-wide, probed to run beside the caller's stream and the engine's weight-gradient streams: see streams.py
            self._ada_stream = streams.get("ada", self.p.device, beside=[_t.cuda.current_stream()] + list(q.get("beside", ())))
        wait(self._ada_stream)
        return self._ada_stream

    def _join(self, side):
        ev = torch.cuda.Event()
        ev.record(side)
        torch.cuda.current_stream().wait_event(ev)

    def grad_norm_clip_coef(self, max_norm, use_mask):
        """Launch the norm pre-pass; leaves (norm, coef, sumsq) in self.stats on device (no host sync)."""
        L = _lib.lib()
        mask = self.mask if use_mask else None
        s = stream_ptr()
        fs = self.fused_sumsq
        if fs is not None:
            # [GEMM-written partials | ranges | rank-R product]: the first block was filled by the backward pass
            assert self.g2 is None and self.lowrank is not None, "fused sums of squares: single-chain pass with the factored adaLN gradient"
            q, n_gemm, n_rng = self.lowrank, fs["partials"].numel(), fs["n_ranges"]
            buf = fs["buffer"]
            assert buf.numel() >= n_gemm + n_rng + q["NM"] // 8 and fs["partials"].data_ptr() == buf.data_ptr()
            check(L.sfron_sumsq_masked_ranges(ptr(self.g), ptr(mask), ptr(fs["ranges"]), n_rng, ptr(buf[n_gemm:]), s), "sumsq_masked_ranges")
            nblk = ctypes.c_int(0)
            lo, hi = q["lo"], q["lo"] + q["NM"] * q["D"]
            side = self._early_stream(q)
            # the adaLN matrix's share (a rank-(batch) product on the matrix core, 0.17 ms at DiT-XL/2) beside the embedders' backward, which is
            # still running on this stream: it needs the two factors only (engine.ada_wait)
            check(L.sfron_sumsq_lowrank(ptr(q["dmod"]), ptr(q["sc"]), q["R"], q["NM"], q["D"], ptr(None if mask is None else mask[lo:hi]),
                                        ptr(buf[n_gemm + n_rng:]), ctypes.byref(nblk), s if side is None else ctypes.c_void_p(side.cuda_stream)),
                  "sumsq_lowrank")
            if side is not None:
                self._join(side)
            check(L.sfron_clip_coef(ptr(buf), n_gemm + n_rng + nblk.value, float(max_norm), ptr(self.stats), s), "clip_coef")
            return
        segs = self._segments()
        need = sum((self.lowrank["NM"] // 8) if lr else L.sfron_sweep_partials_len() for _, _, lr in segs)
        if self._partials.numel() < need:
            self._partials = torch.empty(need, dtype=torch.float64, device=self.p.device)
        used = 0
        sl = lambda t, lo, hi: None if t is None else t[lo:hi]
        for lo, hi, lr in segs:
            nblk = ctypes.c_int(0)
            if lr:
                q = self.lowrank
                check(L.sfron_sumsq_lowrank(ptr(q["dmod"]), ptr(q["sc"]), q["R"], q["NM"], q["D"], ptr(sl(mask, lo, hi)),
                                            ptr(self._partials[used:]), ctypes.byref(nblk), s), "sumsq_lowrank")
            else:
                check(L.sfron_sumsq_masked(ptr(self.g[lo:hi]), ptr(sl(self.g2, lo, hi)), ptr(sl(mask, lo, hi)), hi - lo,
                                           ptr(self._partials[used:]), ctypes.byref(nblk), s), "sumsq_masked")
            used += nblk.value
        check(L.sfron_clip_coef(ptr(self._partials), used, float(max_norm), ptr(self.stats), s), "clip_coef")

    def step(self, max_norm=None, use_mask=False, ema=None, ema_decay=0.0, ema_mode=0, split=None, pipeline=None):
        """One optimizer step on the current grads.  max_norm=None -> no clipping (DiT remain stage).
        split (optional): dict(ranges=[(lo, hi)] consecutive element ranges (the DiT blocks, in forward order), stream=, events=[one
        torch event per range], max_workgroups=): those ranges are swept on ``stream`` (bounded grid), one event each, while the
        rest of the arena is swept on the current stream -- the next forward pass waits for block l's event only when it
        reaches block l (engine.forward(block_ready=...)).
        pipeline (optional, data-parallel runs): [(lo, hi, event)] consecutive ranges covering the arena whose gradients become final
        (all-reduced on a communication stream) when their event fires: the norm pre-pass / the unclipped update of range i runs
        on the current stream as soon as event i has fired, while the later ranges are still being exchanged."""
        L = _lib.lib()
        if use_mask and self.mask is None:
            use_mask = False
        if pipeline is not None:
            assert split is None and self.lowrank is None and self.g2 is None and self.ranges is None
            return self._step_pipelined(pipeline, max_norm, use_mask, ema, ema_decay, ema_mode)
        if max_norm is not None:
            self.grad_norm_clip_coef(max_norm, use_mask)
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        decay_mul = 1.0 - self.lr * self.wd
        ev = None
        if self.timed is not None:
            import torch as _t
            ev = (_t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.ranges is not None and max_norm is None and self.g2 is None:
            sl = lambda t, lo, hi: None if t is None else t[lo:hi]
            for lo, hi in self.ranges:
                check(L.sfron_masked_clip_adam(ptr(self.p[lo:hi]), ptr(self.g[lo:hi]), None, ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                               ptr(sl(self.mask if use_mask else None, lo, hi)), None, hi - lo, b1, b2, self.eps, step_size,
                                               bc2_sqrt, decay_mul, ptr(sl(self.w_bf16, lo, hi)), ptr(sl(ema, lo, hi)), float(ema_decay),
                                               int(ema_mode if ema is not None else 0), stream_ptr()), "masked_clip_adam")
        else:
            sl = lambda t, lo, hi: None if t is None else t[lo:hi]
            mask = self.mask if use_mask else None
            stats = self.stats if max_norm is not None else None
            emode = int(ema_mode if ema is not None else 0)
            excl = None
            quant = None
            if split is not None:
                import torch as _t
                rngs, quant = split["ranges"], split.get("quant")
                side = split.get("stream")
                # the first `head` ranges stay on the current stream (the next forward pass needs them first); with `quant` (config 5)
                # every range goes through the per-tensor launches below, without a second stream all of them on this one
                head = int(split.get("head", 0)) if side is not None else len(rngs)
                first_own = 0 if quant is not None else head
                excl = (rngs[first_own][0], rngs[-1][1]) if first_own < len(rngs) else None
            # ada_side: the adaLN matrix (a third of the parameters, ~1.1 ms at the HBM rate) is swept on the second stream, in FRONT of the block
            # ranges that go there, at full grid: what this stream still holds -- the head ranges, then the next pass's noising and conditioning
            # prologue, ~20 small launches -- runs beside it; the pass waits for split["ada_done"] in front of its adaLN product (engine.forward)
            ada_side = (split is not None and excl is not None and bool(split.get("ada_side")) and split.get("stream") is not None
                        and head < len(rngs))
            if split is not None:
                split["ada_done"] = None
            for lo, hi, lr in self._segments(excl):
                if lr:
                    q = self.lowrank
                    sp = stream_ptr()
                    if ada_side:
                        fork = _t.cuda.Event()
                        fork.record(_t.cuda.current_stream())       # behind the clip coefficient and everything that read the old matrix
                        split["stream"].wait_event(fork)
                        sp = ctypes.c_void_p(split["stream"].cuda_stream)
                    check(L.sfron_adam_lowrank(ptr(self.p[lo:hi]), ptr(self.m[lo:hi]), ptr(self.v[lo:hi]), ptr(sl(mask, lo, hi)), ptr(stats),
                                               ptr(q["dmod"]), ptr(q["sc"]), q["R"], q["NM"], q["D"], b1, b2, self.eps, step_size, bc2_sqrt,
                                               decay_mul, ptr(sl(self.w_bf16, lo, hi)), ptr(sl(ema, lo, hi)), float(ema_decay), emode,
                                               sp), "adam_lowrank")
                    if ada_side:
                        split["ada_done"] = _t.cuda.Event()
                        split["ada_done"].record(split["stream"])
                else:
                    check(L.sfron_masked_clip_adam(ptr(self.p[lo:hi]), ptr(self.g[lo:hi]), ptr(sl(self.g2, lo, hi)), ptr(self.m[lo:hi]),
                                                   ptr(self.v[lo:hi]), ptr(sl(mask, lo, hi)), ptr(stats), hi - lo, b1, b2, self.eps,
                                                   step_size, bc2_sqrt, decay_mul, ptr(sl(self.w_bf16, lo, hi)), ptr(sl(ema, lo, hi)),
                                                   float(ema_decay), emode, stream_ptr()), "masked_clip_adam")
            if split is not None and excl is not None:
                # the block ranges: the first `head` on this stream, the rest on the second stream behind everything above (what this
                # stream swept at full rate is what the forward pass needs first), bounded grid, one event per range
                cur = _t.cuda.current_stream()
                cap = int(split.get("max_workgroups", 0))

                def sweep_range(i, sp, wg):
                    lo, hi = rngs[i]
                    if quant is None:
                        check(L.sfron_masked_clip_adam_wg(ptr(self.p[lo:hi]), ptr(self.g[lo:hi]), None, ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                                          ptr(sl(mask, lo, hi)), ptr(stats), hi - lo, b1, b2, self.eps, step_size, bc2_sqrt,
                                                          decay_mul, ptr(sl(self.w_bf16, lo, hi)), ptr(sl(ema, lo, hi)), float(ema_decay), emode,
                                                          wg, sp), "masked_clip_adam_wg")
                        return
                    for tlo, thi, si in quant["tensors"][i]:          # one weight tensor (+ bias) per launch: ONE e4m3 scale each
                        check(L.sfron_masked_clip_adam_q(ptr(self.p[tlo:thi]), ptr(self.g[tlo:thi]), ptr(self.m[tlo:thi]), ptr(self.v[tlo:thi]),
                                                         ptr(sl(mask, tlo, thi)), ptr(stats), thi - tlo, b1, b2, self.eps, step_size, bc2_sqrt,
                                                         decay_mul, ptr(sl(self.w_bf16, tlo, thi)), ptr(sl(ema, tlo, thi)), float(ema_decay), emode,
                                                         ptr(quant["w8"][tlo:thi]), ptr(quant["scales"][si:si + 1]), wg, sp), "masked_clip_adam_q")
                cur_p = ctypes.c_void_p(cur.cuda_stream)
                side_p = ctypes.c_void_p(side.cuda_stream) if side is not None else None
                for i in range(min(head, len(rngs))):
                    if i >= first_own:
                        sweep_range(i, cur_p, 0)
                    if side is not None:
                        split["events"][i].record(cur)

                def launch_side():
                    """the ranges that go beside the next forward pass: behind everything the current stream holds AT THIS CALL"""
                    if side is not None and head < len(rngs):
                        ready = _t.cuda.Event()
                        ready.record(_t.cuda.current_stream())
                        side.wait_event(ready)
                    for i in range(head, len(rngs)):
                        sweep_range(i, side_p, cap)
                        split["events"][i].record(side)
                if split.get("defer") and side is not None and head < len(rngs):
                    # the caller launches them itself, behind the conditioning prologue of the forward pass they run beside (step.py:
                    # beside a sweep every boundary of that chain of small launches costs 60-100 us instead of ~5)
                    self.deferred = launch_side
                else:
                    launch_side()
            self.lowrank = None
        self.fused_sumsq = None
        if ev is not None:
            ev[1].record()
            # (start, end, whole): whole = a remain-stage step (EMA fused) whose every byte went through the current stream in ONE pass
            self.timed.append((ev[0], ev[1], (split is None or split.get("stream") is None and split.get("quant") is None) and ema is not None))


    def _step_pipelined(self, pipeline, max_norm, use_mask, ema, ema_decay, ema_mode):
        import torch as _t
        L = _lib.lib()
        cur = _t.cuda.current_stream()
        s = stream_ptr()
        mask = self.mask if use_mask else None
        sl = lambda t, lo, hi: None if t is None else t[lo:hi]
        stats = None
        if max_norm is not None:
            # clip_grad_norm_ needs every range: partial sums per range as each arrives, then one coefficient, then the update
            per = L.sfron_sweep_partials_len()
            if self._partials.numel() < per * len(pipeline):
                self._partials = _t.empty(per * len(pipeline), dtype=_t.float64, device=self.p.device)
            used = 0
            for lo, hi, ev in pipeline:
                cur.wait_event(ev)
                nblk = ctypes.c_int(0)
                check(L.sfron_sumsq_masked(ptr(self.g[lo:hi]), None, ptr(sl(mask, lo, hi)), hi - lo, ptr(self._partials[used:]),
                                           ctypes.byref(nblk), s), "sumsq_masked")
                used += nblk.value
            check(L.sfron_clip_coef(ptr(self._partials), used, float(max_norm), ptr(self.stats), s), "clip_coef")
            stats = self.stats
        self.step_count += 1
        b1, b2 = self.betas
        step_size = self.lr / (1 - b1 ** self.step_count)
        bc2_sqrt = math.sqrt(1 - b2 ** self.step_count)
        decay_mul = 1.0 - self.lr * self.wd
        emode = int(ema_mode if ema is not None else 0)
        for lo, hi, ev in pipeline:
            if max_norm is None:
                cur.wait_event(ev)
            check(L.sfron_masked_clip_adam(ptr(self.p[lo:hi]), ptr(self.g[lo:hi]), None, ptr(self.m[lo:hi]), ptr(self.v[lo:hi]),
                                           ptr(sl(mask, lo, hi)), ptr(stats), hi - lo, b1, b2, self.eps, step_size, bc2_sqrt, decay_mul,
                                           ptr(sl(self.w_bf16, lo, hi)), ptr(sl(ema, lo, hi)), float(ema_decay), emode, s), "masked_clip_adam")


def ema_update(ema, p, decay, mode=1):
    check(_lib.lib().sfron_ema_update(ptr(ema), ptr(p), p.numel(), float(decay), int(mode), stream_ptr()), "ema_update")


def fisher_accum(fisher, g, n_iters):
    check(_lib.lib().sfron_fisher_accum(ptr(fisher), ptr(g), g.numel(), float(n_iters), stream_ptr()), "fisher_accum")


def mask_from_fisher(forget_fisher, remain_fisher, th):
    """((F_f + 1e-15) / (F_r + 1e-15)) >= th as a torch.bool tensor (DiT/generate_mask.py:34-35)."""
    ff = forget_fisher.contiguous().view(-1)
    rf = remain_fisher.contiguous().view(-1)
    out = torch.empty(ff.numel(), dtype=torch.uint8, device=ff.device)
    check(_lib.lib().sfron_mask_from_fisher(ptr(ff), ptr(rf), ff.numel(), float(th), ptr(out), stream_ptr()),
          "mask_from_fisher")
    return out.view(forget_fisher.shape).to(torch.bool)


def cast_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().sfron_cast_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "cast_bf16")
    return dst
