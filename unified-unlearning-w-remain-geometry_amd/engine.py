"""Host handle of the whole-model DiT pass (csrc/dit_engine.hip): arenas, name <-> offset map, workspace.

The parameter arena is one flat fp32 GPU tensor; every reference state_dict key
(/root/reference/DiT/models.py) is a view into it, so pretrained checkpoints load
unchanged while the sweep kernels see one contiguous range.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib, streams
from ._lib import check, ptr, stream_ptr

_LAYOUT_KEYS = ["total", "trainable", "pe_w", "pe_b", "t0_w", "t0_b", "t2_w", "t2_b", "table", "ada_w", "ada_b",
                "blocks", "blk_stride", "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                "fin_w", "fin_b", "pos"]


class DitEngine:
    def __init__(self, batch, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4.0, num_classes=1000, learn_sigma=True, device="cuda", share=None, grads=None):
        """share: another DitEngine whose parameter arenas (fp32 + bf16) this one uses (own workspace, aux; own gradient
        arena unless ``grads`` hands one over)."""
        self._ctor = dict(input_size=input_size, patch_size=patch_size, in_channels=in_channels, hidden_size=hidden_size,
                          depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio, num_classes=num_classes,
                          learn_sigma=learn_sigma, device=device)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SfronError("DitEngine needs a GPU (no CPU fallback)")
        c = _lib.DitCfg()
        c.batch, c.in_channels, c.input_size, c.patch = batch, in_channels, input_size, patch_size
        c.hidden, c.depth, c.heads = hidden_size, depth, num_heads
        c.mlp_hidden, c.num_classes, c.freq_dim = int(hidden_size * mlp_ratio), num_classes, 256
        c.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.cfg = c
        L = _lib.lib()
        lay = (ctypes.c_int64 * _lib.DIT_LAYOUT_LEN)()
        check(L.sfron_dit_param_layout(ctypes.byref(c), lay, _lib.DIT_LAYOUT_LEN), "dit_param_layout (unsupported DiT config?)")
        self.layout = dict(zip(_LAYOUT_KEYS, [int(v) for v in lay]))
        self.n_total, self.n_trainable = self.layout["total"], self.layout["trainable"]
        self.tokens = (input_size // patch_size) ** 2
        self.index = self._build_index()
        if share is None:
            self.params = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
            self.params_bf16 = torch.zeros(self.n_total, dtype=torch.bfloat16, device=self.device)
            # state that belongs to the PARAMETER ARENAS, not to one engine over them: the stream of an optimizer sweep that a runner
            # left in flight (step.DiTSFRon.sweep_across_steps).  Every engine over the same arenas -- a sibling() of a micro-batch chain
            # or of the joint method, the replacement set_batch_size() builds -- orders itself behind it (drain_sweep)
            self._shared = {"sweep": None}
        else:
            assert share.n_total == self.n_total
            self.params, self.params_bf16 = share.params, share.params_bf16
            self._shared = share._shared
        self.grads = grads if grads is not None else torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        assert self.grads.numel() == self.n_total
        ws = L.sfron_dit_workspace_bytes(ctypes.byref(c))
        if ws < 0:
            raise _lib.SfronError("unsupported DiT config")
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=self.device)
        self.out_shape = (batch, c.out_channels, input_size, input_size)
        self.probe = self.wprobe = None
        self.fp8 = None
        h = ctypes.c_void_p()
        check(L.sfron_aux_create(ctypes.byref(h)), "aux_create")     # side stream + events for concurrent wgrads
        self.aux = h

    def close(self):
        """Release the library-side handles (side stream + events, probe events).  Idempotent; also run by __del__."""
        aux, probe, wprobe = getattr(self, "aux", None), getattr(self, "probe", None), getattr(self, "wprobe", None)
        self.aux = self.probe = self.wprobe = None
        try:
            if aux is not None:
                _lib.lib().sfron_aux_destroy(aux)
            for h in (probe, wprobe):
                if h is not None:
                    _lib.lib().sfron_probe_destroy(h)
        except Exception:          # interpreter shutdown: the library may already be gone
            pass

    def __del__(self):
        self.close()

    def sibling(self, batch):
        """A second engine over the SAME parameters (own workspace / gradient arena / side stream) for a micro-batch chain."""
        e = DitEngine(batch, share=self, **self._ctor)
        e._share_fp8(self)
        return e

    # ------------------------------------------------------------------ names
    def _build_index(self):
        """OrderedDict name -> (offset, shape, trainable) in the reference's named_parameters() order."""
        c, lay = self.cfg, self.layout
        D, F, L, p = c.hidden, c.mlp_hidden, c.depth, c.patch
        idx = OrderedDict()
        idx["pos_embed"] = (lay["pos"], (1, self.tokens, D), False)
        idx["x_embedder.proj.weight"] = (lay["pe_w"], (D, c.in_channels, p, p), True)
        idx["x_embedder.proj.bias"] = (lay["pe_b"], (D,), True)
        idx["t_embedder.mlp.0.weight"] = (lay["t0_w"], (D, c.freq_dim), True)
        idx["t_embedder.mlp.0.bias"] = (lay["t0_b"], (D,), True)
        idx["t_embedder.mlp.2.weight"] = (lay["t2_w"], (D, D), True)
        idx["t_embedder.mlp.2.bias"] = (lay["t2_b"], (D,), True)
        idx["y_embedder.embedding_table.weight"] = (lay["table"], (c.num_classes + 1, D), True)
        for l in range(L):
            b = lay["blocks"] + l * lay["blk_stride"]
            pre = f"blocks.{l}."
            idx[pre + "attn.qkv.weight"] = (b + lay["qkv_w"], (3 * D, D), True)
            idx[pre + "attn.qkv.bias"] = (b + lay["qkv_b"], (3 * D,), True)
            idx[pre + "attn.proj.weight"] = (b + lay["proj_w"], (D, D), True)
            idx[pre + "attn.proj.bias"] = (b + lay["proj_b"], (D,), True)
            idx[pre + "mlp.fc1.weight"] = (b + lay["fc1_w"], (F, D), True)
            idx[pre + "mlp.fc1.bias"] = (b + lay["fc1_b"], (F,), True)
            idx[pre + "mlp.fc2.weight"] = (b + lay["fc2_w"], (D, F), True)
            idx[pre + "mlp.fc2.bias"] = (b + lay["fc2_b"], (D,), True)
            idx[pre + "adaLN_modulation.1.weight"] = (lay["ada_w"] + l * 6 * D * D, (6 * D, D), True)
            idx[pre + "adaLN_modulation.1.bias"] = (lay["ada_b"] + l * 6 * D, (6 * D,), True)
        idx["final_layer.linear.weight"] = (lay["fin_w"], (p * p * c.out_channels, D), True)
        idx["final_layer.linear.bias"] = (lay["fin_b"], (p * p * c.out_channels,), True)
        idx["final_layer.adaLN_modulation.1.weight"] = (lay["ada_w"] + L * 6 * D * D, (2 * D, D), True)
        idx["final_layer.adaLN_modulation.1.bias"] = (lay["ada_b"] + L * 6 * D, (2 * D,), True)
        return idx

    def view(self, arena, name):
        off, shape, _ = self.index[name]
        n = 1
        for s in shape:
            n *= s
        return arena[off:off + n].view(shape)

    def arena_mask(self, valid_only=True):
        """bool [n_total]: True on elements that belong to a named tensor (arena padding is False)."""
        m = torch.zeros(self.n_total, dtype=torch.bool, device=self.device)
        for name in self.index:
            self.view(m, name).fill_(True)
        return m

    def sync_bf16(self):
        check(_lib.lib().sfron_cast_bf16(ptr(self.params), ptr(self.params_bf16), self.n_total, stream_ptr()), "cast_bf16")
        if getattr(self, "fp8", None) is not None:
            self.fp8_requantize(fresh=True)

    # ------------------------------------------------------------------ config 5: fp8 (e4m3) forward GEMMs
    FP8_ACT_SCALES = (8.0, 32.0, 16.0)     # LayerNorm+modulate output, attention output, gelu(fc1): static powers of two

    def enable_fp8(self, act_scales=None):
        """Run the four token GEMMs of every block's FORWARD pass on the fp8 matrix core (BASELINE config 5): an e4m3 shadow of
        the block weights with one power-of-two scale per tensor (re-quantised after every optimizer step: fp8_requantize), e4m3
        activations written by their producers.  The backward pass keeps bf16 operands (straight-through estimator)."""
        L = _lib.lib()
        c, lay = self.cfg, self.layout
        M = c.batch * self.tokens
        D, F = c.hidden, c.mlp_hidden
        for N, K in ((3 * D, D), (D, D), (F, D), (D, F)):
            if not L.sfron_fp8_gemm_supported(M, N, K):
                raise _lib.SfronError(f"fp8 path: GEMM {M}x{N}x{K} is not a multiple of the 256x128x128 fp8 tile")
        act = tuple(float(a) for a in (act_scales or self.FP8_ACT_SCALES))
        rows = []
        for l in range(c.depth):
            b = lay["blocks"] + l * lay["blk_stride"]
            rows += [(b + lay["qkv_w"], 3 * D * D), (b + lay["proj_w"], D * D), (b + lay["fc1_w"], F * D), (b + lay["fc2_w"], D * F)]
        # per block: the (weight + bias) element ranges with the index of their scale -- what a re-quantising sweep launches over
        tens = []
        for l in range(c.depth):
            b = lay["blocks"] + l * lay["blk_stride"]
            ends = [b + lay["proj_w"], b + lay["fc1_w"], b + lay["fc2_w"], b + lay["blk_stride"]]
            starts = [b + lay["qkv_w"], b + lay["proj_w"], b + lay["fc1_w"], b + lay["fc2_w"]]
            tens.append([(starts[i], ends[i], 4 * l + i) for i in range(4)])
        ws = L.sfron_dit_fp8_workspace_bytes(ctypes.byref(c))
        self.fp8 = dict(
            tensors=tens, refresh_every=16, sweeps=0,
            act=(ctypes.c_float * 3)(*act), act_scales=act,
            table=torch.tensor(rows, dtype=torch.int64, device=self.device), n=len(rows),
            w8=torch.zeros(self.n_total, dtype=torch.uint8, device=self.device),
            scales=torch.ones(len(rows), dtype=torch.float32, device=self.device),
            amax=torch.zeros(len(rows), dtype=torch.int32, device=self.device),
            # this engine's activation-range words (three uint32 the quantising kernels raise with atomicMax: fp8_activation_range)
            act_amax=torch.zeros(3, dtype=torch.int32, device=self.device),
            ws=torch.empty(ws, dtype=torch.uint8, device=self.device))
        self.fp8_requantize(fresh=True)
        return self

    def fp8_activation_range(self, reset=True):
        """How much of the e4m3 range the activations used since the last reset (the activation scales are static and the conversion
        saturates silently): dict site -> max |x * scale| / 448 for the LayerNorm+modulate outputs, the attention output and gelu(fc1), plus
        ``saturated`` (any site above 1: values were clipped -- raise the matching entry of FP8_ACT_SCALES' divisor, i.e. pass a smaller
        scale to enable_fp8(act_scales=...)).  The counters are this engine's own (round 6: a device array handed to every quantising launch,
        include/sfron.h sfron_fp8_activation_amax; engines that share a shadow -- a sibling() -- share them); synchronises the stream."""
        out = (ctypes.c_float * 3)()
        check(_lib.lib().sfron_fp8_activation_amax(ptr(self.fp8["act_amax"]), out, int(bool(reset)), stream_ptr()), "fp8_activation_amax")
        r = {"ln_modulate": out[0] / 448.0, "attention_out": out[1] / 448.0, "gelu": out[2] / 448.0}
        r["saturated"] = any(v > 1.0 for v in r.values())
        return r

    def fp8_requantize(self, fresh=False):
        """e4m3 shadow <- fp32 masters, after every optimizer step.  One pass: the scales come from the amax the PREVIOUS call
        collected (delayed scaling: a weight moves by at most lr per step and the scale keeps 2x headroom under 448), and this
        pass collects the amax for the next call.  fresh=True (after loading / initialising weights): an extra amax-only pass
        first, so the scales fit the current weights exactly."""
        f, L, s = self.fp8, _lib.lib(), stream_ptr()
        if fresh:
            f["amax"].zero_()
            check(L.sfron_fp8_quant_tensors(ptr(self.params), ptr(f["table"]), f["n"], None, ptr(f["amax"]), None, 0, s), "fp8_amax")
        check(L.sfron_fp8_update_scales(ptr(f["amax"]), f["n"], ptr(f["scales"]), s), "fp8_update_scales")
        check(L.sfron_fp8_quant_tensors(ptr(self.params), ptr(f["table"]), f["n"], ptr(f["scales"]), ptr(f["amax"]), ptr(f["w8"]), 1, s),
              "fp8_quant")

    def fp8_refresh_scales(self):
        """amax pass over the fp32 masters + new power-of-two scales (no quantisation): what a run whose optimizer sweep writes the e4m3
        shadow itself (FlatAdam.step(split=dict(quant=...))) calls every ``refresh_every`` sweeps, BEFORE the sweep -- the sweep then
        rewrites every e4m3 weight with the new scale."""
        f, L, s = self.fp8, _lib.lib(), stream_ptr()
        f["amax"].zero_()
        check(L.sfron_fp8_quant_tensors(ptr(self.params), ptr(f["table"]), f["n"], None, ptr(f["amax"]), None, 0, s), "fp8_amax")
        check(L.sfron_fp8_update_scales(ptr(f["amax"]), f["n"], ptr(f["scales"]), s), "fp8_update_scales")

    def _share_fp8(self, other):
        """Another engine over the same parameters (another batch size / a micro-batch chain) uses the same e4m3 shadow and scales."""
        if getattr(other, "fp8", None) is not None:
            ws = _lib.lib().sfron_dit_fp8_workspace_bytes(ctypes.byref(self.cfg))
            self.fp8 = dict(other.fp8, ws=torch.empty(ws, dtype=torch.uint8, device=self.device))

    # ------------------------------------------------------------------ passes
    def block_sweep_setup(self):
        """Per-block events + a stream for an optimizer sweep that runs beside the next forward pass (FlatAdam.step(split=...))."""
        if getattr(self, "_bs", None) is None:
            L, lay = self.cfg.depth, self.layout
            evs = [torch.cuda.Event(enable_timing=False) for _ in range(L)]
            # process-wide, probed to run beside the caller's stream (the pass it overlaps) and the weight-gradient streams: see streams.py
            st = streams.get("sweep", self.device, beside=[torch.cuda.current_stream()] + self.side_streams())
            for e in evs:
                e.record(st)
            self._bs = dict(ranges=[(lay["blocks"] + l * lay["blk_stride"], lay["blocks"] + (l + 1) * lay["blk_stride"]) for l in range(L)],
                            stream=st, events=evs, handles=(ctypes.c_void_p * L)(*[e.cuda_event for e in evs]))
        return self._bs

    @property
    def _sweep_pending(self):
        return self._shared["sweep"]

    @_sweep_pending.setter
    def _sweep_pending(self, stream):
        self._shared["sweep"] = stream

    def drain_sweep(self):
        """The runner leaves the remain-stage AdamW of the block ranges in flight on the sweep stream when step() returns (the next
        step's forward pass waits block by block); every OTHER reader of the parameters / optimizer state goes through here first."""
        fn = self._shared.pop("deferred", None)
        if fn is not None:
            fn()                               # a block sweep whose launch was left to the next forward pass (step.py): launch it now
        st = self._shared["sweep"]
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
            self._shared["sweep"] = None

    def forward(self, x_t, t, y, drop=None, out=None, block_ready=None, between=None, ada_ready=None):
        """block_ready: ctypes array of depth hipEvent_t handles -- block l waits for entry l before it touches its weights.
        between (with block_ready): a callable run BETWEEN the conditioning prologue and block 0 (sfron_dit_forward_phase): the runner starts
        the block sweep that goes beside this pass there -- behind the prologue's chain of small launches, not beside it (step.py).
        ada_ready (with block_ready): a torch event behind the optimizer sweep of the adaLN matrix, which the runner put on the sweep stream
        (FlatAdam.step(split[ada_side])): everything in front of the adaLN product is issued first -- it runs beside that sweep --, then this
        stream waits for the event, then the product and the blocks."""
        if out is None:
            out = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
        if x_t.shape[0] != self.cfg.batch:
            raise _lib.SfronError(f"engine was built for batch {self.cfg.batch}, got {x_t.shape[0]}")
        if block_ready is None:
            assert between is None
            self.drain_sweep()                 # a block sweep may still be rewriting the weights on its own stream (step.py)
        if ada_ready is not None:
            assert block_ready is not None
        if between is not None or ada_ready is not None:
            L, f = _lib.lib(), self.fp8
            for phase in ((1, 2) if ada_ready is None else (3, 4, 2)):
                if phase == 4:
                    torch.cuda.current_stream().wait_event(ada_ready)
                if f is None:
                    check(L.sfron_dit_forward_phase(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(x_t), ptr(t), ptr(y),
                                                    ptr(drop), ptr(self.workspace), ptr(out), block_ready, self.probe, phase, stream_ptr()),
                          "dit_forward_phase")
                else:
                    check(L.sfron_dit_forward_fp8_phase(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(f["w8"]),
                                                        ptr(f["scales"]), f["act"], ptr(f["act_amax"]), ptr(x_t), ptr(t), ptr(y), ptr(drop), ptr(self.workspace),
                                                        ptr(f["ws"]), ptr(out), block_ready, phase, stream_ptr()), "dit_forward_fp8_phase")
                if phase in (1, 4) and between is not None:
                    between()
            return out
        if block_ready is not None and self.fp8 is None:
            check(_lib.lib().sfron_dit_forward_after(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(x_t), ptr(t), ptr(y),
                                                     ptr(drop), ptr(self.workspace), ptr(out), block_ready, self.probe, stream_ptr()),
                  "dit_forward_after")
            return out
        if getattr(self, "fp8", None) is not None:
            f = self.fp8
            check(_lib.lib().sfron_dit_forward_fp8(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(f["w8"]),
                                                   ptr(f["scales"]), f["act"], ptr(f["act_amax"]), ptr(x_t), ptr(t), ptr(y), ptr(drop), ptr(self.workspace),
                                                   ptr(f["ws"]), ptr(out), block_ready, stream_ptr()), "dit_forward_fp8")
            return out
        check(_lib.lib().sfron_dit_forward_probed(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(x_t),
                                                  ptr(t), ptr(y), ptr(drop), ptr(self.workspace), ptr(out), self.probe,
                                                  stream_ptr()), "dit_forward")
        return out

    # ------------------------------------------------------------------ roofline probe (bench.py)
    def probe_enable(self, max_samples):
        h = ctypes.c_void_p()
        check(_lib.lib().sfron_probe_create(int(max_samples), ctypes.byref(h)), "probe_create")
        self.probe = h

    def wgrad_probe_enable(self, max_samples):
        """Event pairs around the four weight-gradient GEMMs of every 9th block, on the weight-gradient stream."""
        h = ctypes.c_void_p()
        check(_lib.lib().sfron_probe_create(int(max_samples), ctypes.byref(h)), "probe_create")
        check(_lib.lib().sfron_aux_set_probe(self.aux, h), "aux_set_probe")
        self.wprobe = h

    def wgrad_probe_read(self):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        check(_lib.lib().sfron_probe_read(self.wprobe, ctypes.byref(n), ctypes.byref(ms)), "probe_read")
        return n.value, ms.value

    def probe_reset(self):
        check(_lib.lib().sfron_probe_reset(self.probe), "probe_reset")

    def probe_read(self):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        check(_lib.lib().sfron_probe_read(self.probe, ctypes.byref(n), ctypes.byref(ms)), "probe_read")
        return n.value, ms.value

    # ------------------------------------------------------------------ data-parallel backward (overlapped all-reduce)
    def dp_setup(self):
        """One torch event per block (recorded by the library on its weight-gradient stream) + the late-bias staging buffer."""
        if getattr(self, "_dp_events", None) is None:
            L, D = self.cfg.depth, self.cfg.hidden
            self._dp_events = [torch.cuda.Event(enable_timing=False) for _ in range(L)]
            for e in self._dp_events:
                e.record()                                   # materialises the handle
            self._dp_handles = (ctypes.c_void_p * L)(*[e.cuda_event for e in self._dp_events])
            self.late_bias = torch.zeros(L, 2, D, dtype=torch.float32, device=self.device)
            NM = (6 * L + 2) * D
            self.ada_dmod = torch.zeros(self.cfg.batch, NM, dtype=torch.bfloat16, device=self.device)     # factors of the adaLN weight gradient
            self.ada_sc = torch.zeros(self.cfg.batch, D, dtype=torch.bfloat16, device=self.device)
            lay = self.layout
            self.block_ranges = [(lay["blocks"] + l * lay["blk_stride"], lay["blocks"] + (l + 1) * lay["blk_stride"]) for l in range(L)]
        return self._dp_events

    def backward_dp(self, d_out, y, drop=None):
        """Backward pass that records per-block completion events and parks proj.bias / fc2.bias gradients in ``late_bias``."""
        self.dp_setup()
        check(_lib.lib().sfron_dit_backward_dp(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(d_out), ptr(y),
                                               ptr(drop), ptr(self.workspace), ptr(self.grads), self.aux, self._dp_handles,
                                               ptr(self.late_bias), ptr(self.ada_dmod), ptr(self.ada_sc), stream_ptr()), "dit_backward_dp")
        return self.grads

    def backward_factored_ada(self, d_out, y, drop=None):
        """Backward pass that does NOT form the adaLN_modulation weight gradient (a [(6L+2)D][D] fp32 matrix, a third of the arena):
        it hands out its two bf16 factors instead -- dmod [batch][(6L+2)D] and silu(c) [batch][D] -- for a sweep that forms the
        rank-(batch) product itself (sweep.FlatAdam.lowrank).  The arena's ada_w range is left untouched (stale)."""
        if getattr(self, "_ada_f", None) is None or self._ada_f[0].shape[0] != self.cfg.batch:
            NM = (6 * self.cfg.depth + 2) * self.cfg.hidden
            self._ada_f = (torch.empty(self.cfg.batch, NM, dtype=torch.bfloat16, device=self.device),
                           torch.empty(self.cfg.batch, self.cfg.hidden, dtype=torch.bfloat16, device=self.device))
        dmod, sc = self._ada_f
        check(_lib.lib().sfron_dit_backward_dp(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(d_out), ptr(y),
                                               ptr(drop), ptr(self.workspace), ptr(self.grads), self.aux, None, None, ptr(dmod), ptr(sc),
                                               stream_ptr()), "dit_backward (factored adaLN gradient)")
        return dict(lo=self.layout["ada_w"], NM=dmod.shape[1], D=self.cfg.hidden, dmod=dmod, sc=sc, R=self.cfg.batch, wait=self.ada_wait,
                    wait_factors=self.ada_wait_factors, beside=self.side_streams())

    def ada_wait(self, stream):
        """Order ``stream`` (a torch stream) behind the point of the last backward pass after which nothing reads the adaLN matrix and its two
        gradient factors are complete (sfron_aux_wait_ada): what is left of the pass then is the embedders' backward."""
        check(_lib.lib().sfron_aux_wait_ada(self.aux, ctypes.c_void_p(stream.cuda_stream)), "aux_wait_ada")

    def side_streams(self):
        """The library's two weight-gradient streams of this engine as torch streams (read-only use: what a caller's own side stream must
        not share a hardware queue with -- streams.py)."""
        if getattr(self, "_side_streams", None) is None:
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            check(_lib.lib().sfron_aux_streams(self.aux, ctypes.byref(a), ctypes.byref(b)), "aux_streams")
            self._side_streams = [torch.cuda.ExternalStream(a.value, device=self.device), torch.cuda.ExternalStream(b.value, device=self.device)]
        return list(self._side_streams)

    def ada_wait_factors(self, stream):
        """Order ``stream`` behind the point of the last backward_factored_ada pass at which the two gradient factors are complete
        (sfron_aux_wait_ada_factors: earlier than ada_wait -- the dgrad through the adaLN Linear and the embedders' backward are still to come)."""
        check(_lib.lib().sfron_aux_wait_ada_factors(self.aux, ctypes.c_void_p(stream.cuda_stream)), "aux_wait_ada_factors")

    # ------------------------------------------------------------------ clip norm taken where the gradients are produced
    def fused_sumsq_plan(self):
        """What a single-process forget stage needs so that clip_grad_norm_ (DiT/forget.py:293-298) costs no pass over the block range of the
        gradient arena: ``n_gemm`` fp64 partials that the next backward pass's weight-gradient GEMMs write (arm_sumsq), and a device table
        of the element ranges of everything ELSE that is trainable outside the adaLN weight matrix (embedders, label table, adaLN bias, every
        block's four bias vectors, final layer), split into pieces of at most 16 K elements -- one sfron_sumsq_masked_ranges launch.  None
        when a block shape does not run on the 192 x 192 weight-gradient tile."""
        if getattr(self, "_sq_plan", None) is None:
            n_gemm = _lib.lib().sfron_dit_sumsq_partials_len(ctypes.byref(self.cfg))
            if n_gemm <= 0:
                self._sq_plan = False
            else:
                lay, c = self.layout, self.cfg
                D, F, L = c.hidden, c.mlp_hidden, c.depth
                NM = (6 * L + 2) * D
                spans = [(0, lay["ada_w"]), (lay["ada_b"], lay["blocks"])]
                for l in range(L):
                    b = lay["blocks"] + l * lay["blk_stride"]
                    spans += [(b + lay["qkv_b"], b + lay["qkv_b"] + 3 * D), (b + lay["proj_b"], b + lay["proj_b"] + D),
                              (b + lay["fc1_b"], b + lay["fc1_b"] + F), (b + lay["fc2_b"], b + lay["fc2_b"] + D)]
                spans.append((lay["fin_w"], self.n_trainable))
                covered = sum(hi - lo for lo, hi in spans) + NM * D + L * (3 * D * D + D * D + 2 * F * D)
                # arena padding (tensor offsets are multiples of 8) holds zero gradients: a span may include it.  None may be MISSING: every
                # trainable element is in a span, in the adaLN matrix or in a block weight -- so what the list does not cover can only be the
                # padding between a block's tensors (< 8 elements behind each of its 8 tensors).  A tensor added to a block or to the layout
                # later fails here instead of dropping out of the clip norm silently.
                missing = self.n_trainable - covered
                assert 0 <= missing < 8 * 8 * L + 64 and all(lo % 4 == 0 and (hi - lo) % 4 == 0 for lo, hi in spans), \
                    f"sumsq plan: {missing} trainable elements are in no span / GEMM (layout changed?)"
                rows = []
                for lo, hi in spans:
                    for s0 in range(lo, hi, 16384):               # one workgroup per piece: short pieces, many workgroups
                        rows.append((s0, min(16384, hi - s0)))
                self._sq_plan = dict(n_gemm=int(n_gemm), n_ranges=len(rows),
                                     ranges=torch.tensor(rows, dtype=torch.int64, device=self.device))
        return self._sq_plan or None

    def arm_sumsq(self, mask_arena, partials):
        """One-shot: the next backward pass through this engine leaves the masked sums of squares of its block weight gradients in ``partials``
        (fp64, fused_sumsq_plan()["n_gemm"] entries)."""
        check(_lib.lib().sfron_aux_arm_sumsq(self.aux, ptr(mask_arena), ptr(partials)), "aux_arm_sumsq")

    def disarm_sumsq(self):
        """Clear a pending arm_sumsq (a pass that raised before its backward must not leave the raw pointer on the handle)."""
        check(_lib.lib().sfron_aux_arm_sumsq(self.aux, None, None), "aux_arm_sumsq(disarm)")

    def scatter_late_bias(self):
        check(_lib.lib().sfron_dit_scatter_late_bias(ctypes.byref(self.cfg), ptr(self.late_bias), ptr(self.grads), stream_ptr()),
              "dit_scatter_late_bias")

    def backward(self, d_out, y, drop=None, grads=None):
        g = self.grads if grads is None else grads
        check(_lib.lib().sfron_dit_backward(ctypes.byref(self.cfg), ptr(self.params), ptr(self.params_bf16), ptr(d_out),
                                            ptr(y), ptr(drop), ptr(self.workspace), ptr(g), self.aux, stream_ptr()), "dit_backward")
        return g
