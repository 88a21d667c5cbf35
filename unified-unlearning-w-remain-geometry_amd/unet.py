"""The DDPM conditional U-Net (BASELINE config 0: CIFAR-10 class forgetting) over the HIP kernels of csrc/conv.hip.

Mirrors /root/reference/DDPM/models/diffusion.py:195-413 ``Conditional_Model``: same constructor (a config object with
``model`` / ``data`` / ``diffusion`` sections, or the same fields as keywords), ``forward(x, t, c, mode, **kwargs)`` with
``mode="train"`` (``cond_drop_prob``) and ``mode="test"`` (``cond_scale``: (1 + s) * cond - s * null, :340-357), identical
``state_dict()`` keys / shapes / order (so ``ckpts/ckpt.pth`` loads with ``load_state_dict``; "module."-prefixed keys of the
DataParallel checkpoints, runners/diffusion.py:1055-1061, are accepted).  Every nn.Parameter is a view into ONE flat fp32
arena (gradients, Adam moments, EMA shadow and the bf16 weight shadow share its offsets), which is what the mask -> clip ->
Adam -> EMA sweep of csrc/sweep.hip runs over.

No autograd inside: ``forward`` records a tape of explicit backward steps (each a few C-ABI launches) and returns a tensor with
one autograd edge, so ``loss.backward()`` of the reference's training code (functions/losses.py, runners/diffusion.py:1098-1170)
drives the HIP backward pass.  Activations are NHWC fp32 rows between blocks and bf16 where they feed a GEMM.
Random draws that the reference makes inside the forward pass -- the classifier-free keep mask (prob_mask_like, :8-14,372-376)
and one dropout mask per ResnetBlock (:131) -- are drawn here on the device, or handed in (``keep_mask`` / ``dropout_masks``)
so that a test can give the oracle the very same draws.
"""
import ctypes
from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr


def _L():
    return _lib.lib()


class _Holder(nn.Module):
    pass


class Act:
    """An NHWC activation: fp32 rows [B*H*W][C] (+ its gradient, allocated by the first backward step that reaches it)."""
    __slots__ = ("t", "B", "H", "W", "C", "grad")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C, self.grad = t, B, H, W, C, None

    @property
    def rows(self):
        return self.B * self.H * self.W

    def grad_buf(self):
        """(tensor, accumulate flag) for a backward step that ADDS its contribution to this activation's gradient."""
        if self.grad is None:
            self.grad = torch.empty_like(self.t)
            return self.grad, 0
        return self.grad, 1


def _pad8(n):
    return (n + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ thin kernel wrappers
# bench.py: a one-element list here makes every product the tape launches add its algorithmic FLOPs (2 M N K; 2 rows n_out taps c_src
# for a convolution; 4 T^2 d per head forward / 10 T^2 d backward for the flash attention kernels) -- None in normal use
PRODUCT_FLOPS = None


def count_flops(x):
    if PRODUCT_FLOPS is not None:
        PRODUCT_FLOPS[0] += float(x)


def bgemm(A, B, M, N, K, *, lda, ldb, a_t=False, b_t=False, batch=1, sa=0, sb=0, sc=0, batch2=1, sa2=0, sb2=0, sc2=0, alpha=1.0, bias=None,
          c_bf16=None, c_f32=None, ldc=None, resid=None, vec=None, ld_vec=0, rows_per_sample=1, accumulate=False):
    """A / B / outputs may be tensors or raw device addresses (column-slice views are passed as address + leading dimension)."""
    d = _lib.BGemmDesc()
    d.A, d.B = _addr(A), _addr(B)
    d.M, d.N, d.K, d.lda, d.ldb = M, N, K, lda, ldb
    d.a_transposed, d.b_transposed, d.batch = int(a_t), int(b_t), batch
    d.stride_a, d.stride_b, d.stride_c, d.alpha = sa, sb, sc, alpha
    d.batch2, d.stride_a2, d.stride_b2, d.stride_c2 = batch2, sa2, sb2, sc2
    d.bias = _addr(bias)
    d.c_bf16, d.c_f32, d.ldc = _addr(c_bf16), _addr(c_f32), ldc
    d.resid, d.sample_vec, d.ld_vec, d.rows_per_sample, d.accumulate = _addr(resid), _addr(vec), ld_vec, rows_per_sample, int(accumulate)
    small_rows = (not a_t) and b_t and M <= 128 and K >= 4096 and bias is None and resid is None and vec is None and not accumulate
    if ((a_t and b_t and K >= 2048) or small_rows) and c_f32 is not None and batch == 1 and batch2 == 1 and ldc == N:
        ws = _split_scratch(M * N, A if not isinstance(A, int) else (B if not isinstance(B, int) else None))
        if ws is not None:
            d.split_ws, d.split_ws_slabs = ws[0].data_ptr(), ws[1]
    elif a_t and b_t and c_bf16 is not None and batch * batch2 > 1 and K >= 1024 and M <= 128 and N <= 256 and not isinstance(A, int):
        ws, nsl = _split_scratch(M * N * batch * batch2, A)          # head-batched single-tile products: split the contraction
        d.split_ws, d.split_ws_slabs = ws.data_ptr(), nsl * batch * batch2
    count_flops(2.0 * M * N * K * batch * batch2)
    check(_L().sfron_bgemm_bf16(ctypes.byref(d), stream_ptr()), "bgemm_bf16")


_SPLIT_WS = {}
_SPLIT_WS_RETIRED = []      # superseded scratch tensors stay alive: a captured HIP graph may hold their addresses
_SPLIT_WS_FLOOR = 16 << 20  # floats (64 MB): the most any conv / Linear caller asks for (rows * n_out <= 2^23, >= 2 slabs)


def _split_scratch(mn, like):
    """fp32 slab scratch for split-K products, one per (device, stream): allocated at its upper bound on first use, so a later,
    larger request (another batch size, a Fisher pass, a second model) does not move it under a captured stage graph; if a
    caller ever needs more, the old tensor is retired, not freed.  Keyed by stream because two streams may run split-K
    products concurrently (launch order on ONE stream is what makes sharing safe)."""
    if like is None:
        return None
    dev = like.device
    want = min(64, max(2, (64 << 20) // (4 * mn))) * mn
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _SPLIT_WS.get(key)
    if ws is None or ws.numel() < want:
        if ws is not None:
            _SPLIT_WS_RETIRED.append(ws)
        ws = torch.empty(max(want, _SPLIT_WS_FLOOR), dtype=torch.float32, device=dev)
        _SPLIT_WS[key] = ws
    return ws, want // mn         # slab count as the callers' split models were tuned with (not what the floor would allow)


def _addr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if not t.is_cuda:
        raise _lib.SfronError("sfron ops need GPU tensors (no CPU fallback)")
    return t.data_ptr()


def _conv_desc(B, hs, ws, cs, ho, wo, n_out, taps=9, stride=1, pad=1, up=0, dil=0, bias=None, resid=None, vec=None, ld_vec=0,
               out_bf16=None, out_f32=None, ld_out=None, accumulate=False, pending=None):
    d = _lib.ConvDesc()
    d.batch, d.h_src, d.w_src, d.c_src, d.h_out, d.w_out, d.n_out = B, hs, ws, cs, ho, wo, n_out
    d.taps, d.stride, d.pad, d.upsample, d.dilate = taps, stride, pad, up, dil
    d.bias, d.resid, d.sample_vec, d.ld_vec = _addr(bias), _addr(resid), _addr(vec), ld_vec
    d.out_bf16, d.out_f32, d.ld_out, d.accumulate = _addr(out_bf16), _addr(out_f32), ld_out or 0, int(accumulate)
    rows, t = B * ho * wo, out_f32 if out_f32 is not None else out_bf16
    count_flops(2.0 * rows * n_out * taps * cs)          # every descriptor is launched exactly once
    if t is not None and not isinstance(t, int) and rows * n_out <= (1 << 23) and taps * cs >= 2048 and ld_out == n_out:
        ws, nsl = _split_scratch(rows * n_out, t)            # few output tiles, deep contraction: split-K slabs
        d.split_ws, d.split_ws_slabs = ws.data_ptr(), nsl
        d._keep = ws
        if pending is not None:        # a ctypes int: the call may leave its split unfinished for the GroupNorm that follows (split_source)
            d.split_pending = ctypes.pointer(pending)
    return d


def split_source(d, pending, rows, n_out, keep=()):
    """The unfinished result of a split convolution (sfron_conv_desc.split_pending) as a sfron_split_src for the GroupNorm that consumes it,
    or None when the call finished its output.  `keep`: tensors the description points at (held by the returned object)."""
    if pending is None or pending.value <= 0:
        return None
    src = _lib.SplitSrc()
    src.slabs, src.n_slabs, src.slab_stride = d.split_ws, pending.value, rows * n_out
    src.bias, src.sample_vec, src.ld_vec, src.resid, src.ld_resid = d.bias, d.sample_vec, d.ld_vec, d.resid, n_out
    src._keep = (d._keep,) + tuple(keep)
    src._rows_per_sample = d.h_out * d.w_out
    return src


def colsum_f32(x, rows, n, ld, out, scratch):
    check(_L().sfron_colsum(_addr(x), 0, rows, n, ld, ptr(scratch), 64, _addr(out), stream_ptr()), "colsum")


def colsum_bf16(x, rows, n, ld, out, scratch):
    check(_L().sfron_colsum(_addr(x), 1, rows, n, ld, ptr(scratch), 64, _addr(out), stream_ptr()), "colsum")


def cast_rows_colsum(x, ldx, rows, C, dev, colsum_out, scratch):
    """bf16 copy of fp32 rows AND their column sums (-> colsum_out, a tensor or device address) in one pass."""
    y = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
    check(_L().sfron_cast_rows_colsum(_addr(x), ldx, rows, C, ptr(y), ptr(scratch), min(512, scratch.numel() // C), _addr(colsum_out), stream_ptr()),
          "cast_rows_colsum")
    return y


def cast_rows(x, ldx, rows, C, dev):
    y = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
    check(_L().sfron_cast_rows_bf16(_addr(x), ldx, rows, C, ptr(y), stream_ptr()), "cast_rows")
    return y


# ------------------------------------------------------------------------------------------------ the model
class _TapeNet(nn.Module):
    """Shared machinery of the convolutional U-Nets (DDPM Conditional_Model, LDM UNetModel): parameters as views into one flat fp32
    arena (+ gradient arena, + bf16 shadow), and the building blocks -- each runs its forward launches and returns the closure
    that runs its backward launches (the caller keeps those on a tape)."""

    GN_EPS = 1e-6

    def _alloc_arena(self, specs, groups):
        """specs: OrderedDict name -> shape in named_parameters() order; groups: lists of names laid out contiguously (so that one
        GEMM can read several of them as one matrix); everything else follows in order.  Tensors start at multiples of 8."""
        off, index = 0, {}
        for grp in list(groups) + [[nm] for nm in specs]:
            fresh = [nm for nm in grp if nm not in index]
            for nm in fresh:
                n = 1
                for d in specs[nm]:
                    n *= d
                index[nm] = (off, specs[nm])
                off += n
            off = _pad8(off)
        self.n_total = off
        self.index = OrderedDict((nm, index[nm]) for nm in specs)
        dev = self.device_
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.params_bf16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        # bf16 GEMM operands of the 3x3 convolutions (re-laid from the OIHW master by sfron_conv_wprep before each pass)
        self.conv3 = {}
        for nm, shp in specs.items():
            if nm.endswith(".weight") and len(shp) == 4 and shp[2] == 3:
                co, ci = shp[0], shp[1]
                cop, cip = _pad8(co), _pad8(ci)
                self.conv3[nm[:-len(".weight")]] = dict(
                    co=co, ci=ci, cop=cop, cip=cip, fwd=torch.zeros(cop * 9 * cip, dtype=torch.bfloat16, device=dev),
                    dgr=torch.zeros(ci * 9 * cop, dtype=torch.bfloat16, device=dev) if ci % 8 == 0 else None)
        mx = max(v["cop"] * 9 * v["cip"] for v in self.conv3.values())
        self._dw = torch.empty(mx, dtype=torch.float32, device=dev)
        self._dw_retired = []
        self._cs = torch.empty(64 * 16384, dtype=torch.float32, device=dev)      # column-sum partials (64 row chunks x widest output)

    def _register_views(self):
        for name, (off, shape) in self.index.items():
            node, parts = self, name.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Holder())
                node = getattr(node, part)
            n = 1
            for s in shape:
                n *= s
            node.register_parameter(parts[-1], nn.Parameter(self.params[off:off + n].view(shape)))

    def view(self, arena, name):
        off, shape = self.index[name]
        n = 1
        for s in shape:
            n *= s
        return arena[off:off + n].view(shape)

    def flat_arena(self):
        """(params, grads, bf16 shadow, name -> (offset, shape)): what sfron.ddpm.FlatParams adopts instead of re-homing."""
        return self.params, self.grads, self.params_bf16, self.index

    def sync_bf16(self):
        check(_L().sfron_cast_bf16(ptr(self.params), ptr(self.params_bf16), self.n_total, stream_ptr()), "cast_bf16")
        self._conv_dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        if state_dict and all(k.startswith("module.") for k in state_dict):      # DataParallel checkpoints (runners :1055-1061)
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.sync_bf16()
        return r

    def publish_grads(self):
        for name, p in self.named_parameters():
            p.grad = self.view(self.grads, name)

    def _w(self, name):      # bf16 shadow of a 2-D weight ([out][in], also a 1x1 conv's [out][in][1][1])
        off, _ = self.index[name]
        return self.params_bf16.data_ptr() + 2 * off

    def _p(self, name):      # fp32 master (bias, GroupNorm affine, embedding tables)
        off, _ = self.index[name]
        return self.params.data_ptr() + 4 * off

    def _g(self, name):
        off, _ = self.index[name]
        return self.grads.data_ptr() + 4 * off

    def weights_updated(self, convs=True):
        """Tell the model its fp32 arena changed (an optimizer step): the bf16 operands of the 3x3 convolutions are re-laid before
        the next pass.  With ``auto_prep`` (default) every pass re-lays them anyway; the SFR-on loops turn that off and call this
        after each Adam step, so the two forward passes of a forget stage share one re-layout.  ``convs=False``: the step did not touch the
        convolution kernels (SD train_method "xattn"), their operands stay valid."""
        if convs:
            self._conv_dirty = True

    auto_prep = True
    # name -> bool, or None: which weights receive a gradient.  The SD loop with train_method "xattn" hands only the attn2 parameters to
    # Adam (nsfw_removal.py:66-77); the reference's autograd still forms every gradient, here the products for frozen convolution
    # kernels and Linear weights (a third of a training pass) are skipped -- their arena entries then hold stale values nobody reads.
    wgrad_filter = None

    def _trains(self, pname):
        return self.wgrad_filter is None or bool(self.wgrad_filter(pname))

    def _prep_conv_weights(self):
        if not self.auto_prep and not getattr(self, "_conv_dirty", True):
            return
        self._conv_dirty = False
        if getattr(self, "_wprep_table", None) is None:
            # device-resident table of every 3x3 kernel: one launch re-lays them all (pointers are into the fixed arenas)
            items, tile0 = (_lib.WprepItem * len(self.conv3))(), 0
            for it, (base, v) in zip(items, self.conv3.items()):
                it.w, it.fwd, it.dgr = self._p(base + ".weight"), v["fwd"].data_ptr(), (v["dgr"].data_ptr() if v["dgr"] is not None else None)
                it.co, it.ci, it.co_p, it.ci_p, it.tile0 = v["co"], v["ci"], v["cop"], v["cip"], tile0
                tile0 += _L().sfron_conv_wprep_tiles(v["cop"], v["cip"])
            raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device_)
            self._wprep_table = (raw, len(self.conv3), tile0)
        raw, n_items, n_tiles = self._wprep_table
        check(_L().sfron_conv_wprep_batch(ptr(raw), n_items, n_tiles, stream_ptr()), "conv_wprep_batch")

    # ------------------------------------------------------------------ parameter-gradient finishes, one launch per backward pass
    # The fixed-order sums that finish a parameter gradient from per-sample / per-chunk partials (GroupNorm / LayerNorm affine gradients, conv1's
    # bias and the per-sample projection gradient of every ResBlock) are read by nobody before the end of the tape: a backward pass collects
    # them (self._red is a list while one runs) and issues them as ONE sfron_reduce_batch launch -- the same sums in the same order, ~190
    # launches of ~5 us fewer per DDPM step.  Outside a backward pass (self._red is None) a reduction is launched where it is asked for.
    _red = None
    _scat = None
    BATCH_REDUCTIONS = True
    SCATTER_BATCH_MAX_BYTES = 64 << 20      # slabs of a layer above this (the LDM UNet's 1280-wide kernels) are scattered at once, from the Infinity Cache

    def _reduce(self, partials, groups, per_group, D, out_addr, ldout):
        if self._red is not None:
            self._red.append((partials, groups, per_group, D, out_addr, ldout))      # keeps `partials` alive until the flush
            return
        check(_L().sfron_reduce_chunks(ptr(partials), groups, per_group, D, out_addr, ldout, 0, stream_ptr()), "reduce_chunks")

    def _cast_rows_colsum(self, x, ldx, rows, C, colsum_out):
        """bf16 copy of fp32 rows + their column sums -> colsum_out (a bias gradient); inside a backward pass the sums' finish joins the pass's one
        reduction launch (partials in a buffer of their own until then)."""
        dev = self.device_
        if self._red is None:
            return cast_rows_colsum(x, ldx, rows, C, dev, colsum_out, self._cs)
        y = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
        nmax = max(1, min(512 // ((C + 255) // 256), (rows + 31) // 32))
        part = torch.empty(nmax * C, dtype=torch.float32, device=dev)
        nch = ctypes.c_int(0)
        check(_L().sfron_cast_rows_colsum_partials(_addr(x), ldx, rows, C, ptr(y), ptr(part), nmax, ctypes.byref(nch), stream_ptr()),
              "cast_rows_colsum_partials")
        self._reduce(part, 1, nch.value, C, _addr(colsum_out), C)
        return y

    def _reduce_begin(self):
        self._red = [] if self.BATCH_REDUCTIONS else None
        self._scat = []

    def _reduce_flush(self):
        items, self._red = self._red, None
        scat, self._scat = self._scat, None
        if scat:
            arr = (_lib.WgradScatterItem * len(scat))()
            for a, (buf, out_addr, slab, co, ci, taps, cip, nsl) in zip(arr, scat):
                a.dw_gemm, a.dw_oihw, a.slab_stride, a.c_out, a.c_in, a.taps, a.c_in_p, a.n_slabs = buf.data_ptr(), out_addr, slab, co, ci, taps, cip, nsl
            check(_L().sfron_conv_wgrad_scatter_batch(ctypes.cast(arr, ctypes.c_void_p), len(scat), stream_ptr()), "conv_wgrad_scatter_batch")
        if not items:
            return
        arr = (_lib.ReduceItem * len(items))()
        for a, (t, groups, per_group, D, out_addr, ldout) in zip(arr, items):
            a.partials, a.out, a.groups, a.per_group, a.D, a.ldout = t.data_ptr(), out_addr, groups, per_group, D, ldout
        check(_L().sfron_reduce_batch(ctypes.cast(arr, ctypes.c_void_p), len(items), stream_ptr()), "reduce_batch")

    def _finish_or_fuse(self, src, t, B, HW, C):
        """src: an unfinished split-K result that tensor t [B * HW][C] is to be formed from (split_source), or None.  Returns src when the
        GroupNorm over t takes the one-launch form (which then finishes the sum in its first pass and stores t itself); otherwise launches the
        finish here and returns None."""
        if src is None:
            return None
        if _L().sfron_groupnorm_one_launch(B, HW, C, 32):
            return src
        check(_L().sfron_split_finish(ctypes.byref(src), B * HW, C, src._rows_per_sample, ptr(t), None, C, stream_ptr()), "split_finish")
        return None

    def _gn(self, tape, x, name, swish, drop_mask=None, eps=None, src=None):
        """y = bf16(act(GroupNorm32(x)) [* dropout]); the backward step adds to x.grad.
        src: x.t has not been written yet -- it is this unfinished split-K result (_conv3(defer_finish=True)); the GroupNorm forms and stores it."""
        eps = self.GN_EPS if eps is None else eps
        dev = self.device_
        y = torch.empty(x.rows, x.C, dtype=torch.bfloat16, device=dev)
        mean = torch.empty(x.B * 32, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        scale = 1.0 / (1.0 - self.dropout_p) if drop_mask is not None else 1.0
        gam, bet = self._p(name + ".weight"), self._p(name + ".bias")
        trains = self._trains(name + ".weight")          # decided when the tape is built; a frozen norm (SD train_method "xattn") skips the
                                                         # reduction of its affine gradients: nothing reads them
        src = self._finish_or_fuse(src, x.t, x.B, x.H * x.W, x.C)
        if src is not None:
            check(_L().sfron_groupnorm_fwd_src(ctypes.byref(src), ptr(x.t), gam, bet, x.B, x.H * x.W, x.C, 32, eps, int(swish), ptr(drop_mask), scale,
                                               ptr(y), ptr(mean), ptr(rstd), stream_ptr()), "groupnorm_fwd_src")
        else:
            ws = torch.empty(_L().sfron_groupnorm_scratch_bytes(x.B, x.H * x.W, x.C, 32) // 8 + 2, dtype=torch.float64, device=dev)   # per-chunk partial sums
            check(_L().sfron_groupnorm_fwd(ptr(x.t), x.C, gam, bet, x.B, x.H * x.W, x.C, 32, eps, int(swish), ptr(drop_mask), scale, ptr(y),
                                           ptr(mean), ptr(rstd), ptr(ws), stream_ptr()), "groupnorm_fwd")

        def bwd(dy, extra=None, ld_extra=0, src=None):
            """dy: fp32 [rows][C]; extra: fp32 [rows][ld_extra], one more term of x's gradient (the residual branch's) added in
            the same pass instead of a separate x.grad (+)= extra.  src: dy has not been written yet (conv backward with defer=True)."""
            gbuf, acc = x.grad_buf()
            pg = torch.empty(x.B, x.C, dtype=torch.float32, device=dev)
            pb = torch.empty_like(pg)
            src = self._finish_or_fuse(src, dy, x.B, x.H * x.W, x.C)
            if src is not None and (extra is None or (ld_extra % 4 == 0 and extra.data_ptr() % 16 == 0)):
                check(_L().sfron_groupnorm_bwd_res_src(ctypes.byref(src), ptr(dy), ptr(x.t), x.C, gam, bet, ptr(mean), ptr(rstd), x.B, x.H * x.W, x.C, 32,
                                                       int(swish), ptr(drop_mask), scale, ptr(gbuf), x.C, acc, ptr(extra), ld_extra, ptr(pg), ptr(pb),
                                                       stream_ptr()), "groupnorm_bwd_src")
            else:
                if src is not None:
                    check(_L().sfron_split_finish(ctypes.byref(src), x.rows, x.C, src._rows_per_sample, ptr(dy), None, x.C, stream_ptr()), "split_finish")
                ws2 = torch.empty(_L().sfron_groupnorm_scratch_bytes(x.B, x.H * x.W, x.C, 32) // 8 + 2, dtype=torch.float64, device=dev)
                check(_L().sfron_groupnorm_bwd_res(ptr(dy), ptr(x.t), x.C, gam, bet, ptr(mean), ptr(rstd), x.B, x.H * x.W, x.C, 32, int(swish),
                                                   ptr(drop_mask), scale, ptr(gbuf), x.C, acc, ptr(extra), ld_extra, ptr(pg), ptr(pb), ptr(ws2),
                                                   stream_ptr()), "groupnorm_bwd")
            if trains:
                self._reduce(pg, 1, x.B, x.C, self._g(name + ".weight"), x.C)
                self._reduce(pb, 1, x.B, x.C, self._g(name + ".bias"), x.C)

        def bwd_cast(dy, src=None):
            """For an x that only this norm consumes and that a convolution produced: x's gradient as that convolution's bf16 d_out
            operand plus its column sums per (sample, pixel chunk) -- (d_x bf16 [rows][C], partials fp32 [B][chunks][C], chunks) --
            without an fp32 x.grad; None when the shape is not eligible (the caller then takes bwd and casts; a pending dy is finished first)."""
            if not _L().sfron_groupnorm_bwd_cast_ok(x.C, x.C, 32):
                if src is not None:
                    check(_L().sfron_split_finish(ctypes.byref(src), x.rows, x.C, src._rows_per_sample, ptr(dy), None, x.C, stream_ptr()), "split_finish")
                return None
            src = self._finish_or_fuse(src, dy, x.B, x.H * x.W, x.C)
            nch = _L().sfron_groupnorm_chunks(x.B, x.H * x.W)
            d16 = torch.empty(x.rows, x.C, dtype=torch.bfloat16, device=dev)
            cpart = torch.empty(x.B * nch, x.C, dtype=torch.float32, device=dev)
            pg = torch.empty(x.B, x.C, dtype=torch.float32, device=dev)
            pb = torch.empty_like(pg)
            if src is not None:
                check(_L().sfron_groupnorm_bwd_cast_src(ctypes.byref(src), ptr(dy), ptr(x.t), x.C, gam, bet, ptr(mean), ptr(rstd), x.B, x.H * x.W, x.C,
                                                        32, int(swish), ptr(drop_mask), scale, ptr(d16), ptr(cpart), ptr(pg), ptr(pb), stream_ptr()),
                      "groupnorm_bwd_cast_src")
            else:
                ws2 = torch.empty(_L().sfron_groupnorm_scratch_bytes(x.B, x.H * x.W, x.C, 32) // 8 + 2, dtype=torch.float64, device=dev)
                check(_L().sfron_groupnorm_bwd_cast(ptr(dy), ptr(x.t), x.C, gam, bet, ptr(mean), ptr(rstd), x.B, x.H * x.W, x.C, 32, int(swish),
                                                    ptr(drop_mask), scale, ptr(d16), ptr(cpart), ptr(pg), ptr(pb), ptr(ws2), stream_ptr()),
                      "groupnorm_bwd_cast")
            if trains:
                self._reduce(pg, 1, x.B, x.C, self._g(name + ".weight"), x.C)
                self._reduce(pb, 1, x.B, x.C, self._g(name + ".bias"), x.C)
            return d16, cpart, nch
        bwd.cast = bwd_cast
        return y, bwd

    FUSE_SPLIT_FINISH = True     # False (tools / tests): every split convolution finishes its own output, as before round 6

    def _conv3(self, src, B, hs, ws, name, ho, wo, stride=1, pad=1, up=0, vec=None, ld_vec=0, resid=None, defer_finish=False):
        """3x3 convolution of a bf16 NHWC source -> fp32 rows [B*ho*wo][Cout_p]; returns (out, backward(d_out fp32) -> d_src fp32
        at the source resolution, or None for the input layer).  defer_finish (the caller hands `out` to a GroupNorm next): returns
        (out, backward, source) -- source = the unfinished split-K result `out` is to be formed from (split_source), or None."""
        v = self.conv3[name]
        dev = self.device_
        rows = B * ho * wo
        out = torch.empty(rows, v["cop"], dtype=torch.float32, device=dev)
        if v["cop"] == v["co"]:
            bias = self._p(name + ".bias")
        else:                                        # conv_out: 3 output channels computed as 8 (zero rows / zero bias beyond Cout)
            bias = torch.zeros(v["cop"], dtype=torch.float32, device=dev)
            bias[:v["co"]] = self.view(self.params, name + ".bias")
        pend = ctypes.c_int(0) if defer_finish and self.FUSE_SPLIT_FINISH else None
        d = _conv_desc(B, hs, ws, v["cip"], ho, wo, v["cop"], 9, stride, pad, up, 0, bias=bias, resid=resid, vec=vec, ld_vec=ld_vec,
                       out_f32=out, ld_out=v["cop"], pending=pend)
        check(_L().sfron_conv_fwd(ctypes.byref(d), ptr(src), ptr(v["fwd"]), stream_ptr()), "conv_fwd")
        out_src = split_source(d, pend, rows, v["cop"], keep=(bias, resid, vec))

        trains = self._trains(name + ".weight")          # decided when the tape is built, not when it runs

        def bwd(d_out, want_dsrc=True, d_bf=None, defer=False):
            """d_bf: d_out as the bf16 operand [rows][Cout_p] when the caller has it already (then d_out is not read and the bias
            gradient is the caller's: _resblock forms both in the GroupNorm backward, sfron_groupnorm_bwd_cast).
            defer (the caller hands the result to a GroupNorm backward next): returns (d_src, source) as _conv3(defer_finish=True)."""
            if d_bf is not None:
                dyb = d_bf
            elif trains and v["cop"] == v["co"] and v["co"] % 4 == 0:
                dyb = self._cast_rows_colsum(d_out, v["cop"], rows, v["cop"], self._g(name + ".bias"))     # bf16 operand + bias gradient
            else:
                dyb = cast_rows(d_out, v["cop"], rows, v["cop"], dev)
            if not trains or d_bf is not None:
                pass                                     # frozen kernel: only the input gradient below
            elif v["cop"] == v["co"] and v["co"] % 4 == 0:
                pass
            elif v["cop"] == v["co"]:
                colsum_f32(d_out, rows, v["co"], v["cop"], self._g(name + ".bias"), self._cs)
            else:
                bg = torch.empty(v["cop"], dtype=torch.float32, device=dev)
                colsum_f32(d_out, rows, v["cop"], v["cop"], bg, self._cs)
                self.view(self.grads, name + ".bias").copy_(bg[:v["co"]])
            if trains:
                wd = _conv_desc(B, hs, ws, v["cip"], ho, wo, v["cop"], 9, stride, pad, up, 0)
                nsl = _L().sfron_conv_wgrad_splits(ctypes.byref(wd))
                slab = v["cop"] * 9 * v["cip"]
                if self._red is not None and nsl * slab * 4 <= self.SCATTER_BATCH_MAX_BYTES:
                    # inside a backward pass: the slabs stay in a buffer of this layer's own and their scatter into the OIHW gradient joins the
                    # pass's one scatter launch (~100 launches of ~7 us fewer per DDPM step; the price: 1.2 GB of slabs at DDPM batch 64)
                    buf = v.get("dwbuf")
                    if buf is None or buf.numel() < nsl * slab:
                        if buf is not None:
                            self._dw_retired.append(buf)   # a captured stage graph may still write through the old address
                        buf = v["dwbuf"] = torch.empty(nsl * slab, dtype=torch.float32, device=dev)
                    check(_L().sfron_conv_wgrad(ctypes.byref(wd), ptr(dyb), v["cop"], ptr(src), ptr(buf), stream_ptr()), "conv_wgrad")
                    self._scat.append((buf, self._g(name + ".weight"), slab, v["co"], v["ci"], 9, v["cip"], nsl))
                else:
                    if self._dw.numel() < nsl * slab:
                        self._dw_retired.append(self._dw)   # a captured stage graph may still write through the old address
                        self._dw = torch.empty(nsl * slab, dtype=torch.float32, device=dev)
                    check(_L().sfron_conv_wgrad(ctypes.byref(wd), ptr(dyb), v["cop"], ptr(src), ptr(self._dw), stream_ptr()), "conv_wgrad")
                    check(_L().sfron_conv_wgrad_scatter(ptr(self._dw), v["co"], v["ci"], 9, v["cip"], nsl, slab, self._g(name + ".weight"),
                                                        stream_ptr()), "conv_wgrad_scatter")
            if not want_dsrc or v["dgr"] is None:
                return (None, None) if defer else None
            if stride == 2:        # Downsample: gradient = flipped kernel over the zero-dilated dY, padding 2 - pad
                ds = torch.empty(B * hs * ws, v["ci"], dtype=torch.float32, device=dev)
                dd = _conv_desc(B, ho, wo, v["cop"], hs, ws, v["ci"], 9, 1, 2 - pad, 0, 1, out_f32=ds, ld_out=v["ci"])
                check(_L().sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(v["dgr"]), stream_ptr()), "conv_dgrad")
                return (ds, None) if defer else ds
            if up:                 # Upsample (:56-60): gradient wrt the upsampled image, summed over each 2x2 block
                du = torch.empty(rows, v["ci"], dtype=torch.float32, device=dev)
                dd = _conv_desc(B, ho, wo, v["cop"], ho, wo, v["ci"], 9, 1, 1, 0, 0, out_f32=du, ld_out=v["ci"])
                check(_L().sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(v["dgr"]), stream_ptr()), "conv_dgrad")
                ds = torch.empty(B * hs * ws, v["ci"], dtype=torch.float32, device=dev)
                check(_L().sfron_pool2_sum(ptr(du), B, hs, ws, v["ci"], ptr(ds), 0, stream_ptr()), "pool2_sum")
                return (ds, None) if defer else ds
            ds = torch.empty(rows, v["ci"], dtype=torch.float32, device=dev)
            pend_b = ctypes.c_int(0) if defer and self.FUSE_SPLIT_FINISH else None
            dd = _conv_desc(B, ho, wo, v["cop"], ho, wo, v["ci"], 9, 1, 1, 0, 0, out_f32=ds, ld_out=v["ci"], pending=pend_b)
            check(_L().sfron_conv_fwd(ctypes.byref(dd), ptr(dyb), ptr(v["dgr"]), stream_ptr()), "conv_dgrad")
            return (ds, split_source(dd, pend_b, rows, v["ci"])) if defer else ds
        return (out, bwd, out_src) if defer_finish else (out, bwd)

    def _linear(self, x_bf, rows, name, cin, cout, out=None, ldc=None, resid=None, x_lda=None, w_ptr=None, b_ptr=None, g_w=None, g_b=None,
                bias=True):
        """out fp32 [rows][cout] = x_bf [rows][cin] W^T + b (+ resid).  Returns (out, backward(d_out fp32 tensor/addr, ld) ->
        d_x fp32 [rows][cin])."""
        dev = self.device_
        if out is None:
            out = torch.empty(rows, cout, dtype=torch.float32, device=dev)
            ldc = cout
        w = w_ptr if w_ptr is not None else self._w(name + ".weight")
        b = (b_ptr if b_ptr is not None else self._p(name + ".bias")) if bias else None
        lda = x_lda if x_lda is not None else cin
        bgemm(x_bf, w, rows, cout, cin, lda=lda, ldb=cin, bias=b, c_f32=out, ldc=ldc, resid=resid)
        gw = g_w if g_w is not None else self._g(name + ".weight")
        gb = (g_b if g_b is not None else self._g(name + ".bias")) if bias else None

        frozen = name is not None and not self._trains(name + ".weight")      # decided when the tape is built

        def bwd(d_out, ld_d, want_dx=True, d_bf=None, dx_bf16=False):
            """d_out fp32 [rows][ld_d] (or None when only the bf16 form d_bf [rows][cout] exists).  dx_bf16: the input gradient
            leaves the product as bf16 [rows][cin] (its consumer is another product's operand: the rounding a cast pass would do)."""
            fused = d_bf is None and bias and not frozen and cout % 4 == 0 and ld_d % 4 == 0 and not isinstance(d_out, int)
            if fused:
                d_bf = self._cast_rows_colsum(d_out, ld_d, rows, cout, gb)
            elif d_bf is None:
                d_bf = cast_rows(d_out, ld_d, rows, cout, dev)
            if frozen:
                pass                                     # frozen layer: only the input gradient below
            else:
                if fused:
                    pass
                elif bias and d_out is not None:
                    colsum_f32(d_out, rows, cout, ld_d, gb, self._cs)
                elif bias:
                    colsum_bf16(d_bf, rows, cout, cout, gb, self._cs)
                bgemm(d_bf, x_bf, cout, cin, rows, lda=cout, ldb=lda, a_t=True, b_t=True, c_f32=gw, ldc=cin)
            if not want_dx:
                return None
            if dx_bf16:
                dx = torch.empty(rows, cin, dtype=torch.bfloat16, device=dev)
                bgemm(d_bf, w, rows, cin, cout, lda=cout, ldb=cin, b_t=True, c_bf16=dx, ldc=cin)
                return dx
            dx = torch.empty(rows, cin, dtype=torch.float32, device=dev)
            bgemm(d_bf, w, rows, cin, cout, lda=cout, ldb=cin, b_t=True, c_f32=dx, ldc=cin)
            return dx
        return out, bwd

    # sub-module names of a residual block: (norm1, conv1, norm2, conv2, 1x1 shortcut); the DDPM model's, overridden by the LDM one
    RES_NAMES = (".norm1", ".conv1", ".norm2", ".conv2", ".nin_shortcut")

    def _resblock(self, tape, name, x, cin, cout, proj, d_proj, drop_mask):
        """GroupNorm -> swish -> conv3x3 (+ per-sample embedding projection) -> GroupNorm -> swish [-> dropout] -> conv3x3, plus the
        (1x1-projected) input (DDPM ResnetBlock, models/diffusion.py:85-145; LDM ResBlock, openaimodel.py:177-288)."""
        n_norm1, n_conv1, n_norm2, n_conv2, n_short = (name + sfx for sfx in self.RES_NAMES)
        dev, B, H, W = self.device_, x.B, x.H, x.W
        a1, gn1_b = self._gn(tape, x, n_norm1, True)
        c0, _ = self.proj_slices[name]
        v1 = self.conv3[n_conv1]
        fuse1 = v1["cop"] == v1["co"] == cout           # h1 has one consumer, norm2, which then also finishes conv1's split-K sum (if any)
        if fuse1:
            h1_t, conv1_b, h1_src = self._conv3(a1, B, H, W, n_conv1, H, W, vec=proj.data_ptr() + 4 * c0, ld_vec=self.proj_total, defer_finish=True)
        else:
            (h1_t, conv1_b), h1_src = self._conv3(a1, B, H, W, n_conv1, H, W, vec=proj.data_ptr() + 4 * c0, ld_vec=self.proj_total), None
        h1 = Act(h1_t, B, H, W, cout)
        a2, gn2_b = self._gn(tape, h1, n_norm2, True, drop_mask, src=h1_src)
        if cin != cout:
            xb = cast_rows(x.t, cin, x.rows, cin, dev)
            sc, sc_b = self._linear(xb, x.rows, n_short, cin, cout)
        else:
            sc, sc_b = x.t, None
        out_t, conv2_b = self._conv3(a2, B, H, W, n_conv2, H, W, resid=sc)
        out = Act(out_t, B, H, W, cout)

        trains1 = self._trains(n_conv1 + ".weight")       # decided when the tape is built, as in _conv3
        v2 = self.conv3[n_conv2]
        defer2 = v2["ci"] == cout and v2["ci"] % 4 == 0   # conv2's input gradient is [rows][cout] contiguous: norm2's backward may finish it
        defer1 = v1["ci"] == cin and cin % 4 == 0
        use_cast = v1["cop"] == v1["co"] == cout and bool(_L().sfron_groupnorm_bwd_cast_ok(cout, cout, 32))

        def bwd():
            d_out = out.grad
            # (the shortcut first: a pending split result lives in the stream's shared slab scratch, which the next split product overwrites --
            # nothing may run between a deferring convolution and the GroupNorm that finishes it)
            d_skip = d_out if sc_b is None else sc_b(d_out, cout)      # the shortcut's share of x.grad: added by norm1's backward pass below
            d_a2, a2_src = conv2_b(d_out, defer=True) if defer2 else (conv2_b(d_out), None)
            fused = gn2_b.cast(d_a2, src=a2_src) if use_cast else None       # (eligible shape: never None)
            if fused is not None:
                # h1 = conv1(.) + proj has one consumer (norm2): its gradient leaves norm2's backward pass as conv1's bf16 d_out,
                # with column sums per (sample, chunk) that finish as d_proj's slice (per sample) and conv1's bias gradient (all)
                dh1_bf, cpart, nch = fused
                self._reduce(cpart, B, nch, cout, d_proj.data_ptr() + 4 * c0, self.proj_total)
                if trains1:
                    self._reduce(cpart, 1, B * nch, cout, self._g(n_conv1 + ".bias"), cout)
                d_a1, a1_src = conv1_b(None, d_bf=dh1_bf, defer=True) if defer1 else (conv1_b(None, d_bf=dh1_bf), None)
            else:
                gn2_b(d_a2, src=a2_src)                   # -> h1.grad
                dh1 = h1.grad
                check(_L().sfron_sample_colsum(ptr(dh1), cout, B, H * W, cout, d_proj.data_ptr() + 4 * c0, self.proj_total, ptr(self._cs),
                                               self._cs.numel(), stream_ptr()), "sample_colsum")
                d_a1, a1_src = conv1_b(dh1, defer=True) if defer1 else (conv1_b(dh1), None)
            gn1_b(d_a1, d_skip, cin, src=a1_src)          # -> x.grad (+)= d_skip + norm1's gradient
        tape.append(bwd)
        return out


class _GuidedFn(torch.autograd.Function):
    """mode="test" with gradients: out = (1 + s) f(x, t, c | keep all) - s f(x, t, c | drop all) through BOTH branches, as the
    reference's _forward_with_cond_scale (DDPM/models/diffusion.py:340-357), which its Fisher loop back-propagates through
    (runners/diffusion.py:1260-1276).  Two tapes over the same weights; the backward pass runs them one after the other with d_out
    scaled by (1 + s) and -s -- each overwrites the gradient arena, so the first branch's arena is parked and added back."""

    @staticmethod
    def forward(ctx, anchor, model, x, t, c, cond_scale):
        B, dev = x.shape[0], model.device_
        out_c, bwd_c = model._run(x, t, c, torch.ones(B, dtype=torch.uint8, device=dev), None, need_grad=True)
        if cond_scale == 0:                          # models/diffusion.py:349-350: the conditional branch alone
            ctx.model, ctx.bwd, ctx.s = model, (bwd_c, None), 0.0
            return out_c
        out_n, bwd_n = model._run(x, t, c, torch.zeros(B, dtype=torch.uint8, device=dev), None, need_grad=True)
        out = torch.empty_like(out_c)
        check(_L().sfron_axpby(ptr(out_c), ptr(out_n), 1.0 + cond_scale, -float(cond_scale), out.numel(), ptr(out), stream_ptr()), "axpby")
        ctx.model, ctx.bwd, ctx.s = model, (bwd_c, bwd_n), float(cond_scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        m, (bwd_c, bwd_n), s = ctx.model, ctx.bwd, ctx.s
        d_out = d_out.float().contiguous()
        if bwd_n is None:
            bwd_c(d_out)
        else:
            bwd_c(d_out * (1.0 + s))
            first = m.grads.clone()
            bwd_n(d_out * (-s))
            m.grads.add_(first)
        m.publish_grads()
        ctx.bwd = None
        return None, None, None, None, None, None


class Conditional_Model(_TapeNet):
    def __init__(self, config=None, *, ch=128, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), dropout=0.1,
                 in_channels=3, resolution=32, resamp_with_conv=True, n_classes=10, cond_drop_prob=0.1, device="cuda"):
        super().__init__()
        if config is not None:                       # reference constructor (models/diffusion.py:196-224)
            m, d = config.model, config.data
            ch, out_ch, ch_mult = m.ch, m.out_ch, tuple(m.ch_mult)
            num_res_blocks, attn_resolutions, dropout = m.num_res_blocks, tuple(m.attn_resolutions), m.dropout
            in_channels, resolution, resamp_with_conv = m.in_channels, d.image_size, m.resamp_with_conv
            n_classes, cond_drop_prob = d.n_classes, m.cond_drop_prob
            if getattr(m, "type", "simple") == "bayesian":
                raise NotImplementedError("model.type 'bayesian' (a logvar parameter) is not used by the unlearning configs")
        self.config = config
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise _lib.SfronError("Conditional_Model needs a GPU (no CPU fallback)")
        if not resamp_with_conv:
            raise NotImplementedError("cifar10_sfron.yml uses resamp_with_conv: True")
        if ch % 32 or ch * 4 != 512:
            # reference quirk (:93-110): ResnetBlock's cemb_channels stays at its default 512, so the class only runs at ch = 128
            raise ValueError("the reference Conditional_Model only runs with ch = 128 (temb_ch = cemb_ch = 512)")
        self.ch, self.out_ch, self.ch_mult = ch, out_ch, tuple(ch_mult)
        self.temb_ch = self.cemb_ch = ch * 4
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.n_classes = resolution, in_channels, n_classes
        self.attn_resolutions, self.dropout_p, self.cond_drop_prob = tuple(attn_resolutions), float(dropout), cond_drop_prob
        self._plan()
        self._alloc()
        self._register_views()
        self.reset_parameters()

    # -------------------------------------------------------------------------------------------- structure
    def _plan(self):
        """Walks the reference's construction order (:226-327) and records (a) the parameter list in named_parameters()
        order, (b) the block graph the forward pass executes."""
        ch, emb = self.ch, self.temb_ch
        P = []                                      # (name, shape)

        def lin(name, o, i):
            P.extend([(name + ".weight", (o, i)), (name + ".bias", (o,))])

        def conv(name, o, i, k):
            P.extend([(name + ".weight", (o, i, k, k)), (name + ".bias", (o,))])

        def gn(name, c):
            P.extend([(name + ".weight", (c,)), (name + ".bias", (c,))])

        def res(name, cin, cout):
            gn(name + ".norm1", cin); conv(name + ".conv1", cout, cin, 3); lin(name + ".temb_cemb_proj", cout, emb + 512)
            gn(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
            if cin != cout:
                conv(name + ".nin_shortcut", cout, cin, 1)
            self.res_blocks.append((name, cin, cout))

        def attn(name, c):
            gn(name + ".norm", c)
            for w in ("q", "k", "v", "proj_out"):
                conv(name + "." + w, c, c, 1)

        self.res_blocks = []
        P.append(("null_classes_emb", (ch,)))          # the root module's own parameter comes first in named_parameters()
        lin("temb.dense.0", emb, ch); lin("temb.dense.1", emb, emb)
        P.append(("classes_emb.weight", (self.n_classes, ch)))
        lin("cemb.dense.0", emb, ch); lin("cemb.dense.1", emb, emb)
        conv("conv_in", ch, self.in_channels, 3)
        res_now = self.resolution
        in_mult = (1,) + self.ch_mult
        self.down_plan, self.up_plan = [], []
        bin_ = None
        for lvl in range(self.num_resolutions):
            blocks = []
            bin_, bout = ch * in_mult[lvl], ch * self.ch_mult[lvl]
            has_attn = res_now in self.attn_resolutions
            for ib in range(self.num_res_blocks):
                res(f"down.{lvl}.block.{ib}", bin_, bout)
                bin_ = bout
                blocks.append((f"down.{lvl}.block.{ib}", f"down.{lvl}.attn.{ib}" if has_attn else None))
            # the reference registers the attn ModuleList after the block ModuleList: parameters follow in that order
            self.down_plan.append(dict(blocks=blocks, c=bin_, down=None))
            for _, a in blocks:
                if a:
                    attn(a, bin_)
            if lvl != self.num_resolutions - 1:
                conv(f"down.{lvl}.downsample.conv", bin_, bin_, 3)
                self.down_plan[-1]["down"] = f"down.{lvl}.downsample.conv"
                res_now //= 2
        res("mid.block_1", bin_, bin_); attn("mid.attn_1", bin_); res("mid.block_2", bin_, bin_)
        self.mid_c = bin_
        up_params = {}
        for lvl in reversed(range(self.num_resolutions)):
            blocks, saved = [], len(P)
            bout, skip = ch * self.ch_mult[lvl], ch * self.ch_mult[lvl]
            has_attn = res_now in self.attn_resolutions
            for ib in range(self.num_res_blocks + 1):
                if ib == self.num_res_blocks:
                    skip = ch * in_mult[lvl]
                res(f"up.{lvl}.block.{ib}", bin_ + skip, bout)
                bin_ = bout
                blocks.append((f"up.{lvl}.block.{ib}", f"up.{lvl}.attn.{ib}" if has_attn else None))
            for _, a in blocks:
                if a:
                    attn(a, bin_)
            plan = dict(blocks=blocks, c=bin_, up=None)
            if lvl != 0:
                conv(f"up.{lvl}.upsample.conv", bin_, bin_, 3)
                plan["up"] = f"up.{lvl}.upsample.conv"
                res_now *= 2
            self.up_plan.append((lvl, plan))
            up_params[lvl] = P[saved:]
            del P[saved:]
        for lvl in range(self.num_resolutions):      # self.up.insert(0, ...) (:316): the ModuleList is ordered by level
            P.extend(up_params[lvl])
        gn("norm_out", bin_); conv("conv_out", self.out_ch, bin_, 3)
        self.final_c = bin_
        # the up-path residual blocks were appended to res_blocks in execution order (levels descending): what forward needs
        self.param_specs = P

    def _alloc(self):
        """Arena order: all temb_cemb_proj weights (one stacked [sum Cout][1024] matrix: ONE projection GEMM for the whole
        network, the input swish(temb || cemb) being the same for every block), their biases, the q / k / v 1x1 convolutions of
        each AttnBlock as one [3C][C] matrix + [3C] bias, then everything else in named_parameters() order."""
        specs = OrderedDict(self.param_specs)
        proj_names = [n for n, _, _ in self.res_blocks]
        groups = [[n + ".temb_cemb_proj.weight" for n in proj_names], [n + ".temb_cemb_proj.bias" for n in proj_names]]
        self.attn_names = [n[:-len(".q.weight")] for n in specs if n.endswith(".q.weight")]
        for a in self.attn_names:
            groups.append([a + ".q.weight", a + ".k.weight", a + ".v.weight"])
            groups.append([a + ".q.bias", a + ".k.bias", a + ".v.bias"])
        self._alloc_arena(specs, groups)
        self.proj_w_off = self.index[proj_names[0] + ".temb_cemb_proj.weight"][0]
        self.proj_b_off = self.index[proj_names[0] + ".temb_cemb_proj.bias"][0]
        self.proj_slices, c0 = {}, 0
        for n, _, cout in self.res_blocks:
            self.proj_slices[n] = (c0, cout)
            c0 += cout
        self.proj_total = c0

    def reset_parameters(self):
        """torch's default initialisers of the layer types the reference builds (Linear / Conv2d: kaiming_uniform(a=sqrt 5) +
        fan-in bias; GroupNorm 1 / 0; Embedding N(0,1); null_classes_emb randn)."""
        import math
        with torch.no_grad():
            for name, p in self.named_parameters():
                base = name.rsplit(".", 1)[0]
                if name == "null_classes_emb" or name == "classes_emb.weight":
                    p.normal_()
                elif ".norm" in name or name.startswith("norm_out"):
                    p.fill_(1.0 if name.endswith(".weight") else 0.0)
                elif name.endswith(".weight"):
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                else:
                    w = self.view(self.params, base + ".weight")
                    fan_in = w[0].numel()
                    bound = 1 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound)
        self.sync_bf16()

    # -------------------------------------------------------------------------------------------- parameter access
    # -------------------------------------------------------------------------------------------- building blocks
    def _attn(self, tape, name, x):
        dev, B, C, T = self.device_, x.B, x.C, x.H * x.W
        rows = x.rows
        hn, gn_b = self._gn(tape, x, name + ".norm", False)
        qkv = torch.empty(rows, 3 * C, dtype=torch.bfloat16, device=dev)
        bgemm(hn, self._w(name + ".q.weight"), rows, 3 * C, C, lda=C, ldb=C, bias=self._p(name + ".q.bias"), c_bf16=qkv, ldc=3 * C)
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C
        S = torch.empty(B * T, T, dtype=torch.float32, device=dev)
        bgemm(q, k, T, T, C, lda=3 * C, ldb=3 * C, batch=B, sa=T * 3 * C, sb=T * 3 * C, sc=T * T, c_f32=S, ldc=T)
        Pm = torch.empty(B * T, T, dtype=torch.bfloat16, device=dev)
        scale = float(int(C) ** (-0.5))
        check(_L().sfron_softmax_fwd(ptr(S), B * T, T, T, scale, ptr(Pm), stream_ptr()), "softmax_fwd")
        O = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
        bgemm(Pm, v, T, C, T, lda=T, ldb=3 * C, b_t=True, batch=B, sa=T * T, sb=T * 3 * C, sc=T * C, c_bf16=O, ldc=C)
        out_t = torch.empty(rows, C, dtype=torch.float32, device=dev)
        bgemm(O, self._w(name + ".proj_out.weight"), rows, C, C, lda=C, ldb=C, bias=self._p(name + ".proj_out.bias"), c_f32=out_t, ldc=C,
              resid=x.t)
        out = Act(out_t, B, x.H, x.W, C)

        def bwd():
            q, k, v = qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C     # (the closure keeps qkv alive)
            d_out = out.grad
            d_bf = cast_rows(d_out, C, rows, C, dev)
            colsum_f32(d_out, rows, C, C, self._g(name + ".proj_out.bias"), self._cs)
            bgemm(d_bf, O, C, C, rows, lda=C, ldb=C, a_t=True, b_t=True, c_f32=self._g(name + ".proj_out.weight"), ldc=C)
            dO = torch.empty(rows, C, dtype=torch.bfloat16, device=dev)
            bgemm(d_bf, self._w(name + ".proj_out.weight"), rows, C, C, lda=C, ldb=C, b_t=True, c_bf16=dO, ldc=C)
            dP = torch.empty(B * T, T, dtype=torch.float32, device=dev)
            bgemm(dO, v, T, T, C, lda=C, ldb=3 * C, batch=B, sa=T * C, sb=T * 3 * C, sc=T * T, c_f32=dP, ldc=T)
            dS = torch.empty(B * T, T, dtype=torch.bfloat16, device=dev)
            check(_L().sfron_softmax_bwd(ptr(Pm), ptr(dP), B * T, T, scale, ptr(dS), stream_ptr()), "softmax_bwd")
            dqkv = torch.empty(rows, 3 * C, dtype=torch.bfloat16, device=dev)
            dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 2 * C, dqkv.data_ptr() + 4 * C
            bgemm(Pm, dO, T, C, T, lda=T, ldb=C, a_t=True, b_t=True, batch=B, sa=T * T, sb=T * C, sc=T * 3 * C, c_bf16=dv, ldc=3 * C)
            bgemm(dS, k, T, C, T, lda=T, ldb=3 * C, b_t=True, batch=B, sa=T * T, sb=T * 3 * C, sc=T * 3 * C, c_bf16=dq, ldc=3 * C)
            bgemm(dS, q, T, C, T, lda=T, ldb=3 * C, a_t=True, b_t=True, batch=B, sa=T * T, sb=T * 3 * C, sc=T * 3 * C, c_bf16=dk, ldc=3 * C)
            colsum_bf16(dqkv, rows, 3 * C, 3 * C, self._g(name + ".q.bias"), self._cs)
            bgemm(dqkv, hn, 3 * C, C, rows, lda=3 * C, ldb=C, a_t=True, b_t=True, c_f32=self._g(name + ".q.weight"), ldc=C)
            d_hn = torch.empty(rows, C, dtype=torch.float32, device=dev)
            bgemm(dqkv, self._w(name + ".q.weight"), rows, C, 3 * C, lda=3 * C, ldb=C, b_t=True, c_f32=d_hn, ldc=C)
            gn_b(d_hn, d_out, C)                          # x.grad (+)= d_out (residual) + the norm's gradient
        tape.append(bwd)
        return out

    # -------------------------------------------------------------------------------------------- forward / backward
    def _draw_keep(self, b, p):
        if p is None or p <= 0:
            return None
        q = 1.0 - p
        if q >= 1:
            return None
        if q <= 0:
            return torch.zeros(b, dtype=torch.uint8, device=self.device_)
        return (torch.rand(b, device=self.device_) < q).to(torch.uint8)

    def _run(self, x, t, c, keep_mask, dropout_masks, need_grad):
        L, dev = _L(), self.device_
        B = x.shape[0]
        ch, emb = self.ch, self.temb_ch
        tape = []
        self._prep_conv_weights()
        # ---- timestep / class embeddings (:365-379)
        te = torch.empty(B, ch, dtype=torch.bfloat16, device=dev)
        check(L.sfron_ddpm_timestep_embed(ptr(t.float().contiguous()), B, ch, ptr(te), stream_ptr()), "ddpm_timestep_embed")
        ce = torch.empty(B, ch, dtype=torch.bfloat16, device=dev)
        c = c.contiguous()
        check(L.sfron_class_embed_fwd(self._p("classes_emb.weight"), self._p("null_classes_emb"), ptr(c), ptr(keep_mask), self.n_classes, B, ch,
                                      ptr(ce), stream_ptr()), "class_embed_fwd")
        cat = torch.empty(B, 2 * emb, dtype=torch.float32, device=dev)       # temb || cemb
        mlp_b = []
        for which, src, col in (("temb", te, 0), ("cemb", ce, emb)):
            h0, b0 = self._linear(src, B, which + ".dense.0", ch, emb)
            h0s = torch.empty(B, emb, dtype=torch.bfloat16, device=dev)
            check(L.sfron_silu_fwd(ptr(h0), B * emb, ptr(h0s), stream_ptr()), "silu_fwd")
            _, b1 = self._linear(h0s, B, which + ".dense.1", emb, emb, out=cat.data_ptr() + 4 * col, ldc=2 * emb)
            mlp_b.append((which, b0, b1, h0, col))
        sact = torch.empty(B, 2 * emb, dtype=torch.bfloat16, device=dev)
        check(L.sfron_silu_fwd(ptr(cat), B * 2 * emb, ptr(sact), stream_ptr()), "silu_fwd")
        PT = self.proj_total
        proj = torch.empty(B, PT, dtype=torch.float32, device=dev)           # temb_cemb_proj of every ResnetBlock, one GEMM
        bgemm(sact, self.params_bf16.data_ptr() + 2 * self.proj_w_off, B, PT, 2 * emb, lda=2 * emb, ldb=2 * emb,
              bias=self.params.data_ptr() + 4 * self.proj_b_off, c_f32=proj, ldc=PT)
        d_proj = torch.zeros(B, PT, dtype=torch.float32, device=dev) if need_grad else None     # filled slice by slice in backward

        masks = iter(dropout_masks) if dropout_masks is not None else None

        def next_mask(rows, C):
            if masks is not None:
                m = next(masks)
                return None if m is None else m.to(device=dev, dtype=torch.uint8).contiguous()
            if self.training and self.dropout_p > 0:
                salt[0] += 1
                if drop_plan is not None:                 # drawn by the one launch at the top of the pass
                    psalt, n, off = drop_plan["items"][salt[0] - 1]
                    if (psalt, n) != (salt[0], rows * C):
                        raise RuntimeError("the dropout plan of this batch size does not match the pass (model changed after its first pass?)")
                    return drawn[off:off + n].view(rows, C)
                m = torch.empty(rows, C, dtype=torch.uint8, device=dev)
                check(L.sfron_dropout_mask(self._drop_seed, ptr(self._drop_counter), salt[0], rows * C, self.dropout_p, ptr(m), stream_ptr()),
                      "dropout_mask")
                recorded.append((salt[0], rows * C))
                return m
            return None

        salt = [0]
        drop_plan, drawn, recorded = None, None, []
        if getattr(self, "_drop_plans", None) is None:
            self._drop_plans = {}
        if masks is None and self.training and self.dropout_p > 0:
            if getattr(self, "_drop_counter", None) is None:
                self._drop_seed = torch.initial_seed() & ((1 << 63) - 1)           # follows torch.manual_seed
                self._drop_counter = torch.zeros(1, dtype=torch.int64, device=dev)
            self._drop_counter.add_(1)                                             # one device-side tick per pass (graph-replay safe)
            # the first pass at a batch size draws mask by mask and records what it asked for; later passes draw all of them in ONE launch
            # (the same bits: sfron_dropout_mask_batch)
            drop_plan = self._drop_plans.get(B)
            if drop_plan is not None:
                drawn = torch.empty(drop_plan["bytes"], dtype=torch.uint8, device=dev)
                check(L.sfron_dropout_mask_batch(self._drop_seed, ptr(self._drop_counter), ptr(drop_plan["table"]), len(drop_plan["items"]), drop_plan["max_n"],
                                                 self.dropout_p, ptr(drawn), stream_ptr()), "dropout_mask_batch")

        # ---- input
        S = self.resolution
        cip = self.conv3["conv_in"]["cip"]
        xr = torch.empty(B * S * S, cip, dtype=torch.bfloat16, device=dev)
        check(L.sfron_nchw_to_rows_bf16(ptr(x.float().contiguous()), B, self.in_channels, S * S, cip, ptr(xr), stream_ptr()), "nchw_to_rows")
        h0_t, conv_in_b = self._conv3(xr, B, S, S, "conv_in", S, S)
        hs = [Act(h0_t, B, S, S, ch)]
        tape.append(lambda a=hs[0]: conv_in_b(a.grad, want_dsrc=False))
        # ---- down path (:381-392)
        res = S
        for lvl, plan in enumerate(self.down_plan):
            for blk, att in plan["blocks"]:
                _, cin, cout = next(r for r in self.res_blocks if r[0] == blk)
                h = self._resblock(tape, blk, hs[-1], cin, cout, proj, d_proj, next_mask(hs[-1].rows, cout))
                if att:
                    h = self._attn(tape, att, h)
                hs.append(h)
            if plan["down"]:
                src = hs[-1]
                sb = cast_rows(src.t, src.C, src.rows, src.C, dev)
                o_t, db = self._conv3(sb, B, res, res, plan["down"], res // 2, res // 2, stride=2, pad=0)
                res //= 2
                o = Act(o_t, B, res, res, src.C)
                hs.append(o)

                def down_bwd(o=o, src=src, db=db):
                    ds = db(o.grad)
                    g, acc = src.grad_buf()
                    check(L.sfron_copy_cols(ptr(ds), src.C, src.rows, src.C, ptr(g), src.C, acc, stream_ptr()), "copy_cols")
                tape.append(down_bwd)
        # ---- middle (:394-398)
        C = self.mid_c
        h = self._resblock(tape, "mid.block_1", hs[-1], C, C, proj, d_proj, next_mask(hs[-1].rows, C))
        h = self._attn(tape, "mid.attn_1", h)
        h = self._resblock(tape, "mid.block_2", h, C, C, proj, d_proj, next_mask(h.rows, C))
        # ---- up path (:400-409)
        for lvl, plan in self.up_plan:
            for blk, att in plan["blocks"]:
                _, cin, cout = next(r for r in self.res_blocks if r[0] == blk)
                skip = hs.pop()
                c1, c2 = h.C, skip.C
                cat_t = torch.empty(h.rows, c1 + c2, dtype=torch.float32, device=dev)
                check(L.sfron_copy_cols2(ptr(h.t), c1, c1, ptr(cat_t), c1 + c2, 0, ptr(skip.t), c2, c2, cat_t.data_ptr() + 4 * c1, c1 + c2, 0, h.rows,
                                           stream_ptr()), "copy_cols2")
                ca = Act(cat_t, B, h.H, h.W, c1 + c2)

                def cat_bwd(ca=ca, a=h, s=skip, c1=c1, c2=c2):
                    ga, acc_a = a.grad_buf()
                    gs, acc_s = s.grad_buf()
                    check(L.sfron_copy_cols2(ptr(ca.grad), c1 + c2, c1, ptr(ga), c1, acc_a, ca.grad.data_ptr() + 4 * c1, c1 + c2, c2, ptr(gs), c2, acc_s,
                                               a.rows, stream_ptr()), "copy_cols2")
                tape.append(cat_bwd)
                h = self._resblock(tape, blk, ca, cin, cout, proj, d_proj, next_mask(ca.rows, cout))
                if att:
                    h = self._attn(tape, att, h)
            if plan["up"]:
                src = h
                sb = cast_rows(src.t, src.C, src.rows, src.C, dev)
                o_t, ub = self._conv3(sb, B, src.H, src.W, plan["up"], 2 * src.H, 2 * src.W, up=1)
                h = Act(o_t, B, 2 * src.H, 2 * src.W, src.C)

                def up_bwd(o=h, src=src, ub=ub):
                    ds = ub(o.grad)
                    g, acc = src.grad_buf()
                    check(L.sfron_copy_cols(ptr(ds), src.C, src.rows, src.C, ptr(g), src.C, acc, stream_ptr()), "copy_cols")
                tape.append(up_bwd)
        # ---- output (:411-413)
        a, gn_b = self._gn(tape, h, "norm_out", True)
        v = self.conv3["conv_out"]
        o_t, co_b = self._conv3(a, B, h.H, h.W, "conv_out", h.H, h.W)
        out = torch.empty(B, self.out_ch, h.H, h.W, dtype=torch.float32, device=dev)
        check(L.sfron_rows_to_nchw(ptr(o_t), v["cop"], B, self.out_ch, h.H * h.W, ptr(out), stream_ptr()), "rows_to_nchw")
        # first pass at this batch size: remember what the pass asks for.  Never while a stream is capturing (graphs.StageGraph re-captures
        # without a warm-up pass when the input signature changes, e.g. a new batch size): the table's host-to-device copy is illegal
        # there, and a table allocated inside the capture would live in the graph's private pool, which other stages overwrite.
        # Such a pass keeps its per-mask launches (all capturable); the next eager pass at this batch size builds the plan.
        if recorded and drop_plan is None and not torch.cuda.is_current_stream_capturing():
            items, off = [], 0
            for sl_, n in recorded:
                items.append((sl_, n, off))
                off += (n + 3) // 4 * 4
            self._drop_plans[B] = dict(items=items, bytes=off, max_n=max(n for _, n in recorded),
                                       table=torch.tensor([v for it in items for v in it], dtype=torch.int64, device=dev))
        if not need_grad:
            return out, None

        def backward(d_out):
            dr = torch.empty(B * h.H * h.W, v["cop"], dtype=torch.float32, device=dev)
            check(L.sfron_nchw_to_rows_f32(ptr(d_out.float().contiguous()), B, self.out_ch, h.H * h.W, v["cop"], ptr(dr), stream_ptr()),
                  "nchw_to_rows_f32")
            self._reduce_begin()
            try:
                gn_b(co_b(dr))
                hook = getattr(self, "_tape_hook", None)           # debugging aid: called after every backward step
                for i, step in enumerate(reversed(tape)):
                    step()
                    if hook is not None:
                        hook(i, step)
            except BaseException:
                self._red = self._scat = None
                raise
            self._reduce_flush()                                   # every collected parameter-gradient finish, d_proj's slices included
            # ---- temb_cemb_proj of all blocks, then the two embedding MLPs
            dp = d_proj
            dpb = cast_rows(dp, PT, B, PT, dev)
            colsum_f32(dp, B, PT, PT, self.grads.data_ptr() + 4 * self.proj_b_off, self._cs)
            bgemm(dpb, sact, PT, 2 * emb, B, lda=PT, ldb=2 * emb, a_t=True, b_t=True, c_f32=self.grads.data_ptr() + 4 * self.proj_w_off, ldc=2 * emb)
            d_sact = torch.empty(B, 2 * emb, dtype=torch.float32, device=dev)
            bgemm(dpb, self.params_bf16.data_ptr() + 2 * self.proj_w_off, B, 2 * emb, PT, lda=PT, ldb=2 * emb, b_t=True, c_f32=d_sact, ldc=2 * emb)
            d_cat = torch.empty(B, 2 * emb, dtype=torch.float32, device=dev)
            d_cat_bf = torch.empty(B, 2 * emb, dtype=torch.bfloat16, device=dev)
            check(L.sfron_silu_bwd(ptr(d_sact), ptr(cat), B * 2 * emb, ptr(d_cat_bf), ptr(d_cat), stream_ptr()), "silu_bwd")
            self.view(self.grads, "classes_emb.weight").zero_()        # the label scatter below accumulates
            for which, b0, b1, h0, col in mlp_b:
                # dense.1: input h0s = silu(h0); its d_out is the column slice [col, col + emb) of d_cat
                d1 = torch.empty(B, emb, dtype=torch.float32, device=dev)
                check(L.sfron_copy_cols(d_cat.data_ptr() + 4 * col, 2 * emb, B, emb, ptr(d1), emb, 0, stream_ptr()), "copy_cols")
                d_h0s = b1(d1, emb)
                d_h0 = torch.empty(B, emb, dtype=torch.float32, device=dev)
                check(L.sfron_silu_bwd(ptr(d_h0s), ptr(h0), B * emb, None, ptr(d_h0), stream_ptr()), "silu_bwd")
                d_in = b0(d_h0, emb, want_dx=(which == "cemb"))
                if which == "cemb":
                    check(L.sfron_class_embed_bwd(ptr(d_in), ptr(c), ptr(keep_mask), self.n_classes, B, ch, self._g("classes_emb.weight"),
                                                  self._g("null_classes_emb"), stream_ptr()), "class_embed_bwd")
        return out, backward

    def _forward(self, x, t, c, cond_drop_prob=None, keep_mask=None, dropout_masks=None):
        p = self.cond_drop_prob if cond_drop_prob is None else cond_drop_prob
        if keep_mask is None:
            keep_mask = self._draw_keep(x.shape[0], p)
        else:
            keep_mask = keep_mask.to(device=self.device_, dtype=torch.uint8).contiguous()
        if torch.is_grad_enabled():
            return _UNetFn.apply(self._anchor(), self, x, t, c, keep_mask, dropout_masks)
        out, _ = self._run(x, t, c, keep_mask, dropout_masks, need_grad=False)
        return out

    def _anchor(self):
        if not hasattr(self, "_anc"):
            self._anc = torch.zeros((), device=self.device_, requires_grad=True)
        return self._anc

    def forward(self, x, t, c, mode="train", **kwargs):
        assert mode in ("train", "test")
        if mode == "train":
            return self._forward(x, t, c, cond_drop_prob=kwargs.get("cond_drop_prob"), keep_mask=kwargs.get("keep_mask"),
                                 dropout_masks=kwargs.get("dropout_masks"))
        cond_scale = kwargs.get("cond_scale", 1.0)
        if torch.is_grad_enabled():
            # differentiable, as the reference's (models/diffusion.py:340-357): gradients flow through both guidance branches
            return _GuidedFn.apply(self._anchor(), self, x, t, c, float(cond_scale))
        B = x.shape[0]
        logits = self._forward(x, t, c, keep_mask=torch.ones(B, dtype=torch.uint8, device=self.device_))
        if cond_scale == 0:
            return logits
        null = self._forward(x, t, c, keep_mask=torch.zeros(B, dtype=torch.uint8, device=self.device_))
        out = torch.empty_like(logits)
        check(_L().sfron_axpby(ptr(logits), ptr(null), 1.0 + cond_scale, -float(cond_scale), logits.numel(), ptr(out), stream_ptr()), "axpby")
        return out


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, c, keep_mask, dropout_masks):
        out, bwd = model._run(x, t, c, keep_mask, dropout_masks, need_grad=True)
        ctx.model, ctx.bwd = model, bwd
        return out

    @staticmethod
    def backward(ctx, d_out):
        ctx.bwd(d_out)
        ctx.model.publish_grads()
        ctx.bwd = None
        return None, None, None, None, None, None, None


def config_namespace(**kw):
    """A config object with the reference's sections from keyword fields (handy for tests / scripts without the yml)."""
    model = dict(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[16],
                 dropout=0.1, resamp_with_conv=True, cond_drop_prob=0.1)
    data = dict(image_size=32, n_classes=10)
    diffusion = dict(num_diffusion_timesteps=1000)
    for k, v in kw.items():
        (model if k in model else data if k in data else diffusion)[k] = v
    return SimpleNamespace(model=SimpleNamespace(**model), data=SimpleNamespace(**data), diffusion=SimpleNamespace(**diffusion))
