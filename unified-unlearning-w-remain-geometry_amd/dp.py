"""Data-parallel plumbing: one process per GPU, replicated weights / optimizer state / EMA / mask, the two
gradient streams of an SFR-on iteration SUM-all-reduced over RCCL (torch.distributed backend "nccl").

The reference's multi-GPU path is single-process nn.DataParallel (DiT/forget.py:193): scatter the batch,
gather outputs, mean over the GLOBAL batch, reduce grads to GPU 0.  Here every rank scales its loss
gradient by 1/global_batch (csrc/loss.hip grad_scale) and sums, which is the same mean-over-global-batch
gradient up to summation order.  Backend-agnostic on purpose: the CPU tests drive it with gloo.
"""
import os
import sys
import threading
import time

import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_(t, group=None, transport="fp32", scratch=None):
    """In-place SUM all-reduce of one contiguous fp32 range.  transport "bf16": the range travels as bf16 (rounded to nearest
    even into ``scratch``, a bf16 tensor of at least t.numel() elements, summed by the backend, widened back) -- half the bytes on
    the xGMI links for a relative error of 2^-9 per addend; the optimizer still sees fp32 values.  Stream-ordered on the
    current stream, so one scratch buffer serves consecutive calls."""
    if transport == "fp32":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    if transport != "bf16":
        raise ValueError(f"unknown gradient transport {transport!r}")
    buf = (scratch if scratch is not None else torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device))[:t.numel()]
    buf.copy_(t)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    t.copy_(buf)


def allreduce_flat_(flat, bucket_elems, group=None, async_op=False, transport="fp32", scratch=None):
    """In-place SUM all-reduce of a flat tensor in buckets of `bucket_elems` elements.  xGMI is point-to-point
    (7 links/GPU), so buckets are large (default 256 MiB in step.py): few, bandwidth-bound collectives."""
    if world_size(group) == 1:
        return []
    works = []
    for s in range(0, flat.numel(), bucket_elems):
        if transport != "fp32":
            allreduce_(flat[s:s + bucket_elems], group, transport, scratch)
            continue
        w = dist.all_reduce(flat[s:s + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def allreduce_ranges_(flat, ranges, bucket_elems, group=None, transport="fp32", scratch=None):
    """SUM all-reduce of the element ranges [(lo, hi)] of a flat arena only (each in buckets): a run that trains a subset of the
    parameters (SD train_method "xattn": 44 M of 860 M) exchanges what its optimizer owns; the rest of the arena keeps this rank's
    values instead of being re-summed -- and growing by a factor of world -- on every exchange."""
    if world_size(group) == 1:
        return
    for lo, hi in ranges:
        allreduce_flat_(flat[lo:hi], bucket_elems, group, transport=transport, scratch=scratch)


def shard(global_batch, rank, world):
    """Contiguous shard of a dict of tensors whose dim 0 is the global batch."""
    n = next(iter(global_batch.values())).shape[0]
    assert n % world == 0, "global batch must divide evenly over ranks"
    per = n // world
    return {k: v[rank * per:(rank + 1) * per].contiguous() for k, v in global_batch.items()}


class Watchdog:
    """Deadline per phase of a multi-rank run: a collective that one rank never joins blocks the others until the backend's own
    timeout (minutes with RCCL's defaults) and then surfaces as an exception somewhere inside torch.  This thread instead ends
    THIS process with a non-zero code and one clear line as soon as a phase overruns its budget, so the launcher tears the
    job down.  ``phase(name, budget_s)`` (re)arms it, ``stop()`` disarms it.  Host-side only: no device work, no collectives."""

    EXIT_CODE = 124

    def __init__(self, rank=0, out=sys.stderr):
        self.rank, self.out = rank, out
        self._lock = threading.Lock()
        self._name, self._deadline = None, None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="sfron-dp-watchdog", daemon=True)
        self._thread.start()

    def phase(self, name, budget_s):
        with self._lock:
            self._name, self._deadline = name, time.monotonic() + float(budget_s)

    def stop(self):
        with self._lock:
            self._deadline = None
        self._stop.set()

    def _expired(self):
        with self._lock:
            return self._deadline is not None and time.monotonic() > self._deadline, self._name

    def _run(self):
        while not self._stop.wait(0.5):
            late, name = self._expired()
            if late:
                print(f"[sfron.dp] rank {self.rank}: watchdog -- phase '{name}' overran its budget (a collective that not every "
                      f"rank joined?); exiting {self.EXIT_CODE}", file=self.out, flush=True)
                os._exit(self.EXIT_CODE)
