"""Fail-loud checks of an unlearning step that do not stall the host.

The reference raises where something is off (a bad label is an IndexError inside nn.Embedding, DiT/models.py:89-93; a NaN
loss shows in its per-step log line, DiT/forget.py:329-336).  The fast path never reads a device value on the host, so the
checks are device-side counters (two tiny kernels per step, csrc/embed.hip) whose snapshot travels to a pinned host buffer behind a HIP
event; ``poll()`` looks at the snapshot of an EARLIER step once its event has completed -- no synchronisation -- and raises
``SfronError``.  ``poll(block=True)`` (end of a run, bench.py, tests) waits for the newest snapshot.
"""
import torch

from . import _lib
from ._lib import SfronError, check, ptr, stream_ptr

REASONS = ("non-finite loss", "non-finite gradient norm", "label outside [0, num_classes)", "timestep outside [0, num_timesteps)")


class StepGuard:
    def __init__(self, device):
        self.flags = torch.zeros(len(REASONS), dtype=torch.float32, device=device)
        self.host = torch.zeros(len(REASONS), dtype=torch.float32).pin_memory()
        self.event = None
        self.published_step = -1

    def note(self, idx, bad):
        """bad: 0-dim bool / number tensor on the device (True = violation).  Accumulates; never reset."""
        self.flags[idx] += bad.to(torch.float32)

    def check_inputs(self, y, t, num_classes, num_timesteps):
        """Labels / timesteps clamped into range (new tensors) + the violation counters, in ONE launch (csrc/embed.hip)."""
        # the kernel reads both as int64 vectors of one length (the reference indexes nn.Embedding / the schedule tables with
        # LongTensors: models.py:89-93, gaussian_diffusion.py:861-873): anything else is a caller error, not something to reinterpret
        if y.dtype != torch.int64 or t.dtype != torch.int64:
            raise SfronError(f"labels / timesteps must be int64 tensors (got {y.dtype}, {t.dtype})")
        if y.numel() != t.numel():
            raise SfronError(f"labels and timesteps differ in length ({y.numel()} vs {t.numel()})")
        y, t = y.contiguous(), t.contiguous()
        y_safe, t_safe = torch.empty_like(y), torch.empty_like(t)
        check(_lib.lib().sfron_guard_inputs(ptr(y), ptr(t), y.numel(), int(num_classes), int(num_timesteps), ptr(y_safe), ptr(t_safe),
                                            ptr(self.flags), stream_ptr()), "guard_inputs")
        return y_safe, t_safe

    def check_finite(self, terms, stats=None):
        """terms: up to four per-sample fp32 loss vectors of equal length; stats: the optimizer's clip statistics (or None)."""
        terms = [x.contiguous() for x in terms]
        assert 1 <= len(terms) <= 4 and all(x.numel() == terms[0].numel() and x.dtype == torch.float32 for x in terms)
        p = [ptr(x) for x in terms] + [None] * (4 - len(terms))
        check(_lib.lib().sfron_guard_finite(p[0], p[1], p[2], p[3], terms[0].numel(), ptr(stats), ptr(self.flags), stream_ptr()),
              "guard_finite")

    def publish(self, step_no):
        self.host.copy_(self.flags, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.published_step = step_no

    def poll(self, block=False):
        if self.event is None:
            return
        if block:
            self.event.synchronize()
        elif not self.event.query():
            return
        bad = [REASONS[i] for i in range(len(REASONS)) if float(self.host[i]) != 0.0]
        if bad:
            raise SfronError(f"SFR-on step guard (by step {self.published_step}): " + "; ".join(bad))
