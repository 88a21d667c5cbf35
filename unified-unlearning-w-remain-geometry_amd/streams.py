"""Process-wide named side streams, one set per device, each PROBED to run beside the streams it must not serialise with.

Every runner of this package overlaps work on a handful of extra streams (the optimizer sweep beside the next forward pass, the clip
norm's adaLN share beside the embedders' backward, the gradient exchange beside the backward pass, the two micro-batch chains).  The HIP
runtime maps streams onto FOUR hardware queues (per priority); two streams on one queue serialise, whatever events say.  Two failures of
"take the next torch.cuda.Stream()" were measured in round 6:
  * the THIRD runner of a process (bench.py's configuration legs) got a colliding pair and ran 4 ms / step slower than the same runner in a
    fresh process (profiles/r06_fp8_leg.txt) -- so a named stream is taken ONCE per (device, name) and every runner reuses it;
  * the data-parallel exchange stream landed on the hardware queue of the engine's second weight-gradient stream: a stand-in for the
    exchange's footprint slowed the step by exactly its own duration, i.e. the "overlapped" exchange did not overlap at all
    (profiles/r06_hw_queues.txt) -- so a named stream is CHOSEN by a probe: a spin kernel on the candidate and one on each stream it has to
    run beside; if the pair takes about one spin the queues are independent, about two and they are shared.  (Raising the runtime's queue
    count instead, GPU_MAX_HW_QUEUES = 8, makes the headline step 14 ms slower.)
The library's own two weight-gradient streams follow the reuse rule too (csrc/dit_engine.hip: free list in sfron_aux_create / _destroy).
"""
import time

import torch

_streams = {}
_probed = set()            # keys of streams that were chosen by the probe (candidates for later names)
_SPIN = 400_000            # cycles per probe kernel (~0.2 ms): long against a launch, short against everything else
_CANDIDATES = 12


def _spin_pair_ms(a, b):
    """Wall time of one spin kernel on ``a`` and one on ``b`` launched back to back (min of 3), in ms."""
    best = None
    for _ in range(3):
        a.synchronize(); b.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(_SPIN)
        with torch.cuda.stream(b):
            torch.cuda._sleep(_SPIN)
        a.synchronize(); b.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


def runs_beside(cand, others):
    """True if a spin on ``cand`` overlaps a spin on each of ``others`` (independent hardware queues)."""
    if not others:
        return True
    one = _spin_pair_ms(cand, cand) / 2.0            # two spins on ONE stream are serial by definition: half of it is one spin
    for o in others:
        if o is None or o.cuda_stream == cand.cuda_stream:
            return False
        if _spin_pair_ms(cand, o) > 1.5 * one:
            return False
    return True


PRIORITY = {}              # name -> torch stream priority for streams created under that name (A-B knob of the tools; empty = default)


def get(name, device=None, beside=()):
    """The process-wide stream ``name`` of ``device`` (default: the current device).  On first use a stream is chosen that runs beside every
    stream in ``beside`` (torch streams; None entries ignored) -- the first of up to 12 candidates that passes the probe, else the last one
    tried -- and the choice is kept for the life of the process.  A stream already chosen under another name is a candidate again: names
    whose work never overlaps in time (the beside-forward sweep, the clip norm's adaLN share, the gradient exchange) may end up on ONE stream
    when the queues are taken -- which is what four hardware queues and three busy streams leave."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, name)
    st = _streams.get(key)
    if st is not None:
        return st
    beside = [s for s in beside if s is not None]
    with torch.cuda.device(dev):
        mk = lambda: torch.cuda.Stream(device=dev, priority=PRIORITY[name]) if name in PRIORITY else torch.cuda.Stream(device=dev)
        if not beside:                                   # nothing to run beside (the micro-batch chains): a stream of its own
            st = mk()
        else:
            cands = []                                   # streams that already passed a probe first (distinct ones)
            for k, s in _streams.items():
                if k[0] == dev.index and k in _probed and all(s.cuda_stream != c.cuda_stream for c in cands):
                    cands.append(s)
            st = None
            for i in range(_CANDIDATES):
                c = cands[i] if (i < len(cands) and name not in PRIORITY) else mk()
                st = c
                if runs_beside(c, beside):
                    break
    if beside:
        _probed.add(key)
    _streams[key] = st
    return st


def describe():
    """{(device index, name): stream pointer} of what has been chosen so far (bench.py prints it)."""
    return {f"{d}:{n}": int(s.cuda_stream) for (d, n), s in _streams.items()}
