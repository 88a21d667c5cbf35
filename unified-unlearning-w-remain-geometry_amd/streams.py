"""Process-wide named side streams, one set per device.

Every runner of this package overlaps work on a handful of extra streams (the optimizer sweep beside the next forward pass, the clip
norm's adaLN share beside the embedders' backward, the gradient exchange, the two micro-batch chains).  The HIP runtime maps streams onto
a few hardware queues; two streams of one step that land on the same queue serialise.  When each runner took fresh streams
(``torch.cuda.Stream()`` hands out the next entry of torch's 32-stream pool), the THIRD runner of a process -- bench.py's configuration
legs, a test session -- could end up with a colliding pair and ran measurably slower than the same runner in a fresh process
(profiles/r06_fp8_leg.txt: DiT-XL/2 fp8 63.8 ms / step as the third runner, 59.3 as the first; DiT-B/4 10.1 against 9.2).  So the streams
are taken ONCE per (device, name) and every runner reuses them -- a later runner sees what the first one saw.  Runners of one process run
one after another; two engines that do run concurrently (micro-batch chains) ask for different names.  The library's own two
weight-gradient streams follow the same rule (csrc/dit_engine.hip: free list in sfron_aux_create / _destroy).
"""
import torch

_streams = {}


def get(name, device=None):
    """The process-wide stream ``name`` of ``device`` (default: the current device), created on first use."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, name)
    st = _streams.get(key)
    if st is None:
        st = _streams[key] = torch.cuda.Stream(device=dev)
    return st
