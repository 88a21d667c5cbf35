"""HIP-graph replay of one stage of an SFR-on iteration on the convolutional U-Nets.

A stage of the DDPM / SD loops (DDPM/runners/diffusion.py:1081-1136 or :1140-1174; SD/train-scripts/nsfw_removal.py:108-147 or
:164-168) is the forward pass(es), the loss and the backward pass up to this rank's final gradients: 2 000 - 8 000 kernel launches
that the Python tape (unet._TapeNet) issues one by one through the C ABI, about 20 us of host time each -- more than the GPU needs for
them once the products run on the pipelined tiles.  ``StageGraph`` runs the stage eagerly ``warmup`` times (lazily sized scratch,
per-device function attributes), then captures it once into a HIP graph on a capture stream and afterwards replays it: one
hipGraphLaunch per stage.  Inputs are copied into the static tensors the graph reads; outputs are returned as clones (stages that
share a memory pool overwrite each other's intermediates).  Everything a stage does is stream-ordered device work -- no host
synchronisation, no host-side scalars that change from step to step (the decayed forget alpha travels as a device scalar).  The
optimizer sweep stays outside (its bias corrections are host scalars) and so does the data-parallel gradient exchange.
"""
import gc

import torch


class StageGraph:
    def __init__(self, fn, warmup=2, pool=None):
        self.fn, self.warmup, self.pool = fn, warmup, pool
        self.calls = 0
        self.graph = self.static_in = self.static_out = self.sig = None

    @staticmethod
    def _sig(inputs):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items()))

    def __call__(self, **inputs):
        """inputs: device tensors (a value of None is dropped).  Returns what ``fn(**inputs)`` returns (a tensor or a dict of tensors)."""
        inputs = {k: v for k, v in inputs.items() if v is not None}
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            return self.fn(**inputs)
        sig = self._sig(inputs)
        if self.graph is None or sig != self.sig:
            self.sig = sig
            self.static_in = {k: v.clone() for k, v in inputs.items()}
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # No cyclic garbage collection inside the capture: a dead CUDAGraph (an earlier loop's stage: the closures of a StageGraph form
            # reference cycles, so it dies by the collector, not by reference count) destroyed WHILE a stream is capturing raises in its
            # destructor and ends the process (hipErrorStreamCaptureUnsupported; torch.cuda.graph no longer collects on entry by default).
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(self.graph, pool=self.pool):
                    self.static_out = self.fn(**self.static_in)
            finally:
                if gc_was_on:
                    gc.enable()
        else:
            for k, v in inputs.items():
                self.static_in[k].copy_(v)
        self.graph.replay()
        out = self.static_out
        if isinstance(out, dict):
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        return out.clone()


def shared_pool():
    """One private memory pool for the stages of a loop: they run strictly one after another, so their activations may share memory."""
    return torch.cuda.graph_pool_handle()
