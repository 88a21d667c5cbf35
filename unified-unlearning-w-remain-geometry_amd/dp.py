"""Data-parallel plumbing: one process per GPU, replicated weights / optimizer state / EMA / mask, the two
gradient streams of an SFR-on iteration SUM-all-reduced over RCCL (torch.distributed backend "nccl").

The reference's multi-GPU path is single-process nn.DataParallel (DiT/forget.py:193): scatter the batch,
gather outputs, mean over the GLOBAL batch, reduce grads to GPU 0.  Here every rank scales its loss
gradient by 1/global_batch (csrc/loss.hip grad_scale) and sums, which is the same mean-over-global-batch
gradient up to summation order.  Backend-agnostic on purpose: the CPU tests drive it with gloo.
"""
import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_flat_(flat, bucket_elems, group=None, async_op=False):
    """In-place SUM all-reduce of a flat tensor in buckets of `bucket_elems` elements.  xGMI is point-to-point
    (7 links/GPU), so buckets are large (default 256 MiB in step.py): few, bandwidth-bound collectives."""
    if world_size(group) == 1:
        return []
    works = []
    for s in range(0, flat.numel(), bucket_elems):
        w = dist.all_reduce(flat[s:s + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard(global_batch, rank, world):
    """Contiguous shard of a dict of tensors whose dim 0 is the global batch."""
    n = next(iter(global_batch.values())).shape[0]
    assert n % world == 0, "global batch must divide evenly over ranks"
    per = n // world
    return {k: v[rank * per:(rank + 1) * per].contiguous() for k, v in global_batch.items()}
