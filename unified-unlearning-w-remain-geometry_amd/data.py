"""Synthetic, sharding-invariant inputs for parity tests and bench.

Every draw is keyed by (seed, step, stream) and generated for the GLOBAL batch; a rank then takes
its contiguous shard, so any world size sees the same global batch."""
import torch

STREAMS = {"forget": 0, "remain": 1}


def synthetic_batch(seed, step, stream, global_batch, rank=0, world=1, input_size=32, in_channels=4, num_classes=1000,
                    forget_class=207, drop_prob=0.1, num_timesteps=1000, device="cpu"):
    g = torch.Generator().manual_seed((seed * 1_000_003 + step) * 2 + STREAMS[stream])
    x0 = torch.randn(global_batch, in_channels, input_size, input_size, generator=g)
    noise = torch.randn(global_batch, in_channels, input_size, input_size, generator=g)
    t = torch.randint(0, num_timesteps, (global_batch,), generator=g)
    if stream == "forget":
        y = torch.full((global_batch,), forget_class, dtype=torch.int64)
    else:
        y = torch.randint(0, num_classes - 1, (global_batch,), generator=g)
        y = y + (y >= forget_class).long()          # uniform over classes != forget_class
    drop = (torch.rand(global_batch, generator=g) < drop_prob).to(torch.uint8)
    per = global_batch // world
    sl = slice(rank * per, (rank + 1) * per)
    return {k: v[sl].contiguous().to(device) for k, v in dict(x0=x0, noise=noise, t=t, y=y, drop=drop).items()}
