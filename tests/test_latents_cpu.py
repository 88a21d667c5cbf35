"""Host logic of the latent front-end (no GPU): the class split of DiT/unlearn_dataset.py:277-292 and the cache format."""
import os

import numpy as np
import pytest


def test_class_split_follows_sorted_directory_order(tmp_path):
    from sfron import latents
    for c in ("n02", "n01", "n10", "n03"):
        os.makedirs(tmp_path / "train" / c)
    forget, remain, idx = latents.class_split(str(tmp_path), 2)
    assert idx == {"n01": 0, "n02": 1, "n03": 2, "n10": 3}
    assert forget == ["n03"] and remain == ["n01", "n02", "n10"]
    with pytest.raises(FileNotFoundError):
        latents.find_classes(str(tmp_path / "train" / "n01"))


def test_cache_round_trip_and_rank_shares(tmp_path):
    from sfron import latents
    rng = np.random.default_rng(0)
    for i, c in enumerate(("a", "b", "c")):
        latents.write_shard(str(tmp_path), c, i, rng.standard_normal((5 + i, 8, 4, 4)).astype(np.float16))
    cache = latents.LatentCache(str(tmp_path))
    assert cache.classes == ["a", "b", "c"] and cache.split(1) == (["b"], ["a", "c"])
    assert len(cache.samples(["a", "c"])) == 5 + 7
    # the same global batch is cut into strided shares whatever the world size
    full = latents.UnlearnLatentLoader(cache, 1, global_batch=4, device="cpu")._host_batch("remain", 3)
    parts = [latents.UnlearnLatentLoader(cache, 1, global_batch=4, rank=r, world=2, device="cpu")._host_batch("remain", 3) for r in range(2)]
    for k in full:
        assert np.array_equal(full[k][0::2].numpy(), parts[0][k].numpy()) and np.array_equal(full[k][1::2].numpy(), parts[1][k].numpy())
    assert set(full["y"].tolist()) <= {0, 2}
    bad = tmp_path / "bad"
    latents.write_shard(str(bad), "z", 3, np.zeros((1, 8, 4, 4), np.float32))
    with pytest.raises(ValueError):
        latents.LatentCache(str(bad))


def test_sd_unet_parameter_plan_matches_oracle():
    """Names / shapes / order of sfron.sd_unet.UNetModel (built without a device: only the plan) against oracle.sd_ref.UNetModel,
    which tests/golden/sd_unet.npz pins to the reference class (v1-inference.yaml config and a small one)."""
    import torch
    from types import SimpleNamespace
    from oracle import sd_ref
    from sfron import sd_unet
    for kw in (dict(), dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(2, 1), num_res_blocks=1, num_heads=2, context_dim=24),
               dict(model_channels=64, channel_mult=(1, 2, 4), attention_resolutions=(4, 1), num_res_blocks=2, num_heads=4, context_dim=40)):
        full = dict(in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4),
                    num_heads=8, context_dim=768)
        full.update(kw)
        ns = SimpleNamespace(mc=full["model_channels"], ted=4 * full["model_channels"], in_channels=4, out_channels=4, ctx_dim=full["context_dim"],
                             num_res_blocks=full["num_res_blocks"], attn_res=tuple(full["attention_resolutions"]), channel_mult=tuple(full["channel_mult"]))
        sd_unet.UNetModel._plan(ns)
        with torch.device("meta"):
            ref = sd_ref.UNetModel(**full)
        assert [(n, tuple(p.shape)) for n, p in ref.named_parameters()] == [(n, tuple(s)) for n, s in ns.param_specs]
