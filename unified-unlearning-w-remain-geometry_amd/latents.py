"""Latent front-end of the DiT class-forgetting loop.

The reference feeds the step from an ImageFolder split by class (DiT/unlearn_dataset.py:277-292 ``get_unlearn_dataset``: the
forget set is the class whose index -- position in the alphabetically sorted class-directory list -- equals ``forget_class``,
the remain set is every other class) and encodes each image batch with the frozen VAE inside the loop
(DiT/forget.py:265-267,305-307: ``vae.encode(x).latent_dist.sample().mul_(0.18215)``).  The encoder never changes, so here
its OUTPUT is cached once, offline, as the posterior moments (mean || logvar, [8, 32, 32] fp16/fp32 per image) in per-class
shards, and the step's inputs come from that cache: the class split is the reference's, the posterior sample + 0.18215
scaling is sfron_latent_sample on the device, and batches reach the GPU through pinned buffers on a copy stream one batch
ahead of the step (data-parallel ranks take strided shares of each global batch).  The VAE itself (diffusers AutoencoderKL,
absent here) is outside the path: ``write_shard`` takes whatever moments the caller's encoder produced.
"""
import json
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def find_classes(directory):
    """torchvision.datasets.folder.find_classes as the reference uses it: sorted sub-directory names -> index."""
    classes = sorted(e.name for e in os.scandir(directory) if e.is_dir())
    if not classes:
        raise FileNotFoundError(f"Couldn't find any class folder in {directory}.")
    return classes, {c: i for i, c in enumerate(classes)}


def class_split(data_path, forget_class):
    """DiT/unlearn_dataset.py:277-292: (forget class names, remain class names, class_to_idx) of ``data_path/train``."""
    classes, class_to_idx = find_classes(os.path.join(data_path, "train"))
    forget = [c for c in classes if class_to_idx[c] == forget_class]
    remain = [c for c in classes if class_to_idx[c] != forget_class]
    return forget, remain, class_to_idx


def write_shard(cache_dir, class_name, class_index, moments):
    """One class's cached VAE posterior moments [N, 2C, H, W] (mean || logvar) -> <cache_dir>/<class_name>.npy + index entry."""
    os.makedirs(cache_dir, exist_ok=True)
    m = np.asarray(moments)
    assert m.ndim == 4 and m.shape[1] % 2 == 0
    np.save(os.path.join(cache_dir, class_name + ".npy"), m)
    idx_path = os.path.join(cache_dir, "index.json")
    idx = json.load(open(idx_path)) if os.path.isfile(idx_path) else {}
    idx[class_name] = {"index": int(class_index), "count": int(m.shape[0]), "shape": list(m.shape[1:]), "dtype": str(m.dtype)}
    json.dump(idx, open(idx_path, "w"), indent=1, sort_keys=True)


class LatentCache:
    """Memory-mapped per-class shards written by write_shard; ``split(forget_class)`` gives the reference's two datasets."""

    def __init__(self, cache_dir):
        self.dir = cache_dir
        self.index = json.load(open(os.path.join(cache_dir, "index.json")))
        self.classes = sorted(self.index)                                  # alphabetical, as find_classes
        for i, c in enumerate(self.classes):
            if self.index[c]["index"] != i:
                raise ValueError(f"class {c!r} was cached with index {self.index[c]['index']}, the sorted position is {i}")
        self._maps = {}

    def shard(self, name):
        if name not in self._maps:
            self._maps[name] = np.load(os.path.join(self.dir, name + ".npy"), mmap_mode="r")
        return self._maps[name]

    def split(self, forget_class):
        forget = [c for c in self.classes if self.index[c]["index"] == forget_class]
        remain = [c for c in self.classes if self.index[c]["index"] != forget_class]
        return forget, remain

    def samples(self, names):
        """[(class name, class index, position in shard)] in ImageFolder order (classes sorted, files in shard order)."""
        return [(c, self.index[c]["index"], i) for c in names for i in range(self.index[c]["count"])]


class UnlearnLatentLoader:
    """Infinite forget / remain batch streams for DiTSFRon.step (the reference cycles two shuffled DataLoaders,
    DiT/forget.py:219-228,241-246): each ``next()`` returns ``dict(x0, y, t, noise, drop)`` on the device for this rank's share
    of a global batch.  Shuffling, timesteps, noise, posterior noise and label-dropout draws are keyed by (seed, epoch / step,
    stream) on the host generator, so every world size sees the same global batch."""

    def __init__(self, cache, forget_class, global_batch, rank=0, world=1, seed=0, num_timesteps=1000, drop_prob=0.1, scale=0.18215,
                 device="cuda"):
        assert global_batch % world == 0
        self.cache, self.gb, self.rank, self.world, self.seed = cache, global_batch, rank, world, seed
        self.T, self.p, self.scale, self.dev = num_timesteps, drop_prob, scale, torch.device(device)
        f, r = cache.split(forget_class)
        self.sets = {"forget": cache.samples(f), "remain": cache.samples(r)}
        if not self.sets["forget"] or not self.sets["remain"]:
            raise ValueError("empty forget or remain set")
        self.step = {"forget": 0, "remain": 0}
        self._copy = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self._ahead = {}

    def _host_batch(self, stream, step):
        s = self.sets[stream]
        per_epoch = max(1, len(s) // self.gb)                                   # drop_last, as the reference's loaders
        epoch, pos = divmod(step, per_epoch)
        g = torch.Generator().manual_seed(((self.seed * 1_000_003 + epoch) * 2 + (stream == "remain")) & 0x7FFFFFFF)
        perm = torch.randperm(len(s), generator=g)
        ids = perm[pos * self.gb:(pos + 1) * self.gb] if len(s) >= self.gb else perm[torch.arange(self.gb) % len(s)]
        g2 = torch.Generator().manual_seed(((self.seed * 7_000_003 + step) * 2 + (stream == "remain")) & 0x7FFFFFFF)
        shp = self.cache.index[s[0][0]]["shape"]
        C = shp[0] // 2
        t = torch.randint(0, self.T, (self.gb,), generator=g2)
        noise = torch.randn(self.gb, C, shp[1], shp[2], generator=g2)
        eps = torch.randn(self.gb, C, shp[1], shp[2], generator=g2)
        drop = (torch.rand(self.gb, generator=g2) < self.p).to(torch.uint8)
        mine = list(range(self.rank, self.gb, self.world))                     # strided share of the global batch
        mom = np.stack([np.asarray(self.cache.shard(s[int(ids[j])][0])[s[int(ids[j])][2]], dtype=np.float32) for j in mine])
        y = torch.tensor([s[int(ids[j])][1] for j in mine], dtype=torch.int64)
        return dict(moments=torch.from_numpy(mom), eps=eps[mine].contiguous(), y=y, t=t[mine].contiguous(), noise=noise[mine].contiguous(),
                    drop=drop[mine].contiguous())

    def _stage(self, stream, step):
        hb = self._host_batch(stream, step)
        if self._copy is None:
            return hb, None
        pinned = {k: v.pin_memory() for k, v in hb.items()}
        with torch.cuda.stream(self._copy):
            dv = {k: v.to(self.dev, non_blocking=True) for k, v in pinned.items()}
            ev = torch.cuda.Event()
            ev.record(self._copy)
        return dv, (ev, pinned)

    def next(self, stream):
        step = self.step[stream]
        staged = self._ahead.pop((stream, step), None) or self._stage(stream, step)
        self._ahead[(stream, step + 1)] = self._stage(stream, step + 1)        # one batch ahead, on the copy stream
        self.step[stream] = step + 1
        dv, sync = staged
        if sync is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(sync[0])
            # the staged tensors were allocated on the copy stream: tell the caching allocator that the consumer stream
            # reads them, so a dropped batch is not handed to a later staging copy while queued step kernels still use it
            # (the fast path never syncs the host and may run several steps ahead of the GPU)
            for v in dv.values():
                v.record_stream(cur)
        mom = dv["moments"].contiguous()
        n, c2, h, w = mom.shape
        x0 = torch.empty(n, c2 // 2, h, w, dtype=torch.float32, device=mom.device)
        check(_lib.lib().sfron_latent_sample(ptr(mom), ptr(dv["eps"]), n, c2 // 2, h * w, float(self.scale), ptr(x0), stream_ptr()),
              "latent_sample")
        return dict(x0=x0, y=dv["y"], t=dv["t"], noise=dv["noise"], drop=dv["drop"])
