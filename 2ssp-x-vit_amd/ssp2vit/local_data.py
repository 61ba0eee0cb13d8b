"""Local image data for the CLI ("real CIFAR/ImageNet bytes (when present)").

The reference builds its loaders from the network (`datasets.load_dataset`, adaptation-for-Pures-framework/auto_2ssp.py:268-350);
that fetch is out of scope here.  What is in scope is the part behind it: raw uint8 images that already lie on disk go through
the GPU input pipeline (`ssp2vit.preprocess.GpuPreprocessor` = the reference's Resize(BICUBIC) -> [RandomHorizontalFlip] ->
ToTensor -> Normalize chain, :290-301) in batches with the reference's loader semantics (:345-348):

  * test loader:         batch 64, NOT shuffled, test transform (no flip)
  * calibration loader:  batch 64, shuffled, TRAIN transform (random horizontal flip, p = 0.5)

The shuffle is drawn from `seed` and the epoch number (every `iter()` is an epoch, as with a DataLoader), the flips once per loader
from `seed` (the reference materialises its train transform in `datasets.map`, :334-336), identically on every rank; with `world > 1` a rank yields only the batches it owns (`dist.rank_batch_indices`: global batch b -> rank b % world),
to be consumed with `sharded=True`.

File formats (no pickles — `numpy.load(allow_pickle=False)`):
  * `.npz` with an image array under one of  images / x / data / pixel_values  (uint8 [n, H, W, 3]) and labels under one of
    labels / y / targets  (integer [n]);
  * `.npy` holding the images (memory-mapped, so an ImageNet-sized array is not read whole), labels from `labels_path` or from
    `<stem>_labels.npy` beside it.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist as _dist

IMAGE_KEYS = ("images", "x", "data", "pixel_values")
LABEL_KEYS = ("labels", "y", "targets")


def load_uint8_dataset(path: str, labels_path: Optional[str] = None) -> Tuple[np.ndarray, torch.Tensor]:
    """-> (uint8 images [n, H, W, 3] (numpy; memory-mapped for .npy), int64 labels [n])."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        ik = next((k for k in IMAGE_KEYS if k in z.files), None)
        lk = next((k for k in LABEL_KEYS if k in z.files), None)
        if ik is None:
            raise ValueError(f"{path}: no image array under any of {IMAGE_KEYS} (found {z.files})")
        images = z[ik]
        if labels_path is not None:
            labels = np.load(labels_path, allow_pickle=False)
        elif lk is not None:
            labels = z[lk]
        else:
            raise ValueError(f"{path}: no labels under any of {LABEL_KEYS}; pass labels_path")
    elif path.endswith(".npy"):
        images = np.load(path, mmap_mode="r", allow_pickle=False)
        if labels_path is None:
            labels_path = path[:-4] + "_labels.npy"
        if not os.path.exists(labels_path):
            raise FileNotFoundError(f"labels for {path}: {labels_path}")
        labels = np.load(labels_path, allow_pickle=False)
    else:
        raise ValueError(f"{path}: expected a .npz or .npy file")
    if images.dtype != np.uint8 or images.ndim != 4 or images.shape[-1] != 3:
        raise ValueError(f"{path}: images must be uint8 [n, H, W, 3], got {images.dtype} {tuple(images.shape)}")
    labels = np.asarray(labels)
    if labels.ndim != 1 or labels.shape[0] != images.shape[0] or not np.issubdtype(labels.dtype, np.integer):
        raise ValueError(f"{path}: labels must be an integer vector of length {images.shape[0]}, got {labels.dtype} {tuple(labels.shape)}")
    return images, torch.from_numpy(labels.astype(np.int64))


class Uint8BatchLoader:
    """Iterable of {"pixel_values": uint8 [b, H, W, 3] (pinned host memory), "labels": int64 [b], "preprocess": GpuPreprocessor,
    ["hflip": uint8 [b]]} — what `core._pixels_to_device` turns into fp32 NCHW on the copy stream."""

    def __init__(self, images: np.ndarray, labels: torch.Tensor, batch_size: int = 64, *, shuffle: bool = False, random_flip: bool = False,
                 seed: int = 0, out_size: int = 224, mean: Sequence[float] = (0.5, 0.5, 0.5), std: Sequence[float] = (0.5, 0.5, 0.5),
                 device="cuda", rank: int = 0, world: int = 1, limit: Optional[int] = None, preprocess="auto", pin: bool = True):
        self.images, self.labels = images, labels
        self.n = int(images.shape[0])
        self.batch_size, self.shuffle, self.random_flip, self.seed = int(batch_size), bool(shuffle), bool(random_flip), int(seed)
        self.rank, self.world, self.limit = int(rank), int(world), limit
        self.out_size, self.mean, self.std, self.device = int(out_size), tuple(mean), tuple(std), device
        self._pp = preprocess                # "auto": built at the first batch (needs the GPU); None: batches carry no preprocessor (host tests)
        self.pin = bool(pin) and torch.cuda.is_available()
        self.epoch = 0

    sharded = property(lambda self: self.world > 1)

    def __len__(self) -> int:
        return len(_dist.rank_batch_indices(self.n, self.batch_size, self.rank, self.world, self.limit))

    def order(self, epoch: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(permutation of the n items, flip bit per ITEM) of one epoch — the same on every rank.  The ORDER is redrawn per epoch
        (DataLoader(shuffle=True), reference :348); the FLIPS are drawn once per loader from `seed`: the reference applies its train
        transform inside `datasets.map(...)` (:334-336), which materialises it, so an image keeps its flip over every later pass."""
        g = torch.Generator().manual_seed(self.seed * 1000003 + epoch)
        perm = torch.randperm(self.n, generator=g) if self.shuffle else torch.arange(self.n)
        if self.random_flip:
            gf = torch.Generator().manual_seed(self.seed * 1000003 + 500009)
            flips = (torch.rand(self.n, generator=gf) < 0.5).to(torch.uint8)
        else:
            flips = torch.zeros(self.n, dtype=torch.uint8)
        return perm, flips

    def batch_items(self, epoch: int) -> List[torch.Tensor]:
        """Item indices of the batches THIS rank yields in `epoch`, in order."""
        perm, _ = self.order(epoch)
        return [perm[torch.tensor(ix, dtype=torch.int64)] for ix in _dist.rank_batch_indices(self.n, self.batch_size, self.rank, self.world, self.limit)]

    def _preprocessor(self):
        if self._pp == "auto":
            from .preprocess import GpuPreprocessor
            self._pp = GpuPreprocessor(self.images.shape[1:3], self.out_size, self.mean, self.std, device=self.device)
        return self._pp

    def peek_global(self, n_batches: int = 1) -> List[dict]:
        """The first `n_batches` GLOBAL batches of the NEXT epoch — the same images on every rank, whatever it owns — without
        advancing the epoch (fp8 calibration: every rank must measure the same images, or its e4m3 scales differ from its peers')."""
        perm, flips = self.order(self.epoch)
        idx = _dist.rank_batch_indices(self.n, self.batch_size, 0, 1, self.limit)[: int(n_batches)]
        return [self._batch(perm[torch.tensor(ix, dtype=torch.int64)], flips) for ix in idx]

    def __iter__(self) -> Iterator[dict]:
        epoch, self.epoch = self.epoch, self.epoch + 1
        _, flips = self.order(epoch)
        for items in self.batch_items(epoch):
            yield self._batch(items, flips)

    def _batch(self, items: torch.Tensor, flips: torch.Tensor) -> dict:
        idx = items.numpy()
        srt = np.argsort(idx, kind="stable")                      # one ascending pass over a memory-mapped file, then back in batch order
        block = np.empty((len(idx),) + tuple(self.images.shape[1:]), dtype=np.uint8)
        block[srt] = self.images[idx[srt]]
        px = torch.from_numpy(block)
        if self.pin:
            px = px.pin_memory()
        batch = {"pixel_values": px, "labels": self.labels[items]}
        pp = self._preprocessor()
        if pp is not None:
            batch["preprocess"] = pp
        if self.random_flip:
            batch["hflip"] = flips[items]
        return batch
