"""Data modules: the reference's YAML-listed image pairs (reference dataset.py:11-134) and the
synthetic pairs used by the benchmark.

Reference defect handled here: dataset.py:58 normalises 1-channel images with
3-tuples, which cannot broadcast; this module normalises with (0.5,)/(0.5,), i.e. x*2-1.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .lightning import LightningDataModule


def load_gray_256(path: str, size: int = 256, normalize: bool = True) -> torch.Tensor:
    """read_image(GRAY) -> Resize((256,256), antialias=True) on uint8 -> float32/255 -> [-1,1]
    (reference dataset.py:51-59,129-132).  The resize is the reference's own operator: torchvision 0.15 casts the
    uint8 tensor to float32, runs aten's antialiased bilinear kernel (``interpolate(..., mode="bilinear",
    antialias=True, align_corners=False)``), rounds (half to even) and casts back to uint8 BEFORE the division by
    255 -- byte work, reproduced bit for bit (PIL's BILINEAR filter accumulates with 8-bit fixed-point weights and is
    one uint8 step off on ~1 % of the pixels; PIL only decodes here)."""
    from PIL import Image
    img = torch.from_numpy(np.asarray(Image.open(path).convert("L"), dtype=np.uint8).copy())
    if tuple(img.shape) != (size, size):
        img = torch.nn.functional.interpolate(img[None, None].to(torch.float32), size=(size, size), mode="bilinear",
                                              antialias=True, align_corners=False).round_().to(torch.uint8)[0, 0]
    x = img.to(torch.float32).div_(255).unsqueeze(0)
    return x * 2 - 1 if normalize else x


class ImageDataset(Dataset):
    """Pairs of (input, ground truth) image files (reference dataset.py:110-134)."""

    def __init__(self, data_tuples, normalize=True, size=256):
        super().__init__()
        self.data_tuples, self.normalize, self.size = data_tuples, normalize, size

    def __len__(self):
        return len(self.data_tuples)

    def __getitem__(self, idx):
        inp, gt = self.data_tuples[idx]
        return load_gray_256(inp, self.size, self.normalize), load_gray_256(gt, self.size, self.normalize)


def _read_list(list_file):
    import yaml
    with open(list_file, "r") as f:
        items = yaml.safe_load(f)
    base = os.path.dirname(str(list_file))
    return [(os.path.join(base, it["input"]), os.path.join(base, it["ground_truth"])) for it in items]


class ShardedLoader:
    """A DataLoader over this rank's shard of a dataset.  With ``world > 1`` the indices come from a
    ``DistributedSampler`` (what Lightning's DDP strategy wraps around the reference's loaders, reference
    main.py:123-136 + dataset.py:77-83): every epoch is ONE pass over the data split across the ranks, shuffled with
    a seed all ranks share and the epoch number (``set_epoch``), padded by wrap-around so that every rank draws the
    same number of batches (the collective gradient exchange needs that)."""

    def __init__(self, dataset, batch_size, shuffle, world=1, rank=0, seed=0, **loader_kw):
        self.sampler = (DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed,
                                           drop_last=False) if world > 1 else None)
        self.loader = DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and self.sampler is None,
                                 sampler=self.sampler, drop_last=False, **loader_kw)
        self.dataset, self.batch_size = dataset, batch_size

    def set_epoch(self, epoch: int):
        if self.sampler is not None:
            self.sampler.set_epoch(epoch)

    def __iter__(self):
        return iter(self.loader)

    def __len__(self):
        return len(self.loader)


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


class ImageDataModule(LightningDataModule):
    """``ImageDataModule(data_list_file, val_list_file, batch_size, normalize)`` (reference
    dataset.py:11-107).  ``num_workers`` / ``pin_memory`` are additions: the reference decodes on
    the main process, which cannot feed a GPU at >2k images/s.  ``world`` / ``rank`` (default: taken from
    torch.distributed when it is initialised) shard every split across the data-parallel ranks."""

    def __init__(self, data_list_file: str, val_list_file: Optional[str] = None, batch_size: int = 1,
                 normalize: bool = True, num_workers: int = 0, pin_memory: bool = True,
                 world: Optional[int] = None, rank: Optional[int] = None, seed: int = 0):
        super().__init__()
        self.data_tuples = _read_list(data_list_file)
        self.val_tuples = _read_list(val_list_file) if val_list_file is not None else None
        self.batch_size, self.normalize = batch_size, normalize
        self.num_workers, self.pin_memory = num_workers, pin_memory
        dw, dr = _dist_info()
        self.world, self.rank, self.seed = (dw if world is None else world), (dr if rank is None else rank), seed

    def setup(self, stage: str):
        if stage == "fit":
            self.train_split, self.val_split = self.data_tuples, self.val_tuples
        if stage == "validate":
            self.val_split = self.data_tuples
        if stage == "test":
            self.test_split = self.data_tuples
        if stage == "predict":
            self.pred_split = self.data_tuples

    def _loader(self, split, shuffle):
        return ShardedLoader(ImageDataset(split, self.normalize), self.batch_size, shuffle, self.world, self.rank,
                             self.seed, num_workers=self.num_workers,
                             pin_memory=self.pin_memory and torch.cuda.is_available())

    def train_dataloader(self):
        return self._loader(self.train_split, True)

    def val_dataloader(self):
        return self._loader(self.val_split, False) if self.val_split else None

    def test_dataloader(self):
        return self._loader(self.test_split, False)

    def predict_dataloader(self):
        return self._loader(self.pred_split, False)


def synthetic_pairs(n: int, size: int = 256, seed: int = 1234, kind: str = "uniform"):
    """Canonical synthetic PAI pairs: numpy default_rng, fp32 in [-1, 1).
    ``kind='blobs'`` makes depth-attenuated blob images so that training has something to learn."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        x = rng.random((n, 1, size, size), dtype=np.float32) * 2 - 1
        t = rng.random((n, 1, size, size), dtype=np.float32) * 2 - 1
        return torch.from_numpy(x), torch.from_numpy(t)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    t = np.zeros((n, 1, size, size), np.float32)
    for i in range(n):
        for _ in range(int(rng.integers(3, 9))):
            cy, cx, r = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(3, 18)
            t[i, 0] += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r)) * rng.uniform(0.4, 1.0)
    t = np.clip(t, 0, 1)
    depth = np.exp(-yy / size * 2.5)[None, None]
    x = np.clip(t * depth + 0.05 * rng.standard_normal(t.shape).astype(np.float32), 0, 1)
    return torch.from_numpy(x * 2 - 1), torch.from_numpy(t * 2 - 1)


class _TensorPairs(Dataset):
    def __init__(self, x, t):
        self.x, self.t = x, t

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.t[i]


class SyntheticDataModule(LightningDataModule):
    """Synthetic pairs, the same data set on every rank (one seed), sharded across the ranks like ImageDataModule."""

    def __init__(self, n_train=256, n_val=32, batch_size=8, size=256, seed=1234, kind="blobs",
                 world: Optional[int] = None, rank: Optional[int] = None):
        super().__init__()
        self.n_train, self.n_val, self.batch_size, self.size, self.seed, self.kind = \
            n_train, n_val, batch_size, size, seed, kind
        dw, dr = _dist_info()
        self.world, self.rank = (dw if world is None else world), (dr if rank is None else rank)

    def setup(self, stage: str):
        self.train = _TensorPairs(*synthetic_pairs(self.n_train, self.size, self.seed, self.kind))
        self.val = _TensorPairs(*synthetic_pairs(self.n_val, self.size, self.seed + 1, self.kind))

    def train_dataloader(self):
        return ShardedLoader(self.train, self.batch_size, True, self.world, self.rank, self.seed)

    def val_dataloader(self):
        return ShardedLoader(self.val, self.batch_size, False, self.world, self.rank, self.seed)

    def predict_dataloader(self):
        return DataLoader(self.val, batch_size=self.batch_size, shuffle=False)
