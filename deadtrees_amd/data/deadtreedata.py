"""``DeadtreesDataModule`` surface (reference deadtrees/data/deadtreedata.py:192-405) with a synthetic source.

The reference streams webdataset shards through albumentations on CPU workers; neither package nor any shard
is available here, and at >400 tiles/s/GPU the real loader is the limiter anyway (§8 f2).  This
module keeps the constructor / ``setup`` / ``*_dataloader`` surface and the batch formats
(``{"main": (img, mask, distmap, lu, stats)}`` for train/val :348-395, a bare tuple for test :397-405) and
fills them with synthetic tiles of the reference's shape; ``val_transform`` is the reference normalisation
(:148-154) in numpy.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .distmap import distmaps_for_batch, distmaps_on_device
from .synthetic import MEAN, STD, synth_batch


class DeadtreeDatasetConfig:
    """reference deadtreedata.py:27-34"""
    mean = np.array(MEAN)
    std = np.array(STD)
    tile_size = 256
    fractions = [0.7, 0.2, 0.1]


def val_transform(image: np.ndarray, mask: Optional[np.ndarray] = None):
    """albumentations ``Normalize(mean, std)`` + ``ToTensorV2`` of deadtreedata.py:148-154: HWC uint8 -> CHW f32"""
    c = image.shape[-1]
    mean = (np.asarray(MEAN[:c], dtype=np.float32) * 255.0)
    inv = 1.0 / (np.asarray(STD[:c], dtype=np.float32) * 255.0)
    img = (image.astype(np.float32) - mean) * inv
    out = {"image": torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))}
    if mask is not None:
        out["mask"] = torch.from_numpy(mask)
    return out


def draw_train_params(batch: int, rng: np.random.Generator):
    """Per-sample random parameters of the reference's ``train_transform`` (deadtreedata.py:128-146):
    OneOf([HorizontalFlip, VerticalFlip], p=0.5) -> 25 % each; RandomRotate90(p=0.5) -> k = randint(0, 3);
    RandomBrightnessContrast(p=0.5, brightness_limit=0.2, contrast_limit=0.15): alpha = 1 + U(-.15, .15),
    beta = U(-.2, .2).  Returns (geo int32 [B,2], bc float32 [B,2]) for ``train_transform_device``."""
    geo = np.zeros((batch, 2), np.int32)
    bc = np.tile(np.array([1.0, 0.0], np.float32), (batch, 1))
    for b in range(batch):
        if rng.random() < 0.5:
            geo[b, 0] = 1 if rng.random() < 0.5 else 2
        if rng.random() < 0.5:
            geo[b, 1] = int(rng.integers(0, 4))
        if rng.random() < 0.5:
            bc[b] = (1.0 + rng.uniform(-0.15, 0.15), rng.uniform(-0.2, 0.2))
    return torch.from_numpy(geo), torch.from_numpy(bc)


def train_transform_device(tiles_u8_nhwc: torch.Tensor, mask: torch.Tensor, lu: Optional[torch.Tensor],
                           rng: np.random.Generator, in_channels: int = 3):
    """``train_transform`` + ``transform`` (deadtreedata.py:128-146,165-176) for a whole batch already in HBM:
    uint8 [B,H,W,4] tiles, int64 masks (and land-use maps) -> (NCHW fp32 image batch, mask, lu) with ONE fused
    flip/rot90/brightness-contrast/normalise pass and one gather per label map (kernels `dt_augment_*`)."""
    from .. import ops
    B = tiles_u8_nhwc.shape[0]
    geo, bc = draw_train_params(B, rng)
    geo, bc = geo.to(tiles_u8_nhwc.device), bc.to(tiles_u8_nhwc.device)
    img = ops.augment_normalize_u8(tiles_u8_nhwc, geo, bc, MEAN, STD, in_channels).permute(0, 3, 1, 2)
    mask = ops.augment_labels(mask.long(), geo)
    lu = ops.augment_labels(lu.long(), geo) if lu is not None else None
    return img, mask, lu


_TRAIN_RNG = np.random.default_rng()


def train_transform(image: np.ndarray, mask: Optional[np.ndarray] = None, masks=None):
    """The reference's module-level ``train_transform`` (deadtreedata.py:128-146, an albumentations ``Compose``) as a
    callable on ONE sample: HWC uint8 ``image`` (+ ``mask`` or a list ``masks``, as the reference's ``transform`` passes
    them, :165-176) -> ``{"image": CHW f32 tensor, "mask": ..., "masks": [...]}`` with the same random flip / rot90 /
    brightness-contrast draw for image and label maps.  Runs ``train_transform_device`` on a batch of one — the HIP
    augmentation kernels; there is no CPU path (raises without a HIP device)."""
    if not torch.cuda.is_available():
        raise RuntimeError("deadtrees_amd train_transform runs the HIP augmentation kernels: no HIP device, no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    c = image.shape[-1]
    maps = ([] if mask is None else [mask]) + list(masks or [])
    tiles = torch.from_numpy(np.ascontiguousarray(image))[None].to(dev)
    geo, bc = draw_train_params(1, _TRAIN_RNG)
    geo, bc = geo.to(dev), bc.to(dev)
    from .. import ops
    img = ops.augment_normalize_u8(tiles, geo, bc, MEAN, STD, c).permute(0, 3, 1, 2)[0].contiguous()
    outs = [ops.augment_labels(torch.from_numpy(np.ascontiguousarray(m))[None].to(dev).long(), geo)[0].to(
        torch.from_numpy(np.asarray(m)).dtype) for m in maps]
    out = {"image": img}
    if mask is not None:
        out["mask"] = outs[0]
    if masks is not None:
        out["masks"] = outs[(0 if mask is None else 1):]
    return out


class _SyntheticLoader:
    def __init__(self, n_batches, batch_size, size, in_channels, classes, seed, wrap_main, device, with_distmap):
        self.n, self.bs, self.size, self.c, self.k = n_batches, batch_size, size, in_channels, classes
        self.seed, self.wrap, self.device, self.with_distmap = seed, wrap_main, device, with_distmap

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            img, mask = synth_batch(self.bs, self.size, self.size, self.c, self.k, seed=self.seed + i)
            on_gpu = bool(self.device) and str(self.device).startswith("cuda")
            # labels headed for HBM get their maps from the device EDT kernel below; a host-only loader attaches
            # them the way the reference's loader does (scipy, data/deadtreedata.py:182-185)
            dist = None if (on_gpu or not self.with_distmap) else distmaps_for_batch(mask, self.k)
            lu = torch.ones_like(mask)
            stats = [{"file": f"synthetic_{self.seed + i}_{j}", "frac": float((mask[j] > 0).float().mean())}
                     for j in range(self.bs)]
            if self.device:
                img, mask, lu = (t.to(self.device) for t in (img, mask, lu))
                if self.with_distmap:
                    dist = distmaps_on_device(mask, self.k) if on_gpu else dist.to(self.device)
            item = (img, mask, dist, lu, stats)
            yield {"main": item} if self.wrap else item


class DeadtreesDataModule:
    def __init__(self, data_dir=None, pattern=None, pattern_extra=None, batch_size_extra=None,
                 train_dataloader_conf=None, val_dataloader_conf=None, test_dataloader_conf=None,
                 synthetic_batches: int = 8, tile_size: int = 256, device: Optional[str] = None):
        self.data_dir, self.pattern = data_dir, pattern
        self.train_conf = dict(train_dataloader_conf or {})
        self.val_conf = dict(val_dataloader_conf or {})
        self.test_conf = dict(test_dataloader_conf or {})
        self.synthetic_batches, self.tile_size, self.device = synthetic_batches, tile_size, device
        self.in_channels, self.classes = 3, 2

    def setup(self, stage=None, split_fractions=None, in_channels: int = 3, classes: int = 2):
        if data_dir_has_shards(self.data_dir):
            raise NotImplementedError("webdataset shards need the `webdataset`/`albumentations` packages "
                                      "(absent here); only the synthetic source is built")
        self.in_channels, self.classes = in_channels, classes

    def _loader(self, conf, seed, wrap):
        return _SyntheticLoader(self.synthetic_batches, int(conf.get("batch_size", 8)), self.tile_size,
                                self.in_channels, self.classes, seed, wrap, self.device, True)

    def train_dataloader(self):
        return self._loader(self.train_conf, 1000, True)

    def val_dataloader(self):
        return self._loader(self.val_conf, 2000, True)

    def test_dataloader(self):
        return self._loader(self.test_conf, 3000, False)


def data_dir_has_shards(data_dir) -> bool:
    import glob
    import os
    return bool(data_dir) and bool(glob.glob(os.path.join(str(data_dir), "*.tar")))
