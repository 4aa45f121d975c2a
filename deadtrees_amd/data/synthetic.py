"""Synthetic 512x512 tiles of the shape the reference's loader produces.

Image: uint8 noise normalised with the dataset mean/std of reference
deadtrees/data/deadtreedata.py:31-32 (``DeadtreeDatasetConfig``), first ``C`` of the 4 RGBN channels,
f32 NCHW.  Labels: sparse blocks (32x32 Bernoulli(0.05) grid, nearest-upsampled), int64; tile 0 of
every batch is all background (exercises the +1e-9 paths of the Dice losses).
"""
from __future__ import annotations

import torch

MEAN = (0.3661029729, 0.3875165941, 0.3501133538, 0.5797285859)   # deadtreedata.py:31
STD = (0.2388708549, 0.2103625723, 0.2050272174, 0.2025812523)    # deadtreedata.py:32


def synth_batch(B: int, H: int, W: int, C: int = 3, K: int = 2, seed: int = 1234, p_fg: float = 0.05):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, H, W, 4), generator=g, dtype=torch.uint8)
    mean = torch.tensor(MEAN, dtype=torch.float32)
    std = torch.tensor(STD, dtype=torch.float32)
    img = ((u8.float() - mean * 255.0) * (1.0 / (std * 255.0)))[..., :C].permute(0, 3, 1, 2).contiguous()
    gh, gw = max(H // 16, 1), max(W // 16, 1)
    fg = torch.rand((B, gh, gw), generator=g) < p_fg
    cls = torch.randint(1, K, (B, gh, gw), generator=g) if K > 2 else torch.ones((B, gh, gw), dtype=torch.int64)
    lab = (fg * cls).repeat_interleave(H // gh, 1).repeat_interleave(W // gw, 2).to(torch.int64)
    lab[0] = 0
    return img, lab


def synth_u8_batch(B: int, H: int, W: int, seed: int = 1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, H, W, 4), generator=g, dtype=torch.uint8)
