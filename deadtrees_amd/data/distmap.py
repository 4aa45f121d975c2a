"""Signed distance maps for the boundary loss — the loader-side step of the reference
(deadtrees/loss/losses.py:159-178 ``one_hot2dist`` called from data/deadtreedata.py:182-185).

`distmaps_on_device` is the product path: the exact integer EDT kernel `dt_signed_distmap` (csrc/distmap.hip)
builds the maps from the label batch already in HBM, bit-identical to the reference's scipy pass, so the loader
no longer spends 2K scipy EDTs per sample.  `one_hot2dist` / `distmaps_for_batch` keep the reference's
loader-side host function (scipy, as in the reference) for data pipelines that attach maps themselves.
The reference allocates the result with the one-hot's INTEGER dtype,
so fractional distances are truncated toward zero before the cast to float32; reproduced,
because the boundary-loss values of the reference depend on it.
"""
from __future__ import annotations

import numpy as np
import torch


def one_hot2dist(seg_onehot: np.ndarray, resolution=(1, 1)) -> np.ndarray:
    """[K,H,W] one-hot (integer dtype) -> [K,H,W] signed distance, same integer dtype (truncated)."""
    from scipy.ndimage import distance_transform_edt as edt
    res = np.zeros_like(seg_onehot)
    for k in range(seg_onehot.shape[0]):
        pos = seg_onehot[k].astype(bool)
        if pos.any():
            neg = ~pos
            res[k] = edt(neg, sampling=list(resolution)) * neg - (edt(pos, sampling=list(resolution)) - 1) * pos
    return res


def distmaps_for_batch(mask: torch.Tensor, K: int) -> torch.Tensor:
    """int64 labels [B,H,W] -> float32 distance maps [B,K,H,W] (what the reference loader attaches to a batch)."""
    m = mask.cpu().numpy()
    out = np.empty((m.shape[0], K) + m.shape[1:], dtype=np.float32)
    for i in range(m.shape[0]):
        oh = (m[i][None] == np.arange(K)[:, None, None]).astype(np.int32)
        out[i] = one_hot2dist(oh).astype(np.float32)
    return torch.from_numpy(out)


def distmaps_on_device(mask: torch.Tensor, K: int) -> torch.Tensor:
    """int64 labels [B,H,W] on the GPU -> float32 distance maps [B,K,H,W] on the GPU (HIP kernel, no host pass)."""
    from .. import ops
    d, _ = ops.signed_distmap(mask if mask.dtype == torch.int64 else mask.long(), K)
    return d
