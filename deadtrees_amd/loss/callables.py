"""The reference's loss callables (deadtrees/loss/{losses,gdl,gwdl}.py) as thin fronts of the fused HIP loss.

Same names, constructor arguments and call signature ``loss(probs, target) -> 0-d tensor`` (probabilities in, like
segmodel.py:214-218 hands them over), so code written against ``deadtrees.loss`` keeps working; the arithmetic is
the ONE fused reduction pass of ``seg_loss`` (softmax(log p) == p for a probability vector, so the kernels see
the caller's probabilities), differentiable with respect to ``probs``.

Only the class selections ``SemSegment`` uses are built (segmodel.py:113-134): Dice / Boundary over the non-background
classes, Focal / CrossEntropy over all classes; another ``idc`` raises ``NotImplementedError``.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from .seg_loss import _SegLoss, PART_KEYS


def class2one_hot(seg: Tensor, K: int) -> Tensor:
    """reference loss/losses.py:124-141: int labels [B,H,W] -> int32 one-hot [B,K,H,W] (labels outside [0,K) raise)"""
    if seg.dim() == 2:
        seg = seg.unsqueeze(0)
    if bool(((seg < 0) | (seg >= K)).any()):
        raise AssertionError(f"labels outside [0,{K})")
    b, *img_shape = seg.shape
    return torch.zeros((b, K, *img_shape), dtype=torch.int32, device=seg.device).scatter_(1, seg[:, None, ...].long(), 1)


def one_hot2dist(seg: np.ndarray, resolution=None, dtype=None) -> np.ndarray:
    """reference loss/losses.py:159-178 on the host (scipy), incl. its dtype rule (result in seg's dtype unless
    ``dtype`` is given -> int32 truncation for the loader's int32 one-hot); the device version is ``dt_signed_distmap``"""
    from scipy.ndimage import distance_transform_edt as eucl_distance
    K = len(seg)
    res = np.zeros_like(seg, dtype=dtype)
    for k in range(K):
        posmask = seg[k].astype(bool)
        if posmask.any():
            negmask = ~posmask
            res[k] = eucl_distance(negmask, sampling=resolution) * negmask - (
                eucl_distance(posmask, sampling=resolution) - 1) * posmask
    return res


def _labels_of(target: Tensor) -> Tensor:
    return target.argmax(dim=1) if target.dim() == 4 else target


def _fused(probs: Tensor, target: Tensor, distmap: Optional[Tensor], losses: Sequence[str], gamma: float = 2.0):
    if not probs.is_cuda:
        raise RuntimeError("deadtrees_amd losses run only on the HIP device (no CPU fallback)")
    logits = torch.log(probs.float().clamp_min(1e-38))
    total, parts, _ = _SegLoss.apply(logits, _labels_of(target).long(), distmap,
                                     {"losses": tuple(losses), "gamma": gamma, "allow_no_dice": True})
    return total, {k: parts[i] for i, k in enumerate(PART_KEYS)}


def _check_idc(name, idc, want, K):
    if list(idc) != list(want):
        raise NotImplementedError(f"{name}(idc={list(idc)}): the HIP kernels implement the selection SemSegment uses "
                                  f"({list(want)} for {K} classes)")


class GeneralizedDiceLoss(torch.nn.Module):
    """reference loss/gdl.py:6-27"""

    def forward(self, inp: Tensor, targ: Tensor) -> Tensor:
        return _fused(inp, targ, None, ("GDICE",))[0]


class DiceLoss:
    """reference loss/losses.py:226-247 (idc = the non-background classes, segmodel.py:127)"""

    def __init__(self, **kwargs):
        self.idc = list(kwargs["idc"])

    def __call__(self, probs: Tensor, target: Tensor) -> Tensor:
        K = probs.shape[1]
        _check_idc("DiceLoss", self.idc, range(1, K), K)
        return _fused(probs, target, None, ("DICE",))[0]


class FocalLoss:
    """reference loss/losses.py:273-291 (idc = all classes, gamma = 2: segmodel.py:129)"""

    def __init__(self, **kwargs):
        self.idc, self.gamma = list(kwargs["idc"]), float(kwargs["gamma"])

    def __call__(self, probs: Tensor, target: Tensor) -> Tensor:
        K = probs.shape[1]
        _check_idc("FocalLoss", self.idc, range(K), K)
        return _fused(probs, target, None, ("FOCAL",), gamma=self.gamma)[0]


class CrossEntropy:
    """reference loss/losses.py:181-196 = the focal loss with gamma 0"""

    def __init__(self, **kwargs):
        self.idc = list(kwargs["idc"])

    def __call__(self, probs: Tensor, target: Tensor) -> Tensor:
        K = probs.shape[1]
        _check_idc("CrossEntropy", self.idc, range(K), K)
        return _fused(probs, target, None, ("FOCAL",), gamma=0.0)[0]


class SurfaceLoss:
    """reference loss/losses.py:250-267 (idc = the non-background classes)"""

    def __init__(self, **kwargs):
        self.idc = list(kwargs["idc"])

    def __call__(self, probs: Tensor, dist_maps: Tensor) -> Tensor:
        K = probs.shape[1]
        _check_idc("SurfaceLoss", self.idc, range(1, K), K)
        labels = probs.detach().argmax(dim=1)     # the boundary term does not read the labels; any valid map will do
        return _fused(probs, labels, dist_maps.float(), ("BOUNDARY",))[0]


BoundaryLoss = SurfaceLoss


class GeneralizedWassersteinDiceLoss(torch.nn.Module):
    """reference loss/gwdl.py:18-253, ``weighting_mode="default"``, ``reduction="mean"`` and the label-distance matrix of
    segmodel.py:119-121 (the only configuration the reference instantiates); target = int labels [B,H,W]."""

    def __init__(self, dist_matrix, weighting_mode: str = "default", reduction: str = "mean"):
        super().__init__()
        m = np.asarray(dist_matrix, dtype=np.float64)
        want = np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 0.5], [1.0, 0.5, 0.0]])
        if m.shape[0] != m.shape[1] or m.shape[0] not in (2, 3) or not np.array_equal(m, want[:m.shape[0], :m.shape[0]]):
            raise NotImplementedError("GeneralizedWassersteinDiceLoss: only the reference's own distance matrix is built")
        if weighting_mode != "default" or reduction != "mean":
            raise NotImplementedError("GeneralizedWassersteinDiceLoss: weighting_mode 'default' / reduction 'mean' only")
        self.num_classes = m.shape[0]

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if input.shape[1] != self.num_classes:
            raise ValueError(f"input has {input.shape[1]} classes, the distance matrix {self.num_classes}")
        return _fused(input, target, None, ("GWDICE",))[0]
