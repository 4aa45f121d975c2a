"""Fused segmentation losses and metrics on the HIP kernels (dt_seg_loss_fwd / dt_seg_loss_bwd).

Replaces, for ``SemSegment.calculate_loss`` / ``log_metrics`` (reference deadtrees/network/segmodel.py:
169-208), the chain  ``class2one_hot`` (loss/losses.py:124-141) -> ``logits.softmax(dim=1)``
(segmodel.py:216) -> ``GeneralizedDiceLoss`` (loss/gdl.py:10-27) | ``DiceLoss`` (losses.py:232-247)
| ``GeneralizedWassersteinDiceLoss`` (loss/gwdl.py:84-138, "GWDICE", default weighting) -> ``FocalLoss`` (losses.py:280-291) | ``CrossEntropy`` (:187-196) -> ``BoundaryLoss`` (:256-267)
-> smp ``Fscore`` x2 (segmodel.py:145-149): ONE reduction pass over the logits produces every
per-(sample, class) sum in fp64; the scalar algebra below runs on those few numbers on the device
(no host sync); ONE elementwise pass produces d(loss)/d(logits).

The int32 one-hot tensor of the reference is never materialised (``t_k = [label == k]`` in-kernel) and
the two ``torch.unique(a.cpu())`` host syncs of ``class2one_hot``'s asserts are replaced by a device
flag (`label_error`) that callers may check lazily.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from .. import _lib

NACC = 10
GW_EPS = 2.220446049250313e-16  # np.spacing(1), loss/gwdl.py:92
EPS = 1e-10  # reference loss/losses.py:19


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gwdice_matrix(K: int, device) -> torch.Tensor:
    """label-distance matrix of segmodel.py:119-121 (background far from both tree classes, the two tree
    classes 0.5 apart), cut to K classes; its maximum is 1 so gwdl.py:74-79 does not rescale it."""
    if K not in (2, 3):
        raise NotImplementedError("GWDICE: the reference defines the distance matrix for 2 or 3 classes only")
    key = (K, str(device))
    m = _GW_CACHE.get(key)
    if m is None:   # one host-to-device copy per (K, device), not one per step (and none inside a graph capture)
        m = torch.tensor([[0.0, 1.0, 1.0], [1.0, 0.0, 0.5], [1.0, 0.5, 0.0]], dtype=torch.float32)[:K, :K]
        m = _GW_CACHE[key] = m.contiguous().to(device)
    return m


_GW_CACHE: Dict = {}


def loss_sums(logits: torch.Tensor, labels: torch.Tensor, distmap: Optional[torch.Tensor] = None,
              gamma: float = 2.0, want_probs: bool = False, wass_m: Optional[torch.Tensor] = None):
    """-> acc f64 [B,K,10], probs or None, err_flag int32[1]"""
    if not logits.is_cuda:
        raise RuntimeError("deadtrees_amd losses run only on the HIP device (no CPU fallback)")
    lib = _lib.load()
    B, K, H, W = logits.shape
    logits = logits.contiguous().float()
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        labels = labels.long()
    if tuple(labels.shape) != (B, H, W):
        raise RuntimeError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
    if distmap is not None:
        distmap = distmap.contiguous().float()
        if tuple(distmap.shape) != (B, K, H, W):
            raise RuntimeError("distmap must be [B,K,H,W]")
    n = lib.dt_seg_loss_acc_doubles(B, K, H, W)
    acc = torch.empty(n, dtype=torch.float64, device=logits.device)
    probs = torch.empty_like(logits) if want_probs else None
    err = torch.zeros(1, dtype=torch.int32, device=logits.device)
    possum = None
    if wass_m is not None:
        # cross-sample position sums of the reference's [B,1,S] x [B,S] broadcast (gwdl.py:180-198)
        possum = torch.empty(H * W, dtype=torch.float32, device=logits.device)
        _lib.check(lib.dt_gwdice_possum(_p(logits), _p(labels), _p(wass_m), _p(possum), B, K, H, W, _stream()),
                   "dt_gwdice_possum")
    _lib.check(lib.dt_seg_loss_fwd(_p(logits), _p(labels), _p(distmap), _p(wass_m), _p(possum), float(gamma), _p(acc), _p(probs), _p(err),
                                   B, K, H, W, _stream()), "dt_seg_loss_fwd")
    return acc[:B * K * NACC].view(B, K, NACC), probs, err


_DICE_KINDS = {"GDICE": 0, "DICE": 1, "GWDICE": 2}
_KIND_CODE = dict(_DICE_KINDS, NONE=3)


def loss_forward(logits: torch.Tensor, labels: torch.Tensor, distmap: Optional[torch.Tensor], cfg: dict):
    """reduction pass + device-side scalar algebra -> (parts fp32 [8] on the device, err flag, saved-for-backward).
    parts[i] follows PART_KEYS; parts[7] (= total) is the scalar to differentiate.  No ATen arithmetic, no host sync."""
    losses = cfg["losses"]
    alpha = float(cfg.get("alpha", 1.0))
    gamma = float(cfg.get("gamma", 2.0))
    B, K, H, W = logits.shape
    use_bd = ("BOUNDARY" in losses or "BOUNDARY-RAMPED" in losses) and distmap is not None
    dev = logits.device
    dice_kind = [n for n in losses if n in _DICE_KINDS]
    if not dice_kind:
        if not cfg.get("allow_no_dice"):
            raise AssertionError("a dice term (GDICE, DICE or GWDICE) is mandatory")  # segmodel.py:143
        dice_kind = ["NONE"]   # the stand-alone loss callables of loss/callables.py
    dice_kind = dice_kind[-1]   # segmodel.py:113-127: a later entry replaces self.dice_loss
    wass_m = gwdice_matrix(K, dev) if dice_kind == "GWDICE" else None
    acc, _, err = loss_sums(logits, labels, distmap if use_bd else None, gamma, wass_m=wass_m)
    # one fp32 workspace: parts[8] | coef[B*K*2] | wfocal[2] (+2 pad) | wbound[K -> 4] | wass_a[B] | wass_c[B]
    o_coef, o_wf, o_wb, o_wa = 8, 8 + 2 * B * K, 8 + 2 * B * K + 4, 8 + 2 * B * K + 8
    ws = torch.empty(o_wa + 2 * B, dtype=torch.float32, device=dev)
    parts, coef, wf, wbound = ws[:8], ws[o_coef:o_wf], ws[o_wf:o_wf + 2], ws[o_wb:o_wb + K]
    wass_a, wass_c = (ws[o_wa:o_wa + B], ws[o_wa + B:o_wa + 2 * B]) if wass_m is not None else (None, None)
    c = _lib.LossCfg(_KIND_CODE[dice_kind], 1 if use_bd else 0, 1 if "FOCAL" in losses else 0,
                     alpha if "BOUNDARY-RAMPED" in losses else 1.0, gamma)
    _lib.check(_lib.load().dt_seg_loss_algebra(_p(acc), C.byref(c), B, K, H, W, _p(parts), _p(coef), _p(wf), _p(wbound),
                                               _p(wass_a), _p(wass_c), _stream()), "dt_seg_loss_algebra")
    saved = (logits, labels, distmap if use_bd else None, coef, wf, wbound if use_bd else None, wass_m, wass_c, wass_a)
    return parts, err, saved


def loss_backward(saved, gtotal: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d total / d logits (one elementwise pass); gtotal: optional 0-d fp32 device scalar (upstream gradient)"""
    logits, labels, distmap, coef, wf, wbound, wass_m, wass_c, wass_a = saved
    lib = _lib.load()
    B, K, H, W = logits.shape
    logits = logits.contiguous()
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        labels = labels.long()
    dl = torch.empty_like(logits)
    gs = None
    if gtotal is not None:
        gs = gtotal.reshape(1)
        if gs.dtype != torch.float32:
            gs = gs.float()
    posgrad = None
    if wass_m is not None:
        posgrad = torch.empty(H * W, dtype=torch.float32, device=logits.device)
        _lib.check(lib.dt_gwdice_posgrad(_p(labels), _p(wass_a), _p(posgrad), B, H, W, _stream()),
                   "dt_gwdice_posgrad")
    _lib.check(lib.dt_seg_loss_bwd(_p(logits), _p(labels), _p(distmap), _p(coef), _p(wf), _p(wbound), _p(gs),
                                   _p(wass_m), _p(wass_c), _p(posgrad), _p(dl), B, K, H, W, _stream()), "dt_seg_loss_bwd")
    return dl


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, distmap, cfg):
        parts, err, saved = loss_forward(logits, labels, distmap, cfg)
        ctx.save_for_backward(*saved)
        out_parts = parts[:7]
        total = parts[7]
        ctx.mark_non_differentiable(out_parts, err)
        return total, out_parts, err

    @staticmethod
    def backward(ctx, gtotal, _gparts, _gerr):
        return loss_backward(ctx.saved_tensors, gtotal), None, None, None


PART_KEYS = ("dice_loss", "boundary_loss", "focal_loss", "ce_loss", "dice", "dice_with_bg", "total_loss")


def seg_loss(logits: torch.Tensor, labels: torch.Tensor, distmap: Optional[torch.Tensor] = None,
             losses: Sequence[str] = ("GDICE", "FOCAL"), alpha: float = 1.0, gamma: float = 2.0):
    """-> (total loss [differentiable, 0-d f32], {name: 0-d tensor} parts & metrics, label_error flag)."""
    losses = tuple(losses)
    if "GDICE" in losses and "DICE" in losses:
        raise AssertionError(f"Only GDICE _OR_ DICE allowed {losses}")  # segmodel.py:109-111
    for name in losses:
        if name not in ("GDICE", "GWDICE", "DICE", "FOCAL", "BOUNDARY", "BOUNDARY-RAMPED"):
            raise NotImplementedError(f"The loss component <{name}> is not recognized")  # segmodel.py:136-138
    if labels.dtype != torch.int64:
        labels = labels.long()
    total, parts, err = _SegLoss.apply(logits, labels, distmap, {"losses": losses, "alpha": alpha, "gamma": gamma})
    d = {k: parts[i] for i, k in enumerate(PART_KEYS)}
    return total, d, err
