"""Fused segmentation losses and metrics on the HIP kernels (dt_seg_loss_fwd / dt_seg_loss_bwd).

Replaces, for ``SemSegment.calculate_loss`` / ``log_metrics`` (reference deadtrees/network/segmodel.py:
169-208), the chain  ``class2one_hot`` (loss/losses.py:124-141) -> ``logits.softmax(dim=1)``
(segmodel.py:216) -> ``GeneralizedDiceLoss`` (loss/gdl.py:10-27) | ``DiceLoss`` (losses.py:232-247)
-> ``FocalLoss`` (losses.py:280-291) | ``CrossEntropy`` (:187-196) -> ``BoundaryLoss`` (:256-267)
-> smp ``Fscore`` x2 (segmodel.py:145-149): ONE reduction pass over the logits produces every
per-(sample, class) sum in fp64; the scalar algebra below runs on those few numbers on the device
(no host sync); ONE elementwise pass produces d(loss)/d(logits).

The int32 one-hot tensor of the reference is never materialised (``t_k = [label == k]`` in-kernel) and
the two ``torch.unique(a.cpu())`` host syncs of ``class2one_hot``'s asserts are replaced by a device
flag (`label_error`) that callers may check lazily.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from .. import _lib

NACC = 8
EPS = 1e-10  # reference loss/losses.py:19


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def loss_sums(logits: torch.Tensor, labels: torch.Tensor, distmap: Optional[torch.Tensor] = None,
              gamma: float = 2.0, want_probs: bool = False):
    """-> acc f64 [B,K,8], probs or None, err_flag int32[1]"""
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
    _lib.check(lib.dt_seg_loss_fwd(_p(logits), _p(labels), _p(distmap), float(gamma), _p(acc), _p(probs), _p(err),
                                   B, K, H, W, _stream()), "dt_seg_loss_fwd")
    return acc[:B * K * NACC].view(B, K, NACC), probs, err


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, distmap, cfg):
        losses = cfg["losses"]
        alpha = float(cfg.get("alpha", 1.0))
        gamma = float(cfg.get("gamma", 2.0))
        B, K, H, W = logits.shape
        use_bd = ("BOUNDARY" in losses or "BOUNDARY-RAMPED" in losses) and distmap is not None
        acc, _, err = loss_sums(logits, labels, distmap if use_bd else None, gamma)
        cnt, pt, ps, foc, ce, bd, tp, prs = (acc[..., i] for i in range(NACC))
        dev = logits.device
        coef_a = torch.zeros((B, K), dtype=torch.float64, device=dev)
        coef_c = torch.zeros((B, K), dtype=torch.float64, device=dev)
        parts: Dict[str, torch.Tensor] = {}
        total = torch.zeros((), dtype=torch.float64, device=dev)
        if "GDICE" in losses:
            S = cnt.sum(0)
            w = 1.0 / (S * S + 1e-9)
            N = (w * pt.sum(0)).sum()
            D = (w * (cnt.sum(0) + ps.sum(0))).sum()
            parts["dice_loss"] = 1.0 - 2.0 * (N + 1e-9) / (D + 1e-9)
            coef_a += (-2.0 * w / (D + 1e-9))[None, :]
            coef_c += (2.0 * w * (N + 1e-9) / (D + 1e-9) ** 2)[None, :]
        elif "DICE" in losses:
            nfg = K - 1
            I, U = pt[:, 1:], ps[:, 1:] + cnt[:, 1:]
            parts["dice_loss"] = (1.0 - (2.0 * I + EPS) / (U + EPS)).mean()
            coef_a[:, 1:] += -2.0 / (U + EPS) / (B * nfg)
            coef_c[:, 1:] += (2.0 * I + EPS) / (U + EPS) ** 2 / (B * nfg)
        else:
            raise AssertionError("a dice term (GDICE or DICE) is mandatory")  # segmodel.py:143
        total = total + parts["dice_loss"]
        wbound = None
        if use_bd:
            nfg = K - 1
            scale = 1.0 / (B * nfg * H * W)
            parts["boundary_loss"] = bd[:, 1:].sum() * scale
            wa = alpha if "BOUNDARY-RAMPED" in losses else 1.0
            total = total + wa * parts["boundary_loss"]
            wbound = torch.full((K,), wa * scale, dtype=torch.float32, device=dev)
            wbound[0] = 0.0
        wf = torch.zeros(2, dtype=torch.float32, device=dev)
        wf[1] = gamma
        if "FOCAL" in losses:
            M = cnt.sum() + EPS
            parts["focal_loss"] = -foc.sum() / M
            total = total + parts["focal_loss"]
            wf[0] = (1.0 / M).float()
        parts["ce_loss"] = -ce.sum() / (cnt.sum() + EPS)  # losses.py:187-196 (not part of total)
        # smp Fscore (threshold 0.5, beta 1, eps 1e-7): ignore_channels=[0] and all channels
        def fscore(sl):
            tps, prsum, gts = tp[:, sl].sum(), prs[:, sl].sum(), cnt[:, sl].sum()
            return (2.0 * tps + 1e-7) / (2.0 * tps + (gts - tps) + (prsum - tps) + 1e-7)
        parts["dice"] = fscore(slice(1, None))
        parts["dice_with_bg"] = fscore(slice(0, None))
        parts["total_loss"] = total
        coef = torch.stack([coef_a, coef_c], dim=-1).float().contiguous()
        ctx.save_for_backward(logits, labels, distmap if use_bd else None, coef, wf, wbound)
        ctx.use_bd = use_bd
        out_parts = torch.stack([parts.get(k, torch.zeros((), dtype=torch.float64, device=dev)).double() for k in
                                 PART_KEYS]).float()
        ctx.mark_non_differentiable(out_parts, err)
        return total.float(), out_parts, err

    @staticmethod
    def backward(ctx, gtotal, _gparts, _gerr):
        logits, labels, distmap, coef, wf, wbound = ctx.saved_tensors
        lib = _lib.load()
        B, K, H, W = logits.shape
        logits = logits.contiguous()
        labels = labels.contiguous()
        dl = torch.empty_like(logits)
        gs = gtotal.reshape(1).float().contiguous()
        _lib.check(lib.dt_seg_loss_bwd(_p(logits), _p(labels), _p(distmap), _p(coef), _p(wf), _p(wbound), _p(gs),
                                       _p(dl), B, K, H, W, _stream()), "dt_seg_loss_bwd")
        return dl, None, None, None


PART_KEYS = ("dice_loss", "boundary_loss", "focal_loss", "ce_loss", "dice", "dice_with_bg", "total_loss")


def seg_loss(logits: torch.Tensor, labels: torch.Tensor, distmap: Optional[torch.Tensor] = None,
             losses: Sequence[str] = ("GDICE", "FOCAL"), alpha: float = 1.0, gamma: float = 2.0):
    """-> (total loss [differentiable, 0-d f32], {name: 0-d tensor} parts & metrics, label_error flag)."""
    losses = tuple(losses)
    if "GDICE" in losses and "DICE" in losses:
        raise AssertionError(f"Only GDICE _OR_ DICE allowed {losses}")  # segmodel.py:109-111
    for name in losses:
        if name not in ("GDICE", "DICE", "FOCAL", "BOUNDARY", "BOUNDARY-RAMPED"):
            raise NotImplementedError(f"The loss component <{name}> is not recognized")  # segmodel.py:136-138
    if labels.dtype != torch.int64:
        labels = labels.long()
    total, parts, err = _SegLoss.apply(logits, labels, distmap, {"losses": losses, "alpha": alpha, "gamma": gamma})
    d = {k: parts[i] for i, k in enumerate(PART_KEYS)}
    return total, d, err
