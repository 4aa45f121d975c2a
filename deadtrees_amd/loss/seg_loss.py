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
    m = torch.tensor([[0.0, 1.0, 1.0], [1.0, 0.0, 0.5], [1.0, 0.5, 0.0]], dtype=torch.float32)[:K, :K]
    return m.contiguous().to(device)


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


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, distmap, cfg):
        losses = cfg["losses"]
        alpha = float(cfg.get("alpha", 1.0))
        gamma = float(cfg.get("gamma", 2.0))
        B, K, H, W = logits.shape
        use_bd = ("BOUNDARY" in losses or "BOUNDARY-RAMPED" in losses) and distmap is not None
        dev = logits.device
        dice_kind = [n for n in losses if n in ("GDICE", "DICE", "GWDICE")]
        dice_kind = dice_kind[-1] if dice_kind else None   # segmodel.py:113-127: a later entry replaces self.dice_loss
        wass_m = gwdice_matrix(K, dev) if dice_kind == "GWDICE" else None
        acc, _, err = loss_sums(logits, labels, distmap if use_bd else None, gamma, wass_m=wass_m)
        cnt, pt, ps, foc, ce, bd, tp, prs, ws, vs = (acc[..., i] for i in range(NACC))
        wass_c = wass_a = None
        coef_a = torch.zeros((B, K), dtype=torch.float64, device=dev)
        coef_c = torch.zeros((B, K), dtype=torch.float64, device=dev)
        parts: Dict[str, torch.Tensor] = {}
        total = torch.zeros((), dtype=torch.float64, device=dev)
        if dice_kind == "GWDICE":
            # gwdl.py:110-138 (weighting_mode "default": alpha = 0 for background, 1 otherwise; mean over samples)
            al = torch.ones(K, dtype=torch.float64, device=dev)
            al[0:1].fill_(0.0)     # fill_: the scalar travels as a kernel argument (HIP-graph capturable)
            gtp = (al[None, :] * vs).sum(1)      # sum_s alpha_i(s) * sum_j (1 - wass_j(s)): the reference's broadcast
            ae = ws.sum(1)
            den = 2.0 * gtp + ae + GW_EPS
            parts["dice_loss"] = (1.0 - (2.0 * gtp + GW_EPS) / den).mean()
            # d loss / d wass_j(s) = sum_i wass_a[i] * alpha_i(s) + wass_c[j]
            wass_a = (2.0 * ae / (den * den) / B).float().contiguous()
            wass_c = ((2.0 * gtp + GW_EPS) / (den * den) / B).float().contiguous()
        elif dice_kind == "GDICE":
            S = cnt.sum(0)
            w = 1.0 / (S * S + 1e-9)
            N = (w * pt.sum(0)).sum()
            D = (w * (cnt.sum(0) + ps.sum(0))).sum()
            parts["dice_loss"] = 1.0 - 2.0 * (N + 1e-9) / (D + 1e-9)
            coef_a += (-2.0 * w / (D + 1e-9))[None, :]
            coef_c += (2.0 * w * (N + 1e-9) / (D + 1e-9) ** 2)[None, :]
        elif dice_kind == "DICE":
            nfg = K - 1
            I, U = pt[:, 1:], ps[:, 1:] + cnt[:, 1:]
            parts["dice_loss"] = (1.0 - (2.0 * I + EPS) / (U + EPS)).mean()
            coef_a[:, 1:] += -2.0 / (U + EPS) / (B * nfg)
            coef_c[:, 1:] += (2.0 * I + EPS) / (U + EPS) ** 2 / (B * nfg)
        else:
            raise AssertionError("a dice term (GDICE, DICE or GWDICE) is mandatory")  # segmodel.py:143
        total = total + parts["dice_loss"]
        wbound = None
        if use_bd:
            nfg = K - 1
            scale = 1.0 / (B * nfg * H * W)
            parts["boundary_loss"] = bd[:, 1:].sum() * scale
            wa = alpha if "BOUNDARY-RAMPED" in losses else 1.0
            total = total + wa * parts["boundary_loss"]
            wbound = torch.full((K,), wa * scale, dtype=torch.float32, device=dev)
            wbound[0:1].fill_(0.0)
        wf = torch.zeros(2, dtype=torch.float32, device=dev)
        wf[1:2].fill_(gamma)
        if "FOCAL" in losses:
            M = cnt.sum() + EPS
            parts["focal_loss"] = -foc.sum() / M
            total = total + parts["focal_loss"]
            wf[0:1].copy_((1.0 / M).float().reshape(1))
        parts["ce_loss"] = -ce.sum() / (cnt.sum() + EPS)  # losses.py:187-196 (not part of total)
        # smp Fscore (threshold 0.5, beta 1, eps 1e-7): ignore_channels=[0] and all channels
        def fscore(sl):
            tps, prsum, gts = tp[:, sl].sum(), prs[:, sl].sum(), cnt[:, sl].sum()
            return (2.0 * tps + 1e-7) / (2.0 * tps + (gts - tps) + (prsum - tps) + 1e-7)
        parts["dice"] = fscore(slice(1, None))
        parts["dice_with_bg"] = fscore(slice(0, None))
        parts["total_loss"] = total
        coef = torch.stack([coef_a, coef_c], dim=-1).float().contiguous()
        ctx.save_for_backward(logits, labels, distmap if use_bd else None, coef, wf, wbound, wass_m, wass_c, wass_a)
        ctx.use_bd = use_bd
        out_parts = torch.stack([parts.get(k, torch.zeros((), dtype=torch.float64, device=dev)).double() for k in
                                 PART_KEYS]).float()
        ctx.mark_non_differentiable(out_parts, err)
        return total.float(), out_parts, err

    @staticmethod
    def backward(ctx, gtotal, _gparts, _gerr):
        logits, labels, distmap, coef, wf, wbound, wass_m, wass_c, wass_a = ctx.saved_tensors
        lib = _lib.load()
        B, K, H, W = logits.shape
        logits = logits.contiguous()
        labels = labels.contiguous()
        dl = torch.empty_like(logits)
        gs = gtotal.reshape(1).float().contiguous()
        posgrad = None
        if wass_m is not None:
            posgrad = torch.empty(H * W, dtype=torch.float32, device=logits.device)
            _lib.check(lib.dt_gwdice_posgrad(_p(labels), _p(wass_a), _p(posgrad), B, H, W, _stream()),
                       "dt_gwdice_posgrad")
        _lib.check(lib.dt_seg_loss_bwd(_p(logits), _p(labels), _p(distmap), _p(coef), _p(wf), _p(wbound), _p(gs),
                                       _p(wass_m), _p(wass_c), _p(posgrad), _p(dl), B, K, H, W, _stream()), "dt_seg_loss_bwd")
        return dl, None, None, None


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
