"""CPU restatement of the reference's loss / metric arithmetic.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned: ``oracle/make_golden.py``
runs the *imported* reference callables on seeded inputs and stores inputs and
outputs in tests/golden/losses_*.npz; tests/test_oracle_golden.py checks this
file against those vectors.

All functions take softmax *probabilities* ``p`` [B,K,H,W] and integer labels
``mask`` [B,H,W] (the one-hot target of the reference is implied by ``mask``),
accumulate in float64 and return python floats / float64 tensors so they can
serve as the high-precision side of a tolerance budget.
"""
from __future__ import annotations

import numpy as np
import torch

EPS = 1e-10  # reference deadtrees/loss/losses.py:19


def one_hot(mask: torch.Tensor, K: int) -> torch.Tensor:
    """reference losses.py:124-141 ``class2one_hot``: int32 [B,K,H,W]; labels must lie in [0,K)."""
    lo, hi = int(mask.min()), int(mask.max())
    if lo < 0 or hi >= K:
        raise AssertionError((sorted(set(mask.flatten().tolist())), K))
    planes = [(mask == k) for k in range(K)]
    return torch.stack(planes, dim=1).to(torch.int32)


def gdice(p: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """reference deadtrees/loss/gdl.py:10-27 (batch-global generalised Dice)."""
    K = p.shape[1]
    t = one_hot(mask, K).to(torch.float64)
    p = p.to(torch.float64)
    cnt = t.sum(dim=(0, 2, 3))
    w = 1.0 / (cnt * cnt + 1e-9)
    num = (w * (t * p).sum(dim=(0, 2, 3))).sum()
    den = (w * (t + p).sum(dim=(0, 2, 3))).sum()
    return 1.0 - 2.0 * (num + 1e-9) / (den + 1e-9)


def dice(p: torch.Tensor, mask: torch.Tensor, idc) -> torch.Tensor:
    """reference losses.py:232-247 ``DiceLoss`` (per sample, per class in ``idc``; mean)."""
    K = p.shape[1]
    t = one_hot(mask, K).to(torch.float64)[:, idc]
    pc = p.to(torch.float64)[:, idc]
    inter = (pc * t).sum(dim=(2, 3))
    union = pc.sum(dim=(2, 3)) + t.sum(dim=(2, 3))
    return (1.0 - (2.0 * inter + EPS) / (union + EPS)).mean()


def focal(p: torch.Tensor, mask: torch.Tensor, idc, gamma: float) -> torch.Tensor:
    """reference losses.py:280-291 ``FocalLoss``; gamma=0 gives losses.py:187-196 ``CrossEntropy``."""
    K = p.shape[1]
    t = one_hot(mask, K).to(torch.float64)[:, idc]
    pc = p.to(torch.float64)[:, idc]
    logp = torch.log(pc + EPS)
    wgt = (1.0 - pc) ** gamma if gamma != 0 else torch.ones_like(pc)
    return -(wgt * t * logp).sum() / (t.sum() + EPS)


def cross_entropy(p, mask, idc):
    return focal(p, mask, idc, 0.0)


def boundary(p: torch.Tensor, distmap: torch.Tensor, idc) -> torch.Tensor:
    """reference losses.py:256-267 ``SurfaceLoss``: mean over [B,|idc|,H,W] of p*dist."""
    return (p.to(torch.float64)[:, idc] * distmap.to(torch.float64)[:, idc]).mean()


GWDICE_M3 = [[0.0, 1.0, 1.0], [1.0, 0.0, 0.5], [1.0, 0.5, 0.0]]  # reference segmodel.py:119


def gwdice(p: torch.Tensor, mask: torch.Tensor, M=None) -> torch.Tensor:
    """reference loss/gwdl.py:84-138 ``GeneralizedWassersteinDiceLoss`` (weighting_mode "default", reduction
    "mean") as SemSegment calls it (segmodel.py:117-124,176): the input is already a PROBABILITY map and the loss
    applies softmax to it once more (gwdl.py:104); M defaults to the matrix of segmodel.py:119 cut to K classes.
    Differentiable; works in the dtype of ``p``."""
    B, K = p.shape[0], p.shape[1]
    Mt = torch.tensor(GWDICE_M3 if M is None else M, dtype=p.dtype)[:K, :K]
    eps = float(np.spacing(1))
    q = p.reshape(B, K, -1).softmax(dim=1)                      # gwdl.py:100-104
    t = mask.reshape(B, -1)
    wass = (Mt[t].permute(0, 2, 1) * q).sum(dim=1)              # gwdl.py:131-178: sum_l M[t_s, l] q_l
    alpha = torch.ones(K, dtype=p.dtype)
    alpha[0] = 0.0                                              # gwdl.py:246-252
    # gwdl.py:180-198 multiplies alpha [B,1,S] with (1 - wass) [B,S]: the shapes broadcast to [B,B,S], so the
    # "true positives" of sample i sum alpha_i(s) * (1 - wass_j(s)) over EVERY sample j of the batch.  The
    # reference trains with that; it is reproduced (it reduces to the published formula for B = 1).
    tp = (alpha[t] * (1.0 - wass).sum(dim=0, keepdim=True)).sum(dim=1)
    all_err = wass.sum(dim=1)
    dice_ = (2.0 * tp + eps) / (2.0 * tp + all_err + eps)       # gwdl.py:126-128
    return (1.0 - dice_).mean()


def dist_map(onehot_sample: np.ndarray) -> np.ndarray:
    """reference losses.py:159-178 ``one_hot2dist`` with ``resolution=[1,1]`` as called from
    deadtrees/data/deadtreedata.py:182-185.  NOTE the reference allocates the result with the
    one-hot's integer dtype, so fractional distances are truncated toward zero;
    that quirk is reproduced here."""
    from scipy.ndimage import distance_transform_edt as edt

    out = np.zeros_like(onehot_sample)
    for k in range(onehot_sample.shape[0]):
        pos = onehot_sample[k].astype(bool)
        if pos.any():
            neg = ~pos
            out[k] = edt(neg, sampling=[1, 1]) * neg - (edt(pos, sampling=[1, 1]) - 1) * pos
    return out


def fscore(p: torch.Tensor, mask: torch.Tensor, ignore_channels=(), threshold: float = 0.5,
           beta: float = 1.0, eps: float = 1e-7) -> torch.Tensor:
    """smp ``utils.metrics.Fscore`` as used at reference segmodel.py:145-149,202-208.
    PARITY UNPINNED (smp not importable): restated from its published definition."""
    K = p.shape[1]
    keep = [k for k in range(K) if k not in set(ignore_channels)]
    gt = one_hot(mask, K).to(torch.float64)[:, keep]
    pr = (p[:, keep] > threshold).to(torch.float64)
    tp = (gt * pr).sum()
    fp = pr.sum() - tp
    fn = gt.sum() - tp
    b2 = beta * beta
    return ((1 + b2) * tp + eps) / ((1 + b2) * tp + b2 * fn + fp + eps)


def compound_loss(p, mask, losses=("GDICE", "FOCAL"), distmap=None, alpha: float = 1.0):
    """reference segmodel.py:169-200 ``calculate_loss`` for the loss names parsed at :113-138."""
    K = p.shape[1]
    total = 0.0
    parts = {}
    if "GDICE" in losses:
        assert "DICE" not in losses  # segmodel.py:109-111
        parts["dice_loss"] = gdice(p, mask)
    elif "DICE" in losses:
        parts["dice_loss"] = dice(p, mask, list(range(1, K)))
    else:
        raise AssertionError("dice term is mandatory (segmodel.py:143)")
    total = total + parts["dice_loss"]
    if ("BOUNDARY" in losses or "BOUNDARY-RAMPED" in losses) and distmap is not None:
        parts["boundary_loss"] = boundary(p, distmap, list(range(1, K)))
        total = total + (alpha if "BOUNDARY-RAMPED" in losses else 1.0) * parts["boundary_loss"]
    if "FOCAL" in losses:
        parts["focal_loss"] = focal(p, mask, list(range(K)), 2.0)
        total = total + parts["focal_loss"]
    parts["total_loss"] = total
    return total, parts


# ---- block split / merge (reference deadtrees/utils/data_handling.py:9-34) -------------------

def make_blocks(x: np.ndarray, d: int) -> np.ndarray:
    """[C,M,N] -> [(M/d)*(N/d), C, d, d], row-major over block rows then block cols."""
    c, m, n = x.shape
    out = np.empty(((m // d) * (n // d), c, d, d), dtype=x.dtype)
    i = 0
    for by in range(m // d):
        for bx in range(n // d):
            out[i] = x[:, by * d:(by + 1) * d, bx * d:(bx + 1) * d]
            i += 1
    return out


def unmake_blocks(x, d: int, m: int, n: int) -> np.ndarray:
    """inverse for single-plane blocks: sequence of [b,d,d] arrays -> [m,n]."""
    flat = np.concatenate([np.asarray(a) for a in x]) if not isinstance(x, np.ndarray) or x.ndim == 4 else x
    flat = flat.reshape(-1, d, d)
    out = np.empty((m, n), dtype=flat.dtype)
    i = 0
    for by in range(m // d):
        for bx in range(n // d):
            out[by * d:(by + 1) * d, bx * d:(bx + 1) * d] = flat[i]
            i += 1
    return out
