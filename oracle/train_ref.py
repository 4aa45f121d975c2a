"""CPU restatement of the reference's training step (the `cpu_baseline` "port").

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows reference deadtrees/network/segmodel.py:210-229 (``training_step``): logits ->
``class2one_hot`` -> ``softmax(dim=1)`` -> ``calculate_loss`` (:169-200) with the differentiable
fp32 loss arithmetic of deadtrees/loss/gdl.py:10-27 and deadtrees/loss/losses.py:232-291, then
Lightning's ``gradient_clip_val: 0.5`` (configs/trainer/default.yaml:18, norm clipping) and
``torch.optim.Adam(lr)`` (segmodel.py:420-425).  The differentiable losses here are pinned against
the imported reference by tests/golden/losses_*.npz (values and d loss/d logits).
"""
from __future__ import annotations

import time

import torch

from .unet_ref import UNetR34Ref

EPS = 1e-10


def onehot_f32(mask: torch.Tensor, K: int) -> torch.Tensor:
    return torch.zeros((mask.shape[0], K) + tuple(mask.shape[1:]), dtype=torch.int32).scatter_(
        1, mask[:, None], 1)


def gdice_t(p, t):
    # gdl.py:10-27 (computes in the dtype of p; int32 target promoted)
    cnt = t.sum(dim=(0, 2, 3))
    w = 1.0 / (cnt ** 2 + 1e-9)
    num = (w * (t * p).sum(dim=(0, 2, 3))).sum()
    den = (w * (t + p).sum(dim=(0, 2, 3))).sum()
    return 1.0 - 2.0 * (num + 1e-9) / (den + 1e-9)


def dice_t(p, t, idc):
    pc, tc = p[:, idc].float(), t[:, idc].float()
    inter = (pc * tc).sum(dim=(2, 3))
    union = pc.sum(dim=(2, 3)) + tc.sum(dim=(2, 3))
    return (1.0 - (2 * inter + EPS) / (union + EPS)).mean()


def focal_t(p, t, idc, gamma):
    pc, tc = p[:, idc], t[:, idc].float()
    logp = (pc + EPS).log()
    w = (1 - pc) ** gamma
    return -(w * tc * logp).sum() / (tc.sum() + EPS)


def boundary_t(p, dist, idc):
    return (p[:, idc].float() * dist[:, idc].float()).mean()


def loss_from_logits(logits, mask, losses=("GDICE", "FOCAL"), distmap=None, alpha=1.0):
    K = logits.shape[1]
    t = onehot_f32(mask, K)
    p = logits.softmax(dim=1)
    total = 0
    kind = [n for n in losses if n in ("GDICE", "DICE", "GWDICE")]
    kind = kind[-1] if kind else None      # reference segmodel.py:113-127: the later entry wins
    if kind == "GWDICE":
        from .losses_ref import gwdice
        total = total + gwdice(p, mask)
    elif kind == "GDICE":
        total = total + gdice_t(p, t)
    elif kind == "DICE":
        total = total + dice_t(p, t, list(range(1, K)))
    if ("BOUNDARY" in losses or "BOUNDARY-RAMPED" in losses) and distmap is not None:
        total = total + (alpha if "BOUNDARY-RAMPED" in losses else 1.0) * boundary_t(
            p, distmap, list(range(1, K)))
    if "FOCAL" in losses:
        total = total + focal_t(p, t, list(range(K)), 2)
    return total, p


class RefTrainer:
    """Oracle training loop state: model + Adam + clip, mirroring Lightning defaults."""

    def __init__(self, model: UNetR34Ref, lr: float = 3e-4, clip: float = 0.5,
                 losses=("GDICE", "FOCAL")):
        self.model = model
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        self.clip = clip
        self.losses = tuple(losses)

    def step(self, img, mask, distmap=None):
        self.model.train()
        self.opt.zero_grad(set_to_none=True)
        logits = self.model(img)
        loss, _ = loss_from_logits(logits, mask, self.losses, distmap)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        return float(loss), float(gnorm)


def time_cpu_baseline(img: torch.Tensor, mask: torch.Tensor, steps: int = 3, warmup: int = 1,
                      threads: int | None = None):
    """Timed sample for bench.py's ``cpu_baseline`` (kind "port"): BASELINE config 1 (B=2, 512x512)."""
    from .unet_ref import make_oracle

    if threads:
        torch.set_num_threads(threads)
    batch = img.shape[0]
    model = make_oracle(img.shape[1], 2, seed=0, randomize_bn=False)
    tr = RefTrainer(model)
    ts = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        tr.step(img, mask)
        dt = time.perf_counter() - t0
        if i >= warmup:
            ts.append(dt)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"tiles_per_s": batch / med, "median_s": med, "threads": torch.get_num_threads(),
            "batch": batch, "steps": steps}
