"""One data-parallel training step of the reference recipe on the HIP path.

What Lightning does around ``SemSegment.training_step`` in the reference (deadtrees/train.py:113 with
configs/trainer/default.yaml): forward -> loss -> ``loss.backward()`` -> ``clip_grad_norm_(0.5)`` ->
``Adam.step()``.  Here the same sequence runs on the hand-written kernels with one flat gradient
buffer; with ``world_size > 1`` (one process per GPU) the gradient buckets are all-reduced (sum) over
RCCL/xGMI as soon as backward has produced them, overlapped with the rest of backward; the 1/N of the
mean is folded into the optimiser's clip coefficient (Lightning-DDP semantics: per-replica BatchNorm
statistics, mean-reduced gradients — SURVEY §8e).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .data.distmap import distmaps_on_device
from .loss.seg_loss import seg_loss
from .network.unet import UNetHIP
from .ops import FlatAdam


class GradReducer:
    """Bucketed asynchronous all-reduce of ranges of the flat gradient buffer."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.pending: List = []
        self.grads: Optional[torch.Tensor] = None

    def attach(self, grads: torch.Tensor):
        self.grads = grads

    def hook(self, name: str, lo: int, hi: int):
        if self.world == 1:
            return
        view = self.grads[lo:hi]
        self.pending.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending.clear()


class HipTrainer:
    def __init__(self, model: UNetHIP, lr: float = 3e-4, clip: float = 0.5,
                 losses: Sequence[str] = ("GDICE", "FOCAL"), distributed: bool = False, group=None,
                 precision: str = "fp32"):
        if not model.flat_params.is_cuda:
            raise RuntimeError("HipTrainer needs the model on an MI355X (model.to('cuda'))")
        self.model = model
        model.deliver_grad_to_autograd = False   # this trainer reads the engine's flat gradient buffer itself
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision {precision!r}: use 'fp32' or 'bf16'")
        model.precision = precision
        self.losses = tuple(losses)
        self.opt = FlatAdam(model.flat_params.data, lr=lr, max_norm=clip)
        self.reducer = GradReducer(group) if distributed else None
        self.world = self.reducer.world if self.reducer else 1
        if self.reducer:
            self.reducer.attach(model._grad_buffer())
            model.engine.grad_hook = self.reducer.hook
        self.last = {}

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0):
        if self.reducer and self.world > 1:
            self.reducer.dist.broadcast(self.model.flat_params.data, src, group=self.reducer.group)
            self.reducer.dist.broadcast(self.model.bn_state, src, group=self.reducer.group)

    def step(self, img: torch.Tensor, mask: torch.Tensor, distmap: Optional[torch.Tensor] = None,
             alpha: float = 1.0):
        """returns the (device) loss tensor; no host synchronisation happens here."""
        m = self.model
        m.train()
        m.flat_params.grad = None
        logits = m(img)
        if distmap is None and any(n.startswith("BOUNDARY") for n in self.losses):
            distmap = distmaps_on_device(mask, logits.shape[1])
        loss, parts, err = seg_loss(logits, mask, distmap, self.losses, alpha=alpha)
        loss.backward()
        if self.reducer:
            self.reducer.wait()
        # non-finite loss -> skip the update (reference segmodel.py:220-222 returns None)
        skip = (~torch.isfinite(loss.detach())).to(torch.int32).reshape(1)
        norm = self.opt.step(m._grad_buffer(), grad_scale=1.0 / self.world, skip_flag=skip)
        m.engine.mark_weights_changed()   # the fused optimiser wrote the flat buffer behind torch's version counter
        m.flat_params.grad = None
        self.last = {"loss": loss.detach(), "parts": parts, "grad_norm": norm, "label_error": err, "skipped": skip}
        return loss.detach()


def fit(trainer: HipTrainer, loader, epochs: int, base_lr: float = 3e-4, t_max: int = 10, to_device=None,
        on_epoch_end=None):
    """Minimal stand-in for ``Trainer.fit`` on the hot path (reference deadtrees/train.py:113): per-batch
    ``HipTrainer.step`` and the per-epoch ``CosineAnnealingLR(T_max)`` of segmodel.py:426-428."""
    from .network.segmodel import cosine_lr, create_combined_batch
    history = []
    for epoch in range(epochs):
        trainer.opt.lr = cosine_lr(base_lr, epoch, t_max)
        losses = []
        for batch in loader:
            img, mask, distmap, _, _ = create_combined_batch(batch) if isinstance(batch, dict) else batch
            if to_device:
                img, mask = img.to(to_device), mask.to(to_device)
                distmap = distmap.to(to_device) if distmap is not None else None
            alpha = min((epoch + 1) * 0.01, 0.99)
            losses.append(trainer.step(img, mask, distmap, alpha=alpha))
        mean = float(torch.stack(losses).mean()) if losses else float("nan")
        history.append({"epoch": epoch, "lr": trainer.opt.lr, "train/total_loss": mean})
        if on_epoch_end:
            on_epoch_end(history[-1])
    return history
