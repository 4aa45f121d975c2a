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
from .loss.seg_loss import PART_KEYS, loss_backward, loss_forward
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
                 precision: str = "fp32", graph: bool = False):
        """graph=True: after two eager steps the whole step (forward, loss, backward, clip, Adam) is captured into a
        HIP graph and replayed — ~750 kernel launches become one, which matters once the bf16 step is shorter than
        the Python launch path."""
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
        # with distributed=True the bucketed RCCL all-reduces are captured too (rehearsed at world size 1 on RCCL 2.26;
        # bench.py keeps multi-GPU runs eager until that has run on a real 8-GPU node)
        self.use_graph = bool(graph)
        self._graph = None

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0):
        if self.reducer and self.world > 1:
            self.reducer.dist.broadcast(self.model.flat_params.data, src, group=self.reducer.group)
            self.reducer.dist.broadcast(self.model.bn_state, src, group=self.reducer.group)

    def step(self, img: torch.Tensor, mask: torch.Tensor, distmap: Optional[torch.Tensor] = None,
             alpha: float = 1.0):
        """returns the (device) loss tensor; no host synchronisation happens here."""
        if self.use_graph:
            return self._graph_step(img, mask, distmap, alpha)
        return self._eager_step(img, mask, distmap, alpha)

    # ------------------------------------------------------------------ HIP-graph replay of the whole step
    def _graph_step(self, img, mask, distmap, alpha):
        # alpha (the per-epoch boundary ramp of segmodel.py:157-160) is baked into the captured loss blend: it belongs to
        # the key only where it is read (BOUNDARY-RAMPED) — otherwise fit()'s ramp would re-capture the step every epoch
        key = (tuple(img.shape), img.dtype, tuple(mask.shape), mask.dtype,
               None if distmap is None else tuple(distmap.shape),
               float(alpha) if "BOUNDARY-RAMPED" in self.losses else None)
        g = self._graph
        if g is None or g["key"] != key:
            self._graph = g = {"key": key, "warm": 0}     # new shapes / loss blend: drop the old graph, warm up again
        if "graph" not in g:
            if g["warm"] < 2:    # eager warm-up: lazy initialisation (workspaces, autograd) must not be captured
                g["warm"] += 1
                return self._eager_step(img, mask, distmap, alpha)
            self._capture(g, img, mask, distmap, alpha)
        # a loader that writes its batches straight into the graph's static buffers (`static_batch()`) hands them back
        # here: nothing to copy (at B = 64 the image copy alone is 201 MB = 0.2 ms of a 25 ms step)
        if img is not g["img"]:
            g["img"].copy_(img)
        if mask is not g["mask"]:
            g["mask"].copy_(mask)
        if distmap is not None and distmap is not g["distmap"]:
            g["distmap"].copy_(distmap)
        self.opt.sync_lr()                   # a changed learning rate reaches the replay through lr_dev
        self.opt.t += 1                      # host mirror; the authoritative count is the device's t_dev
        g["graph"].replay()
        self.model.engine.mark_weights_changed()
        self.model.num_batches_tracked += 1  # module bookkeeping outside the graph (smp state_dict key)
        self.last = g["last"]
        return g["last"]["loss"]

    def static_batch(self):
        """(img, mask, distmap) buffers the captured step reads — available once the graph exists (after the eager
        warm-up steps); a data pipeline that fills THESE tensors (H2D copies, device-side augmentation) and passes them
        to ``step`` saves the per-step staging copy.  None before capture or without graph replay."""
        g = self._graph
        if not self.use_graph or g is None or "graph" not in g:
            return None
        return g["img"], g["mask"], g["distmap"]

    def _capture(self, g, img, mask, distmap, alpha):
        eng = self.model.engine
        g["img"], g["mask"] = img.clone(), mask.clone()
        g["distmap"] = None if distmap is None else distmap.clone()
        self.opt.sync_lr()
        t_host = self.opt.t
        eager_ws, eng._ws = eng._ws, {}       # workspaces of the captured step live (and stay) in the graph's pool
        graph = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                self._eager_step(g["img"], g["mask"], g["distmap"], alpha, capturing=True)
        finally:
            g["ws"], eng._ws = eng._ws, eager_ws
        self.opt.t = t_host                   # capture launched nothing: the step count has not moved
        g["graph"], g["last"] = graph, self.last

    def _eager_step(self, img, mask, distmap, alpha, capturing: bool = False):
        """forward -> fused loss -> hand-scheduled backward -> (all-reduce) -> clip + Adam, straight on the C ABI:
        no autograd graph, no ATen arithmetic; every launch is the same for every step (HIP-graph capturable)."""
        m, eng, opt = self.model, self.model.engine, self.opt
        m.train()
        m._require_gpu(img)
        params = m.flat_params.detach()
        grads = m._grad_buffer()
        with torch.no_grad():
            x = img if img.dtype == torch.float32 else img.float()
            if m.precision == "bf16":
                logits = eng.forward_bf16_train(x, params, m.bn_state)
            else:
                logits, _ = eng.forward(x, params, m.bn_state, True, save=True)
            if not capturing:
                m.num_batches_tracked += 1
            if distmap is None and any(n.startswith("BOUNDARY") for n in self.losses):
                distmap = distmaps_on_device(mask, logits.shape[1])
            parts, err, saved = loss_forward(logits, mask, distmap, {"losses": self.losses, "alpha": alpha})
            dl = loss_backward(saved)
            if m.precision == "bf16":
                eng.backward_bf16(dl, params, grads)
            else:
                eng.backward(dl, params, grads)
            if self.reducer:
                self.reducer.wait()
            # non-finite loss -> skip the update (reference segmodel.py:220-222 returns None).  The decision must be
            # GLOBAL: the gradient buckets are already summed over the replicas, so one rank's NaN poisons everyone's
            # update — all ranks skip together (one 4-byte MAX all-reduce on the same group)
            loss = parts[7]
            skip = opt.skip_from_loss(loss)
            if self.reducer and self.world > 1:
                self.reducer.dist.all_reduce(skip, op=self.reducer.dist.ReduceOp.MAX, group=self.reducer.group)
            norm = opt.step(grads, grad_scale=1.0 / self.world, skip_flag=skip, capturing=capturing)
        eng.mark_weights_changed()   # the fused optimiser wrote the flat buffer behind torch's version counter
        self.last = {"loss": loss, "parts": {k: parts[i] for i, k in enumerate(PART_KEYS)}, "grad_norm": norm,
                     "label_error": err, "skipped": skip}
        return loss


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
