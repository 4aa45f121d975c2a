"""``SemSegment`` — host-side mirror of the reference LightningModule (deadtrees/network/segmodel.py:57-438)
over the MI355X-native network and fused losses.

Kept from the reference: constructor ``SemSegment(network, training)`` (:58), attributes ``model``,
``encoder_weights``, ``classes``, ``classes_int``, ``in_channels``, ``hparams`` (:85-100), the loss-list
parsing and its errors (:109-143), ``alpha`` (:157-160), ``training_step`` / ``validation_step`` /
``test_step`` return contracts (:210-289), the ``{stage}/...`` log keys (:185-208), ``configure_optimizers``
(:420-429) and the non-finite-loss rule (``training_step`` returns ``None`` :220-222).

Different on purpose (MI355X-first): ``self.model`` is ``UNetHIP`` (hand-written HIP kernels) instead of
``smp.Unet``; one-hot, softmax, every loss term and both F-scores come from ONE fused reduction pass
(deadtrees_amd/loss/seg_loss.py) instead of ~25 ATen launches and two ``torch.unique(.cpu())`` host syncs per
step (reference loss/losses.py:37-42,129,139).  The base class is ``pytorch_lightning.LightningModule`` when that
package is importable and a minimal stand-in otherwise (it is absent from this image).
"""
from __future__ import annotations

import csv
import json
import logging
import math
from collections import Counter
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import torch
from torch import Tensor

from ..data.distmap import distmaps_on_device
from ..loss.seg_loss import seg_loss
from ..utils.config import AttrDict, to_attrdict
from .unet import UNetHIP

log = logging.getLogger(__name__)

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl  # type: ignore
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):  # minimal LightningModule-shaped base
        def __init__(self):
            super().__init__()
            self.hparams = AttrDict()
            self.current_epoch = 0
            self.trainer = None
            self.logger = []
            self.logged: Dict[str, list] = {}

        def save_hyperparameters(self, **kw):
            self.hparams.update(kw)

        def log(self, name, value, on_step=False, on_epoch=True, **_):
            self.logged.setdefault(name, []).append(value.detach() if isinstance(value, Tensor) else value)

        def epoch_means(self) -> Dict[str, float]:
            """mean of per-batch values, what Lightning logs with on_epoch=True"""
            out = {k: float(torch.stack([torch.as_tensor(x, dtype=torch.float64).cpu() for x in v]).mean())
                   for k, v in self.logged.items() if v}
            self.logged.clear()
            return out


def concat_extra(img, mask, distmap, lu, stats, *, extra):
    """reference segmodel.py:31-40"""
    e_img, e_mask, e_dist, e_lu, e_stats = list(zip(*extra))
    img = torch.cat((img, *e_img), dim=0)
    mask = torch.cat((mask, *e_mask), dim=0)
    # a loader may attach no maps (None): the boundary loss then builds them on the device from the labels
    distmap = None if distmap is None or any(d is None for d in e_dist) else torch.cat((distmap, *e_dist), dim=0)
    lu = torch.cat((lu, *e_lu), dim=0)
    stats = list(stats) + sum((list(s) for s in e_stats), [])
    return img, mask, distmap, lu, stats


def create_combined_batch(batch: Dict[str, Any]):
    """reference segmodel.py:43-54: ``{"main": (...), "extra_i": (...)}`` -> concatenated tensors"""
    img, mask, distmap, lu, stats = batch["main"]
    extra = [v for k, v in batch.items() if k.startswith("extra")]
    if extra:
        img, mask, distmap, lu, stats = concat_extra(img, mask, distmap, lu, stats, extra=extra)
    return img, mask, distmap, lu, stats


_SUPPORTED_ELSEWHERE = ("resunetplusplus", "resunet++",
                        "efficientunetplusplus", "efficientunet++")


class SemSegment(_Base):
    def __init__(self, network, training):
        super().__init__()
        network = to_attrdict(network)
        training = to_attrdict(training)
        architecture = network.architecture.lower().strip()
        if architecture == "unet":
            Model = UNetHIP
        elif architecture == "resunet":     # the reference's in-tree ResUnet (segmodel.py:66-67): same encoder, residual
            def Model(**kw):                # decoder blocks + 1x1 head, on the same kernels
                return UNetHIP(decoder="resunet", **kw)
        elif architecture in ("unetplusplus", "unet++"):   # smp.UnetPlusPlus (segmodel.py:64-65): dense nested decoder
            def Model(**kw):
                return UNetHIP(decoder="unetplusplus", **kw)
        elif architecture in _SUPPORTED_ELSEWHERE:
            raise NotImplementedError(
                f"architecture {architecture!r} exists in the reference but has no MI355X kernels in this build "
                "(hot path = unet/resnet34, SURVEY.md §8)")
        else:
            raise NotImplementedError(
                "Currently only Unet, ResUnet, Unet++, ResUnet++, and EfficientUnet++ architectures are supported")

        clean = network.copy()
        del clean.architecture
        del clean.losses
        n_classes = len(clean.classes)
        del clean.classes

        self.model = Model(**clean, classes=n_classes)
        if clean.get("encoder_weights") is None:
            log.info("Initializing unset weights with Kaiming")
            self.model.reset_parameters()
        self.encoder_weights = clean.get("encoder_weights")

        if HAVE_LIGHTNING:
            self.save_hyperparameters()
        else:
            self.save_hyperparameters(network=network, training=training)

        self.classes = self.hparams["network"]["classes"]
        self.classes_int = list(range(len(self.classes)))
        self.classes_int_wout_bg = [c for c in self.classes_int if c != 0]
        self.in_channels = self.hparams["network"]["in_channels"]

        self.initial_alpha = 0.01
        self.boundary_loss_ramped = False
        losses = list(network.losses)
        assert (("GDICE" in losses) and ("DICE" in losses)) is False, f"Only GDICE _OR_ DICE allowed {losses}"
        self.loss_names = []
        for comp in losses:
            if comp in ("GDICE", "GWDICE", "DICE", "FOCAL", "BOUNDARY"):
                self.loss_names.append(comp)
            elif comp == "BOUNDARY-RAMPED":
                self.loss_names.append(comp)
                self.boundary_loss_ramped = True
            else:
                raise NotImplementedError(f"The loss component <{comp}> is not recognized")
        log.info(f"Losses: {losses}")
        assert any(n in ("GDICE", "GWDICE", "DICE") for n in self.loss_names)  # "we require GDICE!" (segmodel.py:143)

        self.stats = {"train": Counter(), "val": Counter(), "test": Counter()}
        # device flags "labels outside [0,K) seen" (class2one_hot's assert, losses.py:129): collected per step
        # without a host sync and checked when an epoch ends (`_check_labels`)
        self.label_error = None
        self._label_errors = []
        # device-side confusion counts [2,K,K] per stage (all pixels / lu == 1), instead of concatenating every
        # int64 mask of the epoch (reference validation_epoch_end / test_epoch_end, segmodel.py:291-407)
        self._cm = {}

    # ------------------------------------------------------------------ reference helpers
    @property
    def alpha(self):
        """blending parameter for boundary loss - ramps 0.01 -> 0.99 by epoch (segmodel.py:157-160)"""
        return min((self.current_epoch + 1) * self.initial_alpha, 0.99)

    @staticmethod
    def _as_logits_labels(y_hat: Tensor, y: Tensor):
        """The reference hands ``calculate_loss`` / ``log_metrics`` softmax PROBABILITIES and the int32 ONE-HOT target
        (segmodel.py:215-218); the fused kernels want logits and integer labels.  A 4-D ``y`` marks a reference-style
        call: log(p) is a valid logit vector for p (softmax(log p) == p), argmax recovers the labels."""
        if y.dim() == y_hat.dim():
            return torch.log(y_hat.float().clamp_min(1e-38)), y.argmax(dim=1)
        return y_hat, y

    def calculate_loss(self, y_hat: Tensor, y: Tensor, stage: str, distmap: Optional[Tensor] = None) -> Tensor:
        """compound loss of segmodel.py:169-200 -> loss tensor; logs the reference's ``{stage}/...`` keys.
        Accepts the reference's arguments (probabilities, one-hot) and, natively, (logits, integer labels); the
        per-term values of the last call stay in ``self.last_parts`` (metrics included: no second pass)."""
        logits, mask = self._as_logits_labels(y_hat, y)
        use_dist = distmap if any(n.startswith("BOUNDARY") for n in self.loss_names) else None
        if use_dist is None and any(n.startswith("BOUNDARY") for n in self.loss_names):
            use_dist = distmaps_on_device(mask, logits.shape[1])   # loader attached none: HIP EDT on the labels
        loss, parts, err = seg_loss(logits, mask, use_dist, self.loss_names, alpha=self.alpha)
        self.label_error = err
        self._note_label_error(err)
        self.last_parts = parts
        self.log(f"{stage}/dice_loss", parts["dice_loss"], on_step=False, on_epoch=True)
        if use_dist is not None:
            self.log(f"{stage}/boundary_loss", parts["boundary_loss"], on_step=False, on_epoch=True)
        if "FOCAL" in self.loss_names:
            self.log(f"{stage}/focal_loss", parts["focal_loss"], on_step=False, on_epoch=True)
        self.log(f"{stage}/total_loss", loss, on_step=False, on_epoch=True)
        return loss

    def log_metrics(self, y_hat, y: Optional[Tensor] = None, *, stage: str):
        """segmodel.py:202-208.  ``log_metrics(y_hat, y, stage=...)`` like the reference (probabilities + one-hot, or
        logits + labels), or ``log_metrics(parts, stage=...)`` with the sums a ``calculate_loss`` call already made."""
        if isinstance(y_hat, dict):
            parts = y_hat
        else:
            logits, mask = self._as_logits_labels(y_hat, y)
            _, parts, err = seg_loss(logits.detach(), mask, None, [n for n in self.loss_names if not n.startswith("BOUNDARY")])
            self._note_label_error(err)
        self.log(f"{stage}/dice", parts["dice"], on_step=False, on_epoch=True)
        self.log(f"{stage}/dice_with_bg", parts["dice_with_bg"], on_step=False, on_epoch=True)

    def _note_label_error(self, err):
        """collect a step's device flag; the list is folded into one device scalar every 256 entries (no host sync), so
        a training-only run without validation epochs cannot grow it without bound"""
        self._label_errors.append(err)
        if len(self._label_errors) >= 256:
            self._label_errors = [torch.stack([e.reshape(()) for e in self._label_errors]).max()]

    def training_epoch_end(self, outputs=None):
        """Lightning hook: evaluate the epoch's label flags (the reference asserts per step, losses.py:129)"""
        self._check_labels()

    def _check_labels(self):
        """the lazily evaluated assert of class2one_hot (losses.py:129): one host sync per epoch instead of two per step"""
        errs, self._label_errors = self._label_errors, []
        if errs and int(torch.stack([e.reshape(()) for e in errs]).max()) != 0:
            raise AssertionError(f"labels outside [0, {len(self.classes)}) were seen this epoch (class2one_hot)")

    # ------------------------------------------------------------------ steps
    def training_step(self, batch, batch_idx):
        img, mask, distmap, _, stats = create_combined_batch(batch)
        logits = self.model(img)
        loss = self.calculate_loss(logits, mask, "train", distmap=distmap)
        if torch.isnan(loss) or torch.isinf(loss):   # the one host sync per step the reference also has
            log.warning("Train loss is NaN! What is going on?")
            return None
        self.log_metrics(self.last_parts, stage="train")
        self.stats["train"].update([x["file"] for x in stats])
        return loss

    def validation_step(self, batch, batch_idx):
        img, mask, distmap, lu, stats = create_combined_batch(batch)
        with torch.no_grad():       # Lightning evaluates under no_grad; the stand-in base class must too, or a
            logits = self.model(img)    # grad-enabled forward would keep a second set of saved activations alive
            loss = self.calculate_loss(logits, mask, stage="val", distmap=distmap)
        self.log_metrics(self.last_parts, stage="val")
        self.stats["val"].update([x["file"] for x in stats])
        pred = logits.argmax(dim=1)
        self._accumulate_cm("val", pred, mask, lu)
        return {"val_loss": loss, "target": mask, "prediction": pred, "lu": lu}

    def test_step(self, batch: Tuple[Tensor], batch_idx) -> Dict[str, Any]:
        img, mask, _, lu, stats = batch
        with torch.no_grad():
            logits = self.model(img)
            self.log_metrics(logits, mask, stage="test")
        self.stats["test"].update([x["file"] for x in stats])
        pred = logits.argmax(dim=1)
        self._accumulate_cm("test", pred, mask, lu)
        return {"target": mask, "prediction": pred, "lu": lu}

    def _accumulate_cm(self, stage, pred, mask, lu):
        from ..ops import confusion_matrix
        self._cm[stage], _ = confusion_matrix(pred, mask, lu, K=len(self.classes), counts=self._cm.get(stage))

    def confusion_matrices(self, stage: str):
        """{"cm_px", "cm_norm", "cm_px_masked", "cm_norm_masked"} as the reference's *_epoch_end builds them
        (normalize="true": rows sum to 1); resets the accumulator."""
        cm = self._cm.pop(stage, None)
        if cm is None:
            return {}
        cm = cm.cpu().double()
        out = {}
        for name, m in (("", cm[0]), ("_masked", cm[1])):
            out[f"cm_px{name}"] = m.to(torch.int64)
            out[f"cm_norm{name}"] = m / m.sum(dim=1, keepdim=True).clamp_min(1.0)
        return out

    def validation_epoch_end(self, outputs=None):
        self._check_labels()
        return self.confusion_matrices("val")

    def test_epoch_end(self, outputs=None):
        self._check_labels()
        return self.confusion_matrices("test")

    def teardown(self, stage=None) -> None:
        """reference segmodel.py:409-418: per-file sample counts of the run -> train_stats.csv / val_stats.csv in the
        working directory (columns filename,count — what ``DataFrame.to_csv(index=False)`` writes)"""
        for name in ("train", "val"):
            log.debug(f"len(stats_{name}): {len(self.stats[name])}")
            with open(f"{name}_stats.csv", "w", newline="") as f:
                w = csv.writer(f, lineterminator="\n")
                w.writerow(["filename", "count"])
                w.writerows(dict(self.stats[name]).items())

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.parameters(), lr=self.hparams["training"]["learning_rate"])
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=self.hparams["training"]["cosineannealing_tmax"])
        return [opt], [sch]

    # ------------------------------------------------------------------ checkpoints (plain, pickle-free payload)
    def save_checkpoint(self, path):
        """writes a ``.ckpt`` that ``load_from_checkpoint`` reads with ``weights_only=True``: tensors under the
        reference's ``state_dict`` keys (``model.`` prefix) + hyper-parameters as a JSON string."""
        sd = {f"model.{k}": v for k, v in self.model.smp_state_dict().items()}
        hp = {"network": dict(self.hparams["network"]), "training": dict(self.hparams["training"])}
        torch.save({"state_dict": sd, "hyper_parameters_json": json.dumps(hp)}, str(path))

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **_):
        """Never unpickles arbitrary objects, with or without Lightning installed (Lightning's own
        ``load_from_checkpoint`` is a full-pickle ``torch.load`` that imports and runs whatever the file names): first
        the ``weights_only`` loader (this module's ``save_checkpoint`` format), then the restricted reader of
        utils/ckpt.py for the reference's Lightning ``.ckpt`` files."""
        try:
            ck = torch.load(str(path), map_location=map_location, weights_only=True)
        except Exception:  # noqa: BLE001 - pickled omegaconf / Lightning objects: weights_only refuses them
            ck = None
        if ck is not None and "hyper_parameters_json" in ck:
            hp = json.loads(ck["hyper_parameters_json"])
            m = cls(hp["network"], hp["training"])
            m.model.load_smp_state_dict({k[len("model."):]: v for k, v in ck["state_dict"].items()})
            return m
        # a checkpoint written by the reference's Lightning trainer: tensors through the restricted reader (nothing
        # in the file is imported or executed), network configuration from the tensor shapes, default training conf
        from ..utils.ckpt import infer_network_conf, lightning_state_dict
        from ..utils.config import default_training
        sd = lightning_state_dict(path)
        net = infer_network_conf(sd)
        net["losses"] = ["GDICE", "FOCAL"]
        m = cls(net, default_training())
        m.model.load_smp_state_dict(sd)
        return m


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 0.0) -> float:
    """closed form of torch CosineAnnealingLR stepped once per epoch (segmodel.py:426-428)"""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def initialize_weights(m):
    """reference segmodel.py:432-438: zero biases, Kaiming-normal conv/linear weights, recursively.  ``UNetHIP`` keeps
    its parameters in one flat buffer and re-draws them itself (same distribution: fan_in, gain sqrt 2)."""
    if isinstance(m, UNetHIP):
        m.reset_parameters()
        return
    if getattr(m, "bias", None) is not None:
        torch.nn.init.constant_(m.bias, 0)
    if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
        torch.nn.init.kaiming_normal_(m.weight)
    for c in m.children():
        initialize_weights(c)
