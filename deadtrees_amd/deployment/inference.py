"""``PyTorchInference`` mirror (reference deadtrees/deployment/inference.py:30-62) on the HIP path.

Same contract: ``PyTorchInference(ckpt).run(tensor, device) -> int64 class map`` (``.squeeze()``-d), RGB slice
for a 3-channel model fed RGBN, ``ValueError`` for non-``.ckpt`` files, ``TypeError`` for non-tensors.
MI355X-first difference: forward + argmax are one fused pass (the head kernel emits the class map; the
logits never leave the GPU), and ``run_u8`` accepts raw uint8 tiles (normalisation fused on the device).
"""
from __future__ import annotations

from pathlib import Path
from typing import Union

import torch

from ..data.synthetic import MEAN, STD
from ..network.segmodel import SemSegment
from .. import ops


class PyTorchInference:
    def __init__(self, model_file: Union[str, Path]) -> None:
        self._model_file = model_file if isinstance(model_file, Path) else Path(model_file)
        if self._model_file.suffix != ".ckpt":
            raise ValueError(f"ckpt file expected, but {self._model_file.suffix} received")
        model = SemSegment.load_from_checkpoint(self._model_file)
        model.eval()
        self._channels = model.in_channels
        self._model = model.model

    @property
    def model_file(self) -> str:
        return self._model_file.name

    def run(self, input_tensor, device: str = "cuda"):
        if not isinstance(input_tensor, torch.Tensor):
            raise TypeError("no pytorch tensor provided")
        self._model.to(device)
        if input_tensor.dim() == 3:
            input_tensor = input_tensor.unsqueeze(0)
        if self._channels == 3 and input_tensor.shape[1] == 4:
            input_tensor = input_tensor[:, 0:3, :, :]     # rgb model but rgbn data (inference.py:57-59)
        out = self._model.predict_classes(input_tensor.to(device).contiguous(), dtype="int64")
        return out.squeeze()

    def run_u8(self, tiles_u8_nhwc: torch.Tensor, device: str = "cuda") -> torch.Tensor:
        """uint8 [B,H,W,4] (what the tiler cuts) -> uint8 class map [B,H,W]; normalisation on the device
        (reference: per-tile albumentations Normalize on the CPU, scripts/inference.py:94-96)."""
        self._model.to(device)
        x = ops.normalize_u8(tiles_u8_nhwc.to(device), MEAN, STD, self._channels)   # NHWC f32
        x = x.permute(0, 3, 1, 2).contiguous()
        return self._model.predict_classes(x, dtype="uint8")
