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


class Inference:
    """reference deployment/inference.py:14-27"""

    @property
    def in_channels(self):
        """band planes the network reads (infer_tile copies only these to the device); None while unknown"""
        return getattr(self, "_channels", None)

    def __init__(self, model_file: Union[str, Path]) -> None:
        self._model_file = model_file if isinstance(model_file, Path) else Path(model_file)

    @property
    def model_file(self) -> str:
        return self._model_file.name

    def run(self, input_tensor):  # pragma: no cover - abstract
        raise NotImplementedError


class ONNXInference(Inference):
    """reference deployment/inference.py:119-144.  The ONNX export of the reference (scripts/create_onnx.py) cannot
    carry the HIP kernels and onnxruntime is not part of this build: the class exists so that imports resolve."""

    def __init__(self, model_file) -> None:
        super().__init__(model_file)
        if self._model_file.suffix != ".onnx":
            raise ValueError(f"onnx file expected, but {self._model_file.suffix} received")
        raise NotImplementedError("ONNX inference is outside the MI355X hot path (SURVEY.md §8b); use PyTorchInference")


class PyTorchInference(Inference):
    def __init__(self, model_file: Union[str, Path]) -> None:
        super().__init__(model_file)
        if self._model_file.suffix != ".ckpt":
            raise ValueError(f"ckpt file expected, but {self._model_file.suffix} received")
        model = SemSegment.load_from_checkpoint(self._model_file)
        model.eval()
        self._channels = model.in_channels
        self._model = model.model

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
        x = ops.normalize_u8(tiles_u8_nhwc.to(device), MEAN, STD, self._channels)   # NHWC f32: the kernels' layout
        return self._model.predict_classes(x, dtype="uint8", nhwc=True)

    def run_blocks(self, raster_chw_u8: torch.Tensor, d: int, first: int, count: int) -> torch.Tensor:
        """sub-tiles ``first`` .. ``first + count - 1`` of a band-major uint8 raster that already lives on the device ->
        uint8 class maps [count,d,d]: block split + zero padding + Normalize + channel selection are ONE gather
        (``dt_split_normalize_u8``) straight into the stem's NHWC input"""
        self._model.to(raster_chw_u8.device)
        x = ops.split_normalize_u8(raster_chw_u8, d, first, count, MEAN, STD, self._channels)
        return self._model.predict_classes(x, dtype="uint8", nhwc=True)

    @staticmethod
    def is_valid_raster(raster_chw_u8: torch.Tensor) -> bool:
        """scripts/inference.py:60-62 ``is_valid_tile`` on a raster in HBM: False when band 1 holds only 0 / 255"""
        return bool(int(ops.band_has_data(raster_chw_u8[0])) != 0)


class PyTorchEnsembleInference:
    """reference deployment/inference.py:65-116: an odd number of checkpoints, per-pixel majority (torch.mode) of
    their class maps.  Here every model emits a uint8 map from the fused head kernel and one vote kernel
    (``dt_ensemble_vote``, ties -> smallest class like torch.mode) replaces stack + mode."""

    def __init__(self, *model_files: Union[str, Path]):
        self._models, self._channels, self._classes = [], None, None
        if len(model_files) % 2 == 0:
            raise ValueError("PyTorchEnsembleInference requires an uneven number of models")
        for model_file in model_files:
            model_file = Path(model_file)
            if model_file.suffix != ".ckpt":
                raise ValueError(f"Ckpt file expected, but {model_file.suffix} received")
            model = SemSegment.load_from_checkpoint(model_file)
            model.eval()
            if not self._channels:
                self._channels, self._classes = model.in_channels, model.model.spec.classes
            if model.in_channels != self._channels or model.model.spec.classes != self._classes:
                raise ValueError("Models are not compatible since they were trained for different channel configs")
            self._models.append(model.model)

    def run(self, input_tensor, device: str = "cuda"):
        if not isinstance(input_tensor, torch.Tensor):
            raise TypeError("No PyTorch tensor provided")
        if input_tensor.dim() == 3:
            input_tensor = input_tensor.unsqueeze(0)
        if self._channels == 3 and input_tensor.shape[1] == 4:
            input_tensor = input_tensor[:, 0:3, :, :]     # rgb model but rgbn data
        x = input_tensor.to(device).contiguous()
        maps = torch.stack([m.to(device).predict_classes(x, dtype="uint8") for m in self._models], dim=0)
        out, _ = ops.ensemble_vote(maps, self._classes, dtype="int64")
        return out.squeeze()


class GraphedTilePredictor:
    """uint8 NHWC tiles -> uint8 class maps with the whole chain (normalise, U-Net forward, argmax) replayed as ONE
    HIP graph per batch shape: ~120 Python launches per batch become one, which keeps the tile queue of
    scripts/inference.py:80-115 independent of host jitter (eight ranks sharing a node).  On an idle host the eager
    chain is already GPU-bound (bench.py --mode infer: same rate with and without the graph).  The returned tensor is the graph's static output: consume or copy it before the next
    call.  Weights are read from the engine's cached images, so a model whose parameters change must run one eager
    ``predict_classes`` (which repacks them) before further replays."""

    def __init__(self, model, in_channels: int = 3, precision: str = "fp32"):
        self.model, self.c, self.precision = model, in_channels, precision
        self.in_channels = in_channels
        self._graphs = {}

    def _eager(self, tiles_u8):
        x = ops.normalize_u8(tiles_u8, MEAN, STD, self.c)
        return self.model.predict_classes(x, dtype="uint8", precision=self.precision, nhwc=True)

    def __call__(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        key = (tuple(tiles_u8.shape), tiles_u8.device)
        g = self._graphs.get(key)
        if g is None:
            eng = self.model.engine
            self._eager(tiles_u8)                      # warm-up: lazy initialisation + weight images packed eagerly
            g = {"inp": tiles_u8.clone()}
            keep = {k: v for k, v in eng._ws.items() if k.startswith("bf16_w")}
            eager_ws, eng._ws = eng._ws, keep           # activations / scratch of the capture live in the graph's pool
            graph = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(graph):
                    g["out"] = self._eager(g["inp"])
            finally:
                g["ws"], eng._ws = eng._ws, eager_ws
            g["graph"] = graph
            self._graphs[key] = g
        g["inp"].copy_(tiles_u8)
        g["graph"].replay()
        return g["out"]
