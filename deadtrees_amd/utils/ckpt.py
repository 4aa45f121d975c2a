"""Reading PyTorch-Lightning ``.ckpt`` files of the reference without pytorch_lightning / omegaconf.

The reference saves checkpoints through Lightning's ``ModelCheckpoint`` (configs/callbacks/default.yaml) and reads
them with ``SemSegment.load_from_checkpoint`` (deployment/inference.py:39).  Besides the ``state_dict`` those files
pickle the Hydra/omegaconf hyper-parameter containers, so ``torch.load(weights_only=True)`` refuses them and a
plain ``torch.load`` would import and run whatever the pickle names.

``read_checkpoint_tensors`` walks the zip archive with a *restricted* unpickler: only the handful of torch
rebuild functions and plain containers are resolved; every other global (omegaconf classes, callbacks, anything
hostile) becomes an inert placeholder that can be constructed, called and mutated without running code.  Nothing
from the file is executed.  The network configuration is then inferred from tensor shapes.
"""
from __future__ import annotations

import collections
import pickle
import zipfile
from typing import Any, Dict

import numpy as np
import torch

_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
    "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
    "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class _Inert:
    """stand-in for any global that is not on the allow-list: absorbs construction, calls and state"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __reduce__(self):  # never pickled back
        raise pickle.PicklingError("placeholder object")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, v):
        pass


class _StorageType:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    # `storage` is the flat typed tensor persistent_load returned for the storage key
    return torch.as_strided(storage, tuple(size), tuple(stride), storage_offset)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, state=None):
    # torch._utils._rebuild_parameter (3 arguments) and _rebuild_parameter_with_state (4: + the Parameter's __dict__)
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, file, archive: zipfile.ZipFile, prefix: str):
        super().__init__(file)
        self._zip, self._prefix, self._cache = archive, prefix, {}

    def find_class(self, module, name):
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name in ("_rebuild_parameter", "_rebuild_parameter_with_state"):
            return _rebuild_parameter
        if module == "torch" and name in _DTYPES:
            return _StorageType(_DTYPES[name])
        if module == "torch" and name == "Size":
            return lambda seq=(): tuple(seq)
        if module == "collections" and name == "OrderedDict":
            return collections.OrderedDict
        if module == "builtins" and name in ("dict", "list", "tuple", "set", "int", "float", "str", "bool"):
            return {"dict": dict, "list": list, "tuple": tuple, "set": set, "int": int, "float": float, "str": str,
                    "bool": bool}[name]
        return _Inert      # never import anything the file names

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel): the bytes live in <prefix>/data/<key>
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage"):
            raise pickle.UnpicklingError("unsupported persistent id")
        stype, key, numel = pid[1], str(pid[2]), int(pid[4])
        dtype = stype.dtype if isinstance(stype, _StorageType) else torch.uint8
        if key not in self._cache:
            raw = self._zip.read(f"{self._prefix}/data/{key}")
            arr = np.frombuffer(raw, dtype=np.uint8).copy()
            self._cache[key] = torch.from_numpy(arr).view(dtype)[:numel] if numel else torch.empty(0, dtype=dtype)
        return self._cache[key]


def read_checkpoint_tensors(path) -> Dict[str, Any]:
    """-> the unpickled top-level object with tensors restored and every non-allow-listed object replaced by an
    inert placeholder.  Raises ``RuntimeError`` for files that are not torch zip archives."""
    try:
        archive = zipfile.ZipFile(str(path))
    except zipfile.BadZipFile as e:
        raise RuntimeError(f"{path}: not a torch zip checkpoint (legacy formats are not read)") from e
    with archive:
        pkl = [n for n in archive.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise RuntimeError(f"{path}: expected one data.pkl, found {len(pkl)}")
        prefix = pkl[0][: -len("/data.pkl")]
        with archive.open(pkl[0]) as f:
            return _Unpickler(f, archive, prefix).load()


def lightning_state_dict(path, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """``state_dict`` of a Lightning checkpoint with the LightningModule attribute prefix removed (the reference
    keeps the smp network in ``self.model``, network/segmodel.py:85)."""
    ck = read_checkpoint_tensors(path)
    sd = ck.get("state_dict") if isinstance(ck, dict) else None
    if not isinstance(sd, dict):
        raise RuntimeError(f"{path}: no state_dict in checkpoint")
    out = {}
    for k, v in sd.items():
        if isinstance(v, torch.Tensor) and isinstance(k, str) and k.startswith(prefix):
            out[k[len(prefix):]] = v
    if not out:
        raise RuntimeError(f"{path}: state_dict has no '{prefix}*' tensors")
    return out


def infer_network_conf(sd: Dict[str, torch.Tensor]) -> Dict[str, Any]:
    """in_channels / classes / encoder of an smp ``Unet`` from its tensors (the pickled hyper-parameters are not
    trusted or needed): stem weight [64, Cin, 7, 7], head weight [K, 16, 3, 3]; resnet34 has 3/4/6/3 blocks."""
    stem, head = sd.get("encoder.conv1.weight"), sd.get("segmentation_head.0.weight")
    if stem is None or head is None:
        raise RuntimeError("checkpoint is not an smp Unet with a torchvision-ResNet encoder")
    blocks = [len({k.split(".")[2] for k in sd if k.startswith(f"encoder.layer{i}.")}) for i in (1, 2, 3, 4)]
    if blocks != [3, 4, 6, 3] or "encoder.layer1.0.conv3.weight" in sd:
        raise NotImplementedError(f"encoder with blocks {blocks}: only resnet34 has HIP kernels")
    K = int(head.shape[0])
    classes = ["background", "deadtree"] if K == 2 else ["background", "conifers", "deciduous"][:K]
    return {"architecture": "unet", "encoder_name": "resnet34", "encoder_depth": 5, "encoder_weights": None,
            "in_channels": int(stem.shape[1]), "classes": classes}
