"""Attribute-style config containers standing in for omegaconf.DictConfig (absent in this image).

``SemSegment(network, training)`` in the reference receives two DictConfigs (deadtrees/network/
segmodel.py:58); anything mapping-like works here: dict, omegaconf DictConfig (if installed) or AttrDict.
"""
from __future__ import annotations

import copy


class AttrDict(dict):
    """dict with attribute access, ``copy()`` and ``del cfg.key`` like a DictConfig."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def copy(self):
        return AttrDict(copy.deepcopy(dict(self)))


def to_attrdict(cfg) -> AttrDict:
    if isinstance(cfg, AttrDict):
        return cfg
    try:  # omegaconf, when available
        from omegaconf import DictConfig, OmegaConf  # type: ignore
        if isinstance(cfg, DictConfig):
            cfg = OmegaConf.to_container(cfg, resolve=True)
    except Exception:
        pass
    if isinstance(cfg, dict):
        return AttrDict({k: (to_attrdict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
    if hasattr(cfg, "__dict__"):
        return to_attrdict(vars(cfg))
    raise TypeError(f"cannot interpret {type(cfg)} as a config mapping")


def default_network(**over) -> AttrDict:
    """canonical hot-path configuration: unet + resnet34, RGB, 2 classes, GDICE+FOCAL"""
    d = AttrDict(architecture="unet", encoder_name="resnet34", encoder_depth=5, encoder_weights=None,
                 decoder_channels=[256, 128, 64, 32, 16], in_channels=3,
                 classes=["background", "deadtree"], losses=["GDICE", "FOCAL"])
    d.update(over)
    return d


def default_training(**over) -> AttrDict:
    d = AttrDict(learning_rate=3e-4, cosineannealing_tmax=10)   # reference configs/model/default.yaml:12-13
    d.update(over)
    return d
