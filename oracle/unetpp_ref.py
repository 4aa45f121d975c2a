"""Torch-primitive restatement of ``smp.UnetPlusPlus("resnet34")`` (reference segmodel.py:64-65, architecture "unet++").

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic lives in the un-vendored segmentation_models_pytorch
(absent here), but its dense decoder wiring is IN the reference tree: ``deadtrees/network/extra/efficientunetplusplus/
decoder.py:102-184`` is smp's ``UnetPlusPlusDecoder`` with another block type (same ``x_{depth}_{layer}`` names, channel
arithmetic :133-153 and dense forward loop :156-184).  ``oracle/make_golden_unetpp.py`` EXECUTES that class (loaded by
file path) with its block type swapped for smp's plain decoder block — nearest x2, ``cat([x, skip])``, two
``Conv2dReLU`` from the reference's ``extra/modules.py`` — and stores inputs / outputs / gradients in
``tests/golden/unetpp_decoder.npz``; ``tests/test_oracle_golden.py`` checks this restatement against them.  So the dense
wiring and channel arithmetic are PINNED by execution of the reference; the block internals are the plain decoder block
pinned in ``resunet_ref.py``'s fixture; the ResNet-34 encoder and the head (3x3 + bias, smp ``SegmentationHead``
default) stay PARITY UNPINNED like ``unet_ref.py``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .unet_ref import DECODER_CHANNELS, DecoderBlock, ResNet34Encoder


class UnetPlusPlusDecoderRef(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        self.in_channels = [enc[0]] + list(decoder_channels[:-1])
        self.skip_channels = list(enc[1:]) + [0]
        self.out_channels = list(decoder_channels)
        blocks = {}
        for layer_idx in range(len(self.in_channels) - 1):           # efficientunetplusplus/decoder.py:133-149
            for depth_idx in range(layer_idx + 1):
                if depth_idx == 0:
                    in_ch = self.in_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1)
                    out_ch = self.out_channels[layer_idx]
                else:
                    out_ch = self.skip_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1 - depth_idx)
                    in_ch = self.skip_channels[layer_idx - 1]
                blocks[f"x_{depth_idx}_{layer_idx}"] = DecoderBlock(in_ch, skip_ch, out_ch)
        blocks[f"x_0_{len(self.in_channels) - 1}"] = DecoderBlock(self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1

    def forward(self, *features):                                      # efficientunetplusplus/decoder.py:156-184
        features = features[1:][::-1]
        dense = {}
        for layer_idx in range(len(self.in_channels) - 1):
            for depth_idx in range(self.depth - layer_idx):
                if layer_idx == 0:
                    dense[f"x_{depth_idx}_{depth_idx}"] = self.blocks[f"x_{depth_idx}_{depth_idx}"](
                        features[depth_idx], features[depth_idx + 1])
                else:
                    li = depth_idx + layer_idx
                    cat = [dense[f"x_{idx}_{li}"] for idx in range(depth_idx + 1, li + 1)]
                    cat = torch.cat(cat + [features[li + 1]], dim=1)
                    dense[f"x_{depth_idx}_{li}"] = self.blocks[f"x_{depth_idx}_{li}"](dense[f"x_{depth_idx}_{li - 1}"], cat)
        dense[f"x_0_{self.depth}"] = self.blocks[f"x_0_{self.depth}"](dense[f"x_0_{self.depth - 1}"])
        return dense[f"x_0_{self.depth}"]


class UNetPPR34Ref(nn.Module):
    """``smp.UnetPlusPlus("resnet34", encoder_weights=None, in_channels=C, classes=K)`` from torch primitives"""

    def __init__(self, in_channels: int = 3, classes: int = 2):
        super().__init__()
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetPlusPlusDecoderRef(self.encoder.out_channels)
        self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1, bias=True))

    def forward(self, x):
        return self.segmentation_head(self.decoder(*self.encoder(x)))


def make_unetpp_oracle(in_channels: int = 3, classes: int = 2, seed: int = 0) -> UNetPPR34Ref:
    """deterministic weights incl. perturbed BatchNorm parameters / running statistics (like unet_ref.make_oracle)"""
    g = torch.Generator().manual_seed(seed)
    m = UNetPPR34Ref(in_channels, classes)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                fan_in = mod.weight[0].numel()
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if mod.bias is not None:
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
            elif isinstance(mod, nn.BatchNorm2d):
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(1.0 + 0.2 * torch.rand(mod.running_var.shape, generator=g))
    return m
