"""Torch-primitive restatement of the reference's in-tree ResUnet decoder and model.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED: ``oracle/make_golden_resunet.py`` loads the reference's own
``deadtrees/network/extra/modules.py`` and ``extra/resunet/decoder.py`` by file path (they need only torch), runs
``ResUnetDecoder`` on seeded feature pyramids and stores inputs / outputs / gradients in
``tests/golden/resunet_decoder.npz``; ``tests/test_oracle_golden.py`` checks this restatement against them.  The
same fixture run with the 1x1 ``identity_conv`` zeroed pins the plain U-Net decoder wiring of ``unet_ref.py``
(feature order, nearest x2, ``cat([x, skip])``, channel arithmetic) by execution rather than by reading.

What the reference computes (deadtrees/network/extra/resunet/decoder.py:8-52, extra/modules.py:10-49):
``PreActivatedConv2dReLU`` is, despite its name, ``nn.Sequential(conv(bias=False), BatchNorm2d, ReLU)`` — the same
order as ``Conv2dReLU``; a decoder block is ``up x2 -> cat skip -> conv1 -> conv2`` PLUS a 1x1 ``identity_conv``
(with bias) of the concatenated input added to the result, no activation after the sum.  The model
(extra/resunet/model.py:57-103) is smp's encoder + this decoder + a ``SegmentationHead`` with ``kernel_size=1``.
The encoder is the torchvision/smp ResNet-34 of ``unet_ref.py`` (PARITY UNPINNED there, see its header).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import DECODER_CHANNELS, ResNet34Encoder, _conv2d_relu


class ResDecoderBlock(nn.Module):
    def __init__(self, in_ch: int, skip_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = _conv2d_relu(in_ch + skip_ch, out_ch)      # extra/modules.py:10-49: conv -> bn -> relu
        self.conv2 = _conv2d_relu(out_ch, out_ch)
        self.identity_conv = nn.Conv2d(in_ch + skip_ch, out_ch, kernel_size=1)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x)) + self.identity_conv(x)


class ResUnetDecoderRef(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = list(enc[1:]) + [0]
        self.blocks = nn.ModuleList(ResDecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels))

    def forward(self, *features):
        features = features[1:][::-1]
        x, skips = features[0], features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class ResUNetR34Ref(nn.Module):
    """reference ``ResUnet("resnet34", encoder_weights=None, in_channels=C, classes=K)`` from torch primitives"""

    def __init__(self, in_channels: int = 3, classes: int = 2):
        super().__init__()
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = ResUnetDecoderRef(self.encoder.out_channels)
        self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 1, bias=True))

    def forward(self, x):
        return self.segmentation_head(self.decoder(*self.encoder(x)))


def make_resunet_oracle(in_channels: int = 3, classes: int = 2, seed: int = 0) -> ResUNetR34Ref:
    """deterministic weights incl. perturbed BatchNorm parameters / running statistics (like unet_ref.make_oracle)"""
    g = torch.Generator().manual_seed(seed)
    m = ResUNetR34Ref(in_channels, classes)
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
        # He-initialised residual branches without a normalisation after the sum make the activations grow block by
        # block (logits of several thousand: a saturated softmax, an ill-conditioned loss gradient).  Parity tests
        # want O(1..10) logits: damp the 1x1 identity convolutions and the head.
        for blk in m.decoder.blocks:
            blk.identity_conv.weight.mul_(0.25)
        m.segmentation_head[0].weight.mul_(0.1)
    return m
