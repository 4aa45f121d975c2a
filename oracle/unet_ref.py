"""Torch-primitive restatement of ``smp.Unet(encoder_name="resnet34")``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for topology:
segmentation_models_pytorch (reference setup.py:47, ``>=0.2.1``, no lock) and
torchvision are not installed, so the topology is restated from their
published structure and corroborated in-tree by

  * reference deadtrees/network/extra/resunet/decoder.py:40-52  (nearest x2,
    ``cat([x, skip], dim=1)``, two Conv2dReLU per block),
  * reference deadtrees/network/extra/resunet/decoder.py:93-104 (channel
    arithmetic in/skip/out),
  * reference deadtrees/network/extra/modules.py:74-92 (Conv2dReLU =
    conv(bias=False) -> BatchNorm2d -> ReLU),
  * reference deadtrees/network/extra/resunet/model.py:57-103 (assembly:
    encoder -> decoder -> SegmentationHead).

Call site being replaced: reference deadtrees/network/segmodel.py:63,85
(``self.model = smp.Unet(**conf, classes=n)``) and :214/:235/:280
(``logits = self.model(img)``).

Module attribute names are chosen so that ``state_dict()`` keys equal smp's
(SURVEY.md Appendix A.2): ``encoder.conv1.weight``, ``encoder.layer2.0.
downsample.0.weight``, ``decoder.blocks.3.conv1.0.weight``,
``segmentation_head.0.bias`` ...
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

RESNET34_LAYERS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class BasicBlock(nn.Module):
    """torchvision ResNet BasicBlock (expansion 1)."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet34Encoder(nn.Module):
    """smp ``ResNetEncoder`` for resnet34, depth 5: returns 6 feature maps."""

    def __init__(self, in_channels: int = 3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, (n, planes) in enumerate(zip(RESNET34_LAYERS, RESNET34_PLANES), 1):
            blocks = []
            for b in range(n):
                stride = 2 if (b == 0 and li > 1) else 1
                blocks.append(BasicBlock(inplanes, planes, stride))
                inplanes = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.out_channels = (in_channels, 64, 64, 128, 256, 512)

    def forward(self, x):
        f0 = x
        f1 = self.relu(self.bn1(self.conv1(x)))
        f2 = self.layer1(self.maxpool(f1))
        f3 = self.layer2(f2)
        f4 = self.layer3(f3)
        f5 = self.layer4(f4)
        return [f0, f1, f2, f3, f4, f5]


def _conv2d_relu(cin: int, cout: int) -> nn.Sequential:
    # reference deadtrees/network/extra/modules.py:74-92
    return nn.Sequential(
        nn.Conv2d(cin, cout, 3, padding=1, bias=False),
        nn.BatchNorm2d(cout),
        nn.ReLU(inplace=True),
    )


class DecoderBlock(nn.Module):
    def __init__(self, in_ch: int, skip_ch: int, out_ch: int):
        super().__init__()
        self.conv1 = _conv2d_relu(in_ch + skip_ch, out_ch)
        self.conv2 = _conv2d_relu(out_ch, out_ch)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        head = enc[0]
        in_ch = [head] + list(decoder_channels[:-1])
        skip_ch = list(enc[1:]) + [0]
        self.blocks = nn.ModuleList(
            DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels)
        )

    def forward(self, *features):
        features = features[1:][::-1]
        x = features[0]
        skips = features[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UNetR34Ref(nn.Module):
    """``smp.Unet("resnet34", encoder_depth=5, decoder_channels=(256,128,64,32,16),
    encoder_weights=None, in_channels=C, classes=K)`` from torch primitives."""

    def __init__(self, in_channels: int = 3, classes: int = 2):
        super().__init__()
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetDecoder(self.encoder.out_channels)
        self.segmentation_head = nn.Sequential(
            nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1, bias=True)
        )

    def forward(self, x):
        return self.segmentation_head(self.decoder(*self.encoder(x)))


def initialize_weights(m: nn.Module) -> None:
    """reference deadtrees/network/segmodel.py:432-438 (recursive Kaiming re-init)."""
    if getattr(m, "bias", None) is not None:
        nn.init.constant_(m.bias, 0)
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.kaiming_normal_(m.weight)
    for c in m.children():
        initialize_weights(c)


def make_oracle(in_channels: int = 3, classes: int = 2, seed: int = 0,
                randomize_bn: bool = True) -> UNetR34Ref:
    """Deterministic oracle weights.  ``randomize_bn`` perturbs BN affine params and
    running stats so parity tests exercise every BN term (Kaiming init leaves
    gamma=1, beta=0 which would hide gamma/beta indexing bugs)."""
    g = torch.Generator().manual_seed(seed)
    m = UNetR34Ref(in_channels, classes)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                fan_in = mod.weight[0].numel()
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if mod.bias is not None:
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
            elif isinstance(mod, nn.BatchNorm2d) and randomize_bn:
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(1.0 + 0.2 * torch.rand(mod.running_var.shape, generator=g))
    return m
