"""Generate tests/golden/unetpp_decoder.npz by EXECUTING the reference's in-tree dense (Unet++) decoder wiring.

TEST INFRASTRUCTURE.  Usage (where /root/reference exists):   python -m oracle.make_golden_unetpp

``smp.UnetPlusPlus`` itself is absent, but ``deadtrees/network/extra/efficientunetplusplus/decoder.py`` carries smp's
``UnetPlusPlusDecoder`` constructor and forward loop verbatim around another block type.  The file is loaded BY FILE
PATH (like make_golden_resunet.py), its module-level name ``DecoderBlock`` is rebound to smp's plain decoder block —
``interpolate(nearest, x2)`` -> ``cat([x, skip])`` -> two ``Conv2dReLU`` (the reference's own ``extra/modules.py``
class) — and ``EfficientUnetPlusPlusDecoder`` is run on a seeded feature pyramid.  Stored: state_dict, features, output,
upstream gradient, every parameter / feature gradient, BatchNorm running statistics after the step.  Data only.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/deadtrees/network/extra"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "unetpp_decoder.npz")
ENC_CH = (3, 16, 16, 32, 64, 128)
DEC_CH = (64, 32, 16, 16, 8)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    for pkg in ("deadtrees", "deadtrees.network", "deadtrees.network.extra"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    md = _load("deadtrees.network.extra.modules", os.path.join(REF, "modules.py"))
    sys.modules["deadtrees.network.extra"].modules = md
    dec_mod = _load("deadtrees.network.extra.efficientunetplusplus.decoder",
                    os.path.join(REF, "efficientunetplusplus", "decoder.py"))

    class PlainDecoderBlock(torch.nn.Module):
        """smp's DecoderBlock (attention_type=None) from the reference's own Conv2dReLU"""

        def __init__(self, in_channels, skip_channels, out_channels, **_unused):
            super().__init__()
            self.conv1 = md.Conv2dReLU(in_channels + skip_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=True)
            self.conv2 = md.Conv2dReLU(out_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=True)

        def forward(self, x, skip=None):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if skip is not None:
                x = torch.cat([x, skip], dim=1)
            return self.conv2(self.conv1(x))

    dec_mod.DecoderBlock = PlainDecoderBlock
    torch.manual_seed(0)
    dec = dec_mod.EfficientUnetPlusPlusDecoder(encoder_channels=ENC_CH, decoder_channels=DEC_CH, n_blocks=5)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for mod in dec.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    B, S = 2, 32
    feats = [torch.randn((B, c, S >> i, S >> i), generator=g).requires_grad_(True) for i, c in enumerate(ENC_CH)]
    dec.train()
    sd0 = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    out = dec(*feats)
    gout = torch.randn(out.shape, generator=g)
    (out * gout).sum().backward()
    data = {"enc_ch": np.array(ENC_CH), "dec_ch": np.array(DEC_CH), "gout": gout.numpy(), "out": out.detach().numpy()}
    for i, f in enumerate(feats):
        if i > 0:
            data[f"feat{i}"] = f.detach().numpy()
            data[f"dfeat{i}"] = f.grad.numpy()
    for k, v in sd0.items():
        data[f"sd:{k}"] = v.numpy()
    for k, p in dec.named_parameters():
        data[f"grad:{k}"] = p.grad.numpy()
    for k, v in dec.state_dict().items():
        if "running" in k:
            data[f"after:{k}"] = v.numpy()
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(sd0), "state tensors;", sorted({k.split('.')[1] for k in sd0}))


if __name__ == "__main__":
    main()
