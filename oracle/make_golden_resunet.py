"""Generate tests/golden/resunet_decoder.npz by EXECUTING the reference's in-tree ResUnet decoder (build container).

TEST INFRASTRUCTURE.  Usage (where /root/reference exists):   python -m oracle.make_golden_resunet

``deadtrees/network/extra/modules.py`` and ``extra/resunet/decoder.py`` need only torch, but importing them the
normal way runs ``deadtrees/network/__init__.py``, which pulls segmentation_models_pytorch (absent).  They are
therefore loaded BY FILE PATH, with the module name ``deadtrees.network.extra.modules`` (what decoder.py imports)
registered in ``sys.modules`` first.  Stored: the decoder's state_dict, a seeded feature pyramid, the upstream
gradient, the decoder output and every gradient (parameters and features), in train mode; and the output of the
same decoder with its 1x1 identity convolutions zeroed (= the plain U-Net decoder wiring).  Data only — no
reference source text is stored.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/deadtrees/network/extra"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resunet_decoder.npz")
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
    dec_mod = _load("deadtrees.network.extra.resunet.decoder", os.path.join(REF, "resunet", "decoder.py"))
    torch.manual_seed(0)
    dec = dec_mod.ResUnetDecoder(encoder_channels=ENC_CH, decoder_channels=DEC_CH, n_blocks=5)
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
        if i > 0:     # the decoder drops the first (full-resolution) feature, decoder.py:123
            data[f"feat{i}"] = f.detach().numpy()
            data[f"dfeat{i}"] = f.grad.numpy()
    for k, v in sd0.items():
        data[f"sd:{k}"] = v.numpy()
    for k, p in dec.named_parameters():
        data[f"grad:{k}"] = p.grad.numpy()
    for k, v in dec.state_dict().items():
        if "running" in k:
            data[f"after:{k}"] = v.numpy()
    # the same decoder without its residual branch = the plain U-Net decoder (conv -> bn -> relu twice per block)
    dec2 = dec_mod.ResUnetDecoder(encoder_channels=ENC_CH, decoder_channels=DEC_CH, n_blocks=5)
    dec2.load_state_dict(sd0)
    with torch.no_grad():
        for blk in dec2.blocks:
            blk.identity_conv.weight.zero_()
            blk.identity_conv.bias.zero_()
    dec2.train()
    data["out_plain"] = dec2(*[f.detach() for f in feats]).detach().numpy()
    dec2.eval()
    data["out_plain_eval"] = dec2(*[f.detach() for f in feats]).detach().numpy()
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
