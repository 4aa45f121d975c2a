"""Static description of smp ``Unet(resnet34)`` for the HIP engine: layer list, flat-buffer offsets and
the smp ``state_dict`` key of every tensor.  Pure python (no GPU), shared by the engine, the
state-dict converter and the CPU tests.

Topology follows SURVEY.md Appendix A (smp ``Unet`` + torchvision ``resnet34``; in-tree corroboration:
reference deadtrees/network/extra/resunet/decoder.py:40-52,93-104 and extra/modules.py:74-92).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

RESNET34_LAYERS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


def _align4(n: int) -> int:
    return (n + 3) & ~3


@dataclass
class ConvSpec:
    key: str                 # smp key of the conv weight
    bn_key: Optional[str]    # smp prefix of the BatchNorm that follows (None for the head)
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    w_off: int = 0           # offset (floats) in the flat parameter buffer, HWIO layout
    w_size: int = 0
    g_off: int = 0           # BN gamma offset in the flat parameter buffer
    b_off: int = 0           # BN beta offset (or conv bias for the head)
    bn_off: int = 0          # offset of this BN's channel block in per-channel workspaces
    index: int = 0
    layout: str = "hwio"     # flat-buffer layout of the weight: "hwio" (convolution kernels) or "ohwi" (head kernel)
    sd_k: int = 0            # kernel size in the state_dict when it differs from k (1x1 head held as a 3x3 centre tap)

    @property
    def state_k(self) -> int:
        return self.sd_k or self.k


@dataclass
class BlockSpec:
    conv1: ConvSpec
    conv2: ConvSpec
    down: Optional[ConvSpec] = None


@dataclass
class DecBlockSpec:
    conv1: ConvSpec
    conv2: ConvSpec
    in_ch: int = 0
    skip_ch: int = 0
    idc: Optional[ConvSpec] = None   # ResUnet: 1x1 identity_conv (with bias) of the block input, added to the output
    name: str = ""                   # Unet++: node name x_{depth}_{layer}
    low: str = ""                    # Unet++: the node / encoder feature that is upsampled into this block
    cat: tuple = ()                  # Unet++: the nodes / encoder features concatenated as the skip, in order


@dataclass
class UNetSpec:
    in_channels: int
    classes: int
    decoder_kind: str = "unet"     # "unet" (smp Unet decoder), "resunet" (reference network/extra/resunet/decoder.py) or
                                   # "unetplusplus" (smp UnetPlusPlus: dense wiring of extra/efficientunetplusplus/decoder.py)
    stem: ConvSpec = None
    layers: List[List[BlockSpec]] = field(default_factory=list)
    decoder: List[DecBlockSpec] = field(default_factory=list)
    head: ConvSpec = None
    convs: List[ConvSpec] = field(default_factory=list)   # every conv incl. head, forward order
    n_params: int = 0          # flat parameter buffer length (floats, incl. alignment padding)
    n_true_params: int = 0     # parameter count as torch would report it
    n_bn_channels: int = 0
    buckets: list = field(default_factory=list)  # gradient-ready order: [(name, lo, hi)] ranges of the flat buffer


def build_spec(in_channels: int = 3, classes: int = 2, decoder: str = "unet") -> UNetSpec:
    if decoder not in ("unet", "resunet", "unetplusplus"):
        raise ValueError(f"decoder {decoder!r}: 'unet', 'resunet' or 'unetplusplus'")
    s = UNetSpec(in_channels, classes, decoder)
    convs: List[ConvSpec] = []

    def conv(key, bn_key, cin, cout, k, stride, pad):
        c = ConvSpec(key, bn_key, cin, cout, k, stride, pad)
        convs.append(c)
        return c

    s.stem = conv("encoder.conv1.weight", "encoder.bn1", in_channels, 64, 7, 2, 3)
    inpl = 64
    for li, (n, planes) in enumerate(zip(RESNET34_LAYERS, RESNET34_PLANES), 1):
        blocks = []
        for b in range(n):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"encoder.layer{li}.{b}"
            c1 = conv(f"{p}.conv1.weight", f"{p}.bn1", inpl, planes, 3, stride, 1)
            c2 = conv(f"{p}.conv2.weight", f"{p}.bn2", planes, planes, 3, 1, 1)
            dn = None
            if stride != 1 or inpl != planes:
                dn = conv(f"{p}.downsample.0.weight", f"{p}.downsample.1", inpl, planes, 1, stride, 0)
            blocks.append(BlockSpec(c1, c2, dn))
            inpl = planes
        s.layers.append(blocks)
    enc = [512, 256, 128, 64, 64]
    in_ch = [enc[0]] + list(DECODER_CHANNELS[:-1])
    skip_ch = enc[1:] + [0]
    if decoder == "unetplusplus":
        # smp UnetPlusPlusDecoder: block x_{d}_{l} for l = 0..3, d = 0..l, plus x_0_4 (channel arithmetic and wiring as
        # executed in reference network/extra/efficientunetplusplus/decoder.py:133-184, which copies it); blocks in
        # FORWARD order, features f0..f4 = encoder outputs from the deepest (512 ch) to the stem (64 ch)
        depth = len(in_ch) - 1
        chans = {}
        for l in range(depth):
            for d in range(l + 1):
                if d == 0:
                    chans[(d, l)] = (in_ch[l], skip_ch[l] * (l + 1), DECODER_CHANNELS[l])
                else:
                    chans[(d, l)] = (skip_ch[l - 1], skip_ch[l] * (l + 1 - d), skip_ch[l])
        chans[(0, depth)] = (in_ch[-1], 0, DECODER_CHANNELS[-1])
        order = []
        for li in range(depth):
            for d in range(depth - li):
                if li == 0:
                    order.append(((d, d), f"f{d}", (f"f{d + 1}",)))
                else:
                    l = d + li
                    order.append(((d, l), f"x_{d}_{l - 1}", tuple(f"x_{i}_{l}" for i in range(d + 1, l + 1)) + (f"f{l + 1}",)))
        order.append(((0, depth), f"x_0_{depth - 1}", ()))
        for (d, l), low, cat in order:
            ic, sc, oc = chans[(d, l)]
            p = f"decoder.blocks.x_{d}_{l}"
            c1 = conv(f"{p}.conv1.0.weight", f"{p}.conv1.1", ic + sc, oc, 3, 1, 1)
            c2 = conv(f"{p}.conv2.0.weight", f"{p}.conv2.1", oc, oc, 3, 1, 1)
            s.decoder.append(DecBlockSpec(c1, c2, ic, sc, None, f"x_{d}_{l}", low, cat))
        in_ch, skip_ch = [], []     # the plain decoder loop below adds nothing
    for i, (ic, sc, oc) in enumerate(zip(in_ch, skip_ch, DECODER_CHANNELS)):
        p = f"decoder.blocks.{i}"
        c1 = conv(f"{p}.conv1.0.weight", f"{p}.conv1.1", ic + sc, oc, 3, 1, 1)
        c2 = conv(f"{p}.conv2.0.weight", f"{p}.conv2.1", oc, oc, 3, 1, 1)
        idc = conv(f"{p}.identity_conv.weight", None, ic + sc, oc, 1, 1, 0) if decoder == "resunet" else None
        s.decoder.append(DecBlockSpec(c1, c2, ic, sc, idc))
    # smp SegmentationHead: 3x3 for smp.Unet; the reference's ResUnet builds it with kernel_size=1
    # (network/extra/resunet/model.py:89-94) — held as the centre tap of the 3x3 head kernel's weights
    s.head = conv("segmentation_head.0.weight", None, DECODER_CHANNELS[-1], classes, 3, 1, 1)
    s.head.layout = "ohwi"
    if decoder == "resunet":
        s.head.sd_k = 1

    off = 0
    bn_off = 0
    true = 0
    for i, c in enumerate(convs):
        c.index = i
        c.w_size = c.k * c.k * c.cin * c.cout
        c.w_off = off
        off = _align4(off + c.w_size)
        true += c.state_k * c.state_k * c.cin * c.cout
        if c.bn_key is not None:
            c.g_off = off
            off = _align4(off + c.cout)
            c.b_off = off
            off = _align4(off + c.cout)
            c.bn_off = bn_off
            bn_off += c.cout
            true += 2 * c.cout
        else:  # head bias
            c.b_off = off
            off = _align4(off + c.cout)
            true += c.cout
    s.convs = convs
    s.n_params = off
    s.n_true_params = true
    s.n_bn_channels = bn_off

    # gradient buckets in the order backward produces them
    def rng(cs):
        lo = min(c.w_off for c in cs)
        hi = max(_align4(c.b_off + c.cout) for c in cs)
        return lo, hi

    dec_convs = [c for d in s.decoder for c in (d.conv1, d.conv2, d.idc) if c is not None] + [s.head]
    s.buckets.append(("head+decoder",) + rng(dec_convs))
    for li in (3, 2, 1, 0):
        cs = [c for b in s.layers[li] for c in (b.conv1, b.conv2, b.down) if c is not None]
        if li == 0:
            cs = cs + [s.stem]
        s.buckets.append((f"layer{li + 1}" + ("+stem" if li == 0 else ""),) + rng(cs))
    return s


def smp_param_shapes(spec: UNetSpec):
    """{smp_key: shape} for every tensor of the smp state_dict (params and BN buffers)."""
    out = {}
    for c in spec.convs:
        out[c.key] = (c.cout, c.cin, c.state_k, c.state_k)
        if c.bn_key is not None:
            for n in ("weight", "bias", "running_mean", "running_var"):
                out[f"{c.bn_key}.{n}"] = (c.cout,)
            out[f"{c.bn_key}.num_batches_tracked"] = ()
        else:
            out[c.key.replace(".weight", ".bias")] = (c.cout,)
    return out
