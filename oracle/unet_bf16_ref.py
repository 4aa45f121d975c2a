"""Rounding-aware CPU oracle of the bf16 training path (BASELINE configs[2]).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for topology like ``unet_ref.py`` (same
restated smp ``Unet``/``resnet34`` graph — the module passed in IS a ``UNetR34Ref``); what this file adds is
the *numerics contract* of the mixed-precision path: the reference trains under Lightning AMP
(reference protocol.md:27, configs/trainer/default.yaml ``precision``) where torch.autocast decides the
rounding points; the HIP path (deadtrees_amd/network/unet.py ``forward_bf16_train`` / ``backward_bf16``)
fixes them explicitly, and this oracle rounds to bf16 at exactly those points while accumulating in
fp64 (or fp32), so a parity test can separate "bf16 numerics" from "a mis-scheduled layer":

forward
  * conv operands: bf16 activations x bf16(round of the fp32 master weights), wide accumulation;
    BatchNorm batch statistics from the UNROUNDED accumulators; the conv output is stored as bf16;
  * a "virtual" activation (conv1 -> conv2 inside every block, decoder block -> next decoder block) is
    ``bf16(relu(float(y_bf16) * scale + shift))`` evaluated in fp32 while staging;
  * block outputs ``bf16(relu(y2*s2+b2 + residual))`` with the residual read as bf16 (identity) or as
    ``yd*sd+bd`` from the bf16 down-sample output; max-pool on bf16 values (first maximum wins);
  * stem: bf16 image x bf16 7x7 weights when the stem output is wider than 16 pixels, otherwise fp32
    operands; head: bf16 activation x fp32 weights, fp32 logits.
backward
  * head: fp32 dlogits -> bf16 activation gradient; dW/db in wide precision;
  * BatchNorm backward: ReLU mask from the stored bf16 activation or from ``bf16(y*scale+shift) > 0``;
    sums over the bf16 gradient; ``dy = bf16(gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)))``;
  * data gradients: bf16 dy x bf16 weights, wide accumulation, ONE rounding at the store; gradient joins
    add the bf16 value already in memory to the unrounded accumulator and round once;
  * nearest-upsample backward: (a+b)+(c+d) of bf16 values, one rounding; max-pool backward joins likewise; the
    narrow decoder block without a skip (dec4, channels 16 / 32, maps >= 8 x 32) sums the UNROUNDED data-gradient
    accumulators instead and rounds once (round 3: one kernel, dt_conv2d_bf16_upsampled_dgrad);
  * Unet++ dense decoder (round 3, ``UNetPPR34Ref`` from unetpp_ref.py): node outputs stored as bf16 activations; in the
    backward a node's gradient collects its consumers' contributions in reverse forward order, bf16(prev + contribution)
    per step (2x2 sums of the block to its right, channel slices of the concatenated skips further right);
  * ResUnet decoder (round 3, ``ResUNetR34Ref`` from resunet_ref.py): the 1x1 identity_conv output stored as bf16
    without its bias, block output bf16(relu(y2*s2+b2) + (idy + bias)); backward: the two branches' data gradients
    stored as bf16, added in fp32, one rounding, then the 2x2 sums; identity_conv dW / db in wide precision;
  * weight gradients: bf16 operands (the staged, rounded activation), wide accumulation, fp32 result.

Two ways to use it (tests/test_bf16_gpu.py):

* free-running (``forced=None``): a complete bf16 training step on the CPU.  bf16 rounding is discontinuous,
  so ANY difference in accumulation order (fp32 MFMA chains vs. fp64 here) flips a few roundings, the flips
  change downstream inputs by a bf16 ulp, which flips more — within ~3 layers two evaluations are as far from
  each other as independent bf16 evaluations (measured: a 1e-6 relative input perturbation moves THIS oracle's
  own logits by 7-8 % relative L2).  The end-to-end distance HIP <-> oracle is therefore bounded by the oracle's
  own sensitivity floor, not by 1e-3.
* teacher-forced (``forced={name: tensor}``): every intermediate tensor the HIP path produced is compared with
  what this oracle computes FROM THE HIP PATH'S OWN INPUTS of that unit, then replaced by the HIP tensor.
  Errors cannot cascade, so each unit must agree to accumulation accuracy (a fraction of a percent of elements
  off by one bf16 ulp): a wrong scale/shift pairing, a dropped residual or a missed gradient join shows up as an
  O(1) error in exactly the unit that has it.

Follows deadtrees_amd/network/unet.py ``forward_bf16_train`` / ``backward_bf16`` step by step.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

from .unet_ref import UNetR34Ref

BN_EPS = 1e-5


def rbf(x: torch.Tensor) -> torch.Tensor:
    """round to bf16 the way the kernels do: from an fp32 value, round-to-nearest-even"""
    return x.float().to(torch.bfloat16).to(x.dtype)


def _aff32(y, sc, sh):
    """float(y)*scale + shift in fp32 (two roundings, no fma: the library is built with -ffp-contract=off)"""
    return y.float() * sc.float()[None, :, None, None] + sh.float()[None, :, None, None]


class Bf16TrainOracle:
    def __init__(self, ref: UNetR34Ref, dtype=torch.float64, forced: Optional[Dict[str, torch.Tensor]] = None,
                 update_running: bool = True, record: bool = False):
        self.ref, self.dt = ref, dtype
        self.forced = forced
        self.rec: Optional[Dict[str, torch.Tensor]] = {} if record else None   # every named tensor of a free run
        self.update_running = update_running
        self.errs: Dict[str, Tuple[float, float]] = {}   # name -> (relative L2 error, max |err| / max |ref|)
        self.unforced = []
        self.grads: Dict[str, torch.Tensor] = {}
        self.saved = None

    # ------------------------------------------------------------------ teacher forcing
    def _t(self, name, val):
        """record/compare the tensor `name`; with teacher forcing return the HIP path's tensor instead"""
        if self.rec is not None:
            self.rec[name] = val.detach().float().clone()
        if self.forced is None:
            return val
        if name not in self.forced:
            self.unforced.append(name)
            return val
        f = self.forced[name].to(val.dtype)
        assert tuple(f.shape) == tuple(val.shape), (name, tuple(f.shape), tuple(val.shape))
        d = (f.double() - val.double())
        den = float(val.double().norm())
        self.errs[name] = (float(d.norm()) / (den + 1e-30), float(d.abs().max()) / (float(val.abs().max()) + 1e-30))
        return f

    def _tss(self, name, ss):
        if self.forced is None and self.rec is None:
            return ss
        return tuple(self._t(f"{name}.{k}", v) for k, v in zip(("scale", "shift", "mean", "invstd"), ss))

    # ------------------------------------------------------------------ helpers
    def _w(self, conv):
        return rbf(conv.weight.detach()).to(self.dt)

    def _conv_bn(self, conv, bn, x, name):
        """x: already the bf16-valued operand.  -> y (bf16 values), (scale, shift, mean, invstd) fp32"""
        v = F.conv2d(x.to(self.dt), self._w(conv), stride=conv.stride, padding=conv.padding)
        return self._finish_bn(v, bn, name)

    def _finish_bn(self, v, bn, name):
        n = v.numel() // v.shape[1]
        v64 = v.double()
        mean = v64.mean(dim=(0, 2, 3))
        var = ((v64 * v64).mean(dim=(0, 2, 3)) - mean * mean).clamp_min(0.0)
        invstd = 1.0 / torch.sqrt(var + BN_EPS)
        mean32, is32 = mean.float(), invstd.float()
        sc = bn.weight.detach().float() * is32
        sh = bn.bias.detach().float() - mean32 * sc
        if self.update_running:
            unb = var * n / (n - 1) if n > 1 else var
            with torch.no_grad():
                bn.running_mean.mul_(0.9).add_(0.1 * mean32)
                bn.running_var.mul_(0.9).add_(0.1 * unb.float())
        y = self._t(f"{name}", rbf(v).to(self.dt))
        return y, self._tss(f"{name}.bn", (sc, sh, mean32, is32))

    def _virt(self, y, ss):
        return rbf(F.relu(_aff32(y, ss[0], ss[1]))).to(self.dt)

    # ------------------------------------------------------------------ forward
    def forward(self, img: torch.Tensor) -> torch.Tensor:
        r, dt = self.ref, self.dt
        enc, dec = r.encoder, r.decoder
        S = {}
        B, Cin, H, W = img.shape
        s2d = (W // 2) > 16 and H % 2 == 0 and W % 2 == 0
        x = img.float()
        if s2d:
            v = F.conv2d(rbf(x).to(dt), self._w(enc.conv1), stride=2, padding=3)
        else:
            v = F.conv2d(x.to(dt), enc.conv1.weight.detach().to(dt), stride=2, padding=3)
        y, ss = self._finish_bn(v, enc.bn1, "stem.y")
        f1 = self._t("stem.z", self._virt(y, ss))   # stored activation: same arithmetic as a virtual one
        S["stem"] = dict(x=x, y=y, z=f1, ss=ss, s2d=s2d)
        pool, idx = F.max_pool2d(f1.double(), 3, 2, 1, return_indices=True)
        S["pool"] = dict(idx=idx, shape=f1.shape)
        cur = self._t("pool", pool.to(dt))
        feats = [f1]
        for li in range(4):
            layer = getattr(enc, f"layer{li + 1}")
            for bi, blk in enumerate(layer):
                n = f"L{li}B{bi}"
                xin = cur
                y1, ss1 = self._conv_bn(blk.conv1, blk.bn1, xin, f"{n}.y1")
                z1 = self._virt(y1, ss1)
                y2, ss2 = self._conv_bn(blk.conv2, blk.bn2, z1, f"{n}.y2")
                f = _aff32(y2, ss2[0], ss2[1])
                if blk.downsample is not None:
                    yd, ssd = self._conv_bn(blk.downsample[0], blk.downsample[1], xin, f"{n}.yd")
                    f = f + _aff32(yd, ssd[0], ssd[1])
                else:
                    yd, ssd = None, None
                    f = f + xin.float()
                out = self._t(f"{n}.out", rbf(F.relu(f)).to(dt))
                S[n] = dict(x=xin, y1=y1, ss1=ss1, y2=y2, ss2=ss2, yd=yd, ssd=ssd, out=out)
                cur = out
            feats.append(cur)
        d, d_ss = feats[4], None
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        dense = isinstance(dec.blocks, torch.nn.ModuleDict)      # smp UnetPlusPlus (unetpp_ref.UNetPPR34Ref)
        if dense:
            nodes = {f"f{k}": feats[4 - k] for k in range(5)}
            for name, low, cat in self._dense_order(dec):
                blk, n = dec.blocks[name], "P" + name
                xin = F.interpolate(nodes[low].to(dt), scale_factor=2, mode="nearest")
                if cat:
                    xin = torch.cat([xin] + [nodes[c].to(dt) for c in cat], dim=1)
                y1, ss1 = self._conv_bn(blk.conv1[0], blk.conv1[1], xin, f"{n}.y1")
                z1 = self._virt(y1, ss1)
                y2, ss2 = self._conv_bn(blk.conv2[0], blk.conv2[1], z1, f"{n}.y2")
                z2 = self._t(f"{n}.z2", self._virt(y2, ss2))      # node outputs are stored activations
                S[n] = dict(xin=xin, cx=nodes[low].shape[1], y1=y1, ss1=ss1, y2=y2, ss2=ss2, z2=z2,
                            widths=[nodes[c].shape[1] for c in cat])
                nodes[name] = z2
            d = nodes[f"x_0_{dec.depth}"]
        for i, blk in enumerate(dec.blocks if not dense else ()):
            n = f"D{i}"
            xa = self._virt(d, d_ss) if d_ss is not None else d
            xin = F.interpolate(xa.to(dt), scale_factor=2, mode="nearest")
            if skips[i] is not None:
                xin = torch.cat([xin, skips[i].to(dt)], dim=1)
            y1, ss1 = self._conv_bn(blk.conv1[0], blk.conv1[1], xin, f"{n}.y1")
            z1 = self._virt(y1, ss1)
            y2, ss2 = self._conv_bn(blk.conv2[0], blk.conv2[1], z1, f"{n}.y2")
            if hasattr(blk, "identity_conv"):
                # ResUnet block (reference extra/resunet/decoder.py:40-52) as the HIP path rounds it: the 1x1 identity
                # convolution of the (up-sampled, concatenated) input is stored as bf16 WITHOUT its bias; the join is
                # bf16(relu(y2 * scale + shift) + (idy * 1 + bias)) in fp32 (dt_bn_act_bf16, relu = 2)
                idc = blk.identity_conv
                idy = rbf(F.conv2d(xin.to(dt), self._w(idc))).float()
                f = F.relu(_aff32(y2, ss2[0], ss2[1])) + (idy * 1.0 + idc.bias.detach().float()[None, :, None, None])
                out = self._t(f"{n}.out", rbf(f).to(dt))
                S[n] = dict(xin=xin, cx=xa.shape[1], y1=y1, ss1=ss1, y2=y2, ss2=ss2, z2=None)
                d, d_ss = out, None
                continue
            last = i == len(dec.blocks) - 1
            z2 = self._t(f"{n}.z2", self._virt(y2, ss2)) if last else None
            S[n] = dict(xin=xin, cx=xa.shape[1], y1=y1, ss1=ss1, y2=y2, ss2=ss2, z2=z2)
            d, d_ss = (z2, None) if last else (y2, ss2)
        head = r.segmentation_head[0]
        logits = F.conv2d(d.to(dt), head.weight.detach().to(dt), head.bias.detach().to(dt), padding=head.padding)
        logits = self._t("logits", logits.float())
        S["head"] = dict(x=d)
        self.saved = S
        return logits.float()

    # ------------------------------------------------------------------ backward units
    def _bn_bwd(self, bn, pname, g, y, ss, mask_from=None, virtual=False):
        """g: bf16-valued gradient of the activation; returns (dy bf16-valued [unforced], masked g)"""
        dt = self.dt
        if mask_from is not None:
            g = torch.where(mask_from > 0, g, torch.zeros_like(g))
        elif virtual:
            g = torch.where(self._virt(y, ss) > 0, g, torch.zeros_like(g))
        n = g.numel() // g.shape[1]
        xh32 = (y.float() - ss[2][None, :, None, None]) * ss[3][None, :, None, None]    # fp32 like the kernel
        gd = g.double()
        sg = gd.sum(dim=(0, 2, 3))
        sx = (gd * xh32.double()).sum(dim=(0, 2, 3))
        self.grads[f"{pname}.weight"] = sx.float()
        self.grads[f"{pname}.bias"] = sg.float()
        dgam, dbet = sx.float(), sg.float()
        if self.forced is not None and f"grad:{pname}.weight" in self.forced:
            # the kernel normalises with ITS sums (fp32 accumulation): use them so that only this pass is compared
            dgam, dbet = self.forced[f"grad:{pname}.weight"].float(), self.forced[f"grad:{pname}.bias"].float()
        gi = (bn.weight.detach().float() * ss[3])[None, :, None, None]
        kb = (dbet * float(torch.tensor(1.0 / n, dtype=torch.float32)))[None, :, None, None]
        kg = (dgam * float(torch.tensor(1.0 / n, dtype=torch.float32)))[None, :, None, None]
        dy = rbf(gi * (g.float() - kb - xh32 * kg)).to(dt)
        return dy, g

    def _wgrad(self, conv, name, x, dy):
        self.grads[name] = conv2d_weight(x.to(self.dt), conv.weight.shape, dy.to(self.dt), stride=conv.stride,
                                         padding=conv.padding).float()

    def _dgrad(self, conv, in_shape, dy):
        return conv2d_input(in_shape, self._w(conv), dy.to(self.dt), stride=conv.stride, padding=conv.padding)

    @staticmethod
    def _dense_order(dec):
        """(node, lower node, names on its level) in forward order — the loops of smp UnetPlusPlusDecoder.forward (in-tree
        twin: reference network/extra/efficientunetplusplus/decoder.py:156-184); f0 = deepest encoder feature"""
        order, depth = [], dec.depth
        for layer_idx in range(depth):
            for depth_idx in range(depth - layer_idx):
                if layer_idx == 0:
                    order.append((f"x_{depth_idx}_{depth_idx}", f"f{depth_idx}", [f"f{depth_idx + 1}"]))
                else:
                    li = depth_idx + layer_idx
                    cat = [f"x_{idx}_{li}" for idx in range(depth_idx + 1, li + 1)] + [f"f{li + 1}"]
                    order.append((f"x_{depth_idx}_{li}", f"x_{depth_idx}_{li - 1}", cat))
        order.append((f"x_0_{depth}", f"x_0_{depth - 1}", []))
        return order

    def _dense_backward(self, dec, S, g_head, skip_grads):
        """reverse of the dense decoder with the HIP path's accumulation order (backward_bf16): blocks in reverse forward
        order; a node's gradient collects, one rounding per contribution, the 2x2 sums of the block to its right and the
        slices of the concatenated skips further right.  Fills skip_grads, returns the gradient of f0."""
        dt = self.dt
        G = {f"x_0_{dec.depth}": g_head}
        for name, low, cat in reversed(self._dense_order(dec)):
            blk, d, p, n = dec.blocks[name], S["P" + name], f"decoder.blocks.{name}", "P" + name
            g = self._t(f"{n}.g", G.pop(name))
            dy2, _ = self._bn_bwd(blk.conv2[1], f"{p}.conv2.1", g, d["y2"], d["ss2"], virtual=True)
            dy2 = self._t(f"{n}.dy2", dy2)
            z1 = self._virt(d["y1"], d["ss1"])
            self._wgrad(blk.conv2[0], f"{p}.conv2.0.weight", z1, dy2)
            dz1 = self._t(f"{n}.dz1", rbf(self._dgrad(blk.conv2[0], z1.shape, dy2)).to(dt))
            dy1, _ = self._bn_bwd(blk.conv1[1], f"{p}.conv1.1", dz1, d["y1"], d["ss1"], virtual=True)
            dy1 = self._t(f"{n}.dy1", dy1)
            self._wgrad(blk.conv1[0], f"{p}.conv1.0.weight", d["xin"], dy1)
            dxin = rbf(self._dgrad(blk.conv1[0], d["xin"].shape, dy1)).to(dt)
            cx = d["cx"]
            dskip = self._t(f"{n}.dskip", dxin[:, cx:].contiguous()) if cat else None
            dup = self._t(f"{n}.dup", dxin[:, :cx].contiguous()).float()
            quad = (dup[:, :, 0::2, 0::2] + dup[:, :, 0::2, 1::2]) + (dup[:, :, 1::2, 0::2] + dup[:, :, 1::2, 1::2])
            G[low] = rbf(quad).to(dt) if low not in G else rbf(G[low].float() + quad).to(dt)
            off = 0
            for cname, Cn in zip(cat, d["widths"]):
                part = dskip[:, off:off + Cn]
                G[cname] = part.contiguous() if cname not in G else rbf(G[cname].float() + part.float()).to(dt)
                off += Cn
        for k in range(1, 5):
            skip_grads[4 - k] = self._t(f"Pf{k}.g", G[f"f{k}"])
        return self._t("Pf0.g", G["f0"])

    def _resunet_block_backward(self, blk, d, p, n, g, skip_grads, slot):
        """reverse of one ResUnet decoder block with the HIP path's rounding points (backward_bf16): both branches' data
        gradients are stored as bf16, added in fp32 and rounded once; then the 2x2 sums of the up-sampling"""
        dt = self.dt
        idc, xin, cx = blk.identity_conv, d["xin"], d["cx"]
        self._wgrad(idc, f"{p}.identity_conv.weight", xin, g)
        self.grads[f"{p}.identity_conv.bias"] = g.double().sum(dim=(0, 2, 3)).float()
        dy2, _ = self._bn_bwd(blk.conv2[1], f"{p}.conv2.1", g, d["y2"], d["ss2"], virtual=True)
        dy2 = self._t(f"{n}.dy2", dy2)
        z1 = self._virt(d["y1"], d["ss1"])
        self._wgrad(blk.conv2[0], f"{p}.conv2.0.weight", z1, dy2)
        dz1 = self._t(f"{n}.dz1", rbf(self._dgrad(blk.conv2[0], z1.shape, dy2)).to(dt))
        dy1, _ = self._bn_bwd(blk.conv1[1], f"{p}.conv1.1", dz1, d["y1"], d["ss1"], virtual=True)
        dy1 = self._t(f"{n}.dy1", dy1)
        self._wgrad(blk.conv1[0], f"{p}.conv1.0.weight", xin, dy1)
        da = rbf(self._dgrad(blk.conv1[0], xin.shape, dy1)).float()
        db = rbf(self._dgrad(idc, xin.shape, g)).float()
        if xin.shape[1] > cx:
            skip_grads[slot] = self._t(f"{n}.dskip", rbf(da[:, cx:] + db[:, cx:]).to(dt))
        dup = self._t(f"{n}.dup", rbf(da[:, :cx] + db[:, :cx]).to(dt))
        a, b_, c, e = (dup[:, :, 0::2, 0::2].float(), dup[:, :, 0::2, 1::2].float(), dup[:, :, 1::2, 0::2].float(),
                       dup[:, :, 1::2, 1::2].float())
        return self._t(f"{n}.g", rbf((a + b_) + (c + e)).to(dt))

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
        r, dt, S = self.ref, self.dt, self.saved
        enc, dec = r.encoder, r.decoder
        self.grads = {}
        head = r.segmentation_head[0]
        hx = S["head"]["x"]
        dl = dlogits.to(dt)
        self.grads["segmentation_head.0.weight"] = conv2d_weight(hx.to(dt), head.weight.shape, dl, padding=head.padding).float()
        self.grads["segmentation_head.0.bias"] = dl.sum(dim=(0, 2, 3)).float()
        g = self._t("head.g", rbf(conv2d_input(hx.shape, head.weight.detach().to(dt), dl, padding=head.padding)).to(dt))

        skip_grads = [None] * 5
        dense = isinstance(dec.blocks, torch.nn.ModuleDict)
        if dense:
            g = self._dense_backward(dec, S, g, skip_grads)
        for i in (range(4, -1, -1) if not dense else ()):
            blk, d = dec.blocks[i], S[f"D{i}"]
            p, n = f"decoder.blocks.{i}", f"D{i}"
            if hasattr(blk, "identity_conv"):
                g = self._resunet_block_backward(blk, d, p, n, g, skip_grads, 3 - i)
                continue
            dy2, _ = self._bn_bwd(blk.conv2[1], f"{p}.conv2.1", g, d["y2"], d["ss2"], mask_from=d["z2"],
                                  virtual=d["z2"] is None)
            dy2 = self._t(f"{n}.dy2", dy2)
            z1 = self._virt(d["y1"], d["ss1"])
            self._wgrad(blk.conv2[0], f"{p}.conv2.0.weight", z1, dy2)
            dz1 = self._t(f"{n}.dz1", rbf(self._dgrad(blk.conv2[0], z1.shape, dy2)).to(dt))
            dy1, _ = self._bn_bwd(blk.conv1[1], f"{p}.conv1.1", dz1, d["y1"], d["ss1"], virtual=True)
            dy1 = self._t(f"{n}.dy1", dy1)
            self._wgrad(blk.conv1[0], f"{p}.conv1.0.weight", d["xin"], dy1)
            dxin_acc = self._dgrad(blk.conv1[0], d["xin"].shape, dy1)
            cx = d["cx"]
            cv = blk.conv1[0]
            if (dxin_acc.shape[1] == cx and i >= 1 and cv.in_channels in (16, 32) and cv.out_channels in (16, 32)
                    and dy1.shape[3] >= 32 and dy1.shape[2] >= 8
                    and (self.forced is None or f"{n}.dup" not in self.forced)):   # (DT_BF16_FUSE_UPSAMPLE_BWD=0: the chain)
                # the narrow layer without a skip (dec4.conv1): the HIP path takes the 2x2 sums of the up-sampling's backward
                # on the fp32 accumulators of the data gradient (dt_conv2d_bf16_upsampled_dgrad) — ONE rounding, no
                # full-resolution gradient tensor
                u = dxin_acc.to(dt)
                g = self._t(f"{n}.g", rbf((u[:, :, 0::2, 0::2] + u[:, :, 0::2, 1::2]) +
                                          (u[:, :, 1::2, 0::2] + u[:, :, 1::2, 1::2])).to(dt))
                continue
            dxin = rbf(dxin_acc).to(dt)
            dup = self._t(f"{n}.dup", dxin[:, :cx].contiguous())
            if dxin.shape[1] > cx:
                skip_grads[3 - i] = self._t(f"{n}.dskip", dxin[:, cx:].contiguous())
            a, b_, c, e = (dup[:, :, 0::2, 0::2].float(), dup[:, :, 0::2, 1::2].float(), dup[:, :, 1::2, 0::2].float(),
                           dup[:, :, 1::2, 1::2].float())
            g = self._t(f"{n}.g", rbf((a + b_) + (c + e)).to(dt))

        for li in (3, 2, 1, 0):
            layer = getattr(enc, f"layer{li + 1}")
            for bi in range(len(layer) - 1, -1, -1):
                blk, s = layer[bi], S[f"L{li}B{bi}"]
                p, n = f"encoder.layer{li + 1}.{bi}", f"L{li}B{bi}"
                gin = None
                if bi == 0 and li > 0 and skip_grads[li] is not None:
                    gin = skip_grads[li]
                dy2, gm = self._bn_bwd(blk.bn2, f"{p}.bn2", g, s["y2"], s["ss2"], mask_from=s["out"])
                dy2 = self._t(f"{n}.dy2", dy2)
                if blk.downsample is None:
                    gin = gm if gin is None else rbf(gin.float() + gm.float()).to(dt)
                    gin = self._t(f"{n}.gres", gin)
                    dyd = None
                else:
                    gm = self._t(f"{n}.gres", gm)
                    dyd, _ = self._bn_bwd(blk.downsample[1], f"{p}.downsample.1", gm, s["yd"], s["ssd"])
                    dyd = self._t(f"{n}.dyd", dyd)
                z1 = self._virt(s["y1"], s["ss1"])
                self._wgrad(blk.conv2, f"{p}.conv2.weight", z1, dy2)
                dz1 = self._t(f"{n}.dz1", rbf(self._dgrad(blk.conv2, z1.shape, dy2)).to(dt))
                dy1, _ = self._bn_bwd(blk.bn1, f"{p}.bn1", dz1, s["y1"], s["ss1"], virtual=True)
                dy1 = self._t(f"{n}.dy1", dy1)
                self._wgrad(blk.conv1, f"{p}.conv1.weight", s["x"], dy1)
                dx = self._dgrad(blk.conv1, s["x"].shape, dy1)
                gin = rbf(dx).to(dt) if gin is None else rbf(dx.float() + gin.float()).to(dt)
                gin = self._t(f"{n}.gin1", gin)
                if dyd is not None:
                    self._wgrad(blk.downsample[0], f"{p}.downsample.0.weight", s["x"], dyd)
                    dxd = self._dgrad(blk.downsample[0], s["x"].shape, dyd)
                    gin = self._t(f"{n}.gin", rbf(dxd.float() + gin.float()).to(dt))
                g = gin

        pl, st = S["pool"], S["stem"]
        Bq, Cq, Hq, Wq = pl["shape"]
        scat = torch.zeros((Bq, Cq, Hq * Wq), dtype=torch.float32)
        scat.scatter_add_(2, pl["idx"].reshape(Bq, Cq, -1), g.float().reshape(Bq, Cq, -1))
        gf1 = self._t("gf1", rbf(skip_grads[0].float() + scat.reshape(Bq, Cq, Hq, Wq)).to(dt))
        dy, _ = self._bn_bwd(enc.bn1, "encoder.bn1", gf1, st["y"], st["ss"], mask_from=st["z"])
        dy = self._t("stem.dy", dy)
        xs = rbf(st["x"]) if st["s2d"] else st["x"]
        self.grads["encoder.conv1.weight"] = conv2d_weight(xs.to(dt), enc.conv1.weight.shape, dy.to(dt), stride=2,
                                                           padding=3).float()
        self.saved = None
        return self.grads


def bf16_train_step_oracle(ref: UNetR34Ref, img: torch.Tensor, mask: torch.Tensor, losses=("GDICE", "FOCAL"),
                           dtype=torch.float64, update_running: bool = True):
    """free-running forward + loss + backward of the bf16 path on the CPU
    -> (logits fp32, loss, {smp name: gradient})"""
    from .train_ref import loss_from_logits
    o = Bf16TrainOracle(ref, dtype, update_running=update_running)
    logits = o.forward(img)
    lg = logits.clone().requires_grad_(True)
    loss, _ = loss_from_logits(lg, mask, losses)
    loss.backward()
    grads = o.backward(lg.grad)
    return logits, float(loss.detach()), grads
