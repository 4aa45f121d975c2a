"""MI355X-native U-Net (smp ``Unet`` + ``resnet34`` topology) — the drop-in for ``SemSegment.model``.

Replaces ``smp.Unet(**conf, classes=n)`` built at reference deadtrees/network/segmodel.py:63,85 and
called at :214/:235/:280 and deployment/inference.py:60.  Same module boundary:
``forward(x: f32[B,Cin,H,W]) -> logits f32[B,K,H,W]`` and smp-named ``state_dict``.

Inside, nothing is ATen: every convolution / BatchNorm / ReLU / pool / upsample / concat runs in the
hand-written gfx950 kernels of ``libdeadtrees_hip.so`` through its C ABI (include/deadtrees_hip.h),
NHWC, with all parameters in ONE flat fp32 buffer (HWIO conv weights) so the optimiser and the RCCL
gradient all-reduce work on contiguous ranges.  PyTorch only provides device memory, streams and the
autograd entry point (one ``autograd.Function`` for the whole network; the backward pass is an explicit
hand-scheduled chain, not an autograd graph).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from .. import _lib
from .spec import ConvSpec, UNetSpec, build_spec

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Saved:
    """activations kept from forward for the hand-written backward"""
    __slots__ = ("d",)

    def __init__(self):
        self.d = {}


class UNetEngine:
    """Executes the layer list of ``UNetSpec`` on the C ABI.  Holds no parameters itself."""

    def __init__(self, spec: UNetSpec):
        self.spec = spec
        self.lib = _lib.load()
        self._ws = {}
        self.saved: Optional[_Saved] = None
        self.grad_hook: Optional[Callable[[str, int, int], None]] = None
        # when a list: every dt_conv2d launch appends (kernel name, algorithmic FLOPs, start, end events)
        self.profile: Optional[list] = None
        self._weights_epoch = 0
        self._bn_epoch = 0            # bumped by every training-mode forward (running statistics written on the device)
        self._affine_fresh = False
        self._u_all = self._ud_all = None     # Winograd weight images of the current forward / backward pass
        # Weight gradients on a side stream, concurrent with the data-gradient / BatchNorm chain.  Round 1 (direct
        # kernels, 2 workgroups per CU): SLOWER (485 vs 528 tiles/s fp32, 1505 vs 1566 bf16) — co-resident wgrad / dgrad
        # workgroups halve each other's occupancy and share the matrix pipe.  Round 2, fp32 Winograd path: the kernels
        # own a whole CU (148 KB LDS, 512 registers per lane), nothing co-resides, and the second stream only fills
        # the launch gaps and tails of the ~200 short kernels of the backward pass: 716.6 vs 701.6 tiles/s on one box.
        # fp32: on with the Winograd kernels (DT_OVERLAP_WGRAD=0/1 overrides); bf16: off.
        self.overlap_wgrad = os.environ.get("DT_OVERLAP_WGRAD", "1" if os.environ.get("DT_FP32_WINOGRAD", "1") != "0" else "0") != "0"
        self.overlap_wgrad_bf16 = bool(os.environ.get("DT_OVERLAP_WGRAD_BF16"))
        self._bwd_training = True
        # BatchNorm-backward reduction of a block-output layer inside the fp32 gradient-JOIN epilogue: measured slower
        # than the separate pass (526 vs 530 tiles/s, same box: 48 extra loads per lane in the read-modify-write
        # epilogue); the bf16 path keeps it (its join epilogue is LDS-staged, +0.5 %).  Kernel support stays tested.
        # Round 3, Winograd engine: the join epilogue form 3 of conv3x3_wino_kernel carries the sums at +0.7 % of the step
        # (763.9 vs 758.8 tiles/s, same box) and removes 13 of the 23 remaining bn_bwd_reduce passes: on with Winograd.
        self.fuse_join_fp32 = os.environ.get("DT_FUSE_JOIN_FP32", "1" if os.environ.get("DT_FP32_WINOGRAD", "1") != "0" else "0") != "0"
        self._fuse_bn = not os.environ.get("DT_NO_BN_FUSE")   # A/B switch for the plain fused reductions
        # fp32 3x3 stride-1 layers (forward + data gradient) on the Winograd F(2x2,3x3) kernel where its shape conditions
        # hold (conv_wino.hip: 1.6-2.0x the direct kernel per layer); DT_FP32_WINOGRAD=0 keeps the exact-fma direct kernel
        self.winograd = os.environ.get("DT_FP32_WINOGRAD", "1") != "0"
        # conv1 activations of the blocks whose conv2 runs on the Winograd kernels are materialised (bn_act) instead of
        # being applied while conv2 / its weight gradient stage their input: the fused form costs those kernels 11-13 %
        # (one wave per SIMD: the staging instructions are not free behind the MFMAs), the extra pass 0.4 ms — measured 713 vs 704
        # tiles/s; DT_MATERIALIZE_Z1=0 restores the fused form (a gain with the direct kernels: +2 % in round 1)
        # inference (eval mode, nothing saved): BatchNorm + ReLU (+ residual) in the Winograd epilogue, DT_FUSE_EVAL=0 = A/B
        self._fuse_eval_opt = os.environ.get("DT_FUSE_EVAL", "1") != "0"
        self._bf16_images_fused = os.environ.get("DT_BF16_IMAGES_FUSED", "1") != "0"   # four bf16 weight images in one launch
        self._fuse_pool_bn = os.environ.get("DT_FUSE_POOL_BN", "1") != "0"   # stem BatchNorm-backward sums in the max-pool backward
        self._fuse_eval = False
        self._mat_z1 = os.environ.get("DT_MATERIALIZE_Z1", "1" if self.winograd else "0") != "0"
        # the same for the decoder block outputs that feed a Winograd conv1 (716.6 vs 712.8 tiles/s)
        self._mat_z2 = os.environ.get("DT_MATERIALIZE_Z2", "1" if self.winograd else "0") != "0"
        # bf16: the input-transforming form of the LDS-DMA kernel stages its input through registers (no DMA); a stored
        # bf16 activation (2 + 2 B per element) lets conv2 and its weight gradient run the pure-DMA form: 2,235 vs 2,203
        self._mat_z1_bf16 = os.environ.get("DT_BF16_MAT_Z1", "1") != "0"
        self._mat_dec_bf16 = os.environ.get("DT_BF16_MAT_DEC", "1") != "0"
        # when a dict: the bf16 training pass stores a copy of every intermediate tensor it produces under the
        # names of oracle/unet_bf16_ref.py (teacher-forced parity test); None in production
        self.trace: Optional[dict] = None

    # ------------------------------------------------------------------ helpers
    def _tr(self, name: str, t: Optional[torch.Tensor]):
        if self.trace is not None and t is not None:
            self.trace[name] = t.clone()

    def _buf(self, name: str, numel: int, dtype=torch.float32, device=None) -> torch.Tensor:
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.device != device:
            t = torch.empty(max(numel, 1), dtype=dtype, device=device)
            self._ws[name] = t
        return t

    # ---- per-launch profiling (bench.py's roofline table): `self.profile` is None in production; as a list it receives
    # (kernel / family name, algorithmic FLOPs, start event, end event, algorithmic HBM bytes) per bracketed launch group
    def _pb(self):
        if self.profile is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def _pe(self, e0, name: str, flops: float, nbytes: float):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.profile.append((name, float(flops), e0, e1, float(nbytes)))

    @staticmethod
    def _conv_work(desc, elt: int = 4):
        """(algorithmic FLOPs, algorithmic HBM bytes) of a convolution / its weight gradient described by `desc`:
        2 k^2 Cin Cout per output pixel; stored input(s) once (an upsampled source at its stored size) + output once
        (+ the read of a read-modify-write join) + weights once"""
        flops = 2.0 * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout * desc.Ho * desc.Wo * desc.B
        if desc.mode0 == 2:
            flops /= 4.0   # transposed conv: 3/4 of the zero-inserted input does no algorithmic work
        sdiv = 4 if desc.mode0 else 1
        nbytes = elt * desc.B * (desc.Hin * desc.Win * (desc.C0 / sdiv + desc.C1) +
                                 desc.Ho * desc.Wo * desc.Cout * (2 if desc.accumulate else 1)) + \
            elt * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout
        return flops, nbytes

    def _desc(self, B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, k, stride, pad, split=0, acc=0):
        return _lib.ConvDesc(B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, k, stride, pad, split, acc)

    def _conv_kernel_name(self, desc, transformed: bool = False) -> str:
        """name of the kernel instantiation dt_conv2d launches, spelled like rocprofv3 prints it"""
        tw, tn, ck = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.dt_conv2d_config(C.byref(desc), C.byref(tw), C.byref(tn), C.byref(ck)), "dt_conv2d_config")
        if ck.value >= 2000:     # its sub-pixel form for the up-sampled input: ck = 2000 + 10 CB + NB
            cb, nbk = (ck.value - 2000) // 10, (ck.value - 2000) % 10
            return f"conv3x3_f32_upc_kernel<{cb}, {nbk}, {'true' if transformed else 'false'}>"
        if ck.value >= 1000:     # the lean narrow-layer kernel (conv_narrow.hip): ck = 1000 + 10 CB + NB
            cb, nbk = (ck.value - 1000) // 10, (ck.value - 1000) % 10
            return f"conv3x3_f32_narrow_kernel<{cb}, {nbk}, {'true' if transformed else 'false'}, false>"
        if tn.value == 16:
            return "conv_fwd_n16_kernel"
        zi = "true" if self.lib.dt_conv2d_uses_zi(C.byref(desc)) else "false"
        tf = "true" if transformed else "false"
        return f"conv_fwd_kernel<{desc.ksize}, {desc.stride}, {tw.value}, {tn.value}, {ck.value}, {zi}, {tf}>"

    def _use_wino(self, desc, u) -> bool:
        return u is not None and bool(self.lib.dt_conv2d_winograd_supported(C.byref(desc)))

    def _stat_rows(self, desc, u=None) -> int:
        """rows of the BatchNorm partial-statistics buffer the convolution launch for `desc` writes"""
        P = (self.lib.dt_conv2d_winograd_stat_rows if self._use_wino(desc, u) else self.lib.dt_conv2d_stat_rows)(C.byref(desc))
        if P <= 0:
            raise RuntimeError(f"dt_conv2d_stat_rows: {self.lib.dt_last_error().decode()}")
        return P

    def _conv(self, desc, src0, src1, w, out0, out1=None, stats=None, in_ss=None, u=None):
        """u: the layer's Winograd weight image (or None): used when the kernel supports the descriptor"""
        prof = self.profile
        wino = self._use_wino(desc, u)
        if prof is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if wino:
            _lib.check(self.lib.dt_conv2d_winograd(C.byref(desc), _p(src0), _p(src1), _p(u), _p(out0), _p(out1),
                                                   _p(stats), _p(in_ss[0]) if in_ss else None,
                                                   _p(in_ss[1]) if in_ss else None, _stream()), "dt_conv2d_winograd")
        else:
            _lib.check(self.lib.dt_conv2d(C.byref(desc), _p(src0), _p(src1), _p(w), _p(out0), _p(out1), _p(stats),
                                          _p(in_ss[0]) if in_ss else None, _p(in_ss[1]) if in_ss else None,
                                          _stream()), "dt_conv2d")
        if prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            flops = 2.0 * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout * desc.Ho * desc.Wo * desc.B
            if desc.mode0 == 2:
                flops /= 4.0   # transposed conv: 3/4 of the zero-inserted input does no algorithmic work
            # algorithmic HBM bytes: stored input(s) once + output once (+ read-modify-write) + weights once
            sdiv = 4 if desc.mode0 else 1
            nbytes = 4.0 * desc.B * (desc.Hin * desc.Win * (desc.C0 / sdiv + desc.C1) +
                                     desc.Ho * desc.Wo * desc.Cout * (2 if desc.accumulate else 1)) + \
                4.0 * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout
            name = self._wino_kernel_name(in_ss is not None, 2 if desc.accumulate else 0) if wino else \
                self._conv_kernel_name(desc, in_ss is not None)
            prof.append((name, flops, e0, e1, nbytes))

    @staticmethod
    def _wino_kernel_name(transformed: bool, epi: int) -> str:
        return f"conv3x3_wino_kernel<{'true' if transformed else 'false'}, {epi}>"

    # ------------------------------------------------------------------ Winograd weight images (conv_wino.hip)
    def _wino_table(self, device, dgrad: bool):
        """(device table, rows, blocks, total floats, {conv key: offset}) of the layers whose forward conv (dgrad False)
        or stride-1 data gradient (dgrad True: Cin / Cout swapped, read from the dt_weight_images mode-0 buffer) can run
        on the Winograd kernel: 3x3 stride 1 pad 1, input channels a multiple of 16, output channels a multiple of 64"""
        key = ("wino", bool(dgrad), str(device))
        if not hasattr(self, "_tables"):
            self._tables = {}
        tab = self._tables.get(key)
        if tab is None:
            rows, blocks, off, offs = [], 0, 0, {}
            for c in self.spec.convs:
                if c is self.spec.stem or c is self.spec.head or c.k != 3 or c.stride != 1 or c.pad != 1:
                    continue
                cin, cout = (c.cout, c.cin) if dgrad else (c.cin, c.cout)
                if cin % 16 or cout % 64:
                    continue
                rows.append([c.w_off, off, cin, cout, blocks])
                offs[c.key] = (off, 16 * cin * cout)
                blocks += ((cout + 63) // 64) * ((cin // 4 + 3) // 4)
                off += 16 * cin * cout
            tab = (torch.tensor(rows, dtype=torch.int32).to(device) if rows else None, len(rows), blocks, off, offs)
            self._tables[key] = tab
        return tab

    def _wino_images(self, weights: torch.Tensor, name: str, dgrad: bool):
        tab, n, blocks, total, _ = self._wino_table(weights.device, dgrad)
        if n == 0:
            return None
        buf = self._buf(name, total, device=weights.device)
        _lib.check(self.lib.dt_winograd_weight_images(_p(weights), _p(buf), _p(tab), n, blocks, _stream()),
                   "dt_winograd_weight_images")
        return buf

    def _wino_fwd_weights(self, params: torch.Tensor):
        """forward images of every eligible layer, rebuilt when the flat parameter buffer changed (one launch)"""
        if not self.winograd:
            return None
        key = (params.data_ptr(), params._version, self._weights_epoch)
        if self._ws.get("wino_u_key") != key:
            self._ws["wino_u_val"] = self._wino_images(params, "wino_u", False)
            self._ws["wino_u_key"] = key
        return self._ws["wino_u_val"]

    def _u(self, c: ConvSpec, dgrad: bool = False):
        """the Winograd image of conv c (forward / data gradient) or None"""
        buf = self._ud_all if dgrad else self._u_all
        if buf is None:
            return None
        ent = self._wino_table(buf.device, dgrad)[4].get(c.key)
        return None if ent is None else buf[ent[0]:ent[0] + ent[1]]

    # ------------------------------------------------------------------ forward units
    def _ss(self, c: ConvSpec, bnws):
        """(scale, shift) slices of conv c's BatchNorm in the per-forward workspace"""
        nb = self.spec.n_bn_channels
        return (bnws[2 * nb + c.bn_off: 2 * nb + c.bn_off + c.cout], bnws[3 * nb + c.bn_off: 3 * nb + c.bn_off + c.cout])

    def _conv_bn(self, c: ConvSpec, params, bnstate, bnws, src0, src1, mode0, B, Hin, Win, training, in_ss=None,
                 save_stats=False):
        """y = conv(x); BN statistics -> per-channel scale/shift in bnws.  Returns y, (Ho, Wo).
        in_ss: (scale, shift) of the layer that produced src0 when src0 is a RAW conv output whose
        BatchNorm-apply + ReLU is fused into this conv's LDS staging (virtual activation)."""
        dev = src0.device
        Ho = (Hin + 2 * c.pad - c.k) // c.stride + 1
        Wo = (Win + 2 * c.pad - c.k) // c.stride + 1
        C0 = src0.shape[-1]
        C1 = 0 if src1 is None else src1.shape[-1]
        assert C0 + C1 == c.cin, (c.key, C0, C1, c.cin)
        desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
        y = torch.empty((B, Ho, Wo, c.cout), dtype=torch.float32, device=dev)
        w = params[c.w_off:c.w_off + c.w_size]
        nb = self.spec.n_bn_channels
        mean = bnws[0 * nb + c.bn_off: 0 * nb + c.bn_off + c.cout]
        invstd = bnws[1 * nb + c.bn_off: 1 * nb + c.bn_off + c.cout]
        scale = bnws[2 * nb + c.bn_off: 2 * nb + c.bn_off + c.cout]
        shift = bnws[3 * nb + c.bn_off: 3 * nb + c.bn_off + c.cout]
        gamma = params[c.g_off:c.g_off + c.cout]
        beta = params[c.b_off:c.b_off + c.cout]
        rmean = bnstate[2 * c.bn_off: 2 * c.bn_off + c.cout]
        rvar = bnstate[2 * c.bn_off + c.cout: 2 * c.bn_off + 2 * c.cout]
        u = self._u(c)
        if training:
            P = self._stat_rows(desc, u)
            stats = self._buf("bn_stats", self.lib.dt_bn_stats_floats(P, c.cout), device=dev)
            self._conv(desc, src0, src1, w, y, None, stats, in_ss, u=u)
            _lib.check(self.lib.dt_bn_finalize(_p(stats), P, c.cout, float(B * Ho * Wo), _p(gamma), _p(beta),
                                               BN_EPS, BN_MOMENTUM, _p(rmean), _p(rvar), _p(mean), _p(invstd),
                                               _p(scale), _p(shift), _stream()), "dt_bn_finalize")
        else:
            self._conv(desc, src0, src1, w, y, None, None, in_ss, u=u)
            if not self._affine_fresh:   # inference: the coefficients of the previous call are still valid (see forward)
                _lib.check(self.lib.dt_bn_eval_affine(_p(gamma), _p(beta), _p(rmean), _p(rvar), BN_EPS, c.cout,
                                                      _p(scale), _p(shift), _stream()), "dt_bn_eval_affine")
            if save_stats:   # a backward pass may follow (frozen-BatchNorm fine-tuning): xhat uses the running stats
                _lib.check(self.lib.dt_bn_eval_stats(_p(rmean), _p(rvar), BN_EPS, c.cout, _p(mean), _p(invstd),
                                                     _stream()), "dt_bn_eval_stats")
        return y, Ho, Wo, (scale, shift)

    def _bn_act(self, y, ss, res=None, res_ss=None, relu=True, out=None):
        """relu: True/1 = ReLU after the residual add, 2 = ReLU on the main branch only (ResUnet decoder), 0 = none"""
        B, H, W, Cc = y.shape
        z = torch.empty_like(y) if out is None else out
        e0 = self._pb()
        _lib.check(self.lib.dt_bn_act(_p(y), _p(ss[0]), _p(ss[1]), _p(res),
                                      _p(res_ss[0]) if res_ss else None, _p(res_ss[1]) if res_ss else None,
                                      _p(z), B * H * W, Cc, int(relu), _stream()), "dt_bn_act")
        self._pe(e0, "bn_act_kernel", 0.0, 4.0 * y.numel() * (2 + (res is not None)))
        return z

    def _conv_affine_direct(self, c: ConvSpec, params, bnstate, bnws, src0, src1, mode0, B, Hin, Win, relu=True):
        """inference: [relu](bn_eval(conv(x))) in one launch of the direct kernel (dt_conv2d_affine) — the layers that are
        neither Winograd nor narrow layers: stem, stride-2 3x3, 1x1 down-sample.  Returns (activation, Ho, Wo)."""
        Ho = (Hin + 2 * c.pad - c.k) // c.stride + 1
        Wo = (Win + 2 * c.pad - c.k) // c.stride + 1
        C0 = src0.shape[-1]
        C1 = 0 if src1 is None else src1.shape[-1]
        desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
        scale, shift = self._ss(c, bnws)
        if not self._affine_fresh:
            _lib.check(self.lib.dt_bn_eval_affine(_p(params[c.g_off:c.g_off + c.cout]), _p(params[c.b_off:c.b_off + c.cout]),
                                                  _p(bnstate[2 * c.bn_off: 2 * c.bn_off + c.cout]),
                                                  _p(bnstate[2 * c.bn_off + c.cout: 2 * c.bn_off + 2 * c.cout]), BN_EPS,
                                                  c.cout, _p(scale), _p(shift), _stream()), "dt_bn_eval_affine")
        z = torch.empty((B, Ho, Wo, c.cout), dtype=torch.float32, device=src0.device)
        e0 = self._pb()
        _lib.check(self.lib.dt_conv2d_affine(C.byref(desc), _p(src0), _p(src1), _p(params[c.w_off:c.w_off + c.w_size]), _p(z),
                                             _p(scale), _p(shift), 1 if relu else 0, _stream()), "dt_conv2d_affine")
        if e0 is not None:
            fl, nb = self._conv_work(desc)
            self._pe(e0, self._conv_kernel_name(desc, False), fl, nb)
        return z, Ho, Wo

    def _conv_affine_eval(self, c: ConvSpec, params, bnstate, bnws, src0, src1, mode0, B, Hin, Win, res=None, in_ss=None):
        """inference: relu(bn_eval(conv(x)) [+ res]) in ONE Winograd launch (dt_conv2d_winograd_affine) — no raw output, no
        bn_act pass.  Returns the activation, or None when the layer is not a Winograd layer (caller: conv + bn_act)."""
        if not self._fuse_eval or c.k != 3 or c.stride != 1 or c.pad != 1:
            return None
        C0 = src0.shape[-1]
        C1 = 0 if src1 is None else src1.shape[-1]
        desc = self._desc(B, Hin, Win, C0, C1, mode0, Hin, Win, c.cout, 3, 1, 1)
        u = self._u(c)
        narrow = res is None and not self._use_wino(desc, u) and bool(self.lib.dt_conv2d_narrow_supported(C.byref(desc)))
        if not narrow and (in_ss is not None or not self._use_wino(desc, u)):
            return None
        scale, shift = self._ss(c, bnws)
        if not self._affine_fresh:
            _lib.check(self.lib.dt_bn_eval_affine(_p(params[c.g_off:c.g_off + c.cout]), _p(params[c.b_off:c.b_off + c.cout]),
                                                  _p(bnstate[2 * c.bn_off: 2 * c.bn_off + c.cout]),
                                                  _p(bnstate[2 * c.bn_off + c.cout: 2 * c.bn_off + 2 * c.cout]), BN_EPS,
                                                  c.cout, _p(scale), _p(shift), _stream()), "dt_bn_eval_affine")
        z = torch.empty((B, Hin, Win, c.cout), dtype=torch.float32, device=src0.device)
        e0 = self._pb()
        if narrow:     # the narrow decoder layers (dec3.conv2, dec4): the lean kernel's inference epilogue
            _lib.check(self.lib.dt_conv2d_narrow_affine(C.byref(desc), _p(src0), _p(params[c.w_off:c.w_off + c.w_size]), _p(z),
                                                        _p(scale), _p(shift), _p(in_ss[0]) if in_ss else None,
                                                        _p(in_ss[1]) if in_ss else None, _stream()), "dt_conv2d_narrow_affine")
            if e0 is not None:
                fl, nb = self._conv_work(desc)
                self._pe(e0, f"conv3x3_f32_narrow_kernel<{C0 // 16}, {c.cout // 16}, {'true' if in_ss else 'false'}, 4>", fl, nb)
            return z
        _lib.check(self.lib.dt_conv2d_winograd_affine(C.byref(desc), _p(src0), _p(src1), _p(u), _p(z), _p(scale),
                                                      _p(shift), _p(res), _stream()), "dt_conv2d_winograd_affine")
        if e0 is not None:
            fl, nb = self._conv_work(desc)
            self._pe(e0, self._wino_kernel_name(False, 5 if res is not None else 4), fl, nb + (4.0 * z.numel() if res is not None else 0.0))
        return z

    def _const_vec(self, value: float, n: int, device) -> torch.Tensor:
        key = f"const_{value}"
        t = self._ws.get(key)
        if t is None or t.numel() < n or t.device != device:
            t = torch.full((max(n, 512),), float(value), dtype=torch.float32, device=device)
            self._ws[key] = t
        return t[:n]

    # ------------------------------------------------------------------ forward
    def forward(self, x_nchw: torch.Tensor, params: torch.Tensor, bnstate: torch.Tensor, training: bool,
                save: bool, want_argmax: Optional[str] = None, nhwc: bool = False):
        """nhwc=True: the input already is the kernels' layout [B,H,W,C] (the tiled-inference gather produces it):
        no NCHW -> NHWC pass"""
        sp = self.spec
        if nhwc:
            if x_nchw.dim() != 4 or x_nchw.shape[3] != sp.in_channels:
                raise RuntimeError(f"expected NHWC input [B,H,W,{sp.in_channels}], got {tuple(x_nchw.shape)}")
            B, H, W, Cin = x_nchw.shape
        else:
            if x_nchw.dim() != 4 or x_nchw.shape[1] != sp.in_channels:
                raise RuntimeError(f"expected input [B,{sp.in_channels},H,W], got {tuple(x_nchw.shape)}")
            B, Cin, H, W = x_nchw.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"H and W must be divisible by 32 (encoder depth 5), got {H}x{W}")
        if x_nchw.dtype != torch.float32 or not x_nchw.is_cuda:
            raise RuntimeError("input must be a float32 CUDA/HIP tensor")
        dev = x_nchw.device
        x_nchw = x_nchw.contiguous()
        st = _stream()
        lib = self.lib
        sv = _Saved() if save else None
        self._u_all = self._wino_fwd_weights(params)
        bnws = self._buf("bnws", 4 * sp.n_bn_channels, device=dev)
        # repeated inference calls (tiled prediction): the 46 eval-mode scale/shift launches are skipped while neither the
        # parameters nor the running statistics changed (torch's version counters + the epochs of the raw device writes)
        if training:
            self._bn_epoch += 1
        akey = None if (training or save) else (params.data_ptr(), params._version, self._weights_epoch, bnstate.data_ptr(),
                                                bnstate._version, self._bn_epoch, bnws.data_ptr())
        self._affine_fresh = akey is not None and self._ws.get("affine_key") == akey
        self._ws["affine_key"] = akey
        self._fuse_eval = (self._fuse_eval_opt and not training and not save and self.winograd
                           and sp.decoder_kind not in ("resunet", "unetplusplus"))
        if save:
            # mean/invstd are needed by backward: keep a private copy target per forward
            bnws = torch.empty(4 * sp.n_bn_channels, dtype=torch.float32, device=dev)
            sv.d["bnws"] = bnws

        if nhwc:
            x = x_nchw
        else:
            x = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
            _lib.check(lib.dt_nchw_to_nhwc(_p(x_nchw), _p(x), B, Cin, H, W, st), "dt_nchw_to_nhwc")

        def keep(key, **kw):
            if save:
                sv.d[key] = kw

        # ---- stem
        if self._fuse_eval:     # inference: BatchNorm + ReLU in the stem kernel's epilogue, no raw output
            f1, h, w_ = self._conv_affine_direct(sp.stem, params, bnstate, bnws, x, None, 0, B, H, W)
        else:
            y, h, w_, ss = self._conv_bn(sp.stem, params, bnstate, bnws, x, None, 0, B, H, W, training, save_stats=save)
            f1 = self._bn_act(y, ss)
            keep("stem", x=x, y=y, z=f1, Hin=H, Win=W)
        hp, wp = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
        pool = torch.empty((B, hp, wp, 64), dtype=torch.float32, device=dev)
        amax = torch.empty((B, hp, wp, 64), dtype=torch.uint8, device=dev) if save else None
        _lib.check(lib.dt_maxpool3x3s2(_p(f1), _p(pool), _p(amax), B, h, w_, 64, st), "dt_maxpool3x3s2")
        keep("pool", amax=amax, H=h, W=w_)

        feats = [f1]
        cur, ch, cw = pool, hp, wp
        for li, blocks in enumerate(sp.layers):
            for bi, blk in enumerate(blocks):
                xin = cur
                if self._fuse_eval and blk.conv1.stride == 1 and blk.down is None:
                    z1 = self._conv_affine_eval(blk.conv1, params, bnstate, bnws, xin, None, 0, B, ch, cw)
                    out = None if z1 is None else self._conv_affine_eval(blk.conv2, params, bnstate, bnws, z1, None, 0, B,
                                                                         ch, cw, res=xin)
                    if out is not None:
                        cur = out
                        continue
                if self._fuse_eval and blk.down is not None and blk.conv2.cout % 64 == 0:
                    # first block of layers 2-4: relu(bn1(conv1)) and bn_d(down(x)) from the direct kernel's epilogue, the
                    # join relu(bn2(conv2) + .) in the Winograd kernel's
                    z1, h1, w1 = self._conv_affine_direct(blk.conv1, params, bnstate, bnws, xin, None, 0, B, ch, cw)
                    rd, _, _ = self._conv_affine_direct(blk.down, params, bnstate, bnws, xin, None, 0, B, ch, cw, relu=False)
                    out = self._conv_affine_eval(blk.conv2, params, bnstate, bnws, z1, None, 0, B, h1, w1, res=rd)
                    if out is not None:
                        cur, ch, cw = out, h1, w1
                        continue
                y1, h1, w1, ss1 = self._conv_bn(blk.conv1, params, bnstate, bnws, xin, None, 0, B, ch, cw, training,
                                                        save_stats=save)
                # z1 = relu(bn1(y1)) is virtual: conv2 applies it while staging y1 (A/B switch DT_MATERIALIZE_Z1:
                # a stored activation instead, read by the plain convolution / weight-gradient kernels)
                z1 = self._bn_act(y1, ss1) if self._mat_z1 else None
                y2, h2, w2, ss2 = self._conv_bn(blk.conv2, params, bnstate, bnws, y1 if z1 is None else z1, None, 0, B,
                                                h1, w1, training, in_ss=ss1 if z1 is None else None, save_stats=save)
                if blk.down is not None:
                    yd, _, _, ssd = self._conv_bn(blk.down, params, bnstate, bnws, xin, None, 0, B, ch, cw, training,
                                                  save_stats=save)
                    out = self._bn_act(y2, ss2, res=yd, res_ss=ssd)
                else:
                    yd = None
                    out = self._bn_act(y2, ss2, res=xin)
                keep(f"L{li}B{bi}", x=xin, y1=y1, z1=z1, y2=y2, yd=yd, out=out, Hin=ch, Win=cw, H=h2, W=w2)
                cur, ch, cw = out, h2, w2
            feats.append(cur)
        # feats = [f1, f2, f3, f4, f5]
        if sp.decoder_kind == "unetplusplus":
            d, dh, dw = self._forward_unetpp(feats, params, bnstate, bnws, B, training, save, keep)
            dec_blocks = []
        else:
            d, dh, dw = feats[4], ch, cw
            dec_blocks = sp.decoder
        d_ss = None   # (scale, shift) when d is a raw conv output with a virtual activation
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        for i, blk in enumerate(dec_blocks):
            skip = skips[i]
            Hin, Win = 2 * dh, 2 * dw
            if sp.decoder_kind == "resunet":
                # reference network/extra/resunet/decoder.py:40-52: conv1 -> conv2 (conv-BN-ReLU each, extra/modules.py)
                # plus the 1x1 identity_conv (with bias) of the up-sampled + concatenated input; no activation after
                # the sum.  The block output is a real tensor (the next block and the head read it).
                y1, h1, w1, ss1 = self._conv_bn(blk.conv1, params, bnstate, bnws, d, skip, 1, B, Hin, Win, training,
                                                save_stats=save)
                y2, h2, w2, ss2 = self._conv_bn(blk.conv2, params, bnstate, bnws, y1, None, 0, B, h1, w1, training,
                                                in_ss=ss1, save_stats=save)
                ic = blk.idc
                C0, C1 = d.shape[-1], (0 if skip is None else skip.shape[-1])
                idesc = self._desc(B, Hin, Win, C0, C1, 1, Hin, Win, ic.cout, 1, 1, 0)
                idy = torch.empty((B, Hin, Win, ic.cout), dtype=torch.float32, device=dev)
                self._conv(idesc, d, skip, params[ic.w_off:ic.w_off + ic.w_size], idy)
                out = self._bn_act(y2, ss2, res=idy, res_ss=(self._const_vec(1.0, ic.cout, dev),
                                                             params[ic.b_off:ic.b_off + ic.cout]), relu=2)
                del idy
                keep(f"D{i}", x=d, skip=skip, y1=y1, y2=y2, H=h1, W=w1)
                d, dh, dw, d_ss = out, h2, w2, None
                continue
            if self._fuse_eval and d_ss is None:
                z1 = self._conv_affine_eval(blk.conv1, params, bnstate, bnws, d, skip, 1, B, Hin, Win)
                if z1 is not None:
                    z2 = self._conv_affine_eval(blk.conv2, params, bnstate, bnws, z1, None, 0, B, Hin, Win)
                else:
                    # conv1 is neither a Winograd nor a narrow layer (dec3.conv1: 128 -> 32 from two sources): its raw output
                    # feeds conv2's lean kernel, which applies bn1 + ReLU while staging AND bn2 + ReLU in its epilogue
                    y1, _, _, ss1 = self._conv_bn(blk.conv1, params, bnstate, bnws, d, skip, 1, B, Hin, Win, training)
                    z2 = self._conv_affine_eval(blk.conv2, params, bnstate, bnws, y1, None, 0, B, Hin, Win, in_ss=ss1)
                    if z2 is None:
                        z2 = self._bn_act(self._conv_bn(blk.conv2, params, bnstate, bnws, y1, None, 0, B, Hin, Win, training,
                                                        in_ss=ss1)[0], self._ss(blk.conv2, bnws))
                if z2 is not None:
                    d, dh, dw, d_ss = z2, Hin, Win, None
                    continue
            y1, h1, w1, ss1 = self._conv_bn(blk.conv1, params, bnstate, bnws, d, skip, 1, B, Hin, Win, training,
                                            in_ss=d_ss, save_stats=save)
            z1 = self._bn_act(y1, ss1) if (self._mat_z1 and blk.conv2.cout % 64 == 0) else None
            y2, h2, w2, ss2 = self._conv_bn(blk.conv2, params, bnstate, bnws, y1 if z1 is None else z1, None, 0, B, h1, w1,
                                            training, in_ss=ss1 if z1 is None else None, save_stats=save)
            if i == len(sp.decoder) - 1 or (self._mat_z2 and sp.decoder[i + 1].conv1.cout % 64 == 0):
                z2 = self._bn_act(y2, ss2)      # the head kernel (or a Winograd conv1) reads a materialised activation
                nxt, nxt_ss = z2, None
            else:
                z2 = None                        # virtual: the next block's conv1 applies bn2+relu while staging
                nxt, nxt_ss = y2, ss2
            keep(f"D{i}", x=d, x_virtual=d_ss is not None, skip=skip, y1=y1, z1=z1, y2=y2, z2=z2, H=h1, W=w1)
            d, dh, dw, d_ss = nxt, h2, w2, nxt_ss

        # ---- head
        hd = sp.head
        K = hd.cout
        logits = torch.empty((B, K, dh, dw), dtype=torch.float32, device=dev)
        am64 = am8 = None
        if want_argmax == "int64":
            am64 = torch.empty((B, dh, dw), dtype=torch.int64, device=dev)
        elif want_argmax == "uint8":
            am8 = torch.empty((B, dh, dw), dtype=torch.uint8, device=dev)
        wh = params[hd.w_off:hd.w_off + hd.w_size]
        bh = params[hd.b_off:hd.b_off + K]
        e0 = self._pb()
        _lib.check(lib.dt_head_fwd(_p(d), _p(wh), _p(bh), _p(logits), _p(am64), _p(am8), B, dh, dw, hd.cin, K, st),
                   "dt_head_fwd")
        self._pe(e0, "head_fwd_kernel", 2.0 * 9 * hd.cin * K * B * dh * dw, 4.0 * B * dh * dw * (hd.cin + K))
        keep("head", x=d, H=dh, W=dw)
        if save:
            sv.d["B"] = B
            sv.d["training"] = bool(training)
            self.saved = sv
        return logits, (am64 if am64 is not None else am8)

    # ------------------------------------------------------------------ Unet++ decoder (smp UnetPlusPlus)
    def _cat_channels(self, tensors):
        """torch.cat(dim=1) of NHWC activations through dt_channel_slice -> (wide tensor, [(channel offset, width)])"""
        if len(tensors) == 1:
            return tensors[0], [(0, tensors[0].shape[-1])]
        B, H, W = tensors[0].shape[:3]
        Cw = sum(t.shape[-1] for t in tensors)
        bf = tensors[0].dtype == torch.bfloat16
        wide = torch.empty((B, H, W, Cw), dtype=tensors[0].dtype, device=tensors[0].device)
        slice_fn = self.lib.dt_channel_slice_bf16 if bf else self.lib.dt_channel_slice
        parts, off = [], 0
        for t in tensors:
            Cn = t.shape[-1]
            _lib.check(slice_fn(_p(t), _p(wide), B * H * W, Cn, Cw, off, 1, 0, _stream()), "dt_channel_slice")
            parts.append((off, Cn))
            off += Cn
        return wide, parts

    def _forward_unetpp(self, feats, params, bnstate, bnws, B, training, save, keep):
        """dense decoder of smp.UnetPlusPlus (wiring: reference network/extra/efficientunetplusplus/decoder.py:156-184):
        every node x_{d}_{l} = DecoderBlock(up x2 of its lower node, cat of the nodes / encoder feature on its level).
        Node outputs are materialised activations (they feed several consumers); conv2 reads conv1's raw output with the
        BatchNorm+ReLU fused into its staging like everywhere else."""
        sp = self.spec
        nodes = {f"f{k}": feats[4 - k] for k in range(5)}     # f0 = deepest encoder feature ... f4 = stem output
        for blk in sp.decoder:
            low = nodes[blk.low]
            Hin, Win = 2 * low.shape[1], 2 * low.shape[2]
            skip, parts = (None, []) if not blk.cat else self._cat_channels([nodes[n] for n in blk.cat])
            y1, h1, w1, ss1 = self._conv_bn(blk.conv1, params, bnstate, bnws, low, skip, 1, B, Hin, Win, training,
                                            save_stats=save)
            y2, h2, w2, ss2 = self._conv_bn(blk.conv2, params, bnstate, bnws, y1, None, 0, B, h1, w1, training,
                                            in_ss=ss1, save_stats=save)
            z2 = self._bn_act(y2, ss2)
            keep("P" + blk.name, x=low, skip=skip, parts=parts, y1=y1, y2=y2, z2=z2, H=h1, W=w1)
            nodes[blk.name] = z2
        out = nodes[sp.decoder[-1].name]
        return out, out.shape[1], out.shape[2]

    # ------------------------------------------------------------------ bf16 inference leg
    def _weight_table(self, device):
        """device table of the BatchNorm-ed convolutions except the stem for dt_weight_images:
        (w_off, taps, Cin, Cout, first_tile) rows, built once per device"""
        key = ("wtab", str(device))
        tab = self._tables.get(key) if hasattr(self, "_tables") else None
        if tab is None:
            rows, tiles = [], 0
            for c in self.spec.convs:
                if c is self.spec.stem or c is self.spec.head:
                    continue
                rows.append([c.w_off, c.k * c.k, c.cin, c.cout, tiles])
                tiles += c.k * c.k * ((c.cin + 31) // 32) * ((c.cout + 31) // 32)
            tab = (torch.tensor(rows, dtype=torch.int32).to(device), len(rows), tiles)
            if not hasattr(self, "_tables"):
                self._tables = {}
            self._tables[key] = tab
        return tab

    def _weight_images(self, params: torch.Tensor, out: torch.Tensor, mode: int):
        tab, n, tiles = self._weight_table(params.device)
        _lib.check(self.lib.dt_weight_images(_p(params), _p(out), _p(tab), n, tiles, mode, _stream()),
                   "dt_weight_images")

    def _stem_bf16(self, x, params, y, stats, B, H, W, Cin):
        """7x7/2 stem of the bf16 path into `y` (bf16) with optional BatchNorm partial statistics -> stat rows P.
        Even tiles wider than 32 pixels run on the bf16 MFMA kernels through the 2x2 space-to-depth image
        (dt_stem_s2d_bf16 + a 4x4 window: K = 256 bf16 instead of 147 fp32); anything else on the fp32 stem kernel."""
        lib, stc, st = self.lib, self.spec.stem, _stream()
        w7 = params[stc.w_off:stc.w_off + stc.w_size]
        h, w_ = y.shape[1], y.shape[2]
        if H % 2 == 0 and W % 2 == 0 and w_ > 16 and stc.cout % 64 == 0 and stc.k == 7 and stc.stride == 2:
            s2d = torch.empty((B, h, w_, 16), dtype=torch.bfloat16, device=x.device)
            _lib.check(lib.dt_stem_s2d_bf16(_p(x), _p(s2d), B, H, W, Cin, st), "dt_stem_s2d_bf16")
            wp = self._buf("stem_w4", 16 * stc.cout * 16, dtype=torch.bfloat16, device=x.device)
            _lib.check(lib.dt_stem_pack_weights_bf16(_p(w7), _p(wp), Cin, stc.cout, st), "dt_stem_pack_weights_bf16")
            desc = self._desc(B, h, w_, 16, 0, 0, h, w_, stc.cout, 4, 1, 2)
            P = lib.dt_conv2d_bf16_stat_rows(C.byref(desc))
            sbuf = self._buf("bn_stats", lib.dt_bn_stats_floats(P, stc.cout), device=x.device) if stats else None
            self._conv_bf16(desc, s2d, None, wp, y, None, sbuf, None, "dt_conv2d_bf16(stem)")
            self._stem_s2d = s2d if stats else None     # training: the weight gradient reuses the image
            return P, sbuf
        sdesc = self._desc(B, H, W, Cin, 0, 0, h, w_, stc.cout, stc.k, stc.stride, stc.pad)
        P = lib.dt_conv2d_stat_rows(C.byref(sdesc))
        sbuf = self._buf("bn_stats", lib.dt_bn_stats_floats(P, stc.cout), device=x.device) if stats else None
        _lib.check(lib.dt_conv2d_out_bf16(C.byref(sdesc), _p(x), _p(w7), _p(y), _p(sbuf), st), "dt_conv2d_out_bf16")
        self._stem_s2d = None
        return P, sbuf

    def _resunet_join_bf16(self, blk, params, wb, d, skip, y2, ss2, B, H, W):
        """ResUnet decoder block output under AMP: bf16(relu(y2 * scale2 + shift2) + identity_conv(up(d) | skip) + bias) —
        the 1x1 identity convolution over the virtual (up-sampled, concatenated) input on the bf16 kernels, the join in
        dt_bn_act_bf16 (relu = 2: ReLU on the main branch only)"""
        ic, dev, bf = blk.idc, y2.device, torch.bfloat16
        C0, C1 = d.shape[-1], (0 if skip is None else skip.shape[-1])
        idesc = self._desc(B, H, W, C0, C1, 1, H, W, ic.cout, 1, 1, 0)
        idy = torch.empty((B, H, W, ic.cout), dtype=bf, device=dev)
        self._conv_bf16(idesc, d, skip, wb[ic.w_off:ic.w_off + ic.w_size], idy, None, None, None, "dt_conv2d_bf16(identity_conv)")
        out = torch.empty((B, H, W, ic.cout), dtype=bf, device=dev)
        _lib.check(self.lib.dt_bn_act_bf16(_p(y2), 0, _p(ss2[0]), _p(ss2[1]), _p(idy), _p(self._const_vec(1.0, ic.cout, dev)),
                                           _p(params[ic.b_off:ic.b_off + ic.cout]), _p(out), B * H * W, ic.cout, 2, _stream()),
                   "dt_bn_act_bf16")
        return out

    def _bf16_weights(self, params: torch.Tensor, dgrad: bool = False, chunked: bool = False):
        """bf16 images of every conv weight except stem and head: [tap][Cout][Cin] for the forward convs, or the
        data-gradient image (HWIO with reversed taps).  Repacked when the flat parameter buffer changed: torch's
        version counter catches torch-side writes, ``self.weights_dirty`` the fused optimiser's raw writes."""
        name = ("bf16_wd" if dgrad else "bf16_w") + ("c" if chunked else "")   # chunked: [tap][K/32][N][32] (DMA kernels)
        key = (params.data_ptr(), params._version, self._weights_epoch)
        if self._ws.get(name + "_key") == key:
            return self._ws[name]
        buf = self._ws.get(name)
        if buf is None or buf.device != params.device:
            buf = torch.empty(self.spec.n_params, dtype=torch.bfloat16, device=params.device)
        mode = (4 if dgrad else 3) if chunked else (2 if dgrad else 1)
        self._weight_images(params, buf, mode)     # every layer's image in one launch
        self._ws[name + "_key"], self._ws[name] = key, buf
        return buf

    def _bf16_weights_all(self, params: torch.Tensor):
        """the four bf16 images a training step reads (forward / data gradient, plain / chunked) in ONE launch — one read of
        the fp32 parameters instead of four; fills the caches _bf16_weights() looks at"""
        key = (params.data_ptr(), params._version, self._weights_epoch)
        names = ("bf16_w", "bf16_wd", "bf16_wc", "bf16_wdc")
        if all(self._ws.get(n + "_key") == key for n in names):
            return
        bufs = []
        for n in names:
            b = self._ws.get(n)
            if b is None or b.device != params.device:
                b = torch.empty(self.spec.n_params, dtype=torch.bfloat16, device=params.device)
            bufs.append(b)
        tab, nl, tiles = self._weight_table(params.device)
        _lib.check(self.lib.dt_weight_images_bf16_all(_p(params), _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(bufs[3]), _p(tab), nl,
                                                      tiles, _stream()), "dt_weight_images_bf16_all")
        for n, b in zip(names, bufs):
            self._ws[n + "_key"], self._ws[n] = key, b

    def _bf16_mt(self, desc) -> int:
        """kernel family dt_conv2d_bf16 picks for `desc`: 8 = LDS-DMA staged (conv_bf16_dma.hip), 16 = lean narrow-layer
        kernel (conv_bf16_narrow.hip), else the register-staged kernels' tile multiplier"""
        tw, tn, ck, mt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if self.lib.dt_conv2d_bf16_config(C.byref(desc), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt)) != 0:
            return -1
        return mt.value

    def _uses_dma_kernel(self, desc) -> bool:
        """the LDS-DMA staged kernels (reported as mt == 8) read the CHUNKED weight images"""
        return self._bf16_mt(desc) == 8

    def _conv_bf16(self, desc, src0, src1, w, out0, out1, stats, in_ss, what="dt_conv2d_bf16", w_chunked=None):
        if w_chunked is not None and self._uses_dma_kernel(desc):
            w = w_chunked
        prof = self.profile
        if prof is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(self.lib.dt_conv2d_bf16(C.byref(desc), _p(src0), _p(src1), _p(w), _p(out0), _p(out1), _p(stats),
                                           _p(in_ss[0]) if in_ss else None, _p(in_ss[1]) if in_ss else None,
                                           _stream()), what)
        if prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tw, tn, ck, mt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.lib.dt_conv2d_bf16_config(C.byref(desc), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt))
            flops = 2.0 * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout * desc.Ho * desc.Wo * desc.B
            if desc.mode0 == 2:
                flops /= 4.0
            if desc.ksize == 4:   # space-to-depth stem: the algorithmic work is the 7x7 x in_channels window
                flops = 2.0 * 49 * self.spec.in_channels * desc.Cout * desc.Ho * desc.Wo * desc.B
            sdiv = 4 if desc.mode0 else 1
            nbytes = 2.0 * desc.B * (desc.Hin * desc.Win * (desc.C0 / sdiv + desc.C1) +
                                     desc.Ho * desc.Wo * desc.Cout * (2 if desc.accumulate else 1)) + \
                2.0 * desc.ksize ** 2 * (desc.C0 + desc.C1) * desc.Cout
            if mt.value == 8:     # the LDS-DMA staged 512-pixel kernel (conv_bf16_dma.hip)
                name = f"conv3x3_bf16_dma_kernel<{'true' if in_ss else 'false'}, {2 if desc.accumulate else 0}>"
            elif mt.value == 16:  # the lean narrow-layer kernel (conv_bf16_narrow.hip)
                name = (f"conv3x3_bf16_narrow_kernel<{desc.C0 // 16}, {desc.Cout // 16}, "
                        f"{'true' if in_ss else 'false'}, false>")
            else:
                name = (f"conv_fwd_bf16_kernel<{desc.ksize}, {desc.stride}, {tw.value}, {tn.value}, {ck.value}, "
                        f"{mt.value}, {'true' if in_ss else 'false'}>")
            prof.append((name, flops, e0, e1, nbytes))

    # ---- weight gradients run on a side stream, concurrently with the data-gradient chain of the main stream:
    # both only depend on dy, and the many tiny reduction / finalize launches of either chain otherwise leave
    # the chip idle.  Ordering: side waits for the event recorded after dy was produced; main waits for the side
    # stream before a gradient bucket is handed to the reducer / optimiser.  Tensors touched by the side stream are
    # registered with the caching allocator (record_stream) so they are not recycled while still in use.
    def _side_stream(self, device):
        st = self._ws.get("side_stream")
        if st is None or st.device != device:
            # high priority = its own hardware queue.  ROCm deals streams round-robin onto GPU_MAX_HW_QUEUES (4) hardware
            # queues; once an RCCL process group has created its streams a default-priority side stream lands on the
            # queue of the main stream and the weight gradients serialise behind the chain they should run beside
            # (measured with an RCCL group initialised, same box: 729 tiles/s -> 770; without a group 767 either way)
            st = torch.cuda.Stream(device=device, priority=int(os.environ.get("DT_SIDE_PRIORITY", "-1")))
            self._ws["side_stream"] = st
        return st

    def _on_side(self, fn, *tensors):
        main = torch.cuda.current_stream()
        side = self._side_stream(main.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        for t in tensors:
            if t is not None:
                t.record_stream(side)
        with torch.cuda.stream(side):
            fn()

    def _join_side(self):
        main = torch.cuda.current_stream()
        side = self._ws.get("side_stream")
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)

    def _bucket_done(self, bucket):
        """a contiguous range of the flat gradient buffer is complete: hand it to ``grad_hook`` (the data-parallel
        all-reduce).  Its producers ran on the main stream (BatchNorm / head gradients) AND on the weight-gradient side
        stream; instead of joining main <- side (which drains the overlap at every bucket) the hook is called with the
        SIDE stream current, after that stream has been ordered behind main's work so far: the collective waits for
        both, the main stream waits for nobody."""
        if not self.grad_hook:
            return
        main = torch.cuda.current_stream()
        side = self._ws.get("side_stream")
        if side is None:
            self.grad_hook(*bucket)
            return
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self.grad_hook(*bucket)

    def mark_weights_changed(self):
        """call after writing the flat parameter buffer behind torch's back (fused optimiser step)"""
        self._weights_epoch += 1

    def forward_bf16_eval(self, x_nchw: torch.Tensor, params: torch.Tensor, bnstate: torch.Tensor,
                          want_argmax: Optional[str] = None):
        """eval-mode forward with bf16 activations/weights and fp32 accumulation (stem and head stay fp32)."""
        sp, lib = self.spec, self.lib
        if x_nchw.dim() != 4 or x_nchw.shape[1] != sp.in_channels:
            raise RuntimeError(f"expected input [B,{sp.in_channels},H,W], got {tuple(x_nchw.shape)}")
        B, Cin, H, W = x_nchw.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"H and W must be divisible by 32 (encoder depth 5), got {H}x{W}")
        dev = x_nchw.device
        st = _stream()
        wb = self._bf16_weights(params)
        wbc = self._bf16_weights(params, chunked=True)
        bnws = self._buf("bnws", 4 * sp.n_bn_channels, device=dev)
        bf = torch.bfloat16
        akey = (params.data_ptr(), params._version, self._weights_epoch, bnstate.data_ptr(), bnstate._version,
                self._bn_epoch, bnws.data_ptr())
        fresh = self._ws.get("affine_key") == akey      # same coefficients as the previous inference call: skip 46 launches
        self._ws["affine_key"] = akey

        def affine(c):
            ss = self._ss(c, bnws)
            if fresh:
                return ss
            _lib.check(lib.dt_bn_eval_affine(_p(params[c.g_off:c.g_off + c.cout]), _p(params[c.b_off:c.b_off + c.cout]),
                                             _p(bnstate[2 * c.bn_off:2 * c.bn_off + c.cout]),
                                             _p(bnstate[2 * c.bn_off + c.cout:2 * c.bn_off + 2 * c.cout]), BN_EPS,
                                             c.cout, _p(ss[0]), _p(ss[1]), st), "dt_bn_eval_affine")
            return ss

        def conv(c, src0, src1, mode0, Hin, Win, in_ss=None):
            Ho = (Hin + 2 * c.pad - c.k) // c.stride + 1
            Wo = (Win + 2 * c.pad - c.k) // c.stride + 1
            C0 = src0.shape[-1]
            C1 = 0 if src1 is None else src1.shape[-1]
            desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
            y = torch.empty((B, Ho, Wo, c.cout), dtype=bf, device=dev)
            self._conv_bf16(desc, src0, src1, wb[c.w_off:c.w_off + c.w_size], y, None, None, in_ss,
                            w_chunked=wbc[c.w_off:c.w_off + c.w_size])
            return y, Ho, Wo, affine(c)

        def bn_act(y, ss, res=None, res_ss=None, y_f32=False):
            Bq, Hq, Wq, Cq = y.shape
            z = torch.empty((Bq, Hq, Wq, Cq), dtype=bf, device=dev)
            _lib.check(lib.dt_bn_act_bf16(_p(y), 1 if y_f32 else 0, _p(ss[0]), _p(ss[1]), _p(res),
                                          _p(res_ss[0]) if res_ss else None, _p(res_ss[1]) if res_ss else None, _p(z),
                                          Bq * Hq * Wq, Cq, 1, st), "dt_bn_act_bf16")
            return z

        x = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
        _lib.check(lib.dt_nchw_to_nhwc(_p(x_nchw.contiguous()), _p(x), B, Cin, H, W, st), "dt_nchw_to_nhwc")
        # stem: bf16 MFMA over the space-to-depth image (fp32-MFMA kernel for odd / tiny tiles), bf16 output
        stc = sp.stem
        h, w_ = (H + 2 * stc.pad - stc.k) // stc.stride + 1, (W + 2 * stc.pad - stc.k) // stc.stride + 1
        y = torch.empty((B, h, w_, stc.cout), dtype=bf, device=dev)
        self._stem_bf16(x, params, y, False, B, H, W, Cin)
        f1 = bn_act(y, affine(stc))
        hp, wp = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
        pool = torch.empty((B, hp, wp, 64), dtype=bf, device=dev)
        _lib.check(lib.dt_maxpool3x3s2_bf16(_p(f1), _p(pool), B, h, w_, 64, st), "dt_maxpool3x3s2_bf16")
        feats = [f1]
        cur, ch, cw = pool, hp, wp
        for blocks in sp.layers:
            for blk in blocks:
                y1, h1, w1, ss1 = conv(blk.conv1, cur, None, 0, ch, cw)
                y2, h2, w2, ss2 = conv(blk.conv2, y1, None, 0, h1, w1, in_ss=ss1)
                if blk.down is not None:
                    yd, _, _, ssd = conv(blk.down, cur, None, 0, ch, cw)
                    out = bn_act(y2, ss2, res=yd, res_ss=ssd)
                else:
                    out = bn_act(y2, ss2, res=cur)
                cur, ch, cw = out, h2, w2
            feats.append(cur)
        d, dh, dw, d_ss = feats[4], ch, cw, None
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        if sp.decoder_kind == "unetplusplus":      # dense decoder (fp32 twin: _forward_unetpp): node outputs are stored tensors
            nodes = {f"f{k}": feats[4 - k] for k in range(5)}
            for blk in sp.decoder:
                low = nodes[blk.low]
                skip = None if not blk.cat else self._cat_channels([nodes[n] for n in blk.cat])[0]
                y1, h1, w1, ss1 = conv(blk.conv1, low, skip, 1, 2 * low.shape[1], 2 * low.shape[2])
                y2, h2, w2, ss2 = conv(blk.conv2, y1, None, 0, h1, w1, in_ss=ss1)
                nodes[blk.name] = bn_act(y2, ss2)
            d = nodes[sp.decoder[-1].name]
            dh, dw = d.shape[1], d.shape[2]
        for i, blk in enumerate(sp.decoder if sp.decoder_kind != "unetplusplus" else ()):
            y1, h1, w1, ss1 = conv(blk.conv1, d, skips[i], 1, 2 * dh, 2 * dw, in_ss=d_ss)
            y2, h2, w2, ss2 = conv(blk.conv2, y1, None, 0, h1, w1, in_ss=ss1)
            if sp.decoder_kind == "resunet":     # relu(bn2(conv2(.))) + identity_conv(up + skip) (resunet/decoder.py:40-52)
                d, d_ss = self._resunet_join_bf16(blk, params, wb, d, skips[i], y2, ss2, B, h2, w2), None
                dh, dw = h2, w2
                continue
            if i == len(sp.decoder) - 1:
                d, d_ss = bn_act(y2, ss2), None
            else:
                d, d_ss = y2, ss2
            dh, dw = h2, w2
        hd = sp.head
        K = hd.cout
        logits = torch.empty((B, K, dh, dw), dtype=torch.float32, device=dev)
        am64 = torch.empty((B, dh, dw), dtype=torch.int64, device=dev) if want_argmax == "int64" else None
        am8 = torch.empty((B, dh, dw), dtype=torch.uint8, device=dev) if want_argmax == "uint8" else None
        _lib.check(lib.dt_head_fwd_bf16(_p(d), _p(params[hd.w_off:hd.w_off + hd.w_size]),
                                        _p(params[hd.b_off:hd.b_off + K]), _p(logits), _p(am64), _p(am8), B, dh, dw,
                                        hd.cin, K, st), "dt_head_fwd_bf16")
        return logits, (am64 if am64 is not None else am8)

    # ------------------------------------------------------------------ bf16 training (BASELINE configs[2])
    def forward_bf16_train(self, x_nchw: torch.Tensor, params: torch.Tensor, bnstate: torch.Tensor):
        """training-mode forward with bf16 activations / weights, fp32 accumulation, fp32 BatchNorm statistics
        (taken from the accumulators), fp32 master parameters.  Stem and head run in fp32."""
        sp, lib = self.spec, self.lib
        B, Cin, H, W = x_nchw.shape
        if H % 32 or W % 32 or Cin != sp.in_channels:
            raise RuntimeError(f"bad input {tuple(x_nchw.shape)}")
        dev, st, bf = x_nchw.device, _stream(), torch.bfloat16
        self._bn_epoch += 1          # running statistics are rewritten on the device (invalidates cached eval affines)
        if self._bf16_images_fused:
            self._bf16_weights_all(params)
        wb = self._bf16_weights(params)
        wbc = self._bf16_weights(params, chunked=True)
        sv = _Saved()
        bnws = torch.empty(4 * sp.n_bn_channels, dtype=torch.float32, device=dev)
        sv.d["bnws"] = bnws
        nb = sp.n_bn_channels

        def finalize(c, stats, P, count):
            _lib.check(lib.dt_bn_finalize(_p(stats), P, c.cout, float(count), _p(params[c.g_off:c.g_off + c.cout]),
                                          _p(params[c.b_off:c.b_off + c.cout]), BN_EPS, BN_MOMENTUM,
                                          _p(bnstate[2 * c.bn_off:2 * c.bn_off + c.cout]),
                                          _p(bnstate[2 * c.bn_off + c.cout:2 * c.bn_off + 2 * c.cout]),
                                          _p(bnws[c.bn_off:c.bn_off + c.cout]),
                                          _p(bnws[nb + c.bn_off:nb + c.bn_off + c.cout]),
                                          _p(bnws[2 * nb + c.bn_off:2 * nb + c.bn_off + c.cout]),
                                          _p(bnws[3 * nb + c.bn_off:3 * nb + c.bn_off + c.cout]), st), "dt_bn_finalize")
            return self._ss(c, bnws)

        def conv(c, src0, src1, mode0, Hin, Win, in_ss=None):
            Ho = (Hin + 2 * c.pad - c.k) // c.stride + 1
            Wo = (Win + 2 * c.pad - c.k) // c.stride + 1
            C0 = src0.shape[-1]
            C1 = 0 if src1 is None else src1.shape[-1]
            desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
            P = lib.dt_conv2d_bf16_stat_rows(C.byref(desc))
            if P <= 0:
                raise RuntimeError(lib.dt_last_error().decode())
            stats = self._buf("bn_stats", lib.dt_bn_stats_floats(P, c.cout), device=dev)
            y = torch.empty((B, Ho, Wo, c.cout), dtype=bf, device=dev)
            self._conv_bf16(desc, src0, src1, wb[c.w_off:c.w_off + c.w_size], y, None, stats, in_ss,
                            w_chunked=wbc[c.w_off:c.w_off + c.w_size])
            return y, Ho, Wo, finalize(c, stats, P, B * Ho * Wo)

        def bn_act(y, ss, res=None, res_ss=None, y_f32=False):
            Bq, Hq, Wq, Cq = y.shape
            z = torch.empty((Bq, Hq, Wq, Cq), dtype=bf, device=dev)
            e0 = self._pb()
            _lib.check(lib.dt_bn_act_bf16(_p(y), 1 if y_f32 else 0, _p(ss[0]), _p(ss[1]), _p(res),
                                          _p(res_ss[0]) if res_ss else None, _p(res_ss[1]) if res_ss else None, _p(z),
                                          Bq * Hq * Wq, Cq, 1, st), "dt_bn_act_bf16")
            self._pe(e0, "bn_act_bf16_kernel", 0.0, 2.0 * y.numel() * (2 + (res is not None)))
            return z

        x = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
        _lib.check(lib.dt_nchw_to_nhwc(_p(x_nchw.contiguous()), _p(x), B, Cin, H, W, st), "dt_nchw_to_nhwc")
        # stem: bf16 MFMA over the space-to-depth image (fp32-MFMA kernel for odd / tiny tiles), fp32 statistics
        stc = sp.stem
        h, w_ = (H + 2 * stc.pad - stc.k) // stc.stride + 1, (W + 2 * stc.pad - stc.k) // stc.stride + 1
        ystem = torch.empty((B, h, w_, stc.cout), dtype=bf, device=dev)
        Pst, sstats = self._stem_bf16(x, params, ystem, True, B, H, W, Cin)
        ss = finalize(stc, sstats, Pst, B * h * w_)
        f1 = bn_act(ystem, ss)
        sv.d["stem"] = dict(x=x, y=ystem, z=f1, Hin=H, Win=W, s2d=self._stem_s2d)
        self._stem_s2d = None
        hp, wp = (h + 2 - 3) // 2 + 1, (w_ + 2 - 3) // 2 + 1
        pool = torch.empty((B, hp, wp, 64), dtype=bf, device=dev)
        amax = torch.empty((B, hp, wp, 64), dtype=torch.uint8, device=dev)
        _lib.check(lib.dt_maxpool3x3s2_bf16_amax(_p(f1), _p(pool), _p(amax), B, h, w_, 64, st), "dt_maxpool3x3s2_bf16_amax")
        sv.d["pool"] = dict(amax=amax, H=h, W=w_)
        self._tr("pool", pool)
        feats = [f1]
        cur, ch, cw = pool, hp, wp
        for li, blocks in enumerate(sp.layers):
            for bi, blk in enumerate(blocks):
                xin = cur
                y1, h1, w1, ss1 = conv(blk.conv1, xin, None, 0, ch, cw)
                z1 = bn_act(y1, ss1) if self._mat_z1_bf16 else None      # A/B switch DT_BF16_MAT_Z1 (see __init__)
                y2, h2, w2, ss2 = conv(blk.conv2, y1 if z1 is None else z1, None, 0, h1, w1, in_ss=ss1 if z1 is None else None)
                if blk.down is not None:
                    yd, _, _, ssd = conv(blk.down, xin, None, 0, ch, cw)
                    out = bn_act(y2, ss2, res=yd, res_ss=ssd)
                else:
                    yd = None
                    out = bn_act(y2, ss2, res=xin)
                sv.d[f"L{li}B{bi}"] = dict(x=xin, y1=y1, z1=z1, y2=y2, yd=yd, out=out, Hin=ch, Win=cw, H=h2, W=w2)
                cur, ch, cw = out, h2, w2
            feats.append(cur)
        d, dh, dw, d_ss = feats[4], ch, cw, None
        skips = [feats[3], feats[2], feats[1], feats[0], None]
        if sp.decoder_kind == "unetplusplus":
            # smp UnetPlusPlus under AMP (fp32 twin: _forward_unetpp): every node = DecoderBlock(up x2 of its lower node, cat of
            # the nodes / encoder feature on its level); node outputs are stored bf16 activations (several consumers), conv2
            # reads conv1's raw output with BatchNorm + ReLU applied while staging
            nodes = {f"f{k}": feats[4 - k] for k in range(5)}
            for blk in sp.decoder:
                low = nodes[blk.low]
                skip, parts = (None, []) if not blk.cat else self._cat_channels([nodes[n] for n in blk.cat])
                y1, h1, w1, ss1 = conv(blk.conv1, low, skip, 1, 2 * low.shape[1], 2 * low.shape[2])
                y2, h2, w2, ss2 = conv(blk.conv2, y1, None, 0, h1, w1, in_ss=ss1)
                z2 = bn_act(y2, ss2)
                sv.d["P" + blk.name] = dict(x=low, skip=skip, parts=parts, y1=y1, y2=y2, z2=z2, H=h1, W=w1)
                nodes[blk.name] = z2
            d = nodes[sp.decoder[-1].name]
            dh, dw = d.shape[1], d.shape[2]
        for i, blk in enumerate(sp.decoder if sp.decoder_kind != "unetplusplus" else ()):
            if sp.decoder_kind == "resunet":
                # reference network/extra/resunet/decoder.py:40-52 under AMP: conv1 -> conv2 (conv-BN-ReLU each, both
                # activations virtual) + the 1x1 identity_conv (bias) of the up-sampled + concatenated input; no activation
                # after the sum; the block output is a stored bf16 tensor
                y1, h1, w1, ss1 = conv(blk.conv1, d, skips[i], 1, 2 * dh, 2 * dw)
                y2, h2, w2, ss2 = conv(blk.conv2, y1, None, 0, h1, w1, in_ss=ss1)
                out = self._resunet_join_bf16(blk, params, wb, d, skips[i], y2, ss2, B, h2, w2)
                self._tr(f"D{i}.out", out)
                sv.d[f"D{i}"] = dict(x=d, skip=skips[i], y1=y1, y2=y2, H=h1, W=w1)
                d, dh, dw, d_ss = out, h2, w2, None
                continue
            y1, h1, w1, ss1 = conv(blk.conv1, d, skips[i], 1, 2 * dh, 2 * dw, in_ss=d_ss)
            # round 3: the activations of the 64+-channel decoder blocks are stored (bn_act, 4 B per element moved) so that
            # conv2 / the next conv1 AND their weight gradients run the pure LDS-DMA kernels (a DMA cannot transform);
            # with the register-staged weight gradient this was neutral (2,236 vs 2,234), DT_BF16_MAT_DEC=0 restores it
            wide = self._mat_dec_bf16 and blk.conv2.cout % 64 == 0
            z1 = bn_act(y1, ss1) if wide else None
            y2, h2, w2, ss2 = conv(blk.conv2, y1 if z1 is None else z1, None, 0, h1, w1, in_ss=ss1 if z1 is None else None)
            if i == len(sp.decoder) - 1 or wide:
                z2 = bn_act(y2, ss2)
                nxt, nxt_ss = z2, None
            else:
                z2 = None
                nxt, nxt_ss = y2, ss2
            sv.d[f"D{i}"] = dict(x=d, x_virtual=d_ss is not None, skip=skips[i], y1=y1, z1=z1, y2=y2, z2=z2, H=h1, W=w1)
            d, dh, dw, d_ss = nxt, h2, w2, nxt_ss
        hd = sp.head
        K = hd.cout
        logits = torch.empty((B, K, dh, dw), dtype=torch.float32, device=dev)
        _lib.check(lib.dt_head_fwd_bf16(_p(d), _p(params[hd.w_off:hd.w_off + hd.w_size]),
                                        _p(params[hd.b_off:hd.b_off + K]), _p(logits), None, None, B, dh, dw, hd.cin, K,
                                        st), "dt_head_fwd_bf16")
        sv.d["head"] = dict(x=d, H=dh, W=dw)
        sv.d["B"] = B
        sv.d["bf16"] = True
        self.saved = sv
        return logits

    def backward_bf16(self, dlogits: torch.Tensor, params: torch.Tensor, grads: torch.Tensor,
                      saved: Optional[_Saved] = None):
        """reverse pass of forward_bf16_train: bf16 activation gradients, fp32 parameter gradients"""
        sp, lib = self.spec, self.lib
        sv = saved if saved is not None else self.saved
        if sv is None:
            raise RuntimeError("backward called without a saved forward (was another forward run in between?)")
        S = sv.d
        B, bnws = S["B"], S["bnws"]
        dev, st, bf = dlogits.device, _stream(), torch.bfloat16
        wbd = self._bf16_weights(params, dgrad=True)
        wbdc = self._bf16_weights(params, dgrad=True, chunked=True)
        nb = sp.n_bn_channels

        def bn_bwd(c, dout, out_act, y, dres=None, dres_acc=False, virtual_act=False, reduced=None):
            Bq, Hq, Wq, Cq = y.shape
            n_pix = Bq * Hq * Wq
            mean = bnws[c.bn_off:c.bn_off + Cq]
            invstd = bnws[nb + c.bn_off:nb + c.bn_off + Cq]
            asc, ash = self._ss(c, bnws) if virtual_act else (None, None)
            if reduced is not None:       # partial sums came out of the data-gradient kernel that wrote dout
                red, P = reduced
            else:
                P = lib.dt_bn_bwd_rows_bf16(n_pix)
                red = self._buf("bn_red", lib.dt_bn_stats_floats(P, Cq), device=dev)
                e0 = self._pb()
                _lib.check(lib.dt_bn_bwd_reduce_bf16(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(asc),
                                                     _p(ash), _p(red), n_pix, Cq, st), "dt_bn_bwd_reduce_bf16")
                self._pe(e0, "bn_bwd_reduce_bf16_kernel", 0.0, 2.0 * y.numel() * (2 + (out_act is not None)))
            dy = torch.empty(y.shape, dtype=bf, device=dev)
            e0 = self._pb()
            _lib.check(lib.dt_bn_bwd_apply_bf16(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd),
                                                _p(params[c.g_off:c.g_off + Cq]), _p(asc), _p(ash), _p(red), P,
                                                _p(grads[c.g_off:c.g_off + Cq]), _p(grads[c.b_off:c.b_off + Cq]),
                                                _p(dy), _p(dres), 1 if dres_acc else 0, n_pix, Cq, st),
                       "dt_bn_bwd_apply_bf16")
            self._pe(e0, "bn_bwd_apply_bf16_kernel", 0.0,
                     2.0 * y.numel() * (3 + (out_act is not None) + (dres is not None) * (2 if dres_acc else 1)))
            return dy

        def wgrad(c, src0, src1, mode0, Hin, Win, dy, in_ss=None, side=True):
            if side and self.overlap_wgrad_bf16:
                self._on_side(lambda: wgrad(c, src0, src1, mode0, Hin, Win, dy, in_ss, side=False), src0, src1, dy)
                return
            Ho, Wo = dy.shape[1], dy.shape[2]
            C0 = src0.shape[-1]
            C1 = 0 if src1 is None else src1.shape[-1]
            desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
            nbytes = lib.dt_conv2d_wgrad_bf16_workspace(C.byref(desc))
            if nbytes == 0:
                raise RuntimeError(lib.dt_last_error().decode())
            ws = self._buf("wgrad_ws", nbytes // 4, device=dev)
            e0 = self._pb()
            _lib.check(lib.dt_conv2d_wgrad_bf16(C.byref(desc), _p(src0), _p(src1), _p(dy),
                                                _p(grads[c.w_off:c.w_off + c.w_size]), _p(ws), ws.numel() * 4,
                                                _p(in_ss[0]) if in_ss else None, _p(in_ss[1]) if in_ss else None,
                                                _stream()), "dt_conv2d_wgrad_bf16")
            if e0 is not None:
                fl, wbytes = self._conv_work(desc, 2)
                self._pe(e0, "conv_wgrad_bf16_kernel (+ split-K final)", fl, wbytes + 2.0 * c.w_size)   # fp32 gradient out

        def dgrad_bn(c, dy, Hh, Ww, out0, bn_conv, y, act=None):
            """stride-1 data gradient of conv c with the BatchNorm-backward reduction of bn_conv fused (fp32 twin:
            _dgrad_bn; act = stored block output -> gradient join) -> (red, P)"""
            Cq = bn_conv.cout
            desc = self._desc(B, Hh, Ww, c.cout, 0, 0, Hh, Ww, c.cin, c.k, 1, c.k - 1 - c.pad, 0, 0 if act is None else 1)
            P = lib.dt_conv2d_bf16_stat_rows(C.byref(desc))
            red = self._buf("bn_red_fused", lib.dt_bn_stats_floats(P, Cq), device=dev)
            asc, ash = self._ss(bn_conv, bnws) if act is None else (None, None)
            fuse = _lib.BnBwdFuse(_p(y), _p(bnws[bn_conv.bn_off:bn_conv.bn_off + Cq]),
                                  _p(bnws[nb + bn_conv.bn_off:nb + bn_conv.bn_off + Cq]), _p(asc), _p(ash), _p(act))
            dma = self._uses_dma_kernel(desc)
            wsel = wbdc if dma else wbd
            e0 = self._pb()
            _lib.check(lib.dt_conv2d_bf16_bn_bwd(C.byref(desc), _p(dy), _p(wsel[c.w_off:c.w_off + c.w_size]), _p(out0),
                                                 _p(red), C.byref(fuse), _stream()), "dt_conv2d_bf16_bn_bwd")
            if e0 is not None:
                fl, wbytes = self._conv_work(desc, 2)
                narrow = self._bf16_mt(desc) == 16
                self._pe(e0, f"conv3x3_bf16_dma_kernel<false, {1 if act is None else 3}>" if dma else
                         (f"conv3x3_bf16_narrow_kernel<{desc.C0 // 16}, {desc.Cout // 16}, false, true>" if narrow else
                          "conv_fwd_bf16_kernel (data gradient + BatchNorm-backward sums)"), fl, wbytes + 2.0 * out0.numel())
            return red, P

        def dgrad(c, dy, Hin, Win, out0, out1=None, split=0, acc=False):
            Ho, Wo = dy.shape[1], dy.shape[2]
            pad = c.k - 1 - c.pad
            if c.stride == 1:
                desc = self._desc(B, Ho, Wo, c.cout, 0, 0, Hin, Win, c.cin, c.k, 1, pad, split, 1 if acc else 0)
            else:
                desc = self._desc(B, Hin, Win, c.cout, 0, 2, Hin, Win, c.cin, c.k, 1, pad, split, 1 if acc else 0)
            self._conv_bf16(desc, dy, None, wbd[c.w_off:c.w_off + c.w_size], out0, out1, None, None,
                            "dt_conv2d_bf16(dgrad)", w_chunked=wbdc[c.w_off:c.w_off + c.w_size])

        # ---- head (fp32) -> bf16 gradient of the last decoder activation
        hd, hsv = sp.head, S["head"]
        H, W, K = hsv["H"], hsv["W"], sp.head.cout
        g = torch.empty(hsv["x"].shape, dtype=bf, device=dev)
        P = lib.dt_head_bwd_rows(B, H, W)
        red = self._buf("head_red", lib.dt_head_bwd_red_floats(B, H, W, hd.cin, K), device=dev)
        _lib.check(lib.dt_head_bwd_bf16(_p(hsv["x"]), _p(params[hd.w_off:hd.w_off + hd.w_size]),
                                        _p(dlogits.contiguous()), _p(g), _p(red), B, H, W, hd.cin, K, st),
                   "dt_head_bwd_bf16")
        _lib.check(lib.dt_head_bwd_finalize(_p(red), P, _p(grads[hd.w_off:hd.w_off + hd.w_size]),
                                            _p(grads[hd.b_off:hd.b_off + K]), hd.cin, K, st), "dt_head_bwd_finalize")
        self._tr("head.g", g)

        skip_grads = [None] * 5
        g_red = None
        if sp.decoder_kind == "resunet" and hd.sd_k == 1:
            # the 1x1 head lives in the centre tap of the 3x3 head kernel: the other taps stay zero (like backward())
            gw = grads[hd.w_off:hd.w_off + hd.w_size].view(K, 9, hd.cin)
            gw[:, :4].zero_()
            gw[:, 5:].zero_()
        if sp.decoder_kind == "unetplusplus":
            # reverse of the dense decoder (fp32 twin: _backward_unetpp): blocks in reverse forward order; a node's gradient is
            # the sum over its consumers — as the up-sampled input of the block to its right (accumulating 2x2 sums) and as a
            # slice of the concatenated skip of the blocks further right (accumulating slice copies), one rounding each
            G = {sp.decoder[-1].name: g}

            def slot(name, shape):
                t = G.get(name)
                if t is None:
                    t = G[name] = torch.empty(shape, dtype=bf, device=dev)
                    return t, 0
                return t, 1

            for blk in reversed(sp.decoder):
                d = S["P" + blk.name]
                g = G.pop(blk.name)
                self._tr(f"P{blk.name}.g", g)
                Hh, Ww = d["H"], d["W"]
                dy2 = bn_bwd(blk.conv2, g, None, d["y2"], virtual_act=True)
                self._tr(f"P{blk.name}.dy2", dy2)
                del g
                wgrad(blk.conv2, d["y1"], None, 0, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
                dz1 = torch.empty(d["y1"].shape, dtype=bf, device=dev)
                red1 = dgrad_bn(blk.conv2, dy2, Hh, Ww, dz1, blk.conv1, d["y1"])
                del dy2
                self._tr(f"P{blk.name}.dz1", dz1)
                dy1 = bn_bwd(blk.conv1, dz1, None, d["y1"], virtual_act=True, reduced=red1)
                self._tr(f"P{blk.name}.dy1", dy1)
                del dz1
                wgrad(blk.conv1, d["x"], d["skip"], 1, Hh, Ww, dy1)
                cx = blk.in_ch
                dup = torch.empty((B, Hh, Ww, cx), dtype=bf, device=dev)
                dskip = None
                if d["skip"] is not None:
                    dskip = torch.empty(d["skip"].shape, dtype=bf, device=dev)
                    dgrad(blk.conv1, dy1, Hh, Ww, dup, dskip, split=cx)
                    self._tr(f"P{blk.name}.dskip", dskip)
                else:
                    dgrad(blk.conv1, dy1, Hh, Ww, dup)
                self._tr(f"P{blk.name}.dup", dup)
                del dy1
                glow, acc = slot(blk.low, d["x"].shape)
                _lib.check(lib.dt_upsample2x_bwd_acc_bf16(_p(dup), _p(glow), acc, B, Hh // 2, Ww // 2, cx, st),
                           "dt_upsample2x_bwd_acc_bf16")
                del dup
                if dskip is not None:
                    Cw = dskip.shape[-1]
                    for name, (off, Cn) in zip(blk.cat, d["parts"]):
                        if len(blk.cat) == 1 and name not in G:
                            G[name] = dskip                      # the skip was the tensor itself: its gradient as is
                            continue
                        gm, acc = slot(name, (B, Hh, Ww, Cn))
                        _lib.check(lib.dt_channel_slice_bf16(_p(dskip), _p(gm), B * Hh * Ww, Cn, Cw, off, 0, acc, st),
                                   "dt_channel_slice_bf16")
                S["P" + blk.name] = None
            for k in range(1, 5):
                skip_grads[4 - k] = G[f"f{k}"]      # f_k of the decoder = feats[4 - k]
                self._tr(f"Pf{k}.g", G[f"f{k}"])
            g = G["f0"]
            self._tr("Pf0.g", g)
        for i in (range(4, -1, -1) if sp.decoder_kind == "resunet" else ()):
            # reverse of one ResUnet block (fp32 twin: _backward_resunet_block): g = gradient of the block output
            blk, d = sp.decoder[i], S[f"D{i}"]
            Hh, Ww = d["H"], d["W"]
            ic, cx = blk.idc, blk.in_ch
            sk = 0 if d["skip"] is None else d["skip"].shape[-1]
            n_pix = B * Hh * Ww
            wgrad(ic, d["x"], d["skip"], 1, Hh, Ww, g)                   # identity branch: dW over the virtual input
            cws = self._buf("chsum_ws", int(lib.dt_channel_sums_bf16_workspace(n_pix, ic.cout)), device=dev)
            _lib.check(lib.dt_channel_sums_bf16(_p(g), _p(cws), n_pix, ic.cout, _p(grads[ic.b_off:ic.b_off + ic.cout]), st),
                       "dt_channel_sums_bf16")                            # its bias gradient = sum g
            dy2 = bn_bwd(blk.conv2, g, None, d["y2"], virtual_act=True)   # main branch: both activations virtual
            self._tr(f"D{i}.dy2", dy2)
            wgrad(blk.conv2, d["y1"], None, 0, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
            dz1 = torch.empty(d["y1"].shape, dtype=bf, device=dev)
            red1 = dgrad_bn(blk.conv2, dy2, Hh, Ww, dz1, blk.conv1, d["y1"])
            del dy2
            self._tr(f"D{i}.dz1", dz1)
            dy1 = bn_bwd(blk.conv1, dz1, None, d["y1"], virtual_act=True, reduced=red1)
            self._tr(f"D{i}.dy1", dy1)
            del dz1
            wgrad(blk.conv1, d["x"], d["skip"], 1, Hh, Ww, dy1)
            dup = torch.empty((B, Hh, Ww, cx), dtype=bf, device=dev)
            dup_id = torch.empty((B, Hh, Ww, cx), dtype=bf, device=dev)
            one = self._const_vec(1.0, max(cx, sk, 8), dev)
            zero = self._const_vec(0.0, max(cx, sk, 8), dev)
            if sk:
                dskip = torch.empty(d["skip"].shape, dtype=bf, device=dev)
                dskip_id = torch.empty(d["skip"].shape, dtype=bf, device=dev)
                dgrad(blk.conv1, dy1, Hh, Ww, dup, dskip, split=cx)
                dgrad(ic, g, Hh, Ww, dup_id, dskip_id, split=cx)
                # gradient of the skip feature = the two branches' parts, one rounding
                _lib.check(lib.dt_bn_act_bf16(_p(dskip), 0, _p(one), _p(zero), _p(dskip_id), None, None, _p(dskip), n_pix, sk,
                                              0, st), "dt_bn_act_bf16")
                skip_grads[3 - i] = dskip
                self._tr(f"D{i}.dskip", dskip)
                del dskip_id
            else:
                dgrad(blk.conv1, dy1, Hh, Ww, dup)
                dgrad(ic, g, Hh, Ww, dup_id)
            del dy1
            _lib.check(lib.dt_bn_act_bf16(_p(dup), 0, _p(one), _p(zero), _p(dup_id), None, None, _p(dup), n_pix, cx, 0, st),
                       "dt_bn_act_bf16")
            del dup_id
            self._tr(f"D{i}.dup", dup)
            g = torch.empty(d["x"].shape, dtype=bf, device=dev)
            _lib.check(lib.dt_upsample2x_bwd_bf16(_p(dup), _p(g), B, Hh // 2, Ww // 2, cx, st), "dt_upsample2x_bwd_bf16")
            del dup
            self._tr(f"D{i}.g", g)
            S[f"D{i}"] = None
        for i in (range(4, -1, -1) if sp.decoder_kind == "unet" else ()):
            blk, d = sp.decoder[i], S[f"D{i}"]
            Hh, Ww = d["H"], d["W"]
            # the ReLU mask is recomputed from y2 * scale + shift even where z2 was stored (same arithmetic as bn_act:
            # identical mask, one tensor less to read in the reduce / apply passes)
            dy2 = bn_bwd(blk.conv2, g, None, d["y2"], virtual_act=True, reduced=g_red)
            self._tr(f"D{i}.dy2", dy2)
            if d.get("z1") is not None:
                wgrad(blk.conv2, d["z1"], None, 0, Hh, Ww, dy2)
            else:
                wgrad(blk.conv2, d["y1"], None, 0, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
            dz1 = torch.empty(d["y1"].shape, dtype=bf, device=dev)
            red1 = dgrad_bn(blk.conv2, dy2, Hh, Ww, dz1, blk.conv1, d["y1"])
            del dy2
            self._tr(f"D{i}.dz1", dz1)
            dy1 = bn_bwd(blk.conv1, dz1, None, d["y1"], virtual_act=True, reduced=red1)
            self._tr(f"D{i}.dy1", dy1)
            del dz1
            x_ss = self._ss(sp.decoder[i - 1].conv2, bnws) if d["x_virtual"] else None
            wgrad(blk.conv1, d["x"], d["skip"], 1, Hh, Ww, dy1, in_ss=x_ss)
            cx = blk.in_ch
            if d["skip"] is None and i >= 1:
                # dec4.conv1: data gradient, the 2x2 sums of the up-sampling's backward and the BatchNorm-backward sums of
                # the block below in one launch of the narrow kernel — no full-resolution gradient tensor
                c1 = blk.conv1
                ddesc = self._desc(B, Hh, Ww, c1.cout, 0, 0, Hh, Ww, c1.cin, c1.k, 1, c1.k - 1 - c1.pad, 0, 0)
                if lib.dt_conv2d_bf16_upsampled_dgrad_supported(C.byref(ddesc)):
                    pb = sp.decoder[i - 1].conv2
                    y2p = S[f"D{i - 1}"]["y2"]
                    P = lib.dt_conv2d_bf16_stat_rows(C.byref(ddesc))
                    red = self._buf("bn_red_up", lib.dt_bn_stats_floats(P, cx), device=dev)
                    psc, psh = self._ss(pb, bnws)
                    fuse = _lib.BnBwdFuse(_p(y2p), _p(bnws[pb.bn_off:pb.bn_off + cx]),
                                          _p(bnws[nb + pb.bn_off:nb + pb.bn_off + cx]), _p(psc), _p(psh))
                    g = torch.empty(d["x"].shape, dtype=bf, device=dev)
                    ev = self._pb()
                    _lib.check(lib.dt_conv2d_bf16_upsampled_dgrad(C.byref(ddesc), _p(dy1), _p(wbd[c1.w_off:c1.w_off + c1.w_size]),
                                                                  _p(g), _p(red), C.byref(fuse), st),
                               "dt_conv2d_bf16_upsampled_dgrad")
                    self._pe(ev, f"conv3x3_bf16_narrow_kernel<{c1.cout // 16}, {c1.cin // 16}, false, true, true>",
                             2.0 * 9 * c1.cin * c1.cout * Hh * Ww * B,
                             2.0 * B * (Hh * Ww * c1.cout + (Hh // 2) * (Ww // 2) * cx * 2))
                    g_red = (red, P)
                    del dy1
                    self._tr(f"D{i}.g", g)
                    S[f"D{i}"] = None
                    continue
            dup = torch.empty((B, Hh, Ww, cx), dtype=bf, device=dev)
            if d["skip"] is not None:
                dskip = torch.empty(d["skip"].shape, dtype=bf, device=dev)
                dgrad(blk.conv1, dy1, Hh, Ww, dup, dskip, split=cx)
                skip_grads[3 - i] = dskip
                self._tr(f"D{i}.dskip", dskip)
            else:
                dgrad(blk.conv1, dy1, Hh, Ww, dup)
            self._tr(f"D{i}.dup", dup)
            del dy1
            g = torch.empty(d["x"].shape, dtype=bf, device=dev)
            g_red = None
            if i >= 1:     # also where z2 was stored: the mask is recomputed from y2 * scale + shift either way
                pb = sp.decoder[i - 1].conv2
                y2p = S[f"D{i - 1}"]["y2"]
                P = lib.dt_upsample2x_bwd_bn_bf16_rows(B, Hh // 2, Ww // 2, cx)
                red = self._buf("bn_red_up", lib.dt_bn_stats_floats(P, cx), device=dev)
                psc, psh = self._ss(pb, bnws)
                fuse = _lib.BnBwdFuse(_p(y2p), _p(bnws[pb.bn_off:pb.bn_off + cx]),
                                      _p(bnws[nb + pb.bn_off:nb + pb.bn_off + cx]), _p(psc), _p(psh))
                _lib.check(lib.dt_upsample2x_bwd_bn_bf16(_p(dup), _p(g), C.byref(fuse), _p(red), B, Hh // 2, Ww // 2, cx,
                                                         st), "dt_upsample2x_bwd_bn_bf16")
                g_red = (red, P)
            else:
                _lib.check(lib.dt_upsample2x_bwd_bf16(_p(dup), _p(g), B, Hh // 2, Ww // 2, cx, st),
                           "dt_upsample2x_bwd_bf16")
            del dup
            self._tr(f"D{i}.g", g)
            S[f"D{i}"] = None
        self._bucket_done(sp.buckets[0])

        for li in (3, 2, 1, 0):
            blocks = sp.layers[li]
            for bi in range(len(blocks) - 1, -1, -1):
                blk, r = blocks[bi], S[f"L{li}B{bi}"]
                Hin, Win, Hh, Ww = r["Hin"], r["Win"], r["H"], r["W"]
                gin, gin_has = None, False
                if bi == 0 and li > 0 and skip_grads[li] is not None:
                    gin, gin_has = skip_grads[li], True
                if gin is None:
                    gin = torch.empty(r["x"].shape, dtype=bf, device=dev)
                if blk.down is None:
                    dy2 = bn_bwd(blk.conv2, g, r["out"], r["y2"], dres=gin, dres_acc=gin_has, reduced=g_red)
                    gin_has = True
                    dyd = None
                    self._tr(f"L{li}B{bi}.gres", gin)
                else:
                    gd = torch.empty(r["out"].shape, dtype=bf, device=dev)
                    dy2 = bn_bwd(blk.conv2, g, r["out"], r["y2"], dres=gd, reduced=g_red)
                    dyd = bn_bwd(blk.down, gd, None, r["yd"])
                    self._tr(f"L{li}B{bi}.gres", gd)
                    self._tr(f"L{li}B{bi}.dyd", dyd)
                    del gd
                self._tr(f"L{li}B{bi}.dy2", dy2)
                if r.get("z1") is not None:
                    wgrad(blk.conv2, r["z1"], None, 0, Hh, Ww, dy2)
                else:
                    wgrad(blk.conv2, r["y1"], None, 0, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
                dz1 = torch.empty(r["y1"].shape, dtype=bf, device=dev)
                red1 = dgrad_bn(blk.conv2, dy2, Hh, Ww, dz1, blk.conv1, r["y1"])
                del dy2
                self._tr(f"L{li}B{bi}.dz1", dz1)
                dy1 = bn_bwd(blk.conv1, dz1, None, r["y1"], virtual_act=True, reduced=red1)
                self._tr(f"L{li}B{bi}.dy1", dy1)
                del dz1
                wgrad(blk.conv1, r["x"], None, 0, Hin, Win, dy1)
                g_red = None
                if bi > 0 and blk.down is None and gin_has:
                    # last writer of block bi-1's output gradient: its bn2 reduction (mask: stored output) rides along
                    rp = S[f"L{li}B{bi - 1}"]
                    g_red = dgrad_bn(blk.conv1, dy1, Hin, Win, gin, blocks[bi - 1].conv2, rp["y2"], act=rp["out"])
                else:
                    dgrad(blk.conv1, dy1, Hin, Win, gin, acc=gin_has)
                gin_has = True
                del dy1
                self._tr(f"L{li}B{bi}.gin1", gin)
                if dyd is not None:
                    wgrad(blk.down, r["x"], None, 0, Hin, Win, dyd)
                    dgrad(blk.down, dyd, Hin, Win, gin, acc=True)
                    del dyd
                    self._tr(f"L{li}B{bi}.gin", gin)
                g = gin
                S[f"L{li}B{bi}"] = None
            if li > 0:
                self._bucket_done(sp.buckets[4 - li])

        pl, stem = S["pool"], S["stem"]
        gf1 = skip_grads[0]
        stem_red = None
        P = lib.dt_maxpool3x3s2_bwd_bn_bf16_rows(B, pl["H"], pl["W"], 64) if self._fuse_pool_bn else 0
        if P > 0:      # even maps: the stem's BatchNorm-backward sums ride in the pass that writes its activation gradient
            stq = sp.stem
            red = self._buf("bn_red_pool", lib.dt_bn_stats_floats(P, 64), device=dev)
            ssc, ssh = self._ss(stq, bnws)
            fuse = _lib.BnBwdFuse(_p(stem["y"]), _p(bnws[stq.bn_off:stq.bn_off + 64]),
                                  _p(bnws[nb + stq.bn_off:nb + stq.bn_off + 64]), _p(ssc), _p(ssh))
            _lib.check(lib.dt_maxpool3x3s2_bwd_bn_bf16(_p(g), _p(pl["amax"]), _p(gf1), 1, C.byref(fuse), _p(red), B, pl["H"],
                                                       pl["W"], 64, st), "dt_maxpool3x3s2_bwd_bn_bf16")
            stem_red = (red, P)
        else:
            _lib.check(lib.dt_maxpool3x3s2_bwd_bf16(_p(g), _p(pl["amax"]), _p(gf1), 1, B, pl["H"], pl["W"], 64, st),
                       "dt_maxpool3x3s2_bwd_bf16")
        self._tr("gf1", gf1)
        dy = bn_bwd(sp.stem, gf1, None, stem["y"], virtual_act=True, reduced=stem_red)
        self._tr("stem.dy", dy)
        stc = sp.stem
        if stem.get("s2d") is not None:
            # space-to-depth form on the bf16 MFMA kernels: dW over 16 taps x 16 channels, gathered back to 7x7
            cin = stem["x"].shape[-1]
            d4 = self._desc(B, dy.shape[1], dy.shape[2], 16, 0, 0, dy.shape[1], dy.shape[2], stc.cout, 4, 1, 2)
            nbytes = lib.dt_conv2d_wgrad_bf16_workspace(C.byref(d4))
            if nbytes == 0:
                raise RuntimeError(lib.dt_last_error().decode())
            ws = self._buf("wgrad_ws", nbytes // 4, device=dev)
            dw4 = self._buf("stem_dw4", 16 * 16 * stc.cout, device=dev)
            _lib.check(lib.dt_conv2d_wgrad_bf16(C.byref(d4), _p(stem["s2d"]), None, _p(dy), _p(dw4), _p(ws),
                                                ws.numel() * 4, None, None, st), "dt_conv2d_wgrad_bf16(stem)")
            _lib.check(lib.dt_stem_unpack_wgrad(_p(dw4), _p(grads[stc.w_off:stc.w_off + stc.w_size]), cin, stc.cout, st),
                       "dt_stem_unpack_wgrad")
        else:
            sdesc = self._desc(B, stem["Hin"], stem["Win"], stem["x"].shape[-1], 0, 0, dy.shape[1], dy.shape[2], stc.cout,
                               stc.k, stc.stride, stc.pad)
            nbytes = lib.dt_conv2d_wgrad_workspace(C.byref(sdesc))
            ws = self._buf("wgrad_ws", nbytes // 4, device=dev)
            _lib.check(lib.dt_conv2d_wgrad_stem_dy_bf16(C.byref(sdesc), _p(stem["x"]), _p(dy),
                                                        _p(grads[stc.w_off:stc.w_off + stc.w_size]), _p(ws),
                                                        ws.numel() * 4, st), "dt_conv2d_wgrad_stem_dy_bf16")
        self._join_side()
        if self.grad_hook:
            self.grad_hook(*sp.buckets[4])
        self.saved = None

    # ------------------------------------------------------------------ backward units
    def _bn_bwd(self, c: ConvSpec, params, grads, bnws, dout, out_act, y, dres=None, dres_acc=False,
                virtual_act=False, reduced=None):
        """virtual_act: the activation was never stored; its ReLU mask is recomputed from y*scale+shift.
        reduced = (red, P): the partial sums were already produced by the data-gradient kernel that wrote `dout`
        (`_dgrad_bn`), so the reduction pass over (dout, y) is skipped."""
        B, H, W, Cc = y.shape
        n_pix = B * H * W
        nb = self.spec.n_bn_channels
        mean = bnws[c.bn_off: c.bn_off + Cc]
        invstd = bnws[nb + c.bn_off: nb + c.bn_off + Cc]
        gamma = params[c.g_off:c.g_off + Cc]
        st = _stream()
        asc, ash = self._ss(c, bnws) if virtual_act else (None, None)
        if reduced is not None:
            red, P = reduced
        else:
            P = self.lib.dt_bn_bwd_rows(n_pix, Cc)
            red = self._buf("bn_red", self.lib.dt_bn_bwd_red_floats(n_pix, Cc), device=y.device)
            e0 = self._pb()
            _lib.check(self.lib.dt_bn_bwd_reduce(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(asc), _p(ash),
                                                 _p(red), n_pix, Cc, st), "dt_bn_bwd_reduce")
            self._pe(e0, "bn_bwd_reduce_kernel", 0.0, 4.0 * y.numel() * (2 + (out_act is not None)))
        dy = torch.empty_like(y)
        e0 = self._pb()
        # eval-mode (frozen) BatchNorm: y*scale+shift with constant statistics -> dy = g*gamma*invstd, no mean terms
        fn = self.lib.dt_bn_bwd_apply if self._bwd_training else self.lib.dt_bn_bwd_apply_frozen
        _lib.check(fn(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(gamma), _p(asc),
                      _p(ash), _p(red), P,
                      _p(grads[c.g_off:c.g_off + Cc]), _p(grads[c.b_off:c.b_off + Cc]),
                      _p(dy), _p(dres), 1 if dres_acc else 0, n_pix, Cc, st),
                   "dt_bn_bwd_apply")
        # dout + y (+ stored activation) read, dy written (+ residual-branch gradient written, or read-modify-written)
        self._pe(e0, "bn_bwd_apply_kernel", 0.0, 4.0 * y.numel() * (3 + (out_act is not None) + (dres is not None) * (2 if dres_acc else 1)))
        return dy

    def _wgrad(self, c: ConvSpec, grads, src0, src1, mode0, B, Hin, Win, dy, in_ss=None, side=True):
        if side and self.overlap_wgrad:
            self._on_side(lambda: self._wgrad(c, grads, src0, src1, mode0, B, Hin, Win, dy, in_ss, side=False),
                          src0, src1, dy)
            return
        Ho, Wo = dy.shape[1], dy.shape[2]
        C0 = src0.shape[-1]
        C1 = 0 if src1 is None else src1.shape[-1]
        desc = self._desc(B, Hin, Win, C0, C1, mode0, Ho, Wo, c.cout, c.k, c.stride, c.pad)
        e0 = self._pb()
        fl, nb = self._conv_work(desc) if e0 is not None else (0.0, 0.0)
        if self.winograd and self.lib.dt_conv2d_wgrad_winograd_supported(C.byref(desc)):
            # 3x3 stride-1 layers with 64-channel blocks: the Winograd form (conv_wino_wgrad.hip, 1.6-1.75x the direct one)
            nbytes = self.lib.dt_conv2d_wgrad_winograd_workspace(C.byref(desc))
            ws = self._buf("wgrad_ws", nbytes // 4, device=dy.device)
            _lib.check(self.lib.dt_conv2d_wgrad_winograd(C.byref(desc), _p(src0), _p(src1), _p(dy),
                                                         _p(grads[c.w_off:c.w_off + c.w_size]), _p(ws), ws.numel() * 4,
                                                         _p(in_ss[0]) if in_ss else None,
                                                         _p(in_ss[1]) if in_ss else None, _stream()),
                       "dt_conv2d_wgrad_winograd")
            self._pe(e0, "conv3x3_wino_wgrad_kernel (+ split-K reduce / final)", fl, nb)
            return
        nbytes = self.lib.dt_conv2d_wgrad_workspace(C.byref(desc))
        if nbytes == 0:
            raise RuntimeError(f"dt_conv2d_wgrad_workspace: {self.lib.dt_last_error().decode()}")
        ws = self._buf("wgrad_ws", nbytes // 4, device=dy.device)
        _lib.check(self.lib.dt_conv2d_wgrad(C.byref(desc), _p(src0), _p(src1), _p(dy),
                                            _p(grads[c.w_off:c.w_off + c.w_size]), _p(ws), ws.numel() * 4,
                                            _p(in_ss[0]) if in_ss else None, _p(in_ss[1]) if in_ss else None,
                                            _stream()), "dt_conv2d_wgrad")
        self._pe(e0, "conv_wgrad_stem_kernel (+ reduce)" if c is self.spec.stem else
                 ("conv_wgrad_n16_kernel (+ reduce)" if max(desc.C0 + desc.C1, desc.Cout) <= 32 and min(desc.C0 + desc.C1, desc.Cout) <= 16
                  else "conv_wgrad_kernel (+ split-K reduce)"), fl, nb)

    def _dgrad_bn(self, c: ConvSpec, dy, B, H, W, out0, bn_conv: ConvSpec, y, bnws, act=None):
        """stride-1 data gradient of conv `c` into out0 with the BatchNorm-backward reduction of `bn_conv` (the layer
        whose raw output `y` has out0's shape) fused into the epilogue -> (red, P) for _bn_bwd.  act None: plain store,
        virtual activation (mask from y); act = stored block output: the gradient is ADDED to out0 (join) and the
        sums are taken over the joined tensor."""
        Cc = bn_conv.cout
        assert c.stride == 1 and c.cin == Cc and tuple(y.shape) == tuple(out0.shape)
        wd = self._wd_all[c.w_off:c.w_off + c.w_size]
        desc = self._desc(B, H, W, c.cout, 0, 0, H, W, c.cin, c.k, 1, c.k - 1 - c.pad, 0, 0 if act is None else 1)
        ud = self._u(c, dgrad=True)
        wino = self._use_wino(desc, ud)
        P = self._stat_rows(desc, ud)
        red = self._buf("bn_red_fused", self.lib.dt_bn_stats_floats(P, Cc), device=dy.device)
        nb = self.spec.n_bn_channels
        asc, ash = self._ss(bn_conv, bnws) if act is None else (None, None)
        fuse = _lib.BnBwdFuse(_p(y), _p(bnws[bn_conv.bn_off: bn_conv.bn_off + Cc]),
                              _p(bnws[nb + bn_conv.bn_off: nb + bn_conv.bn_off + Cc]), _p(asc), _p(ash), _p(act))
        prof = self.profile
        if prof is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if wino:
            _lib.check(self.lib.dt_conv2d_winograd_bn_bwd(C.byref(desc), _p(dy), _p(ud), _p(out0), _p(red),
                                                          C.byref(fuse), _stream()), "dt_conv2d_winograd_bn_bwd")
        else:
            _lib.check(self.lib.dt_conv2d_bn_bwd(C.byref(desc), _p(dy), _p(wd), _p(out0), _p(red), C.byref(fuse),
                                                 _stream()), "dt_conv2d_bn_bwd")
        if prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            flops = 2.0 * desc.ksize ** 2 * desc.C0 * desc.Cout * desc.Ho * desc.Wo * desc.B
            nbytes = 4.0 * desc.B * desc.Ho * desc.Wo * (desc.C0 + (2 if act is None else 4) * desc.Cout) + \
                4.0 * desc.ksize ** 2 * desc.C0 * desc.Cout
            name = self._wino_kernel_name(False, 1 if act is None else 3) if wino else self._conv_kernel_name(desc, False)
            prof.append((name, flops, e0, e1, nbytes))
        return red, P

    def _upsampled_dgrad(self, blk, prev_conv: ConvSpec, params, bnws, dy1, y2p, d, B, Hh, Ww) -> bool:
        """decoder block without a skip: gradient of the block input (low resolution) straight from dy1 — the data
        gradient of conv1 and the backward of the nearest x2 upsample in one sub-pixel kernel, the BatchNorm-backward
        sums of the previous block's conv2 in its epilogue.  Fills d["g"], d["g_red"]; False where the layer shape is
        not covered (the generic chain runs)."""
        c = blk.conv1
        cx = blk.in_ch
        desc = self._desc(B, Hh, Ww, cx, 0, 1, Hh, Ww, c.cout, c.k, c.stride, c.pad, 0, 0)
        if not self.lib.dt_conv2d_upsampled_dgrad_supported(C.byref(desc)):
            return False
        lib = self.lib
        P = lib.dt_conv2d_upsampled_dgrad_rows(C.byref(desc))
        red = self._buf("bn_red_up", lib.dt_bn_stats_floats(P, cx), device=dy1.device)
        psc, psh = self._ss(prev_conv, bnws)
        nbq = self.spec.n_bn_channels
        fuse = _lib.BnBwdFuse(_p(y2p), _p(bnws[prev_conv.bn_off: prev_conv.bn_off + cx]),
                              _p(bnws[nbq + prev_conv.bn_off: nbq + prev_conv.bn_off + cx]), _p(psc), _p(psh))
        g = torch.empty_like(d["x"])
        ev = self._pb()
        _lib.check(lib.dt_conv2d_upsampled_dgrad(C.byref(desc), _p(dy1), _p(params[c.w_off:c.w_off + c.w_size]), _p(g),
                                                 _p(red), C.byref(fuse), _stream()), "dt_conv2d_upsampled_dgrad")
        self._pe(ev, "conv3x3_f32_upc_dgrad_kernel", 2.0 * 9 * cx * c.cout * Hh * Ww * B,
                 4.0 * B * Hh * Ww * c.cout + 4.0 * B * (Hh // 2) * (Ww // 2) * cx * 2)
        d["g"], d["g_red"] = g, (red, P)
        return True

    def _dgrad(self, c: ConvSpec, params, dy, B, Hin, Win, out0, out1=None, split=0, acc=False):
        """gradient wrt the conv's logical input [B,Hin,Win,cin] (before virtual upsample handling)."""
        Ho, Wo = dy.shape[1], dy.shape[2]
        wd = self._wd_all[c.w_off:c.w_off + c.w_size]     # flipped / transposed image, built at the start of backward
        pad = c.k - 1 - c.pad
        if c.stride == 1:
            desc = self._desc(B, Ho, Wo, c.cout, 0, 0, Hin, Win, c.cin, c.k, 1, pad, split, 1 if acc else 0)
        else:
            assert Hin == 2 * Ho and Win == 2 * Wo
            desc = self._desc(B, Hin, Win, c.cout, 0, 2, Hin, Win, c.cin, c.k, 1, pad, split, 1 if acc else 0)
        self._conv(desc, dy, None, wd, out0, out1, None, u=self._u(c, dgrad=True) if c.stride == 1 else None)

    def _backward_unetpp(self, S, g_head, params, grads, bnws, B, skip_grads):
        """reverse of _forward_unetpp: the blocks in reverse forward order (every consumer of a node comes before the
        node); a node's gradient is the sum over its consumers — as the upsampled input of the block to its right
        (dt_upsample2x_bwd, accumulating) and as a slice of the concatenated skip of the blocks further right
        (dt_channel_slice, accumulating).  Fills skip_grads (gradients of f1..f4) and returns the gradient of f5."""
        sp, lib, st = self.spec, self.lib, _stream()
        G = {sp.decoder[-1].name: g_head}

        def slot(name, shape, dev):
            t = G.get(name)
            if t is None:
                t = G[name] = torch.empty(shape, dtype=torch.float32, device=dev)
                return t, 0
            return t, 1

        for blk in reversed(sp.decoder):
            d = S["P" + blk.name]
            g = G.pop(blk.name)
            dev = g.device
            Hh, Ww = d["H"], d["W"]
            dy2 = self._bn_bwd(blk.conv2, params, grads, bnws, g, None, d["y2"], virtual_act=True)
            del g
            self._wgrad(blk.conv2, grads, d["y1"], None, 0, B, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
            dz1 = torch.empty_like(d["y1"])
            if self._fuse_bn:
                red1 = self._dgrad_bn(blk.conv2, dy2, B, Hh, Ww, dz1, blk.conv1, d["y1"], bnws)
            else:
                red1 = self._dgrad(blk.conv2, params, dy2, B, Hh, Ww, dz1)
            del dy2
            dy1 = self._bn_bwd(blk.conv1, params, grads, bnws, dz1, None, d["y1"], virtual_act=True, reduced=red1)
            del dz1
            self._wgrad(blk.conv1, grads, d["x"], d["skip"], 1, B, Hh, Ww, dy1)
            cx = blk.in_ch
            dup = torch.empty((B, Hh, Ww, cx), dtype=torch.float32, device=dev)
            dskip = None
            if d["skip"] is not None:
                dskip = torch.empty_like(d["skip"])
                self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup, dskip, split=cx)
            else:
                self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup)
            del dy1
            glow, acc = slot(blk.low, d["x"].shape, dev)
            _lib.check(lib.dt_upsample2x_bwd(_p(dup), _p(glow), acc, B, Hh // 2, Ww // 2, cx, st), "dt_upsample2x_bwd")
            del dup
            if dskip is not None:
                Cw = dskip.shape[-1]
                for name, (off, Cn) in zip(blk.cat, d["parts"]):
                    if len(blk.cat) == 1 and name not in G:
                        G[name] = dskip                      # the skip was the tensor itself: its gradient as is
                        continue
                    gm, acc = slot(name, (B, Hh, Ww, Cn), dev)
                    _lib.check(lib.dt_channel_slice(_p(dskip), _p(gm), B * Hh * Ww, Cn, Cw, off, 0, acc, st), "dt_channel_slice")
            S["P" + blk.name] = None
        for k in range(1, 5):
            skip_grads[4 - k] = G[f"f{k}"]      # f_k of the decoder = feats[4 - k]
        return G["f0"]

    def _backward_resunet_block(self, blk, d, g, params, grads, bnws, B, Hh, Ww, skip_grads, skip_slot):
        """reverse of one ResUnet decoder block (forward: see the decoder loop): g = gradient of the block output
        [B,Hh,Ww,cout] -> returns the gradient of the block's low-resolution input; writes the skip gradient."""
        lib, dev, st = self.lib, g.device, _stream()
        ic, cx = blk.idc, blk.in_ch
        sk = 0 if d["skip"] is None else d["skip"].shape[-1]
        n_pix = B * Hh * Ww
        # identity branch: weight gradient over the virtual (up-sampled + concatenated) input, bias gradient = sum g
        self._wgrad(ic, grads, d["x"], d["skip"], 1, B, Hh, Ww, g)
        ws = self._buf("chsum_ws", int(lib.dt_channel_sums_workspace(n_pix, ic.cout)), device=dev)
        _lib.check(lib.dt_channel_sums(_p(g), _p(ws), n_pix, ic.cout, _p(grads[ic.b_off:ic.b_off + ic.cout]), st),
                   "dt_channel_sums")
        # main branch: relu(bn2(conv2(relu(bn1(conv1(xin))))))  (both activations virtual: masks from y*scale+shift)
        dy2 = self._bn_bwd(blk.conv2, params, grads, bnws, g, None, d["y2"], virtual_act=True)
        self._wgrad(blk.conv2, grads, d["y1"], None, 0, B, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
        dz1 = torch.empty_like(d["y1"])
        red1 = self._dgrad_bn(blk.conv2, dy2, B, Hh, Ww, dz1, blk.conv1, d["y1"], bnws) if self._fuse_bn else \
            self._dgrad(blk.conv2, params, dy2, B, Hh, Ww, dz1)
        del dy2
        dy1 = self._bn_bwd(blk.conv1, params, grads, bnws, dz1, None, d["y1"], virtual_act=True, reduced=red1)
        del dz1
        self._wgrad(blk.conv1, grads, d["x"], d["skip"], 1, B, Hh, Ww, dy1)
        dup = torch.empty((B, Hh, Ww, cx), dtype=torch.float32, device=dev)
        dup_id = torch.empty_like(dup)
        if sk:
            dskip, dskip_id = torch.empty_like(d["skip"]), torch.empty_like(d["skip"])
            self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup, dskip, split=cx)
            self._dgrad(ic, params, g, B, Hh, Ww, dup_id, dskip_id, split=cx)
            # dskip += dskip_id (the split data-gradient kernels accumulate into their first output only)
            one, zero = self._const_vec(1.0, sk, dev), self._const_vec(0.0, sk, dev)
            self._bn_act(dskip, (one, zero), res=dskip_id, relu=0, out=dskip)
            skip_grads[skip_slot] = dskip
            del dskip_id
        else:
            self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup)
            self._dgrad(ic, params, g, B, Hh, Ww, dup_id)
        del dy1
        gx = torch.empty_like(d["x"])
        _lib.check(lib.dt_upsample2x_bwd(_p(dup), _p(gx), 0, B, Hh // 2, Ww // 2, cx, st), "dt_upsample2x_bwd")
        _lib.check(lib.dt_upsample2x_bwd(_p(dup_id), _p(gx), 1, B, Hh // 2, Ww // 2, cx, st), "dt_upsample2x_bwd")
        return gx

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits: torch.Tensor, params: torch.Tensor, grads: torch.Tensor, saved: Optional[_Saved] = None):
        """Hand-scheduled reverse pass.  Writes every parameter gradient into ``grads`` (flat, same layout
        as ``params``) and calls ``grad_hook(name, lo, hi)`` as each bucket of the flat buffer completes.
        ``saved``: the activations of the forward pass this gradient belongs to (default: the engine's last one)."""
        sp, lib = self.spec, self.lib
        sv = saved if saved is not None else self.saved
        if sv is None:
            raise RuntimeError("backward called without a saved forward (was another forward run in between?)")
        S = sv.d
        self._bwd_training = bool(S.get("training", True))
        B = S["B"]
        bnws = S["bnws"]
        dev = dlogits.device
        st = _stream()
        dlogits = dlogits.contiguous()
        # data-gradient weight images of every layer ([tap'][co][ci], taps reversed) in one launch
        self._wd_all = self._buf("wd_all", sp.n_params, device=dev)
        self._weight_images(params, self._wd_all, 0)
        self._ud_all = self._wino_images(self._wd_all, "wino_ud", True) if self.winograd else None

        # ---- head
        hd = sp.head
        h = S["head"]
        H, W = h["H"], h["W"]
        K = hd.cout
        g = torch.empty_like(h["x"])
        P = lib.dt_head_bwd_rows(B, H, W)
        red = self._buf("head_red", lib.dt_head_bwd_red_floats(B, H, W, hd.cin, K), device=dev)
        wh = params[hd.w_off:hd.w_off + hd.w_size]
        e0 = self._pb()
        _lib.check(lib.dt_head_bwd(_p(h["x"]), _p(wh), _p(dlogits), _p(g), _p(red), B, H, W, hd.cin, K, st),
                   "dt_head_bwd")
        self._pe(e0, "head_bwd_kernel", 4.0 * 9 * hd.cin * K * B * H * W, 4.0 * B * H * W * (2 * hd.cin + K))
        _lib.check(lib.dt_head_bwd_finalize(_p(red), P, _p(grads[hd.w_off:hd.w_off + hd.w_size]),
                                            _p(grads[hd.b_off:hd.b_off + K]), hd.cin, K, st), "dt_head_bwd_finalize")

        # ---- decoder (reverse)
        skip_grads = [None] * 5  # gradient of feats[0..4] = f1..f5
        g_red = None             # BatchNorm-backward partial sums that already came with g (fused producers)
        if sp.decoder_kind == "resunet" and hd.sd_k == 1:
            # the 1x1 head lives in the centre tap of the 3x3 head kernel: the other taps stay zero
            gw = grads[hd.w_off:hd.w_off + hd.w_size].view(K, 9, hd.cin)
            gw[:, :4].zero_()
            gw[:, 5:].zero_()
        if sp.decoder_kind == "unetplusplus":
            g = self._backward_unetpp(S, g, params, grads, bnws, B, skip_grads)
        for i in (range(4, -1, -1) if sp.decoder_kind != "unetplusplus" else ()):
            blk = sp.decoder[i]
            d = S[f"D{i}"]
            Hh, Ww = d["H"], d["W"]
            if sp.decoder_kind == "resunet":
                g = self._backward_resunet_block(blk, d, g, params, grads, bnws, B, Hh, Ww, skip_grads, 3 - i)
                S[f"D{i}"] = None
                continue
            # conv2 + BN + ReLU (activation stored only for the last block)
            # mask recomputed from y2 * scale + shift even where z2 was stored (identical to bn_act's; one read less)
            dy2 = self._bn_bwd(blk.conv2, params, grads, bnws, g, None, d["y2"], virtual_act=True, reduced=g_red)
            if d.get("z1") is not None:
                self._wgrad(blk.conv2, grads, d["z1"], None, 0, B, Hh, Ww, dy2)
            else:
                self._wgrad(blk.conv2, grads, d["y1"], None, 0, B, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
            dz1 = torch.empty_like(d["y1"])
            if self._fuse_bn:
                red1 = self._dgrad_bn(blk.conv2, dy2, B, Hh, Ww, dz1, blk.conv1, d["y1"], bnws)
            else:
                red1 = self._dgrad(blk.conv2, params, dy2, B, Hh, Ww, dz1)
            del dy2
            dy1 = self._bn_bwd(blk.conv1, params, grads, bnws, dz1, None, d["y1"], virtual_act=True, reduced=red1)
            del dz1
            x_ss = self._ss(sp.decoder[i - 1].conv2, bnws) if d["x_virtual"] else None
            self._wgrad(blk.conv1, grads, d["x"], d["skip"], 1, B, Hh, Ww, dy1, in_ss=x_ss)
            cx = blk.in_ch
            if d["skip"] is None and i >= 1 and self._fuse_bn and self._upsampled_dgrad(blk, sp.decoder[i - 1].conv2, params,
                                                                                       bnws, dy1, S[f"D{i - 1}"]["y2"], d, B, Hh, Ww):
                g, g_red = d["g"], d["g_red"]
                del dy1
                S[f"D{i}"] = None
                continue
            if d["skip"] is not None and i >= 1 and self._fuse_bn and self.winograd:
                # decoder blocks 1-3: the Winograd data gradient with the up-sampling's backward (2x2 sums) and the
                # BatchNorm-backward sums of the block below in its epilogue; the skip's gradient from a second launch
                c1 = blk.conv1
                ddesc = self._desc(B, Hh, Ww, c1.cout, 0, 0, Hh, Ww, c1.cin, c1.k, 1, c1.k - 1 - c1.pad, cx, 0)
                ud = self._u(c1, dgrad=True)
                if ud is not None and lib.dt_conv2d_winograd_upsampled_dgrad_supported(C.byref(ddesc)):
                    pb = sp.decoder[i - 1].conv2
                    y2p = S[f"D{i - 1}"]["y2"]
                    P = lib.dt_conv2d_winograd_upsampled_dgrad_rows(C.byref(ddesc))
                    red = self._buf("bn_red_up", lib.dt_bn_stats_floats(P, cx), device=dev)
                    psc, psh = self._ss(pb, bnws)
                    nbq = sp.n_bn_channels
                    fuse = _lib.BnBwdFuse(_p(y2p), _p(bnws[pb.bn_off: pb.bn_off + cx]),
                                          _p(bnws[nbq + pb.bn_off: nbq + pb.bn_off + cx]), _p(psc), _p(psh))
                    dskip = torch.empty_like(d["skip"])
                    g = torch.empty_like(d["x"])
                    ev = self._pb()
                    _lib.check(lib.dt_conv2d_winograd_upsampled_dgrad(C.byref(ddesc), _p(dy1), _p(ud), _p(g), _p(dskip), _p(red),
                                                                      C.byref(fuse), 3, st), "dt_conv2d_winograd_upsampled_dgrad")
                    self._pe(ev, self._wino_kernel_name(False, 6), 2.0 * 9 * c1.cin * c1.cout * Hh * Ww * B,
                             4.0 * B * Hh * Ww * (c1.cout + (c1.cin - cx)) + 4.0 * B * (Hh // 2) * (Ww // 2) * cx * 2)
                    skip_grads[3 - i] = dskip
                    g_red = (red, P)
                    del dy1
                    S[f"D{i}"] = None
                    continue
            dup = torch.empty((B, Hh, Ww, cx), dtype=torch.float32, device=dev)
            if d["skip"] is not None:
                dskip = torch.empty_like(d["skip"])
                self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup, dskip, split=cx)
                skip_grads[3 - i] = dskip
            else:
                self._dgrad(blk.conv1, params, dy1, B, Hh, Ww, dup)
            del dy1
            g = torch.empty_like(d["x"])
            g_red = None
            if i >= 1 and self._fuse_bn:     # also where z2 was stored (DT_MATERIALIZE_Z2): the mask is recomputed from y2
                # g is the gradient of relu(bn(y2)) of decoder block i-1 (never stored): its BatchNorm-backward
                # reduction rides along in the pass that writes g
                pb = sp.decoder[i - 1].conv2
                y2p = S[f"D{i - 1}"]["y2"]
                P = lib.dt_upsample2x_bwd_bn_rows(B, Hh // 2, Ww // 2, cx)
                red = self._buf("bn_red_up", lib.dt_bn_stats_floats(P, cx), device=dev)
                psc, psh = self._ss(pb, bnws)
                nbq = sp.n_bn_channels
                fuse = _lib.BnBwdFuse(_p(y2p), _p(bnws[pb.bn_off: pb.bn_off + cx]),
                                      _p(bnws[nbq + pb.bn_off: nbq + pb.bn_off + cx]), _p(psc), _p(psh))
                _lib.check(lib.dt_upsample2x_bwd_bn(_p(dup), _p(g), C.byref(fuse), _p(red), B, Hh // 2, Ww // 2, cx, st),
                           "dt_upsample2x_bwd_bn")
                g_red = (red, P)
            else:
                _lib.check(lib.dt_upsample2x_bwd(_p(dup), _p(g), 0, B, Hh // 2, Ww // 2, cx, st), "dt_upsample2x_bwd")
            del dup
            S[f"D{i}"] = None
        self._bucket_done(sp.buckets[0])

        # g = gradient wrt f5 ; encoder layers in reverse
        for li in (3, 2, 1, 0):
            blocks = sp.layers[li]
            for bi in range(len(blocks) - 1, -1, -1):
                blk = blocks[bi]
                r = S[f"L{li}B{bi}"]
                Hin, Win, Hh, Ww = r["Hin"], r["Win"], r["H"], r["W"]
                # gradient buffer of the block input; a decoder skip gradient may already live there
                gin = None
                gin_has = False
                if bi == 0 and li > 0 and skip_grads[li] is not None:
                    gin, gin_has = skip_grads[li], True   # block input of layer(li+1).0 is f_{li+1} = feats[li]
                if gin is None:
                    gin = torch.empty_like(r["x"])
                if blk.down is None:
                    dy2 = self._bn_bwd(blk.conv2, params, grads, bnws, g, r["out"], r["y2"], dres=gin,
                                       dres_acc=gin_has, reduced=g_red)
                    gin_has = True
                    dyd = None
                else:
                    gd = torch.empty_like(r["out"])
                    dy2 = self._bn_bwd(blk.conv2, params, grads, bnws, g, r["out"], r["y2"], dres=gd, reduced=g_red)
                    dyd = self._bn_bwd(blk.down, params, grads, bnws, gd, None, r["yd"])
                    del gd
                if r.get("z1") is not None:
                    self._wgrad(blk.conv2, grads, r["z1"], None, 0, B, Hh, Ww, dy2)
                else:
                    self._wgrad(blk.conv2, grads, r["y1"], None, 0, B, Hh, Ww, dy2, in_ss=self._ss(blk.conv1, bnws))
                dz1 = torch.empty_like(r["y1"])
                if self._fuse_bn:
                    red1 = self._dgrad_bn(blk.conv2, dy2, B, Hh, Ww, dz1, blk.conv1, r["y1"], bnws)
                else:
                    red1 = self._dgrad(blk.conv2, params, dy2, B, Hh, Ww, dz1)
                del dy2
                dy1 = self._bn_bwd(blk.conv1, params, grads, bnws, dz1, None, r["y1"], virtual_act=True, reduced=red1)
                del dz1
                self._wgrad(blk.conv1, grads, r["x"], None, 0, B, Hin, Win, dy1)
                g_red = None
                if self.fuse_join_fp32 and bi > 0 and blk.down is None and gin_has:
                    # gin becomes the output gradient of block bi-1: this join is its last writer, so the
                    # BatchNorm-backward sums of that block's bn2 (mask: its stored output) ride along
                    rp = S[f"L{li}B{bi - 1}"]
                    g_red = self._dgrad_bn(blk.conv1, dy1, B, Hin, Win, gin, blocks[bi - 1].conv2, rp["y2"], bnws,
                                           act=rp["out"])
                else:
                    self._dgrad(blk.conv1, params, dy1, B, Hin, Win, gin, acc=gin_has)
                gin_has = True
                del dy1
                if dyd is not None:
                    self._wgrad(blk.down, grads, r["x"], None, 0, B, Hin, Win, dyd)
                    self._dgrad(blk.down, params, dyd, B, Hin, Win, gin, acc=True)
                    del dyd
                g = gin
                S[f"L{li}B{bi}"] = None
            if li > 0:
                self._bucket_done(sp.buckets[4 - li])

        # ---- maxpool + stem
        pl = S["pool"]
        stem = S["stem"]
        gf1 = skip_grads[0]
        stem_red = None
        P = lib.dt_maxpool3x3s2_bwd_bn_rows(B, pl["H"], pl["W"], 64) if (self._fuse_bn and self._fuse_pool_bn) else 0
        if P > 0:      # even maps: the stem's BatchNorm-backward sums ride in the pass that writes its activation gradient
            stc, nbq = sp.stem, sp.n_bn_channels
            red = self._buf("bn_red_pool", lib.dt_bn_stats_floats(P, 64), device=dev)
            ssc, ssh = self._ss(stc, bnws)
            fuse = _lib.BnBwdFuse(_p(stem["y"]), _p(bnws[stc.bn_off:stc.bn_off + 64]),
                                  _p(bnws[nbq + stc.bn_off:nbq + stc.bn_off + 64]), _p(ssc), _p(ssh))
            _lib.check(lib.dt_maxpool3x3s2_bwd_bn(_p(g), _p(pl["amax"]), _p(gf1), 1, C.byref(fuse), _p(red), B, pl["H"], pl["W"],
                                                  64, st), "dt_maxpool3x3s2_bwd_bn")
            stem_red = (red, P)
        else:
            _lib.check(lib.dt_maxpool3x3s2_bwd(_p(g), _p(pl["amax"]), _p(gf1), 1, B, pl["H"], pl["W"], 64, st),
                       "dt_maxpool3x3s2_bwd")
        dy = self._bn_bwd(sp.stem, params, grads, bnws, gf1, None, stem["y"], virtual_act=True, reduced=stem_red)
        self._wgrad(sp.stem, grads, stem["x"], None, 0, B, stem["Hin"], stem["Win"], dy)
        self._join_side()
        if self.grad_hook:
            self.grad_hook(*sp.buckets[4])
        self.saved = None


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, module):
        eng = module.engine
        if module.precision == "bf16" and module.training:
            logits = eng.forward_bf16_train(x, flat.detach(), module.bn_state)
        else:
            logits, _ = eng.forward(x, flat.detach(), module.bn_state, module.training, save=True)
        # the activations belong to THIS autograd node, not to the engine: another grad-enabled forward (a validation
        # step, a second loss term) between this forward and its backward must not replace them
        ctx.saved_acts, eng.saved = eng.saved, None
        ctx.module = module
        module._bn_tracked_inc()
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        grads = m._grad_buffer()
        sv, ctx.saved_acts = ctx.saved_acts, None
        if sv is None:
            raise RuntimeError("UNetHIP: backward through the same forward twice (activations are freed after use)")
        if sv.d.get("bf16"):
            m.engine.backward_bf16(dlogits, m.flat_params.detach(), grads, saved=sv)
        else:
            m.engine.backward(dlogits, m.flat_params.detach(), grads, saved=sv)
        # a trainer that consumes the flat buffer directly (HipTrainer) opts out of autograd's copy into .grad
        return None, (grads if m.deliver_grad_to_autograd else None), None


class UNetHIP(nn.Module):
    """Drop-in for ``smp.Unet("resnet34", encoder_depth=5, decoder_channels=(256,128,64,32,16),
    encoder_weights=None, in_channels=C, classes=K)`` on MI355X."""

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights=None,
                 decoder_channels=(256, 128, 64, 32, 16), in_channels: int = 3, classes: int = 2,
                 decoder: str = "unet", decoder_use_batchnorm=True, decoder_attention_type=None, **unused):
        """decoder "unet": smp.Unet; "resunet": the reference's in-tree ResUnet (network/extra/resunet/model.py:57-103 —
        residual decoder blocks with a 1x1 identity_conv, 1x1 segmentation head); "unetplusplus": smp.UnetPlusPlus (dense
        nested decoder x_{depth}_{layer}, 3x3 head).  The two alternatives run on the fp32 path."""
        super().__init__()
        if decoder_use_batchnorm is not True or decoder_attention_type is not None:
            raise NotImplementedError("only decoder_use_batchnorm=True / decoder_attention_type=None have HIP kernels")
        if encoder_name != "resnet34":
            raise NotImplementedError(f"encoder {encoder_name!r}: only resnet34 has HIP kernels")
        if encoder_depth != 5 or tuple(decoder_channels) != (256, 128, 64, 32, 16):
            raise NotImplementedError("only encoder_depth=5 / decoder_channels=(256,128,64,32,16)")
        if encoder_weights is not None:
            raise NotImplementedError("pretrained encoder weights need a network fetch; load a state_dict instead")
        self.spec = build_spec(in_channels, classes, decoder)
        self.flat_params = nn.Parameter(torch.zeros(self.spec.n_params, dtype=torch.float32))
        self.register_buffer("bn_state", torch.zeros(2 * self.spec.n_bn_channels, dtype=torch.float32),
                             persistent=False)
        self.register_buffer("num_batches_tracked", torch.zeros(len(self.spec.convs), dtype=torch.int64),
                             persistent=False)
        self._engine: Optional[UNetEngine] = None
        self._grads: Optional[torch.Tensor] = None
        self.deliver_grad_to_autograd = True
        # "fp32" (BASELINE configs[1]) or "bf16": bf16 activations/weights, fp32 accumulation, fp32 master
        # parameters and optimiser (configs[2]; the AMP setting of the reference's protocol.md:27)
        self.precision = "fp32"
        self.reset_parameters()

    # ------------------------------------------------------------------ init / state_dict
    def reset_parameters(self, seed: Optional[int] = None):
        """Kaiming-normal conv weights (fan_in, gain sqrt 2), zero biases, BN gamma 1 / beta 0 — what the
        reference ends with when ``encoder_weights is None`` (segmodel.py:87-89,432-438)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        sd = {}
        for c in self.spec.convs:
            k = c.state_k
            fan_in = c.cin * k * k
            sd[c.key] = torch.randn((c.cout, c.cin, k, k), generator=g) * (2.0 / fan_in) ** 0.5
            if c.bn_key is not None:
                sd[f"{c.bn_key}.weight"] = torch.ones(c.cout)
                sd[f"{c.bn_key}.bias"] = torch.zeros(c.cout)
                sd[f"{c.bn_key}.running_mean"] = torch.zeros(c.cout)
                sd[f"{c.bn_key}.running_var"] = torch.ones(c.cout)
                sd[f"{c.bn_key}.num_batches_tracked"] = torch.tensor(0)
            else:
                sd[c.key.replace(".weight", ".bias")] = torch.zeros(c.cout)
        self.load_smp_state_dict(sd)

    @torch.no_grad()
    def load_smp_state_dict(self, sd, strict: bool = True):
        """smp/torch layout (OIHW conv weights) -> flat HWIO buffer."""
        flat = torch.zeros(self.spec.n_params, dtype=torch.float32)
        bn = torch.zeros(2 * self.spec.n_bn_channels, dtype=torch.float32)
        nbt = torch.zeros(len(self.spec.convs), dtype=torch.int64)
        missing = []

        def get(k, shape):
            if k not in sd:
                missing.append(k)
                return None
            t = sd[k].detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shape)}")
            return t

        for c in self.spec.convs:
            w = get(c.key, (c.cout, c.cin, c.state_k, c.state_k))
            if w is not None:
                if c.state_k != c.k:     # 1x1 head held as the centre tap of the 3x3 head kernel
                    full = torch.zeros((c.cout, c.cin, c.k, c.k), dtype=torch.float32)
                    full[:, :, c.k // 2, c.k // 2] = w[:, :, 0, 0]
                    w = full
                if c.layout == "hwio":
                    flat[c.w_off:c.w_off + c.w_size] = w.permute(2, 3, 1, 0).reshape(-1)
                else:
                    flat[c.w_off:c.w_off + c.w_size] = w.permute(0, 2, 3, 1).reshape(-1)   # head: OHWI
            if c.bn_key is not None:
                for name, off in (("weight", c.g_off), ("bias", c.b_off)):
                    t = get(f"{c.bn_key}.{name}", (c.cout,))
                    if t is not None:
                        flat[off:off + c.cout] = t
                rm = get(f"{c.bn_key}.running_mean", (c.cout,))
                rv = get(f"{c.bn_key}.running_var", (c.cout,))
                if rm is not None:
                    bn[2 * c.bn_off:2 * c.bn_off + c.cout] = rm
                if rv is not None:
                    bn[2 * c.bn_off + c.cout:2 * c.bn_off + 2 * c.cout] = rv
                k = f"{c.bn_key}.num_batches_tracked"
                if k in sd:
                    nbt[c.index] = int(sd[k])
            else:
                t = get(c.key.replace(".weight", ".bias"), (c.cout,))
                if t is not None:
                    flat[c.b_off:c.b_off + c.cout] = t
        if strict and missing:
            raise RuntimeError(f"missing keys in state_dict: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        self.flat_params.data.copy_(flat.to(self.flat_params.device))
        if self._engine is not None:
            self._engine.mark_weights_changed()
        self.bn_state.copy_(bn.to(self.bn_state.device))
        self.num_batches_tracked.copy_(nbt.to(self.num_batches_tracked.device))
        return missing

    def smp_state_dict(self, prefix: str = ""):
        """flat HWIO buffer -> smp/torch-named tensors (what ``smp.Unet.state_dict()`` would hold)."""
        flat = self.flat_params.detach().cpu()
        bn = self.bn_state.detach().cpu()
        nbt = self.num_batches_tracked.cpu()
        out = {}
        for c in self.spec.convs:
            w = flat[c.w_off:c.w_off + c.w_size]
            if c.bn_key is not None:
                out[prefix + c.key] = w.reshape(c.k, c.k, c.cin, c.cout).permute(3, 2, 0, 1).contiguous()
                out[prefix + f"{c.bn_key}.weight"] = flat[c.g_off:c.g_off + c.cout].clone()
                out[prefix + f"{c.bn_key}.bias"] = flat[c.b_off:c.b_off + c.cout].clone()
                out[prefix + f"{c.bn_key}.running_mean"] = bn[2 * c.bn_off:2 * c.bn_off + c.cout].clone()
                out[prefix + f"{c.bn_key}.running_var"] = bn[2 * c.bn_off + c.cout:2 * c.bn_off + 2 * c.cout].clone()
                out[prefix + f"{c.bn_key}.num_batches_tracked"] = nbt[c.index].clone()
            else:
                out[prefix + c.key] = self._oihw(w, c)
                out[prefix + c.key.replace(".weight", ".bias")] = flat[c.b_off:c.b_off + c.cout].clone()
        return out

    @staticmethod
    def _oihw(w_flat: torch.Tensor, c: ConvSpec) -> torch.Tensor:
        """flat-buffer weight of a convolution without BatchNorm (head, identity_conv) -> torch OIHW, state_dict size"""
        if c.layout == "hwio":
            w = w_flat.reshape(c.k, c.k, c.cin, c.cout).permute(3, 2, 0, 1)
        else:
            w = w_flat.reshape(c.cout, c.k, c.k, c.cin).permute(0, 3, 1, 2)
        if c.state_k != c.k:
            w = w[:, :, c.k // 2:c.k // 2 + 1, c.k // 2:c.k // 2 + 1]
        return w.contiguous()

    # nn.Module protocol: expose smp keys so Lightning checkpoints stay interchangeable with the reference
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination.update(self.smp_state_dict(prefix))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        sub = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        try:
            miss = self.load_smp_state_dict(sub, strict=False)
            missing_keys.extend(prefix + m for m in miss)
        except RuntimeError as e:  # size mismatch
            error_msgs.append(str(e))

    def smp_grad_dict(self):
        """parameter gradients under smp names / OIHW layout (parity tests, debugging)."""
        g = self._grad_buffer().detach().cpu()
        out = {}
        for c in self.spec.convs:
            w = g[c.w_off:c.w_off + c.w_size]
            if c.bn_key is not None:
                out[c.key] = w.reshape(c.k, c.k, c.cin, c.cout).permute(3, 2, 0, 1).contiguous()
                out[f"{c.bn_key}.weight"] = g[c.g_off:c.g_off + c.cout].clone()
                out[f"{c.bn_key}.bias"] = g[c.b_off:c.b_off + c.cout].clone()
            else:
                out[c.key] = self._oihw(w, c)
                out[c.key.replace(".weight", ".bias")] = g[c.b_off:c.b_off + c.cout].clone()
        return out

    # ------------------------------------------------------------------ execution
    @property
    def engine(self) -> UNetEngine:
        if self._engine is None:
            self._engine = UNetEngine(self.spec)
        return self._engine

    def _grad_buffer(self) -> torch.Tensor:
        if self._grads is None or self._grads.device != self.flat_params.device:
            self._grads = torch.zeros_like(self.flat_params.data)
        return self._grads

    def _bn_tracked_inc(self):
        if self.training:
            self.num_batches_tracked += 1

    def _require_gpu(self, x):
        if not x.is_cuda:
            raise RuntimeError("deadtrees_amd.UNetHIP runs only on an MI355X (HIP) device; there is no CPU fallback")
        if self.flat_params.device != x.device:
            raise RuntimeError(f"model on {self.flat_params.device}, input on {x.device}: call model.to(device)")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._require_gpu(x)
        x = x.float()
        if torch.is_grad_enabled() and self.flat_params.requires_grad:
            return _UNetFunction.apply(x, self.flat_params, self)
        logits, _ = self.engine.forward(x, self.flat_params.detach(), self.bn_state, self.training, save=False)
        self._bn_tracked_inc()
        return logits

    @torch.no_grad()
    def predict_classes(self, x: torch.Tensor, dtype: str = "int64", precision: str = "fp32", nhwc: bool = False) -> torch.Tensor:
        """forward + argmax fused in the head kernel (deployment/inference.py:60-62), eval-mode BN.
        precision "bf16": bf16 activations/weights with fp32 accumulation (the AMP setting of the reference's
        training protocol) — class maps agree with fp32 wherever the logit margin exceeds bf16 rounding."""
        self._require_gpu(x)
        if precision == "bf16":
            if nhwc:
                x = x.permute(0, 3, 1, 2).contiguous()
            _, am = self.engine.forward_bf16_eval(x.float(), self.flat_params.detach(), self.bn_state, want_argmax=dtype)
            return am
        if precision != "fp32":
            raise ValueError(f"precision {precision!r}: use 'fp32' or 'bf16'")
        was = self.training
        self.eval()
        try:
            _, am = self.engine.forward(x.float(), self.flat_params.detach(), self.bn_state, False, save=False,
                                        want_argmax=dtype, nhwc=nhwc)
        finally:
            self.train(was)
        return am

    @torch.no_grad()
    def forward_bf16(self, x: torch.Tensor) -> torch.Tensor:
        """eval-mode logits (fp32 tensor) from the bf16 path"""
        self._require_gpu(x)
        logits, _ = self.engine.forward_bf16_eval(x.float(), self.flat_params.detach(), self.bn_state)
        return logits
