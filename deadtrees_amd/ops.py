"""Thin tensor-level wrappers over the C ABI (one python function per ``dt_*`` entry point).

Every function takes/returns torch HIP tensors, validates nothing beyond what the C side validates,
and raises RuntimeError when the library reports an error.  Used by the engine-independent callers
(tiled inference, optimiser) and by the parity tests, which therefore exercise the C ABI itself.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("deadtrees_amd ops need HIP device tensors (no CPU fallback)")


def conv_desc(B, Hin, Win, C0, C1, mode0, Cout, k, stride, pad, split=0, acc=0):
    Ho = (Hin + 2 * pad - k) // stride + 1
    Wo = (Win + 2 * pad - k) // stride + 1
    return _lib.ConvDesc(B, Hin, Win, C0, C1, mode0, Ho, Wo, Cout, k, stride, pad, split, acc)


def conv2d(src0, w_hwio, k, stride, pad, src1=None, mode0=0, split=0, out0=None, out1=None, accumulate=False,
           want_stats=False, in_scale=None, in_shift=None):
    """NHWC conv through dt_conv2d.  Returns (out0, out1, stats[2,P,Cout] or None)."""
    _gpu(src0, src1, w_hwio)
    lib = _lib.load()
    B = src0.shape[0]
    C0 = src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    if mode0 == 0:
        Hin, Win = src0.shape[1], src0.shape[2]
    else:
        Hin, Win = 2 * src0.shape[1], 2 * src0.shape[2]
    Cout = w_hwio.shape[-1]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, k, stride, pad, split, 1 if accumulate else 0)
    dev = src0.device
    if out0 is None:
        out0 = torch.empty((B, d.Ho, d.Wo, split if split else Cout), dtype=torch.float32, device=dev)
    if split and out1 is None:
        out1 = torch.empty((B, d.Ho, d.Wo, Cout - split), dtype=torch.float32, device=dev)
    stats = None
    if want_stats:
        P = lib.dt_conv2d_stat_rows(C.byref(d))
        if P <= 0:
            raise RuntimeError(lib.dt_last_error().decode())
        stats = torch.empty(lib.dt_bn_stats_floats(P, Cout), dtype=torch.float32, device=dev)
    _lib.check(lib.dt_conv2d(C.byref(d), _p(src0), _p(src1), _p(w_hwio.contiguous()), _p(out0), _p(out1), _p(stats),
                             _p(in_scale), _p(in_shift), _st()), "dt_conv2d")
    if stats is not None:
        stats = stats[:2 * P * Cout].view(2, P, Cout)
    return out0, out1, stats


def conv2d_affine(src0, w_hwio, k, stride, pad, scale, shift, relu=True, src1=None, mode0=0):
    """inference form through dt_conv2d_affine: [relu](conv(x) * scale + shift) in one launch (eval-mode BatchNorm folded
    into per-channel scale / shift) -> NHWC activation"""
    _gpu(src0, src1, w_hwio, scale, shift)
    B, C0 = src0.shape[0], src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    Cout = w_hwio.shape[-1]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, k, stride, pad)
    out = torch.empty((B, d.Ho, d.Wo, Cout), dtype=torch.float32, device=src0.device)
    _lib.check(_lib.load().dt_conv2d_affine(C.byref(d), _p(src0), _p(src1), _p(w_hwio.contiguous()), _p(out), _p(scale),
                                            _p(shift), 1 if relu else 0, _st()), "dt_conv2d_affine")
    return out


def conv2d_wgrad_winograd(src0, dy, src1=None, mode0=0, in_scale=None, in_shift=None):
    """3x3 stride-1 pad-1 weight gradient through dt_conv2d_wgrad_winograd -> dw HWIO"""
    _gpu(src0, src1, dy)
    lib = _lib.load()
    B, C0 = src0.shape[0], src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    Cout = dy.shape[-1]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, 3, 1, 1)
    if not lib.dt_conv2d_wgrad_winograd_supported(C.byref(d)):
        raise ValueError("layer shape not supported by the Winograd weight-gradient kernel")
    nbytes = lib.dt_conv2d_wgrad_winograd_workspace(C.byref(d))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
    dw = torch.empty((3, 3, C0 + C1, Cout), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dt_conv2d_wgrad_winograd(C.byref(d), _p(src0), _p(src1), _p(dy.contiguous()), _p(dw), _p(ws), nbytes,
                                            _p(in_scale), _p(in_shift), _st()), "dt_conv2d_wgrad_winograd")
    return dw


def winograd_weights(w_hwio):
    """U = G g G^T of a 3x3 HWIO weight in the Winograd kernel's image order [16][Cin/8][2][Cout][4]."""
    _gpu(w_hwio)
    k, _, cin, cout = w_hwio.shape
    assert k == 3
    u = torch.empty((16, cin // 8, 2, cout, 4), dtype=torch.float32, device=w_hwio.device)
    _lib.check(_lib.load().dt_winograd_weights(_p(w_hwio.contiguous()), _p(u), cin, cout, _st()), "dt_winograd_weights")
    return u


def conv2d_winograd(src0, u, src1=None, mode0=0, split=0, out0=None, out1=None, accumulate=False, want_stats=False,
                    in_scale=None, in_shift=None):
    """3x3 stride-1 pad-1 NHWC conv through dt_conv2d_winograd.  Returns (out0, out1, stats[2,P,Cout] or None)."""
    _gpu(src0, src1, u)
    lib = _lib.load()
    B, C0 = src0.shape[0], src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    Cout = u.shape[3]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, 3, 1, 1, split, 1 if accumulate else 0)
    if not lib.dt_conv2d_winograd_supported(C.byref(d)):
        raise ValueError("layer shape not supported by the Winograd kernel")
    dev = src0.device
    if out0 is None:
        out0 = torch.empty((B, d.Ho, d.Wo, split if split else Cout), dtype=torch.float32, device=dev)
    if split and out1 is None:
        out1 = torch.empty((B, d.Ho, d.Wo, Cout - split), dtype=torch.float32, device=dev)
    stats = None
    if want_stats:
        P = lib.dt_conv2d_winograd_stat_rows(C.byref(d))
        stats = torch.empty(lib.dt_bn_stats_floats(P, Cout), dtype=torch.float32, device=dev)
    _lib.check(lib.dt_conv2d_winograd(C.byref(d), _p(src0), _p(src1), _p(u), _p(out0), _p(out1), _p(stats),
                                      _p(in_scale), _p(in_shift), _st()), "dt_conv2d_winograd")
    if stats is not None:
        stats = stats[:2 * P * Cout].view(2, P, Cout)
    return out0, out1, stats


def weight_flip_transpose(w_hwio):
    _gpu(w_hwio)
    k, _, cin, cout = w_hwio.shape
    wd = torch.empty((k, k, cout, cin), dtype=torch.float32, device=w_hwio.device)
    _lib.check(_lib.load().dt_weight_flip_transpose(_p(w_hwio.contiguous()), _p(wd), k, cin, cout, _st()),
               "dt_weight_flip_transpose")
    return wd


def conv2d_wgrad(src0, dy, k, stride, pad, src1=None, mode0=0, in_scale=None, in_shift=None):
    _gpu(src0, src1, dy)
    lib = _lib.load()
    B = src0.shape[0]
    C0 = src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    Cout = dy.shape[-1]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, k, stride, pad)
    assert (d.Ho, d.Wo) == (dy.shape[1], dy.shape[2]), ((d.Ho, d.Wo), dy.shape)
    nbytes = lib.dt_conv2d_wgrad_workspace(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(lib.dt_last_error().decode())
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
    dw = torch.empty((k, k, C0 + C1, Cout), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dt_conv2d_wgrad(C.byref(d), _p(src0), _p(src1), _p(dy.contiguous()), _p(dw), _p(ws), nbytes,
                                   _p(in_scale), _p(in_shift), _st()), "dt_conv2d_wgrad")
    return dw


def bn_finalize(stats, count, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    _gpu(stats, gamma, beta)
    _, P, Cc = stats.shape
    dev = stats.device
    full = torch.empty(_lib.load().dt_bn_stats_floats(P, Cc), dtype=torch.float32, device=dev)
    full[:2 * P * Cc] = stats.reshape(-1)
    stats = full
    mean, invstd, scale, shift = (torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(4))
    _lib.check(_lib.load().dt_bn_finalize(_p(stats), P, Cc, float(count), _p(gamma), _p(beta), eps, momentum,
                                          _p(running_mean), _p(running_var), _p(mean), _p(invstd), _p(scale),
                                          _p(shift), _st()), "dt_bn_finalize")
    return mean, invstd, scale, shift


def bn_act(y, scale, shift, res=None, rscale=None, rshift=None, relu=True):
    _gpu(y, scale, shift, res)
    out = torch.empty_like(y)
    n_pix = y.numel() // y.shape[-1]
    _lib.check(_lib.load().dt_bn_act(_p(y), _p(scale), _p(shift), _p(res), _p(rscale), _p(rshift), _p(out), n_pix,
                                     y.shape[-1], 1 if relu else 0, _st()), "dt_bn_act")
    return out


def bn_backward(dout, out_act, y, mean, invstd, gamma, want_dres=False, act_scale=None, act_shift=None):
    _gpu(dout, y)
    lib = _lib.load()
    Cc = y.shape[-1]
    n_pix = y.numel() // Cc
    P = lib.dt_bn_bwd_rows(n_pix, Cc)
    red = torch.empty(lib.dt_bn_bwd_red_floats(n_pix, Cc), dtype=torch.float32, device=y.device)
    _lib.check(lib.dt_bn_bwd_reduce(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(act_scale), _p(act_shift),
                                    _p(red), n_pix, Cc, _st()), "dt_bn_bwd_reduce")
    dgamma = torch.empty(Cc, dtype=torch.float32, device=y.device)
    dbeta = torch.empty_like(dgamma)
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    _lib.check(lib.dt_bn_bwd_apply(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(gamma), _p(act_scale),
                                   _p(act_shift), _p(red), P, _p(dgamma), _p(dbeta), _p(dy), _p(dres), 0, n_pix, Cc,
                                   _st()), "dt_bn_bwd_apply")
    return dy, dgamma, dbeta, dres


def maxpool3x3s2(x, want_argmax=True):
    _gpu(x)
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty((B, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    am = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device) if want_argmax else None
    _lib.check(_lib.load().dt_maxpool3x3s2(_p(x), _p(out), _p(am), B, H, W, Cc, _st()), "dt_maxpool3x3s2")
    return out, am


def maxpool3x3s2_bwd(dout, argmax, H, W, dx=None):
    _gpu(dout, argmax)
    B, _, _, Cc = dout.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty((B, H, W, Cc), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.load().dt_maxpool3x3s2_bwd(_p(dout), _p(argmax), _p(dx), 1 if acc else 0, B, H, W, Cc, _st()),
               "dt_maxpool3x3s2_bwd")
    return dx


def upsample2x_bwd(dup):
    _gpu(dup)
    B, H2, W2, Cc = dup.shape
    dx = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=torch.float32, device=dup.device)
    _lib.check(_lib.load().dt_upsample2x_bwd(_p(dup), _p(dx), 0, B, H2 // 2, W2 // 2, Cc, _st()), "dt_upsample2x_bwd")
    return dx


def nchw_to_nhwc(x):
    _gpu(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dt_nchw_to_nhwc(_p(x.contiguous()), _p(out), B, Cc, H, W, _st()), "dt_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    _gpu(x)
    B, H, W, Cc = x.shape
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dt_nhwc_to_nchw(_p(x.contiguous()), _p(out), B, Cc, H, W, _st()), "dt_nhwc_to_nchw")
    return out


def normalize_u8(src_u8_nhwc, mean, std, c_dst):
    """uint8 [..., Csrc] -> f32 [..., c_dst] with (x - 255*mean) / (255*std)  (deadtreedata.py:148-154)."""
    _gpu(src_u8_nhwc)
    cs = src_u8_nhwc.shape[-1]
    n_pix = src_u8_nhwc.numel() // cs
    out = torch.empty(tuple(src_u8_nhwc.shape[:-1]) + (c_dst,), dtype=torch.float32, device=src_u8_nhwc.device)
    m = (C.c_float * c_dst)(*[float(v) for v in mean[:c_dst]])
    s = (C.c_float * c_dst)(*[float(v) for v in std[:c_dst]])
    _lib.check(_lib.load().dt_normalize_u8(_p(src_u8_nhwc.contiguous()), _p(out), n_pix, cs, c_dst, m, s, _st()),
               "dt_normalize_u8")
    return out


def head_fwd(x, w_ohwi, bias, argmax: Optional[str] = None):
    _gpu(x, w_ohwi, bias)
    B, H, W, Cin = x.shape
    K = w_ohwi.shape[0]
    logits = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
    a64 = torch.empty((B, H, W), dtype=torch.int64, device=x.device) if argmax == "int64" else None
    a8 = torch.empty((B, H, W), dtype=torch.uint8, device=x.device) if argmax == "uint8" else None
    _lib.check(_lib.load().dt_head_fwd(_p(x), _p(w_ohwi.contiguous()), _p(bias), _p(logits), _p(a64), _p(a8), B, H, W,
                                       Cin, K, _st()), "dt_head_fwd")
    return logits, (a64 if a64 is not None else a8)


def head_bwd(x, w_ohwi, dlogits):
    _gpu(x, w_ohwi, dlogits)
    lib = _lib.load()
    B, H, W, Cin = x.shape
    K = w_ohwi.shape[0]
    P = lib.dt_head_bwd_rows(B, H, W)
    red = torch.empty(lib.dt_head_bwd_red_floats(B, H, W, Cin, K), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    _lib.check(lib.dt_head_bwd(_p(x), _p(w_ohwi.contiguous()), _p(dlogits.contiguous()), _p(dx), _p(red), B, H, W, Cin,
                               K, _st()), "dt_head_bwd")
    dw = torch.empty_like(w_ohwi)
    db = torch.empty(K, dtype=torch.float32, device=x.device)
    _lib.check(lib.dt_head_bwd_finalize(_p(red), P, _p(dw), _p(db), Cin, K, _st()), "dt_head_bwd_finalize")
    return dx, dw, db


def conv2d_bf16(src0, w_packed, k, stride, pad, cout, src1=None, mode0=0, split=0, out0=None, out1=None,
                accumulate=False, want_stats=False, in_scale=None, in_shift=None):
    """bf16 NHWC conv (fp32 accumulate) through dt_conv2d_bf16; w_packed = [k*k, cout, cin] bf16."""
    _gpu(src0, src1, w_packed)
    lib = _lib.load()
    B, C0 = src0.shape[0], src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    d = conv_desc(B, Hin, Win, C0, C1, mode0, cout, k, stride, pad, split, 1 if accumulate else 0)
    dev, bf = src0.device, torch.bfloat16
    if out0 is None:
        out0 = torch.empty((B, d.Ho, d.Wo, split if split else cout), dtype=bf, device=dev)
    if split and out1 is None:
        out1 = torch.empty((B, d.Ho, d.Wo, cout - split), dtype=bf, device=dev)
    stats = None
    if want_stats:
        P = lib.dt_conv2d_bf16_stat_rows(C.byref(d))
        if P <= 0:
            raise RuntimeError(lib.dt_last_error().decode())
        stats = torch.empty(lib.dt_bn_stats_floats(P, cout), dtype=torch.float32, device=dev)
    w_packed = _dma_image(d, w_packed, k, C0 + C1, cout)
    _lib.check(lib.dt_conv2d_bf16(C.byref(d), _p(src0), _p(src1), _p(w_packed), _p(out0), _p(out1), _p(stats),
                                  _p(in_scale), _p(in_shift), _st()), "dt_conv2d_bf16")
    if stats is not None:
        stats = stats[:2 * P * cout].view(2, P, cout)
    return out0, out1, stats


def _dma_image(d, w_packed, k, cin, cout):
    """the LDS-DMA staged kernels (dt_conv2d_bf16_config reports mt == 8) read the weights in the chunked layout
    [tap][Cin/32][Cout][32] (dt_weight_images modes 3 / 4); this test-level wrapper re-arranges a [tap][Cout][Cin] image"""
    tw, tn, ck, mt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if _lib.load().dt_conv2d_bf16_config(C.byref(d), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt)) != 0 or mt.value != 8:
        return w_packed
    return w_packed.view(k * k, cout, cin // 32, 32).permute(0, 2, 1, 3).contiguous()


def pack_weights_bf16(w_hwio, dgrad=False):
    """fp32 HWIO -> bf16 [taps][Cout][Cin] (forward) or the data-gradient image [taps][Cin][Cout] (taps reversed)"""
    _gpu(w_hwio)
    k, _, cin, cout = w_hwio.shape
    lib = _lib.load()
    if dgrad:
        out = torch.empty((k * k, cin, cout), dtype=torch.bfloat16, device=w_hwio.device)
        _lib.check(lib.dt_pack_dgrad_weights_bf16(_p(w_hwio.contiguous()), _p(out), k, cin, cout, _st()),
                   "dt_pack_dgrad_weights_bf16")
    else:
        out = torch.empty((k * k, cout, cin), dtype=torch.bfloat16, device=w_hwio.device)
        _lib.check(lib.dt_pack_weights_bf16(_p(w_hwio.contiguous()), _p(out), k, cin, cout, _st()),
                   "dt_pack_weights_bf16")
    return out


def conv2d_wgrad_bf16(src0, dy, k, stride, pad, src1=None, mode0=0, in_scale=None, in_shift=None):
    _gpu(src0, src1, dy)
    lib = _lib.load()
    B, C0 = src0.shape[0], src0.shape[-1]
    C1 = 0 if src1 is None else src1.shape[-1]
    Hin, Win = (src0.shape[1], src0.shape[2]) if mode0 == 0 else (2 * src0.shape[1], 2 * src0.shape[2])
    Cout = dy.shape[-1]
    d = conv_desc(B, Hin, Win, C0, C1, mode0, Cout, k, stride, pad)
    assert (d.Ho, d.Wo) == (dy.shape[1], dy.shape[2])
    nbytes = lib.dt_conv2d_wgrad_bf16_workspace(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(lib.dt_last_error().decode())
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device)
    dw = torch.empty((k, k, C0 + C1, Cout), dtype=torch.float32, device=dy.device)
    _lib.check(lib.dt_conv2d_wgrad_bf16(C.byref(d), _p(src0), _p(src1), _p(dy.contiguous()), _p(dw), _p(ws), nbytes,
                                        _p(in_scale), _p(in_shift), _st()), "dt_conv2d_wgrad_bf16")
    return dw


def confusion_matrix(pred, target, lu=None, K=2, counts=None):
    """counts int64 [2,K,K] (+=): [0] all pixels, [1] pixels with lu == 1; rows target, cols prediction."""
    _gpu(pred, target, lu)
    # the kernel reads target / lu as int64 for pred.numel() elements: anything else would run past the allocation
    if target.numel() != pred.numel() or (lu is not None and lu.numel() != pred.numel()):
        raise RuntimeError(f"confusion_matrix: prediction {tuple(pred.shape)}, target {tuple(target.shape)}" +
                           (f", lu {tuple(lu.shape)}" if lu is not None else "") + " must have the same number of pixels")
    if target.dtype != torch.int64:
        target = target.long()
    if lu is not None and lu.dtype != torch.int64:
        lu = lu.long()
    if counts is None:
        counts = torch.zeros((2, K, K), dtype=torch.int64, device=pred.device)
    elif counts.dtype != torch.int64 or tuple(counts.shape) != (2, K, K) or counts.device != pred.device:
        raise RuntimeError(f"confusion_matrix: counts must be int64 [2,{K},{K}] on {pred.device}")
    err = torch.zeros(1, dtype=torch.int32, device=pred.device)
    pred = pred.contiguous()
    p64 = pred if pred.dtype == torch.int64 else None
    p8 = pred if pred.dtype == torch.uint8 else None
    if p64 is None and p8 is None:
        raise RuntimeError("confusion_matrix: prediction must be int64 or uint8")
    _lib.check(_lib.load().dt_confusion_matrix(_p(p64), _p(p8), _p(target.contiguous()),
                                               _p(lu.contiguous()) if lu is not None else None, K, pred.numel(),
                                               _p(counts), _p(err), _st()), "dt_confusion_matrix")
    return counts, err


def conv2d_bn_bwd(src0, w_hwio, y, mean, invstd, act_scale=None, act_shift=None, act=None, join_into=None):
    """3x3 stride-1 conv (a data gradient) whose epilogue also produces the BatchNorm-backward partial sums of the
    layer with raw output `y` (same shape as the result).  Virtual activation: give act_scale/act_shift; block
    output: give the stored activation `act` and the tensor `join_into` the gradient is ADDED to.
    -> (out [B,H,W,Cout], red [2,P,Cout])"""
    _gpu(src0, w_hwio, y, mean, invstd, act_scale, act_shift, act, join_into)
    lib = _lib.load()
    B, H, W, C0 = src0.shape
    Cout = w_hwio.shape[-1]
    d = conv_desc(B, H, W, C0, 0, 0, Cout, 3, 1, 1, 0, 1 if join_into is not None else 0)
    P = lib.dt_conv2d_stat_rows(C.byref(d))
    red = torch.empty(lib.dt_bn_stats_floats(P, Cout), dtype=torch.float32, device=src0.device)
    out = join_into if join_into is not None else torch.empty((B, H, W, Cout), dtype=torch.float32, device=src0.device)
    fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift), _p(act))
    _lib.check(lib.dt_conv2d_bn_bwd(C.byref(d), _p(src0), _p(w_hwio.contiguous()), _p(out), _p(red), C.byref(fuse),
                                    _st()), "dt_conv2d_bn_bwd")
    return out, red[:2 * P * Cout].view(2, P, Cout)


def conv2d_winograd_bn_bwd(src0, u, y, mean, invstd, act_scale=None, act_shift=None, act=None, join_into=None):
    """Winograd form of conv2d_bn_bwd (u = winograd_weights of the data-gradient HWIO image) -> (out, red [2,P,Cout])"""
    _gpu(src0, u, y, mean, invstd, act_scale, act_shift, act, join_into)
    lib = _lib.load()
    B, H, W, C0 = src0.shape
    Cout = u.shape[3]
    d = conv_desc(B, H, W, C0, 0, 0, Cout, 3, 1, 1, 0, 1 if join_into is not None else 0)
    P = lib.dt_conv2d_winograd_stat_rows(C.byref(d))
    red = torch.empty(lib.dt_bn_stats_floats(P, Cout), dtype=torch.float32, device=src0.device)
    out = join_into if join_into is not None else torch.empty((B, H, W, Cout), dtype=torch.float32, device=src0.device)
    fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift), _p(act))
    _lib.check(lib.dt_conv2d_winograd_bn_bwd(C.byref(d), _p(src0), _p(u), _p(out), _p(red), C.byref(fuse), _st()),
               "dt_conv2d_winograd_bn_bwd")
    return out, red[:2 * P * Cout].view(2, P, Cout)


def conv2d_bf16_bn_bwd(src0, w_packed, cout, y, mean, invstd, act_scale=None, act_shift=None, act=None,
                       join_into=None):
    """bf16 twin of conv2d_bn_bwd (src0, y, act, join_into bf16 NHWC; w_packed from pack_weights_bf16)
    -> (out bf16, red [2,P,Cout])"""
    _gpu(src0, w_packed, y, mean, invstd, act_scale, act_shift, act, join_into)
    lib = _lib.load()
    B, H, W, C0 = src0.shape
    d = conv_desc(B, H, W, C0, 0, 0, cout, 3, 1, 1, 0, 1 if join_into is not None else 0)
    P = lib.dt_conv2d_bf16_stat_rows(C.byref(d))
    red = torch.empty(lib.dt_bn_stats_floats(P, cout), dtype=torch.float32, device=src0.device)
    out = join_into if join_into is not None else torch.empty((B, H, W, cout), dtype=torch.bfloat16,
                                                              device=src0.device)
    fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift), _p(act))
    w_packed = _dma_image(d, w_packed, 3, C0, cout)
    _lib.check(lib.dt_conv2d_bf16_bn_bwd(C.byref(d), _p(src0), _p(w_packed), _p(out), _p(red), C.byref(fuse), _st()),
               "dt_conv2d_bf16_bn_bwd")
    return out, red[:2 * P * cout].view(2, P, cout)


def conv2d_bf16_upsampled_dgrad(dy, w_packed_dgrad, cin, y, mean, invstd, act_scale, act_shift):
    """bf16 gradient of x for out = conv3x3(nearest_upsample_x2(x)) from dy [B,2h,2w,cout] (bf16) and the data-gradient
    weight image (pack_weights_bf16(..., dgrad=True)), the narrow layers only: the 2x2 sums are taken on the fp32
    accumulators -> (gx [B,h,w,cin] bf16, red [2,P,cin] = BatchNorm-backward partial sums of the layer with raw output y)"""
    _gpu(dy, w_packed_dgrad, y, mean, invstd, act_scale, act_shift)
    lib = _lib.load()
    B, H, W, cout = dy.shape
    d = conv_desc(B, H, W, cout, 0, 0, cin, 3, 1, 1)
    if not lib.dt_conv2d_bf16_upsampled_dgrad_supported(C.byref(d)):
        raise ValueError("conv2d_bf16_upsampled_dgrad: layer shape not covered by the narrow kernel")
    P = lib.dt_conv2d_bf16_stat_rows(C.byref(d))
    red = torch.empty(lib.dt_bn_stats_floats(P, cin), dtype=torch.float32, device=dy.device)
    gx = torch.empty((B, H // 2, W // 2, cin), dtype=torch.bfloat16, device=dy.device)
    fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift))
    _lib.check(lib.dt_conv2d_bf16_upsampled_dgrad(C.byref(d), _p(dy.contiguous()), _p(w_packed_dgrad), _p(gx), _p(red),
                                                  C.byref(fuse), _st()), "dt_conv2d_bf16_upsampled_dgrad")
    return gx, red[:2 * P * cin].view(2, P, cin)


def upsample2x_bwd_bn(dup, y, mean, invstd, act_scale, act_shift):
    """2x2-sum backward of a nearest x2 upsample + the BatchNorm-backward partial sums of the layer with raw output
    `y` ([B,H,W,C], fp32 or bf16 like dup) -> (dx, red [2,P,C])"""
    _gpu(dup, y, mean, invstd, act_scale, act_shift)
    lib = _lib.load()
    B, H2, W2, Cc = dup.shape
    H, W = H2 // 2, W2 // 2
    bf = dup.dtype == torch.bfloat16
    P = (lib.dt_upsample2x_bwd_bn_bf16_rows if bf else lib.dt_upsample2x_bwd_bn_rows)(B, H, W, Cc)
    red = torch.empty(lib.dt_bn_stats_floats(P, Cc), dtype=torch.float32, device=dup.device)
    dx = torch.empty((B, H, W, Cc), dtype=dup.dtype, device=dup.device)
    fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift))
    fn = lib.dt_upsample2x_bwd_bn_bf16 if bf else lib.dt_upsample2x_bwd_bn
    _lib.check(fn(_p(dup.contiguous()), _p(dx), C.byref(fuse), _p(red), B, H, W, Cc, _st()), "dt_upsample2x_bwd_bn")
    return dx, red[:2 * P * Cc].view(2, P, Cc)


def conv2d_upsampled_dgrad(dy, w_hwio, cin, y=None, mean=None, invstd=None, act_scale=None, act_shift=None):
    """gradient of x for out = conv3x3(nearest_upsample_x2(x)) (pad 1) from dy [B,2h,2w,cout] and the forward weights
    [3,3,cin,cout] in one sub-pixel kernel -> (gx [B,h,w,cin], red [2,P,cin] or None).  With y/mean/invstd/act_*: the
    BatchNorm-backward partial sums of the layer with raw output y (x = relu(y*act_scale+act_shift)) come along."""
    _gpu(dy, w_hwio)
    lib = _lib.load()
    B, H, W, cout = dy.shape
    d = _lib.ConvDesc(B, H, W, cin, 0, 1, H, W, cout, 3, 1, 1, 0, 0)
    if not lib.dt_conv2d_upsampled_dgrad_supported(C.byref(d)):
        raise ValueError("conv2d_upsampled_dgrad: layer shape not covered by the sub-pixel kernel")
    gx = torch.empty((B, H // 2, W // 2, cin), dtype=torch.float32, device=dy.device)
    red, fuse, P = None, None, 0
    if y is not None:
        _gpu(y, mean, invstd, act_scale, act_shift)
        P = lib.dt_conv2d_upsampled_dgrad_rows(C.byref(d))
        red = torch.empty(lib.dt_bn_stats_floats(P, cin), dtype=torch.float32, device=dy.device)
        fuse = _lib.BnBwdFuse(_p(y.contiguous()), _p(mean), _p(invstd), _p(act_scale), _p(act_shift))
    _lib.check(lib.dt_conv2d_upsampled_dgrad(C.byref(d), _p(dy.contiguous()), _p(w_hwio.contiguous()), _p(gx), _p(red),
                                             C.byref(fuse) if fuse is not None else None, _st()), "dt_conv2d_upsampled_dgrad")
    return gx, (red[:2 * P * cin].view(2, P, cin) if red is not None else None)


def stem_conv_bf16(x_nhwc_f32, w_hwio_7x7, want_stats=False):
    """the 7x7 / stride-2 / pad-3 stem on the bf16 MFMA kernels: fp32 NHWC image [B,H,W,Cin<=4] -> bf16
    [B,H/2,W/2,Cout] (+ fp32 BatchNorm partial statistics), via the 2x2 space-to-depth image and a 4x4 window"""
    _gpu(x_nhwc_f32, w_hwio_7x7)
    lib = _lib.load()
    B, H, W, Cin = x_nhwc_f32.shape
    Cout = w_hwio_7x7.shape[-1]
    st = _st()
    s2d = torch.empty((B, H // 2, W // 2, 16), dtype=torch.bfloat16, device=x_nhwc_f32.device)
    _lib.check(lib.dt_stem_s2d_bf16(_p(x_nhwc_f32.contiguous()), _p(s2d), B, H, W, Cin, st), "dt_stem_s2d_bf16")
    wp = torch.empty(16 * Cout * 16, dtype=torch.bfloat16, device=x_nhwc_f32.device)
    _lib.check(lib.dt_stem_pack_weights_bf16(_p(w_hwio_7x7.contiguous()), _p(wp), Cin, Cout, st),
               "dt_stem_pack_weights_bf16")
    d = _lib.ConvDesc(B, H // 2, W // 2, 16, 0, 0, H // 2, W // 2, Cout, 4, 1, 2, 0, 0)
    out = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.bfloat16, device=x_nhwc_f32.device)
    stats = None
    if want_stats:
        P = lib.dt_conv2d_bf16_stat_rows(C.byref(d))
        if P <= 0:
            raise RuntimeError(lib.dt_last_error().decode())
        stats = torch.empty(lib.dt_bn_stats_floats(P, Cout), dtype=torch.float32, device=out.device)
    _lib.check(lib.dt_conv2d_bf16(C.byref(d), _p(s2d), None, _p(wp), _p(out), None, _p(stats), None, None, st),
               "dt_conv2d_bf16(stem)")
    if stats is not None:
        stats = stats[:2 * P * Cout].view(2, P, Cout)
    return out, stats


def stem_wgrad_bf16(x_nhwc_f32, dy_bf16):
    """weight gradient of the 7x7 / stride-2 stem on the bf16 MFMA kernels (space-to-depth form) -> fp32 HWIO
    [7,7,Cin,Cout]; dy bf16 [B,H/2,W/2,Cout]"""
    _gpu(x_nhwc_f32, dy_bf16)
    lib = _lib.load()
    B, H, W, Cin = x_nhwc_f32.shape
    Cout = dy_bf16.shape[-1]
    st = _st()
    s2d = torch.empty((B, H // 2, W // 2, 16), dtype=torch.bfloat16, device=x_nhwc_f32.device)
    _lib.check(lib.dt_stem_s2d_bf16(_p(x_nhwc_f32.contiguous()), _p(s2d), B, H, W, Cin, st), "dt_stem_s2d_bf16")
    d = _lib.ConvDesc(B, H // 2, W // 2, 16, 0, 0, H // 2, W // 2, Cout, 4, 1, 2, 0, 0)
    nbytes = lib.dt_conv2d_wgrad_bf16_workspace(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(lib.dt_last_error().decode())
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=s2d.device)
    dw4 = torch.empty(16 * 16 * Cout, dtype=torch.float32, device=s2d.device)
    _lib.check(lib.dt_conv2d_wgrad_bf16(C.byref(d), _p(s2d), None, _p(dy_bf16.contiguous()), _p(dw4), _p(ws), nbytes,
                                        None, None, st), "dt_conv2d_wgrad_bf16(stem)")
    dw7 = torch.empty((7, 7, Cin, Cout), dtype=torch.float32, device=s2d.device)
    _lib.check(lib.dt_stem_unpack_wgrad(_p(dw4), _p(dw7), Cin, Cout, st), "dt_stem_unpack_wgrad")
    return dw7


# ---- bf16 elementwise kernels (thin wrappers; the engine calls the C ABI directly with its own buffers)
def bn_act_bf16(y, scale, shift, res=None, rscale=None, rshift=None, relu=True):
    """bf16 (or fp32) y [.., C] -> bf16 act(y*scale+shift + (res*rscale+rshift))"""
    _gpu(y, scale, shift, res)
    Cq = y.shape[-1]
    out = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
    _lib.check(_lib.load().dt_bn_act_bf16(_p(y), 1 if y.dtype == torch.float32 else 0, _p(scale), _p(shift), _p(res),
                                          _p(rscale), _p(rshift), _p(out), y.numel() // Cq, Cq, 1 if relu else 0,
                                          _st()), "dt_bn_act_bf16")
    return out


def bn_backward_bf16(dout, out_act, y, mean, invstd, gamma, want_dres=False, act_scale=None, act_shift=None,
                     dres=None):
    """-> (dy bf16, dgamma f32, dbeta f32, dres bf16 or None); same contract as bn_backward, bf16 activations.
    `dres` given: the masked gradient is ADDED to it (gradient join)."""
    _gpu(dout, y, mean, invstd, gamma)
    lib = _lib.load()
    Cq = y.shape[-1]
    n_pix = y.numel() // Cq
    P = lib.dt_bn_bwd_rows_bf16(n_pix)
    red = torch.empty(lib.dt_bn_stats_floats(P, Cq), dtype=torch.float32, device=y.device)
    _lib.check(lib.dt_bn_bwd_reduce_bf16(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(act_scale),
                                         _p(act_shift), _p(red), n_pix, Cq, _st()), "dt_bn_bwd_reduce_bf16")
    dy = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
    dgamma = torch.empty(Cq, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(Cq, dtype=torch.float32, device=y.device)
    acc = dres is not None
    if want_dres and dres is None:
        dres = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
    _lib.check(lib.dt_bn_bwd_apply_bf16(_p(dout), _p(out_act), _p(y), _p(mean), _p(invstd), _p(gamma), _p(act_scale),
                                        _p(act_shift), _p(red), P, _p(dgamma), _p(dbeta), _p(dy), _p(dres),
                                        1 if acc else 0, n_pix, Cq, _st()), "dt_bn_bwd_apply_bf16")
    return dy, dgamma, dbeta, dres


def maxpool3x3s2_bf16(x):
    """bf16 NHWC -> (pooled bf16, argmax bytes [B,Ho,Wo,C] uint8: window position kh*3+kw of the first maximum)"""
    _gpu(x)
    B, H, W, Cq = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, Ho, Wo, Cq), dtype=torch.bfloat16, device=x.device)
    am = torch.empty((B, Ho, Wo, Cq), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().dt_maxpool3x3s2_bf16_amax(_p(x), _p(out), _p(am), B, H, W, Cq, _st()),
               "dt_maxpool3x3s2_bf16_amax")
    return out, am


def maxpool3x3s2_bwd_bf16(dout, argmax, H, W, dx=None):
    """gradient of maxpool3x3s2_bf16 wrt its [B,H,W,C] input; `dx` given: accumulate into it"""
    _gpu(dout, argmax)
    B, _, _, Cq = dout.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty((B, H, W, Cq), dtype=torch.bfloat16, device=dout.device)
    _lib.check(_lib.load().dt_maxpool3x3s2_bwd_bf16(_p(dout), _p(argmax), _p(dx), 1 if acc else 0, B, H, W, Cq, _st()),
               "dt_maxpool3x3s2_bwd_bf16")
    return dx


def upsample2x_bwd_bf16(dup):
    """[B,2H,2W,C] bf16 gradient of a nearest x2 upsample -> [B,H,W,C] (2x2 sums, one rounding)"""
    _gpu(dup)
    B, H2, W2, Cq = dup.shape
    dx = torch.empty((B, H2 // 2, W2 // 2, Cq), dtype=torch.bfloat16, device=dup.device)
    _lib.check(_lib.load().dt_upsample2x_bwd_bf16(_p(dup), _p(dx), B, H2 // 2, W2 // 2, Cq, _st()),
               "dt_upsample2x_bwd_bf16")
    return dx


def split_normalize_u8(raster_chw_u8, d: int, first: int, count: int, mean, std, c_dst: int):
    """band-major uint8 raster [C,h,w] on the device -> the fp32 NHWC sub-tiles [count,d,d,c_dst] ``first`` .. of the
    row-major d x d block grid of the zero-padded raster (tiler.py:121-134 + make_blocks_vectorized + Normalize, fused)"""
    _gpu(raster_chw_u8)
    if raster_chw_u8.dtype != torch.uint8 or raster_chw_u8.dim() != 3:
        raise RuntimeError("split_normalize_u8: raster must be uint8 [C,h,w]")
    cs, h, w = raster_chw_u8.shape
    nbx = -(-w // d)
    nby = -(-h // d)
    if first < 0 or count <= 0 or first + count > nbx * nby:
        raise RuntimeError(f"split_normalize_u8: blocks {first}..{first + count - 1} outside the {nby} x {nbx} grid")
    out = torch.empty((count, d, d, c_dst), dtype=torch.float32, device=raster_chw_u8.device)
    m = (C.c_float * c_dst)(*[float(v) for v in mean[:c_dst]])
    s = (C.c_float * c_dst)(*[float(v) for v in std[:c_dst]])
    _lib.check(_lib.load().dt_split_normalize_u8(_p(raster_chw_u8.contiguous()), _p(out), cs, h, w, d, nbx, first, count,
                                                 c_dst, m, s, _st()), "dt_split_normalize_u8")
    return out


def band_has_data(band_u8) -> torch.Tensor:
    """int32[1] device flag: 1 iff some byte is neither 0 nor 255 (scripts/inference.py:60-62 is_valid_tile, no host pass)"""
    _gpu(band_u8)
    if band_u8.dtype != torch.uint8:
        raise RuntimeError("band_has_data: uint8 band expected")
    flag = torch.zeros(1, dtype=torch.int32, device=band_u8.device)
    _lib.check(_lib.load().dt_band_has_data(_p(band_u8.contiguous()), band_u8.numel(), _p(flag), _st()), "dt_band_has_data")
    return flag


def augment_normalize_u8(src_u8_nhwc, geo, bc, mean, std, c_dst):
    """uint8 [B,H,W,Csrc] -> fp32 [B,H,W,c_dst]: flip / rot90 / brightness-contrast LUT / normalise in one pass
    (data/deadtreedata.py:128-146).  geo int32 [B,2] = (flip, rot k), bc fp32 [B,2] = (alpha, beta)."""
    _gpu(src_u8_nhwc, geo, bc)
    B, H, W, cs = src_u8_nhwc.shape
    if geo.dtype != torch.int32 or bc.dtype != torch.float32 or tuple(geo.shape) != (B, 2) or tuple(bc.shape) != (B, 2):
        raise RuntimeError("augment_normalize_u8: geo must be int32 [B,2] and bc float32 [B,2]")
    if H != W and bool((geo[:, 1] % 2 == 1).any()):
        raise RuntimeError("augment_normalize_u8: odd rot90 counts need square tiles")
    out = torch.empty((B, H, W, c_dst), dtype=torch.float32, device=src_u8_nhwc.device)
    sums = torch.empty(B, dtype=torch.int64, device=src_u8_nhwc.device)
    m = (C.c_float * c_dst)(*[float(v) for v in mean[:c_dst]])
    s = (C.c_float * c_dst)(*[float(v) for v in std[:c_dst]])
    _lib.check(_lib.load().dt_augment_normalize_u8(_p(src_u8_nhwc.contiguous()), _p(out), _p(geo.contiguous()),
                                                   _p(bc.contiguous()), _p(sums), B, H, W, cs, c_dst, m, s, _st()),
               "dt_augment_normalize_u8")
    return out


def augment_labels(labels, geo):
    """int64 [B,H,W] masks / land-use maps through the same flip + rot90 as the image batch"""
    _gpu(labels, geo)
    if labels.dtype != torch.int64 or labels.dim() != 3:
        raise RuntimeError("augment_labels: labels must be int64 [B,H,W]")
    B, H, W = labels.shape
    out = torch.empty_like(labels)
    _lib.check(_lib.load().dt_augment_labels(_p(labels.contiguous()), _p(out), _p(geo.contiguous()), B, H, W, _st()),
               "dt_augment_labels")
    return out


def ensemble_vote(maps_u8: torch.Tensor, K: int, dtype: str = "int64"):
    """uint8 class maps [M, ...] of M models -> per-pixel mode [...] (ties -> smallest class, torch.mode);
    returns (map, err flag).  deployment/inference.py:65-116."""
    _gpu(maps_u8)
    if maps_u8.dtype != torch.uint8:
        raise RuntimeError("ensemble_vote: class maps must be uint8")
    M = maps_u8.shape[0]
    n = maps_u8[0].numel()
    out8 = torch.empty(maps_u8.shape[1:], dtype=torch.uint8, device=maps_u8.device) if dtype == "uint8" else None
    out64 = torch.empty(maps_u8.shape[1:], dtype=torch.int64, device=maps_u8.device) if dtype == "int64" else None
    err = torch.zeros(1, dtype=torch.int32, device=maps_u8.device)
    _lib.check(_lib.load().dt_ensemble_vote(_p(maps_u8.contiguous()), M, n, K, _p(out8), _p(out64), _p(err), _st()),
               "dt_ensemble_vote")
    return (out64 if out64 is not None else out8), err


def signed_distmap(labels: torch.Tensor, K: int):
    """int64 labels [B,H,W] -> (fp32 distance maps [B,K,H,W], err flag) — the boundary-loss maps of
    loss/losses.py:159-178 as attached by data/deadtreedata.py:182-185, computed exactly on the device."""
    _gpu(labels)
    if labels.dtype != torch.int64 or labels.dim() != 3:
        raise RuntimeError("signed_distmap: labels must be int64 [B,H,W]")
    B, H, W = labels.shape
    lib = _lib.load()
    ws = torch.empty(int(lib.dt_signed_distmap_workspace(B, K, H, W)), dtype=torch.uint8, device=labels.device)
    dist = torch.empty((B, K, H, W), dtype=torch.float32, device=labels.device)
    err = torch.zeros(1, dtype=torch.int32, device=labels.device)
    _lib.check(lib.dt_signed_distmap(_p(labels.contiguous()), _p(dist), _p(ws), _p(err), B, K, H, W, _st()),
               "dt_signed_distmap")
    return dist, err


class FlatAdam:
    """clip_grad_norm_(max_norm) + torch.optim.Adam on one flat buffer, two fused HIP passes
    (reference: configs/trainer/default.yaml:18 + segmodel.py:420-425).

    Every per-step scalar lives on the device: the step count ``t_dev`` (advanced by ``dt_adam_advance`` only when the
    step is not skipped — Lightning does not call ``optimizer.step`` when ``training_step`` returns None, so the bias
    correction must not move either), the learning rate ``lr_dev`` (refreshed by a stream-ordered fill when ``lr``
    changes) and the bias corrections.  The launch sequence is therefore identical for every step: no ATen algebra,
    no host synchronisation, capturable in a HIP graph."""

    def __init__(self, params: torch.Tensor, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, max_norm: float = 0.5):
        _gpu(params)
        self.p = params
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.t = 0          # host mirror: number of step() calls (skipped steps included; the device count is t_dev)
        lib = _lib.load()
        dev = params.device
        self.rows = lib.dt_sumsq_rows(params.numel())
        self.partial = torch.empty(self.rows, dtype=torch.float64, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.t_dev = torch.zeros(1, dtype=torch.float64, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self._lr_on_dev = float(lr)
        self.hyper = torch.ones(3, dtype=torch.float32, device=dev)
        self.skip = torch.zeros(1, dtype=torch.int32, device=dev)

    def sync_lr(self):
        """push a changed learning rate to the device (stream-ordered fill: the value travels as a kernel argument).
        Call OUTSIDE graph capture / before a replay."""
        if self._lr_on_dev != float(self.lr):
            self.lr_dev.fill_(float(self.lr))
            self._lr_on_dev = float(self.lr)

    def skip_from_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """device flag int32[1] = loss is NaN/Inf (segmodel.py:220-222); `loss` fp32 device scalar"""
        _gpu(loss)
        if loss.dtype != torch.float32:
            loss = loss.float()
        _lib.check(_lib.load().dt_skip_from_loss(_p(loss), _p(self.skip), _st()), "dt_skip_from_loss")
        return self.skip

    def steps_applied(self) -> int:
        """optimiser steps that were not skipped (host sync)"""
        return int(self.t_dev.item())

    def step(self, grads: torch.Tensor, grad_scale: float = 1.0, skip_flag: Optional[torch.Tensor] = None,
             lr: Optional[float] = None, capturing: bool = False):
        lib = _lib.load()
        if lr is not None:
            self.lr = lr
        if not capturing:
            self.sync_lr()
        n = self.p.numel()
        st = _st()
        if skip_flag is None:          # stand-alone use: the non-finite-gradient guard of dt_clip_coef still applies
            skip_flag = self.skip.zero_()
        _lib.check(lib.dt_sumsq(_p(grads), n, _p(self.partial), st), "dt_sumsq")
        _lib.check(lib.dt_clip_coef(_p(self.partial), self.rows, float(self.max_norm or 0.0), float(grad_scale),
                                    _p(self.norm), _p(self.coef), _p(skip_flag), st), "dt_clip_coef")
        b1, b2 = self.betas
        _lib.check(lib.dt_adam_advance(_p(self.t_dev), _p(skip_flag), _p(self.lr_dev), b1, b2, _p(self.hyper), st),
                   "dt_adam_advance")
        _lib.check(lib.dt_adam_step_dev(_p(self.p), _p(grads), _p(self.m), _p(self.v), n, _p(self.hyper), b1, b2,
                                        self.eps, _p(self.coef), _p(skip_flag), st), "dt_adam_step_dev")
        self.t += 1
        return self.norm
