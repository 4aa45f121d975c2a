"""bf16-storage / fp32-accumulate kernels against fp64 references computed on the SAME bf16-rounded inputs.
Tolerance: one bf16 rounding of the output (2^-8 relative) on top of fp32 accumulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _ops():
    from deadtrees_amd import ops
    return ops


def bf(t):  # NCHW fp32 cpu -> NHWC bf16 gpu, and its exact fp64 value
    q = t.to(BF)
    return q.permute(0, 2, 3, 1).contiguous().to(DEV), q.double()


def to_nchw(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2).double()


def close_bf16(got, ref, extra=0.0):
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + ref.abs().max() * (2.0 ** -9 + extra)
    assert bool((err <= tol).all()), float((err / tol).max())


CASES = [(2, 32, 32, 64, 64, 3, 1, 1), (1, 40, 24, 16, 16, 3, 1, 1), (2, 16, 16, 128, 96, 3, 1, 1),
         (2, 8, 8, 32, 48, 3, 1, 1), (2, 32, 32, 64, 128, 3, 2, 1), (2, 32, 32, 64, 128, 1, 2, 0),
         (1, 34, 70, 512, 64, 3, 1, 1),
         # 512-pixel workgroup tiles (Cout >= 128 and >= 1024 tiles): ragged bottom edge, 2 n-tiles, Cin = 16 (mod 32)
         (16, 120, 128, 48, 128, 3, 1, 1),
         # the lean narrow-layer kernel (conv_bf16_narrow.hip: Cin, Cout in {16, 32}, maps >= 8 x 32): every channel
         # combination, ragged edges, several tiles per persistent workgroup (4 x 128 x 128 -> 256 tiles; 9 x 256 x 256)
         (2, 40, 64, 16, 16, 3, 1, 1), (2, 37, 70, 32, 16, 3, 1, 1), (3, 64, 64, 32, 32, 3, 1, 1),
         (2, 16, 32, 16, 32, 3, 1, 1), (4, 128, 128, 16, 16, 3, 1, 1), (9, 256, 256, 32, 16, 3, 1, 1)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CASES)
def test_conv_bf16_forward_and_stats(B, H, W, Cin, Cout, k, s, p):
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    xg, x64 = bf(x)
    w_hwio = w.permute(2, 3, 1, 0).contiguous().to(DEV)
    wp = ops.pack_weights_bf16(w_hwio)
    w64 = w.to(BF).double()
    ref = F.conv2d(x64, w64, stride=s, padding=p)
    y, _, stats = ops.conv2d_bf16(xg, wp, k, s, p, Cout, want_stats=True)
    close_bf16(to_nchw(y), ref)
    np.testing.assert_allclose(stats[0].double().sum(0).cpu(), ref.sum(dim=(0, 2, 3)), rtol=1e-4,
                               atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()))
    np.testing.assert_allclose(stats[1].double().sum(0).cpu(), (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4)


@pytest.mark.parametrize("B,h,w_,Cout", [(2, 10, 12, 96), (16, 64, 64, 192)])   # the second: 512-pixel tiles
def test_conv_bf16_upsample_concat_transform_split_accumulate(B, h, w_, Cout):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    C0, C1 = 64, 32
    a = torch.randn((B, C0, h, w_), generator=g)
    skip = torch.randn((B, C1, 2 * h, 2 * w_), generator=g)
    sc = 1 + 0.3 * torch.randn(C0, generator=g)
    sh = 0.3 * torch.randn(C0, generator=g) + 0.4
    wt = torch.randn((Cout, C0 + C1, 3, 3), generator=g) * 0.05
    ag, a64 = bf(a)
    sg, s64 = bf(skip)
    # the kernel rounds the transformed activation to bf16 while staging it
    z = F.relu(a64.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).double()
    xin = torch.cat([F.interpolate(z, scale_factor=2, mode="nearest"), s64], dim=1)
    ref = F.conv2d(xin, wt.to(BF).double(), padding=1)
    wp = ops.pack_weights_bf16(wt.permute(2, 3, 1, 0).contiguous().to(DEV))
    y, _, _ = ops.conv2d_bf16(ag, wp, 3, 1, 1, Cout, src1=sg, mode0=1, in_scale=sc.to(DEV), in_shift=sh.to(DEV))
    close_bf16(to_nchw(y), ref)
    base = torch.randn((B, 2 * h, 2 * w_, 64), generator=g).to(BF).to(DEV)
    o0, o1, _ = ops.conv2d_bf16(ag, wp, 3, 1, 1, Cout, src1=sg, mode0=1, in_scale=sc.to(DEV), in_shift=sh.to(DEV),
                                split=64, out0=base.clone(), accumulate=True)
    close_bf16(to_nchw(o0), ref[:, :64] + to_nchw(base))
    close_bf16(to_nchw(o1), ref[:, 64:])


@pytest.mark.parametrize("C0,Cout,h,w_", [(32, 16, 20, 24), (16, 16, 8, 16), (32, 32, 33, 17), (16, 32, 16, 16)])
def test_conv_bf16_narrow_kernel_upsampled_input_and_fused_activation(C0, Cout, h, w_):
    """dec4.conv1 of the U-Net on the lean kernel: nearest-upsampled input (index arithmetic in the LDS fill) with the
    producer's BatchNorm + ReLU applied while staging (rounded to bf16 like the general kernel does); run-to-run identical"""
    ops = _ops()
    g = torch.Generator().manual_seed(C0 + Cout + h)
    B = 3
    a = torch.randn((B, C0, h, w_), generator=g)
    sc = 1 + 0.3 * torch.randn(C0, generator=g)
    sh = 0.3 * torch.randn(C0, generator=g) + 0.2
    wt = torch.randn((Cout, C0, 3, 3), generator=g) * (2.0 / (9 * C0)) ** 0.5
    ag, a64 = bf(a)
    z = F.relu(a64.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).double()
    ref = F.conv2d(F.interpolate(z, scale_factor=2, mode="nearest"), wt.to(BF).double(), padding=1)
    wp = ops.pack_weights_bf16(wt.permute(2, 3, 1, 0).contiguous().to(DEV))
    y, _, st = ops.conv2d_bf16(ag, wp, 3, 1, 1, Cout, mode0=1, in_scale=sc.to(DEV), in_shift=sh.to(DEV), want_stats=True)
    close_bf16(to_nchw(y), ref)
    np.testing.assert_allclose(st[0].double().sum(0).cpu(), ref.sum(dim=(0, 2, 3)), rtol=1e-4,
                               atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()))
    np.testing.assert_allclose(st[1].double().sum(0).cpu(), (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4)
    y2, _, _ = ops.conv2d_bf16(ag, wp, 3, 1, 1, Cout, mode0=1, in_scale=sc.to(DEV), in_shift=sh.to(DEV))
    assert torch.equal(y, y2)
    plain = F.conv2d(F.interpolate(a64, scale_factor=2, mode="nearest"), wt.to(BF).double(), padding=1)
    y3, _, _ = ops.conv2d_bf16(ag, wp, 3, 1, 1, Cout, mode0=1)
    close_bf16(to_nchw(y3), plain)


@pytest.mark.parametrize("h,w_,Cin,Cout", [(20, 24, 32, 16), (21, 19, 32, 16), (64, 64, 16, 16), (9, 40, 16, 32), (4, 16, 32, 32)])
def test_conv_bf16_upsampled_data_gradient_fused(h, w_, Cin, Cout):
    """dec4.conv1 backward under AMP in one launch (the narrow kernel's up-sample-backward epilogue): gradient of x for
    conv3x3(interpolate(x, 2, nearest)) with the 2x2 sums on the fp32 accumulators + the BatchNorm-backward sums of the layer
    that produced x — against float64 autograd on the same bf16-rounded operands; ragged maps; the sums against the
    stored (rounded) gradient; run-to-run identical"""
    ops = _ops()
    g = torch.Generator().manual_seed(377 + h + Cout)
    B = 2
    yraw = torch.randn((B, Cin, h, w_), generator=g)
    yg, y64 = bf(yraw)
    sc = 1 + 0.3 * torch.randn(Cin, generator=g)
    sh = 0.2 * torch.randn(Cin, generator=g)
    mu = 0.1 * torch.randn(Cin, generator=g)
    istd = 1 + 0.2 * torch.rand(Cin, generator=g)
    wt = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.08).to(BF)
    x = torch.randn((B, Cin, h, w_), generator=g, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt.double(), padding=1)
    dy = torch.randn(out.shape, generator=g)
    dyg, dy64 = bf(dy)
    out.backward(dy64)
    wd = ops.pack_weights_bf16(wt.float().permute(2, 3, 1, 0).contiguous().to(DEV), dgrad=True)
    args = (dyg, wd, Cin, yg, mu.to(DEV), istd.to(DEV), sc.to(DEV), sh.to(DEV))
    gx, red = ops.conv2d_bf16_upsampled_dgrad(*args)
    close_bf16(to_nchw(gx), x.grad)
    gx2, red2 = ops.conv2d_bf16_upsampled_dgrad(*args)
    assert torch.equal(gx, gx2) and torch.equal(red, red2)
    # sums over the STORED gradient, mask from bf16(y * scale + shift) like dt_bn_bwd_reduce_bf16
    act = (y64.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).float()
    gm = torch.where(act > 0, to_nchw(gx), torch.zeros((), dtype=torch.float64))
    xhat = (y64 - mu.double()[None, :, None, None]) * istd.double()[None, :, None, None]
    tol = 1e-4 * float(gm.abs().sum(dim=(0, 2, 3)).max())
    np.testing.assert_allclose(red[0].double().sum(0).cpu(), gm.sum(dim=(0, 2, 3)), rtol=1e-4, atol=tol)
    np.testing.assert_allclose(red[1].double().sum(0).cpu(), (gm * xhat).sum(dim=(0, 2, 3)), rtol=1e-4, atol=3 * tol)


@pytest.mark.parametrize("k,p", [(3, 1), (1, 0)])
def test_conv_bf16_stride2_data_gradient(k, p):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 2, 16, 24, 64, 128
    wt = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).to(BF).double()
    x = torch.randn((B, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, wt, stride=2, padding=p)
    dy = torch.randn(y.shape, generator=g)
    dyg, dy64 = bf(dy)
    y.backward(dy64)
    wd = ops.pack_weights_bf16(wt.float().permute(2, 3, 1, 0).contiguous().to(DEV), dgrad=True)
    dx, _, _ = ops.conv2d_bf16(dyg, wd, k, 1, k - 1 - p, Cin, mode0=2)
    close_bf16(to_nchw(dx), x.grad)


WG_CASES = [(2, 32, 32, 64, 64, 3, 1, 1), (2, 16, 16, 128, 256, 3, 1, 1), (1, 40, 24, 16, 16, 3, 1, 1),
            (2, 32, 64, 32, 16, 3, 1, 1), (2, 8, 8, 64, 32, 3, 1, 1), (2, 32, 32, 64, 128, 3, 2, 1),
            (2, 32, 32, 64, 128, 1, 2, 0)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", WG_CASES)
def test_wgrad_bf16(B, H, W, Cin, Cout, k, s, p):
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + Cin * 7 + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    xg, x64 = bf(x)
    wt = (torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(x64, wt, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    dyg, dy64 = bf(dy)
    y.backward(dy64)
    dw = ops.conv2d_wgrad_bf16(xg, dyg, k, s, p)
    got = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((got - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-5


@pytest.mark.parametrize("C0,C1,Cout,h,w_", [(64, 64, 32, 8, 8), (32, 0, 16, 20, 24), (16, 16, 32, 8, 40)])
def test_wgrad_bf16_upsample_concat_transform(C0, C1, Cout, h, w_):
    """decoder shapes incl. the 32-channel-block kernel (Cin, Cout <= 32: waves split the pixel rows)"""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    B = 2
    a = torch.randn((B, C0, h, w_), generator=g)
    skip = torch.randn((B, max(C1, 8), 2 * h, 2 * w_), generator=g)[:, :C1]
    sc = 1 + 0.3 * torch.randn(C0, generator=g)
    sh = 0.3 * torch.randn(C0, generator=g) + 0.4
    ag, a64 = bf(a)
    sg, s64 = bf(skip)
    z = F.relu(a64.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).double()
    wt = (torch.randn((Cout, C0 + C1, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(torch.cat([F.interpolate(z, scale_factor=2, mode="nearest"), s64], 1), wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    dyg, dy64 = bf(dy)
    y.backward(dy64)
    dw = ops.conv2d_wgrad_bf16(ag, dyg, 3, 1, 1, src1=sg if C1 else None, mode0=1, in_scale=sc.to(DEV),
                               in_shift=sh.to(DEV))
    got = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((got - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-5


@pytest.mark.parametrize("C0,Cout,h,w_,mode0,tf", [(16, 16, 40, 64, 0, False), (32, 16, 20, 24, 1, True), (32, 32, 33, 35, 0, True),
                                                   (16, 32, 16, 32, 0, False), (32, 16, 64, 64, 1, False), (16, 16, 128, 128, 0, True)])
def test_wgrad_bf16_narrow_kernel(C0, Cout, h, w_, mode0, tf):
    """the lean weight-gradient kernel of the narrow decoder layers (conv_bf16_narrow.hip) against fp64 on the same
    bf16-rounded operands: every channel combination, ragged maps, up-sampled input, fused producer BatchNorm + ReLU,
    several tiles per persistent workgroup; run-to-run bit-identical"""
    ops = _ops()
    g = torch.Generator().manual_seed(C0 * 3 + Cout + h)
    B = 5 if h >= 64 else 2
    a = torch.randn((B, C0, h, w_), generator=g)
    ag, a64 = bf(a)
    sc = sh = None
    z = a64
    if tf:
        sc = 1 + 0.3 * torch.randn(C0, generator=g)
        sh = 0.3 * torch.randn(C0, generator=g) + 0.3
        z = F.relu(a64.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).double()
    xin = F.interpolate(z, scale_factor=2, mode="nearest") if mode0 else z
    wt = (torch.randn((Cout, C0, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(xin, wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    dyg, dy64 = bf(dy)
    y.backward(dy64)
    kw = dict(mode0=mode0, in_scale=sc.to(DEV) if tf else None, in_shift=sh.to(DEV) if tf else None)
    dw = ops.conv2d_wgrad_bf16(ag, dyg, 3, 1, 1, **kw)
    got = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((got - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-5
    assert torch.equal(dw, ops.conv2d_wgrad_bf16(ag, dyg, 3, 1, 1, **kw))


WG_DMA_CASES = [  # B, h, w (stored size of source 0), C0, C1, mode0, Cout — the LDS-DMA persistent weight-gradient kernel
    (2, 32, 32, 64, 0, 0, 64), (2, 37, 50, 64, 0, 0, 128), (3, 16, 16, 128, 0, 0, 64), (2, 8, 8, 64, 0, 0, 64),
    (5, 13, 11, 64, 0, 0, 64), (1, 12, 10, 128, 64, 1, 64), (2, 8, 8, 64, 64, 1, 128), (1, 20, 44, 64, 64, 0, 64),
    # many tiles per persistent workgroup (2,304 tiles on <= 256 workgroups), several ci / co blocks
    (9, 64, 128, 64, 0, 0, 64), (2, 64, 64, 192, 0, 0, 128)]


@pytest.mark.parametrize("B,h,w_,C0,C1,mode0,Cout", WG_DMA_CASES)
def test_wgrad_bf16_dma_kernel(B, h, w_, C0, C1, mode0, Cout):
    """conv_bf16_wgrad_dma.hip against fp64 on the same bf16-rounded operands: plain / concatenated / up-sampled inputs,
    ragged maps (zero padding = out-of-range DMA offsets), maps of at most 16 pixels (the 8 x 16 tile form), tile ranges
    of several tiles per workgroup; run-to-run bit-identical (no atomics)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + h + C0 + Cout)
    a = torch.randn((B, C0, h, w_), generator=g)
    H, W = (2 * h, 2 * w_) if mode0 else (h, w_)
    ag, a64 = bf(a)
    xin = F.interpolate(a64, scale_factor=2, mode="nearest") if mode0 else a64
    sg = None
    if C1:
        skip = torch.randn((B, C1, H, W), generator=g)
        sg, s64 = bf(skip)
        xin = torch.cat([xin, s64], 1)
    wt = (torch.randn((Cout, C0 + C1, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(xin, wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    dyg, dy64 = bf(dy)
    y.backward(dy64)
    dw = ops.conv2d_wgrad_bf16(ag, dyg, 3, 1, 1, src1=sg, mode0=mode0)
    got = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((got - wt.grad).abs().max() / wt.grad.abs().max()) < 2e-5
    assert torch.equal(dw, ops.conv2d_wgrad_bf16(ag, dyg, 3, 1, 1, src1=sg, mode0=mode0))


def test_bf16_training_step_against_fp32_path():
    """configs[2]: bf16 activations/weights, fp32 accumulate, fp32 master weights.  One step on the same batch as
    the fp32 HIP path.  Batch-statistics BatchNorm re-normalises every layer, so bf16 rounding (2^-9 of a channel's
    MEAN) is amplified in channels whose batch sigma is small: logits differ by ~10 % in relative L2 (1.2 % in eval
    mode, tests/test_model_gpu.py) while the averages agree — measured with scripts/diag_bf16.py: loss within
    3e-4, gradient norm within 1 %, gradient cosine 0.96 at B=4/256x256.  Every bf16 kernel is checked exactly
    above; this test bounds the end-to-end drift and checks that optimisation works."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.loss.seg_loss import seg_loss
    from deadtrees_amd.network.unet import UNetHIP
    from deadtrees_amd.trainer import HipTrainer
    from oracle.unet_ref import make_oracle
    ref = make_oracle(3, 2, seed=0)
    img, mask = synth_batch(4, 256, 256, 3, 2, seed=3)
    img, mask = img.to(DEV), mask.to(DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        m = UNetHIP()
        m.load_state_dict(ref.state_dict())
        m.to(DEV).train()
        m.precision = prec
        logits = m(img)
        loss, _, err = seg_loss(logits, mask, None, ("GDICE", "FOCAL"))
        loss.backward()
        res[prec] = (float(loss.detach()), m._grad_buffer().clone(), m.bn_state.clone(), logits.detach().clone())
    l32, g32, bn32, lg32 = res["fp32"]
    l16, g16, bn16, lg16 = res["bf16"]
    assert l16 == pytest.approx(l32, rel=3e-3)
    assert float((lg16 - lg32).norm() / lg32.norm()) < 0.25
    cos = float((g16.double() * g32.double()).sum() / (g16.double().norm() * g32.double().norm()))
    assert cos > 0.9, cos
    assert float(g16.norm()) == pytest.approx(float(g32.norm()), rel=5e-2)
    assert float((bn16 - bn32).abs().max() / bn32.abs().max()) < 2e-2
    m = UNetHIP()
    m.load_state_dict(ref.state_dict())
    m.to(DEV)
    tr = HipTrainer(m, precision="bf16")
    losses = [float(tr.step(img, mask)) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # the bf16 weight images follow the fp32 master weights after the fused optimiser step
    am_a = m.predict_classes(img, precision="bf16")
    am_b = m.predict_classes(img, precision="fp32")
    assert float((am_a == am_b).float().mean()) > 0.97


# ---------------------------------------------------------------- bf16 elementwise kernels, one by one
def _bf(t):
    """round to bf16; returns (device bf16 NHWC-as-given tensor, fp64 copy of the rounded values)"""
    q = t.to(BF)
    return q.to(DEV), q.double()


@pytest.mark.parametrize("n_pix,C", [(2 * 24 * 40, 16), (3 * 64 * 64, 64), (700, 512), (2 * 512 * 300, 32)])
@pytest.mark.parametrize("mode", ["stored_act", "virtual_act", "linear"])
def test_bn_backward_bf16(n_pix, C, mode):
    """dt_bn_bwd_reduce_bf16 / _apply_bf16 against the fp64 batch-norm backward on the same bf16 operands:
    dgamma/dbeta to fp32-reduction accuracy, dy and the residual gradient to one bf16 rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(n_pix % 97 + C)
    y = torch.randn((n_pix, C), generator=g) * (0.5 + torch.rand(C, generator=g)) + 0.3 * torch.randn(C, generator=g)
    dout = torch.randn((n_pix, C), generator=g)
    gamma = 1 + 0.3 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    yg, y64 = _bf(y)
    dg, d64 = _bf(dout)
    mean = y64.mean(0)
    var = y64.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    sc, sh = gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd
    act, asc, ash = None, None, None
    gm = d64
    if mode != "linear":
        z = torch.relu((y64.float() * sc.float() + sh.float())).to(BF)   # what the forward stored / consumers saw
        gm = torch.where(z.double() > 0, d64, torch.zeros_like(d64))
        if mode == "stored_act":
            act = z.to(DEV)
        else:
            asc, ash = sc.float().to(DEV), sh.float().to(DEV)
    xh = (y64 - mean) * invstd
    dbeta = gm.sum(0)
    dgamma = (gm * xh).sum(0)
    dy = gamma.double() * invstd * (gm - dbeta / n_pix - xh * dgamma / n_pix)
    prev = torch.randn((n_pix, C), generator=g).to(BF)
    got_dy, got_dg, got_db, got_res = ops.bn_backward_bf16(
        dg, act, yg, mean.float().to(DEV), invstd.float().to(DEV), gamma.to(DEV), act_scale=asc, act_shift=ash,
        dres=prev.clone().to(DEV))
    tol = 3e-5 * float(gm.abs().sum(0).max()) + 1e-4
    assert float((got_db.cpu().double() - dbeta).abs().max()) < tol
    assert float((got_dg.cpu().double() - dgamma).abs().max()) < 4 * tol
    close_bf16(got_dy.cpu().double(), dy, extra=1e-4)
    close_bf16(got_res.cpu().double(), prev.double() + gm)


@pytest.mark.parametrize("C", [16, 64, 256])
def test_bn_act_bf16(C):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    y = torch.randn((2, 20, 24, C), generator=g)
    res = torch.randn((2, 20, 24, C), generator=g)
    sc, sh = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    rsc, rsh = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    yg, y64 = _bf(y)
    rg, r64 = _bf(res)
    want = torch.relu(y64 * sc.double() + sh.double() + r64 * rsc.double() + rsh.double())
    got = ops.bn_act_bf16(yg, sc.to(DEV), sh.to(DEV), rg, rsc.to(DEV), rsh.to(DEV))
    close_bf16(got.cpu().double(), want)
    got32 = ops.bn_act_bf16(y.to(DEV), sc.to(DEV), sh.to(DEV), relu=False)      # fp32 input (stem), no residual
    close_bf16(got32.cpu().double(), y.double() * sc.double() + sh.double())


@pytest.mark.parametrize("H,W", [(26, 30), (25, 31)])
def test_maxpool_and_upsample_backward_bf16(H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 16, H, W), generator=g)                    # NCHW reference layout
    xg = x.to(BF).permute(0, 2, 3, 1).contiguous().to(DEV)
    x64 = x.to(BF).double().requires_grad_(True)
    want = F.max_pool2d(x64, 3, 2, 1)
    pooled, am = ops.maxpool3x3s2_bf16(xg)
    assert torch.equal(pooled.cpu().permute(0, 3, 1, 2).double(), want.detach())
    d = torch.randn(want.shape, generator=g).to(BF)
    want.backward(d.double())
    dx = ops.maxpool3x3s2_bwd_bf16(d.permute(0, 2, 3, 1).contiguous().to(DEV), am, H, W)
    close_bf16(dx.cpu().permute(0, 3, 1, 2).double(), x64.grad)
    prev = torch.randn(x.shape, generator=g).to(BF)
    dx2 = ops.maxpool3x3s2_bwd_bf16(d.permute(0, 2, 3, 1).contiguous().to(DEV), am, H, W,
                                    dx=prev.permute(0, 2, 3, 1).contiguous().to(DEV))
    close_bf16(dx2.cpu().permute(0, 3, 1, 2).double(), x64.grad + prev.double())
    up = torch.randn((2, 24, 20, 32), generator=g).to(BF)          # NHWC
    got = ops.upsample2x_bwd_bf16(up.to(DEV))
    want_up = up.double().reshape(2, 12, 2, 10, 2, 32).sum(dim=(2, 4))
    close_bf16(got.cpu().double(), want_up)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 40, 64, 64), (1, 24, 40, 16, 16), (2, 20, 36, 32, 32),
                                            (2, 9, 70, 128, 96), (16, 120, 128, 48, 128)])
def test_conv_bf16_with_fused_bn_backward_reduction(B, H, W, Cin, Cout):
    """dt_conv2d_bf16_bn_bwd: output bit-identical to dt_conv2d_bf16; partial sums = BatchNorm-backward reduction of
    the rounded gradient with the mask of bf16(y*scale+shift) (what dt_bn_bwd_reduce_bf16 computes)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 7 + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g).to(BF)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(BF)
    y = (torch.randn((B, Cout, H, W), generator=g) * 1.5 + 0.2).to(BF)
    mean = y.float().mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.float().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)  # noqa: E731
    wp = ops.pack_weights_bf16(w.float().permute(2, 3, 1, 0).contiguous().to(DEV))
    plain, _, _ = ops.conv2d_bf16(nh(x), wp, 3, 1, 1, Cout)
    out, red = ops.conv2d_bf16_bn_bwd(nh(x), wp, Cout, nh(y), mean.to(DEV), invstd.to(DEV), sc.to(DEV), sh.to(DEV))
    assert torch.equal(out, plain)
    dz = out.float().cpu().permute(0, 3, 1, 2).double()
    act = (y.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).float()
    gm = torch.where(act > 0, dz, torch.zeros_like(dz))
    xh = (y.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    sums = red.sum(dim=1).cpu().double()
    scale = float(gm.abs().sum(dim=(0, 2, 3)).max()) + 1.0
    assert float((sums[0] - gm.sum(dim=(0, 2, 3))).abs().max()) <= 3e-5 * scale
    assert float((sums[1] - (gm * xh).sum(dim=(0, 2, 3))).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("B,H,W,Cin", [(2, 64, 96, 3), (1, 128, 128, 4), (3, 34, 70, 3), (2, 512, 512, 3)])
def test_stem_on_bf16_kernels_via_space_to_depth(B, H, W, Cin):
    """7x7 / stride 2 / pad 3 stem as a 4x4 window over the space-to-depth image: equals the fp64 convolution of the
    bf16-rounded image and weights (one bf16 rounding of the result), statistics from the fp32 accumulators."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((64, Cin, 7, 7), generator=g) * (2.0 / (Cin * 49)) ** 0.5
    ref = F.conv2d(x.to(BF).double(), w.to(BF).double(), stride=2, padding=3)
    y, stats = ops.stem_conv_bf16(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(2, 3, 1, 0).contiguous().to(DEV),
                                  want_stats=True)
    assert tuple(y.shape) == (B, H // 2, W // 2, 64)
    close_bf16(to_nchw(y), ref)
    st = stats.sum(dim=1).cpu().double()
    assert float((st[0] - ref.sum(dim=(0, 2, 3))).abs().max()) <= 1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max())
    assert float((st[1] - (ref * ref).sum(dim=(0, 2, 3))).abs().max()) <= 1e-4 * float((ref * ref).sum(dim=(0, 2, 3)).max())


@pytest.mark.parametrize("B,H,W,Cin", [(2, 64, 96, 3), (1, 128, 128, 4), (3, 34, 70, 3)])
def test_stem_weight_gradient_on_bf16_kernels(B, H, W, Cin):
    """stem dW through the space-to-depth form (16 taps split over two workgroups, unpacked to 7x7) against fp64
    autograd on the bf16-rounded image and gradient"""
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + Cin + 1)
    x = torch.randn((B, Cin, H, W), generator=g)
    wt = (torch.randn((64, Cin, 7, 7), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(x.to(BF).double(), wt, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g).to(BF)
    y.backward(dy.double())
    dw = ops.stem_wgrad_bf16(x.permute(0, 2, 3, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    got = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((got - wt.grad).abs().max() / wt.grad.abs().max()) < 3e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 40, 64, 64), (2, 9, 70, 128, 96), (16, 120, 128, 48, 128)])
def test_conv_bf16_gradient_join_with_fused_bn_backward_reduction(B, H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 5 + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g).to(BF)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(BF)
    y = (torch.randn((B, Cout, H, W), generator=g) * 1.5 + 0.2).to(BF)
    z = torch.relu(torch.randn((B, Cout, H, W), generator=g)).to(BF)
    base = torch.randn((B, Cout, H, W), generator=g).to(BF)
    mean = y.float().mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.float().var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)  # noqa: E731
    wp = ops.pack_weights_bf16(w.float().permute(2, 3, 1, 0).contiguous().to(DEV))
    plain, _, _ = ops.conv2d_bf16(nh(x), wp, 3, 1, 1, Cout, out0=nh(base), accumulate=True)
    out, red = ops.conv2d_bf16_bn_bwd(nh(x), wp, Cout, nh(y), mean.to(DEV), invstd.to(DEV), act=nh(z), join_into=nh(base))
    assert torch.equal(out, plain)
    dz = out.float().cpu().permute(0, 3, 1, 2).double()
    gm = torch.where(z.double() > 0, dz, torch.zeros_like(dz))
    xh = (y.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    sums = red.sum(dim=1).cpu().double()
    scale = float(gm.abs().sum(dim=(0, 2, 3)).max()) + 1.0
    assert float((sums[0] - gm.sum(dim=(0, 2, 3))).abs().max()) <= 3e-5 * scale
    assert float((sums[1] - (gm * xh).sum(dim=(0, 2, 3))).abs().max()) <= 1e-4 * scale


# ---------------------------------------------------------------- LDS-DMA staged 512-pixel kernel (conv_bf16_dma.hip)
def _set_dma(mode):
    from deadtrees_amd import _lib
    _lib.check(_lib.load().dt_set_option(b"bf16_dma", mode), "dt_set_option")


DMA_CASES = [  # B, H, W (stored size of source 0), C0, C1, mode0, Cout, split, transform
    (2, 32, 32, 64, 0, 0, 64, 0, False), (1, 34, 70, 512, 0, 0, 64, 0, True), (3, 17, 33, 32, 0, 0, 128, 0, False),
    (2, 16, 20, 64, 32, 1, 128, 0, True), (2, 16, 24, 128, 64, 1, 64, 0, False), (2, 40, 48, 64, 0, 0, 192, 64, False),
    (1, 64, 64, 96, 0, 2, 64, 0, False),
    # more tiles than persistent workgroups (640 / 320 > 256): BatchNorm partial sums accumulated over a workgroup's tiles
    (40, 64, 64, 32, 0, 0, 128, 0, False), (40, 64, 64, 64, 0, 0, 64, 0, True)]


@pytest.mark.parametrize("B,H,W,C0,C1,mode0,Cout,split,tf", DMA_CASES)
def test_conv_bf16_dma_kernel_is_bit_identical_to_the_register_staged_kernel(B, H, W, C0, C1, mode0, Cout, split, tf):
    """the LDS-DMA staged, double-buffered kernel against the register-staged one on the same operands: same MFMA
    sequence per output element -> bit-identical outputs and BatchNorm partial sums (hence every parity test of the
    old kernel carries over); ragged right / bottom edges, several images, 1..16 channel chunks, the producer's
    BatchNorm + ReLU fused into the staging, nearest x2 upsample + concat, zero insertion, split outputs + gradient join."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H + C0 + Cout)
    Hin, Win = (H, W) if mode0 == 0 else (2 * H, 2 * W)
    x = torch.randn((B, H, W, C0), generator=g).to(BF).to(DEV)   # mode0 1 / 2: stored at half resolution
    s1 = torch.randn((B, Hin, Win, C1), generator=g).to(BF).to(DEV) if C1 else None
    w = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).to(DEV)
    wp = ops.pack_weights_bf16(w)
    sc = (1 + 0.3 * torch.randn(C0, generator=g)).to(DEV) if tf else None
    sh = (0.3 * torch.randn(C0, generator=g) + 0.2).to(DEV) if tf else None
    res = {}
    try:
        for mode in (0, 2):
            _set_dma(mode)
            kw = dict(src1=s1, mode0=mode0, in_scale=sc, in_shift=sh)
            if split:
                base0 = torch.randn((B, Hin, Win, split), generator=torch.Generator().manual_seed(1)).to(BF).to(DEV)
                o0, o1, _ = ops.conv2d_bf16(x, wp, 3, 1, 1, Cout, split=split, out0=base0.clone(), accumulate=True, **kw)
                res[mode] = (o0, o1, None)
            else:
                o0, _, st = ops.conv2d_bf16(x, wp, 3, 1, 1, Cout, want_stats=True, **kw)
                res[mode] = (o0, None, st.sum(1))      # tile sizes differ: compare the per-channel totals
    finally:
        _set_dma(1)
    a, bq = res[0], res[2]
    # the register-staged variants with 16-channel chunks (narrow layers, 512-pixel tiles) accumulate the same products
    # in another order: equal up to fp32 summation order there, bit-identical against the 32-channel-chunk variants
    import ctypes as C
    from deadtrees_amd import _lib
    d = ops.conv_desc(B, Hin, Win, C0, C1, mode0, Cout, 3, 1, 1, split, 1 if split else 0)
    tw, tn, ck, mt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _set_dma(0)
    _lib.load().dt_conv2d_bf16_config(C.byref(d), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt))
    _set_dma(2)
    _lib.load().dt_conv2d_bf16_config(C.byref(d), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt))
    assert mt.value == 8                                    # the DMA kernel really ran in mode 2
    _set_dma(0)
    _lib.load().dt_conv2d_bf16_config(C.byref(d), C.byref(tw), C.byref(tn), C.byref(ck), C.byref(mt))
    _set_dma(1)
    same_order = ck.value == 32

    def check(x, y):
        if same_order:
            assert torch.equal(x, y)
        else:
            xf, yf = x.float(), y.float()
            assert float((xf - yf).abs().max()) <= 2.0 ** -7 * float(yf.abs().max())
            assert float((x != y).float().mean()) < 0.02     # only accumulation-order rounding flips
    check(a[0], bq[0])
    if a[1] is not None:
        check(a[1], bq[1])
    if a[2] is not None:
        np.testing.assert_allclose(a[2].cpu().numpy(), bq[2].cpu().numpy(), rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("join,B,H,W", [(False, 2, 40, 36), (True, 2, 40, 36), (False, 24, 64, 64), (True, 24, 64, 64)])
def test_conv_bf16_dma_kernel_fused_bn_backward_sums(join, B, H, W):
    """dt_conv2d_bf16_bn_bwd on the DMA kernel: gradient bit-identical, BatchNorm-backward partial sums equal to the
    register-staged kernel's per-channel totals (virtual activation / stored activation + gradient join); the large case
    has 384 tiles for 256 persistent workgroups (sums carried over a workgroup's tiles, one row per workgroup)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5 + join)
    Cin, Cout = 64, 128
    dy = torch.randn((B, H, W, Cin), generator=g).to(BF).to(DEV)
    wp = ops.pack_weights_bf16((torch.randn((3, 3, Cin, Cout), generator=g) * 0.05).to(DEV), dgrad=False)
    y = (torch.randn((B, H, W, Cout), generator=g) * 1.5 + 0.2).to(BF).to(DEV)
    mean = y.float().mean(dim=(0, 1, 2)).contiguous()
    invstd = (1.0 / torch.sqrt(y.float().var(dim=(0, 1, 2), unbiased=False) + 1e-5)).contiguous()
    sc = (1 + 0.2 * torch.randn(Cout, generator=g)).to(DEV)
    sh = (0.2 * torch.randn(Cout, generator=g)).to(DEV)
    act = torch.relu(y.float() * sc + sh).to(BF) if join else None
    base = torch.randn((B, H, W, Cout), generator=g).to(BF).to(DEV) if join else None
    res = {}
    try:
        for mode in (0, 2):
            _set_dma(mode)
            out, red = ops.conv2d_bf16_bn_bwd(dy, wp, Cout, y, mean, invstd, None if join else sc, None if join else sh,
                                              act=act, join_into=base.clone() if join else None)
            res[mode] = (out, red.sum(1))
    finally:
        _set_dma(1)
    assert torch.equal(res[0][0], res[2][0])
    np.testing.assert_allclose(res[0][1].cpu().numpy(), res[2][1].cpu().numpy(), rtol=2e-5, atol=2e-3)


@pytest.mark.parametrize("K,B,H,W", [(2, 2, 24, 40), (3, 1, 21, 37), (2, 1, 70, 130)])
def test_head_backward_bf16(K, B, H, W):
    """dt_head_bwd_bf16 (bf16 decoder activation in, bf16 gradient out, fp32 dW / dbias; MFMA kernel of head_loss.hip)
    against fp64 on the same bf16-rounded activation: dx within one bf16 rounding, dW / dbias to fp32 accuracy"""
    import ctypes as C
    from deadtrees_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(10 * K + B)
    xq, x64 = bf(torch.randn((B, 16, H, W), generator=g))
    x64 = x64.requires_grad_(True)
    w = (torch.randn((K, 16, 3, 3), generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.zeros(K, dtype=torch.float64, requires_grad=True)
    dl = torch.randn((B, K, H, W), generator=g)
    F.conv2d(x64, w, b, padding=1).backward(dl.double())
    w_ohwi = w.detach().permute(0, 2, 3, 1).contiguous().float().to(DEV)
    dlg = dl.to(DEV)
    dx = torch.empty_like(xq)
    red = torch.empty(lib.dt_head_bwd_red_floats(B, H, W, 16, K), device=DEV)
    dw, db = torch.empty(K * 9 * 16, device=DEV), torch.empty(K, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.dt_head_bwd_bf16(p(xq), p(w_ohwi), p(dlg), p(dx), p(red), B, H, W, 16, K, st), "dt_head_bwd_bf16")
    _lib.check(lib.dt_head_bwd_finalize(p(red), lib.dt_head_bwd_rows(B, H, W), p(dw), p(db), 16, K, st), "dt_head_bwd_finalize")
    close_bf16(to_nchw(dx), x64.grad)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(-1)
    assert float((dw.cpu().double() - ref_dw).abs().max()) <= 1e-5 * float(ref_dw.abs().max())
    assert float((db.cpu().double() - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max())
