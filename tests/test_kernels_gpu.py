"""Per-kernel parity of the C ABI (through deadtrees_amd.ops) against torch CPU fp64 primitives.
All tests need an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from deadtrees_amd import ops
    return ops


def nhwc(t):  # NCHW cpu -> NHWC gpu f32
    return t.permute(0, 2, 3, 1).contiguous().float().to(DEV)


def to_nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).double()


def hwio(w):  # OIHW -> HWIO gpu
    return w.permute(2, 3, 1, 0).contiguous().float().to(DEV)


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 32, 32, 64, 64, 3, 1, 1),
    (1, 40, 24, 16, 16, 3, 1, 1),     # ragged tiles, narrow channels (dec.4 shape)
    (2, 16, 16, 128, 96, 3, 1, 1),    # TW=16 path, Cout not multiple of 64
    (3, 8, 8, 32, 48, 3, 1, 1),       # TW=8 path
    (2, 2, 2, 64, 64, 3, 1, 1),       # tiny map (layer4 of a 64x64 tile)
    (2, 32, 32, 64, 128, 3, 2, 1),    # layer2.0.conv1
    (2, 12, 20, 32, 64, 3, 2, 1),
    (2, 32, 32, 64, 128, 1, 2, 0),    # downsample
    (2, 64, 96, 3, 64, 7, 2, 3),      # stem
    (1, 32, 32, 4, 64, 7, 2, 3),      # stem, RGBN
    (1, 34, 70, 512, 64, 3, 1, 1),    # many input chunks, width > 2 tiles
    (2, 48, 80, 32, 16, 3, 1, 1),     # dec.4.conv1 shape: 16-wide MFMA kernel, two input chunks
    (1, 24, 40, 8, 12, 3, 1, 1),      # 16-wide kernel with ragged channel counts
    # round 3: the lean persistent kernel of the narrow layers (conv3x3_f32_narrow_kernel: Cin, Cout in {16, 32}, maps of
    # at least 8 x 32) — every channel combination, ragged edges, many tiles per persistent workgroup
    (2, 40, 64, 16, 16, 3, 1, 1), (2, 37, 70, 32, 16, 3, 1, 1), (3, 64, 64, 32, 32, 3, 1, 1), (2, 16, 32, 16, 32, 3, 1, 1),
    (6, 256, 256, 16, 16, 3, 1, 1), (5, 128, 256, 32, 32, 3, 1, 1),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", CONV_CASES)
def test_conv_fwd_and_stats(B, H, W, Cin, Cout, k, s, p):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p)
    y, _, stats = ops.conv2d(nhwc(x), hwio(w), k, s, p, want_stats=True)
    torch.cuda.synchronize()
    got = to_nchw(y)
    assert got.shape == ref.shape
    # sequential fp32 fma chain of K = k*k*Cin terms: error grows ~ sqrt(K) * 2^-24
    assert rel_err(got, ref) < 2e-6 * max(1.0, (k * k * Cin / 1000.0) ** 0.5)
    s1 = stats[0].double().sum(0).cpu()
    s2 = stats[1].double().sum(0).cpu()
    np.testing.assert_allclose(s1, ref.sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-4 * ref.abs().sum(dim=(0, 2, 3)).max())
    np.testing.assert_allclose(s2, (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-5)


def test_conv_upsample_concat_split_accumulate():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, h, w_, C0, C1, Cout = 2, 10, 12, 64, 32, 96
    a = torch.randn((B, C0, h, w_), generator=g)
    skip = torch.randn((B, C1, 2 * h, 2 * w_), generator=g)
    wt = torch.randn((Cout, C0 + C1, 3, 3), generator=g) * 0.05
    xin = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), skip], dim=1)
    ref = F.conv2d(xin.double(), wt.double(), padding=1)
    y, _, _ = ops.conv2d(nhwc(a), hwio(wt), 3, 1, 1, src1=nhwc(skip), mode0=1)
    assert rel_err(to_nchw(y), ref) < 2e-6
    # split outputs at channel 64 + accumulate into out0
    base = torch.randn((B, 2 * h, 2 * w_, 64), generator=g).to(DEV)
    o0, o1, _ = ops.conv2d(nhwc(a), hwio(wt), 3, 1, 1, src1=nhwc(skip), mode0=1, split=64, out0=base.clone(),
                           accumulate=True)
    assert rel_err(to_nchw(o0), ref[:, :64] + to_nchw(base)) < 2e-6
    assert rel_err(to_nchw(o1), ref[:, 64:]) < 2e-6


@pytest.mark.parametrize("Cin,Cout,k,p", [(64, 128, 3, 1), (64, 128, 1, 0)])
def test_conv_dgrad_stride2_via_zero_insert(Cin, Cout, k, p):
    """data gradient of a stride-2 conv = stride-1 conv over the zero-inserted dy with flipped/transposed w."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 16, 24
    x = torch.randn((B, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) * 0.05
    y = F.conv2d(x, wt, stride=2, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    wd = ops.weight_flip_transpose(hwio(wt))
    dx, _, _ = ops.conv2d(nhwc(dy), wd, k, 1, k - 1 - p, mode0=2)
    assert rel_err(to_nchw(dx), x.grad) < 2e-6


@pytest.mark.parametrize("h,w_,Cin,Cout", [(20, 24, 32, 16), (21, 19, 32, 16), (64, 64, 16, 16), (9, 40, 16, 32), (4, 16, 32, 16)])
def test_conv_n16_with_virtual_upsample(h, w_, Cin, Cout):
    """dec.4.conv1: nearest x2 upsample (no skip) feeding the narrow-layer kernels, with BN statistics.  Round 3: the
    sub-pixel form (conv3x3_f32_upc_kernel: 4 combined taps per output parity from the low-resolution halo) — ragged
    maps, borders (the padding of the up-sampled image = out-of-range source pixels), every channel combination it takes"""
    ops = _ops()
    g = torch.Generator().manual_seed(77 + h)
    B = 2
    a = torch.randn((B, Cin, h, w_), generator=g)
    wt = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.08
    ref = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest").double(), wt.double(), padding=1)
    y, _, stats = ops.conv2d(nhwc(a), hwio(wt), 3, 1, 1, mode0=1, want_stats=True)
    assert rel_err(to_nchw(y), ref) < 2e-6
    np.testing.assert_allclose(stats[0].double().sum(0).cpu(), ref.sum(dim=(0, 2, 3)), rtol=1e-5,
                               atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()))
    np.testing.assert_allclose(stats[1].double().sum(0).cpu(), (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("B,h,w_,Cin,Cout", [(5, 16, 16, 64, 128), (8, 16, 16, 256, 512), (2, 13, 11, 32, 64), (3, 16, 9, 16, 32)])
def test_conv_stride2_small_maps_share_a_tile(B, h, w_, Cin, Cout):
    """layer4.0.conv1 of a 256-pixel tile (16x16 -> 8x8): four images share one 8 x 32-pixel tile of the stride-2 kernel
    (ConvArgs::pack) — image borders stay zero padding (no bleed between neighbours in the tile), ragged groups (B % 4),
    maps smaller than 8 x 8, BatchNorm statistics over all images"""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 31 + h)
    a = torch.randn((B, Cin, h, w_), generator=g)
    wt = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.05
    ref = F.conv2d(a.double(), wt.double(), stride=2, padding=1)
    y, _, stats = ops.conv2d(nhwc(a), hwio(wt), 3, 2, 1, want_stats=True)
    assert tuple(y.shape) == (B, ref.shape[2], ref.shape[3], Cout)
    assert rel_err(to_nchw(y), ref) < 2e-6
    np.testing.assert_allclose(stats[0].double().sum(0).cpu(), ref.sum(dim=(0, 2, 3)), rtol=1e-5,
                               atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()))
    np.testing.assert_allclose(stats[1].double().sum(0).cpu(), (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-5)
    # one image alone (no packing: B = 1) gives the same bits as the same image inside a group
    y1, _, _ = ops.conv2d(nhwc(a[B - 1:]), hwio(wt), 3, 2, 1)
    assert torch.equal(y1[0], y[B - 1])


@pytest.mark.parametrize("h,w_,Cin,tf", [(20, 36, 32, True), (21, 33, 32, False), (64, 64, 16, True), (5, 40, 16, False), (4, 32, 32, True)])
def test_conv_upsampled_wgrad_subpixel(h, w_, Cin, tf):
    """dec.4.conv1 weight gradient in its sub-pixel form (conv3x3_wgrad_f32_upc_kernel: 16 products per source pixel folded
    to the nine taps) against float64 autograd of conv3x3(interpolate(x, 2, nearest)): ragged maps, borders, with and
    without the producer's BatchNorm + ReLU applied while staging"""
    ops = _ops()
    g = torch.Generator().manual_seed(277 + h)
    B, Cout = 3, 16
    yraw = torch.randn((B, Cin, h, w_), generator=g)
    sc = 1 + 0.3 * torch.randn(Cin, generator=g)
    sh = 0.2 * torch.randn(Cin, generator=g) + 0.3
    wt = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.07).double().requires_grad_(True)
    x = F.relu(yraw.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]) if tf else yraw.double()
    out = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, padding=1)
    dy = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dy)
    dw = ops.conv2d_wgrad(nhwc(yraw), nhwc(dy), 3, 1, 1, mode0=1, in_scale=sc.to(DEV) if tf else None,
                          in_shift=sh.to(DEV) if tf else None)
    assert rel_err(dw.cpu().permute(3, 2, 0, 1), wt.grad) < 3e-6


@pytest.mark.parametrize("k,s,p,Cin,Cout,h,w_,relu,C1,up", [
    (7, 2, 3, 3, 64, 64, 96, True, 0, False),      # stem
    (7, 2, 3, 4, 64, 37, 41, True, 0, False),      # RGBN stem, ragged
    (3, 2, 1, 64, 128, 32, 48, True, 0, False),    # layerN.0.conv1
    (3, 2, 1, 256, 512, 16, 16, True, 0, False),   # layer4.0.conv1 at a 256-pixel tile
    (1, 2, 0, 64, 128, 32, 48, False, 0, False),   # down-sample branch: BatchNorm without ReLU
    (3, 1, 1, 64, 32, 16, 24, True, 64, True),     # dec3.conv1: up-sampled + skip
    (3, 1, 1, 16, 16, 16, 32, True, 0, False),     # a narrow layer (routed to the lean kernel)
    (3, 1, 1, 16, 16, 16, 32, False, 0, False),    # ... without ReLU: the generic kernel
])
def test_conv_affine_epilogue_equals_conv_then_bn_act(k, s, p, Cin, Cout, h, w_, relu, C1, up):
    """inference: dt_conv2d_affine (eval BatchNorm [+ ReLU] on the accumulators) is bit-identical to dt_conv2d followed by
    dt_bn_act, for every kind of layer the direct kernel takes — and keeps NaN like F.relu does"""
    ops = _ops()
    g = torch.Generator().manual_seed(k * 100 + Cin)
    B = 2
    a = torch.randn((B, h, w_, Cin), generator=g).to(DEV)
    hs, ws = (2 * h, 2 * w_) if up else (h, w_)
    a1 = torch.randn((B, hs, ws, C1), generator=g).to(DEV) if C1 else None
    wt = (torch.randn((k, k, Cin + C1, Cout), generator=g) * 0.05).to(DEV)
    sc = (1 + 0.3 * torch.randn(Cout, generator=g)).to(DEV)
    sh = (0.3 * torch.randn(Cout, generator=g)).to(DEV)
    y, _, _ = ops.conv2d(a, wt, k, s, p, src1=a1, mode0=1 if up else 0)
    ref = ops.bn_act(y, sc, sh, relu=relu)
    z = ops.conv2d_affine(a, wt, k, s, p, sc, sh, relu=relu, src1=a1, mode0=1 if up else 0)
    if Cin == 16 and not relu:   # dt_conv2d runs the lean kernel here, the epilogue without ReLU the generic one: another
        assert rel_err(z, ref) < 2e-6   # (exact-fma) summation order
    else:
        assert torch.equal(z, ref)
    if relu:
        assert float(z.min()) >= 0.0
    a[0, h // 2, w_ // 2, 0] = float("nan")
    z = ops.conv2d_affine(a, wt, k, s, p, sc, sh, relu=relu, src1=a1, mode0=1 if up else 0)
    assert bool(torch.isnan(z).any())


@pytest.mark.parametrize("h,w_,Cin,Cout", [(20, 24, 32, 16), (21, 19, 32, 16), (64, 64, 16, 16), (9, 40, 16, 32), (4, 16, 32, 16)])
def test_conv_upsampled_dgrad_subpixel(h, w_, Cin, Cout):
    """dec.4.conv1 backward: autograd of conv3x3(interpolate(x, 2, nearest)) w.r.t. x in one 4x4 / stride-2 kernel
    (conv3x3_f32_upc_dgrad_kernel, 16 combined weight matrices), with the BatchNorm-backward sums of the layer that
    produced x in the epilogue — ragged maps and borders; against float64 autograd"""
    ops = _ops()
    g = torch.Generator().manual_seed(177 + h)
    B = 2
    yraw = torch.randn((B, Cin, h, w_), generator=g)
    sc = 1 + 0.3 * torch.randn(Cin, generator=g)
    sh = 0.2 * torch.randn(Cin, generator=g)
    mu = 0.1 * torch.randn(Cin, generator=g)
    istd = 1 + 0.2 * torch.rand(Cin, generator=g)
    wt = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.08
    x = F.relu(yraw.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).requires_grad_(True)
    out = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt.double(), padding=1)
    dy = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dy)
    gx, none = ops.conv2d_upsampled_dgrad(nhwc(dy), hwio(wt), Cin)
    assert none is None
    assert rel_err(to_nchw(gx), x.grad) < 2e-6
    gx2, red = ops.conv2d_upsampled_dgrad(nhwc(dy), hwio(wt), Cin, y=nhwc(yraw), mean=mu.to(DEV), invstd=istd.to(DEV),
                                          act_scale=sc.to(DEV), act_shift=sh.to(DEV))
    assert torch.equal(gx, gx2)
    # reference sums: over the masked gradient (mask = activation > 0, recomputed in fp32 like the kernel does)
    mask = (yraw * sc[None, :, None, None] + sh[None, :, None, None]) > 0
    gm = torch.where(mask, x.grad, torch.zeros_like(x.grad))
    xhat = (yraw.double() - mu.double()[None, :, None, None]) * istd.double()[None, :, None, None]
    s1, s2 = gm.sum(dim=(0, 2, 3)), (gm * xhat).sum(dim=(0, 2, 3))
    tol = 2e-5 * float(gm.abs().sum(dim=(0, 2, 3)).max())
    np.testing.assert_allclose(red[0].double().sum(0).cpu(), s1, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(red[1].double().sum(0).cpu(), s2, rtol=1e-5, atol=tol * 3)


@pytest.mark.parametrize("Cin,Cout,up", [(64, 64, False), (32, 16, True), (128, 32, True), (16, 16, False)])
def test_conv_fused_input_bn_relu(Cin, Cout, up):
    """virtual activation: conv / wgrad read relu(src*scale+shift) while staging; zero padding must stay zero"""
    ops = _ops()
    g = torch.Generator().manual_seed(Cin + Cout)
    B, h, w_ = 2, 20, 36
    yraw = torch.randn((B, Cin, h, w_), generator=g)
    sc = 1 + 0.3 * torch.randn(Cin, generator=g)
    sh = 0.2 * torch.randn(Cin, generator=g) + 0.5      # shift != 0: a transformed padding pixel would show up
    wt = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.07).double().requires_grad_(True)
    z = F.relu(yraw.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    zin = F.interpolate(z, scale_factor=2, mode="nearest") if up else z
    ref = F.conv2d(zin, wt, padding=1)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    y, _, _ = ops.conv2d(nhwc(yraw), hwio(wt.detach()), 3, 1, 1, mode0=1 if up else 0, in_scale=sc.to(DEV),
                         in_shift=sh.to(DEV))
    assert rel_err(to_nchw(y), ref.detach()) < 3e-6
    dw = ops.conv2d_wgrad(nhwc(yraw), nhwc(dy), 3, 1, 1, mode0=1 if up else 0, in_scale=sc.to(DEV), in_shift=sh.to(DEV))
    assert rel_err(dw.cpu().permute(3, 2, 0, 1), wt.grad) < 3e-6


def test_bn_backward_mask_from_raw_output():
    """BN backward with the ReLU mask recomputed from y (no stored activation) == with the stored activation"""
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    B, H, W, C = 2, 24, 40, 32
    y = torch.randn((B, H, W, C), generator=g).to(DEV)
    dout = torch.randn((B, H, W, C), generator=g).to(DEV)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    stats = torch.stack([y.sum(dim=(0, 1, 2)), (y * y).sum(dim=(0, 1, 2))]).reshape(2, 1, C).contiguous()
    mean, invstd, scale, shift = ops.bn_finalize(stats, B * H * W, gamma, beta)
    z = ops.bn_act(y, scale, shift, relu=True)
    a = ops.bn_backward(dout, z, y, mean, invstd, gamma)
    b = ops.bn_backward(dout, None, y, mean, invstd, gamma, act_scale=scale, act_shift=shift)
    for u, v in zip(a[:3], b[:3]):
        assert torch.equal(u, v)


def test_conv_dgrad_stride1():
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    B, H, W, Cin, Cout = 2, 16, 16, 32, 64
    x = torch.randn((B, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn((Cout, Cin, 3, 3), generator=g, dtype=torch.float64) * 0.05
    y = F.conv2d(x, wt, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dx, _, _ = ops.conv2d(nhwc(dy), ops.weight_flip_transpose(hwio(wt)), 3, 1, 1)
    assert rel_err(to_nchw(dx), x.grad) < 2e-6


WGRAD_CASES = [
    (2, 32, 32, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 3, 1, 1),
    (1, 40, 24, 16, 16, 3, 1, 1),
    (2, 32, 64, 32, 16, 3, 1, 1),
    (2, 8, 8, 64, 32, 3, 1, 1),
    (2, 4, 4, 32, 128, 3, 1, 1),
    (2, 32, 32, 64, 128, 3, 2, 1),
    (2, 32, 32, 64, 128, 1, 2, 0),
    (2, 64, 96, 3, 64, 7, 2, 3),
    (1, 32, 32, 4, 64, 7, 2, 3),
    (2, 36, 72, 16, 32, 3, 1, 1),     # 16-granular weight-gradient kernel, (1,2) tile arrangement
    (1, 20, 44, 12, 8, 3, 1, 1),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", WGRAD_CASES)
def test_conv_wgrad(B, H, W, Cin, Cout, k, s, p):
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + Cin * 7 + Cout)
    x = torch.randn((B, Cin, H, W), generator=g, dtype=torch.float64)
    wt = (torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dw = ops.conv2d_wgrad(nhwc(x), nhwc(dy), k, s, p)
    got = dw.detach().cpu().permute(3, 2, 0, 1).double()   # HWIO -> OIHW
    assert rel_err(got, wt.grad) < 3e-6


def test_conv_wgrad_upsample_concat():
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    B, h, w_, C0, C1, Cout = 2, 8, 8, 64, 64, 32
    a = torch.randn((B, C0, h, w_), generator=g, dtype=torch.float64)
    skip = torch.randn((B, C1, 2 * h, 2 * w_), generator=g, dtype=torch.float64)
    wt = (torch.randn((Cout, C0 + C1, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), skip], 1), wt, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dw = ops.conv2d_wgrad(nhwc(a), nhwc(dy), 3, 1, 1, src1=nhwc(skip), mode0=1)
    assert rel_err(dw.cpu().permute(3, 2, 0, 1), wt.grad) < 3e-6


@pytest.mark.parametrize("C,shape", [(64, (2, 16, 16)), (16, (3, 40, 24)), (512, (2, 2, 2))])
def test_batchnorm_train_forward_backward(C, shape):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    B, H, W = shape
    y = (torch.randn((B, C, H, W), generator=g, dtype=torch.float64) * 1.5 + 0.3).requires_grad_(True)
    res = torch.randn((B, C, H, W), generator=g, dtype=torch.float64)
    gamma = (1 + 0.2 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm = torch.zeros(C, dtype=torch.float64)
    rv = torch.ones(C, dtype=torch.float64)
    out = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5) + res)
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dout)
    # HIP: statistics from the conv epilogue are emulated with a 1x1 identity... use torch sums as the partials
    yg = nhwc(y.detach())
    stats = torch.stack([yg.sum(dim=(0, 1, 2)), (yg * yg).sum(dim=(0, 1, 2))]).reshape(2, 1, C).contiguous()
    rmg, rvg = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd, scale, shift = ops.bn_finalize(stats, B * H * W, gamma.detach().float().to(DEV),
                                                 beta.detach().float().to(DEV), rmg, rvg)
    z = ops.bn_act(yg, scale, shift, res=nhwc(res), relu=True)
    assert rel_err(to_nchw(z), out.detach()) < 1e-5
    np.testing.assert_allclose(rmg.cpu().double(), rm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvg.cpu().double(), rv, rtol=1e-5, atol=1e-6)
    dy, dgamma, dbeta, dres = ops.bn_backward(nhwc(dout), z, yg, mean, invstd, gamma.detach().float().to(DEV),
                                              want_dres=True)
    assert rel_err(to_nchw(dy), y.grad) < 2e-5
    assert rel_err(dgamma.cpu(), gamma.grad) < 2e-5
    assert rel_err(dbeta.cpu(), beta.grad) < 2e-5
    assert rel_err(to_nchw(dres), dout * (out.detach() > 0)) < 1e-6


@pytest.mark.parametrize("H,W", [(20, 28), (21, 27), (2, 2), (64, 6)])
def test_maxpool_forward_backward_with_ties(H, W):
    """even maps: one thread per 2 x 2 block of input pixels (maxpool_bwd_quad_kernel); odd maps: the per-pixel kernel"""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, C = 2, 64
    x = F.relu(torch.randn((B, C, H, W), generator=g)).double()   # many exact zeros -> ties
    x.requires_grad_(True)
    out = F.max_pool2d(x, 3, 2, 1)
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dout)
    o, am = ops.maxpool3x3s2(nhwc(x.detach()))
    assert torch.equal(to_nchw(o), out.detach().float().double())
    dx = ops.maxpool3x3s2_bwd(nhwc(dout), am, H, W)
    assert rel_err(to_nchw(dx), x.grad) < 1e-6


def test_upsample_bwd_and_layout():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    a = torch.randn((2, 32, 6, 10), generator=g, dtype=torch.float64, requires_grad=True)
    up = F.interpolate(a, scale_factor=2, mode="nearest")
    dup = torch.randn(up.shape, generator=g, dtype=torch.float64)
    up.backward(dup)
    assert rel_err(to_nchw(ops.upsample2x_bwd(nhwc(dup))), a.grad) < 1e-6
    x = torch.randn((3, 3, 8, 16), generator=g)
    assert torch.equal(ops.nchw_to_nhwc(x.to(DEV)).cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(DEV))).cpu(), x)


@pytest.mark.parametrize("K,B,H,W", [(2, 2, 24, 40), (3, 2, 24, 40), (1, 3, 21, 37), (4, 1, 8, 32), (2, 1, 70, 130)])
def test_head_forward_backward(K, B, H, W):
    """K = 1..4 (1 / 2 / 2 / 3 row tiles and 3 / 5 / 7 / 9 K-steps of the MFMA backward), ragged right / bottom edges, tile
    runs of a workgroup that cross image rows and images, fewer tiles than one workgroup takes"""
    ops = _ops()
    g = torch.Generator().manual_seed(K)
    Cin = 16
    x = torch.randn((B, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn((K, Cin, 3, 3), generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = (torch.randn(K, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=1)
    dl = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dl)
    w_ohwi = w.detach().permute(0, 2, 3, 1).contiguous().float().to(DEV)
    logits, am = ops.head_fwd(nhwc(x.detach()), w_ohwi, b.detach().float().to(DEV), argmax="int64")
    assert rel_err(logits.cpu(), ref.detach()) < 2e-6
    assert torch.equal(am.cpu(), logits.cpu().argmax(dim=1))
    dx, dw, db = ops.head_bwd(nhwc(x.detach()), w_ohwi, dl.float().to(DEV))
    assert rel_err(to_nchw(dx), x.grad) < 2e-6
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), w.grad) < 1e-5
    assert rel_err(db.cpu(), b.grad) < 1e-5


def test_flat_adam_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=3e-4)
    pg = p0.clone().to(DEV)
    fa = ops.FlatAdam(pg, lr=3e-4, max_norm=0.5)
    for step in range(3):
        grad = torch.randn(n, generator=g) * (0.001 if step == 1 else 1.0)   # step 1: below the clip norm
        ref_p.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
        opt.step()
        norm = fa.step(grad.to(DEV))
        assert float(norm) == pytest.approx(float(grad.norm()), rel=1e-5)
    assert rel_err(pg.cpu(), ref_p.detach()) < 1e-6


def test_flat_adam_update_size_and_nonfinite_gradient_guard():
    """(1) the UPDATE p - p0 (3e-4 of the parameter) against torch.optim.Adam at 2e-6: catches a bias correction computed
    from float betas ((double)(float)0.999 moves 1 - beta2^t by 1.3e-5 at t = 1); (2) a gradient with a NaN
    and a finite loss: dt_clip_coef raises the skip flag, parameters, moments and the step count stay put."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n = 50_001
    p0 = 1e-3 * torch.randn(n, generator=g)   # small parameters: the fp32 grid of p (1e-10) is far below the update (3e-4)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=3e-4)
    pg = p0.clone().to(DEV)
    fa = ops.FlatAdam(pg, lr=3e-4, max_norm=0.0)
    for step in range(2):
        grad = torch.randn(n, generator=g)
        ref_p.grad = grad.clone()
        opt.step()
        fa.step(grad.to(DEV))
        upd, upd_ref = (pg.cpu() - p0).double(), (ref_p.detach() - p0).double()
        assert float((upd - upd_ref).abs().max()) <= 2e-6 * float(upd_ref.abs().max())
    before, m0, v0, t0 = pg.clone(), fa.m.clone(), fa.v.clone(), fa.steps_applied()
    bad = torch.randn(n, generator=g)
    bad[123] = float("nan")
    skip = torch.zeros(1, dtype=torch.int32, device=DEV)
    fa.step(bad.to(DEV), skip_flag=skip)
    assert int(skip) == 1
    assert torch.equal(pg, before) and torch.equal(fa.m, m0) and torch.equal(fa.v, v0) and fa.steps_applied() == t0
    fa.step(bad.to(DEV))                       # stand-alone call (no flag given): the guard still holds
    assert torch.equal(pg, before) and fa.steps_applied() == t0
    fa.step(torch.randn(n, generator=g).to(DEV))
    assert not torch.equal(pg, before) and fa.steps_applied() == t0 + 1


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 40, 64, 64), (1, 24, 40, 16, 16), (2, 20, 36, 32, 32),
                                            (2, 9, 70, 128, 96), (1, 33, 17, 64, 128)])
def test_conv_with_fused_bn_backward_reduction(B, H, W, Cin, Cout):
    """dt_conv2d_bn_bwd: output bit-identical to dt_conv2d; the partial sums equal the BatchNorm-backward
    reduction (sum g, sum g*xhat with the ReLU mask recomputed from y) of dt_bn_bwd_reduce / fp64."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 7 + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5
    y = torch.randn((B, Cout, H, W), generator=g) * 1.5 + 0.2
    mean = y.mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    xg, wg, yg = nhwc(x), hwio(w), nhwc(y)
    plain, _, _ = ops.conv2d(xg, wg, 3, 1, 1)
    out, red = ops.conv2d_bn_bwd(xg, wg, yg, mean.to(DEV), invstd.to(DEV), sc.to(DEV), sh.to(DEV))
    assert torch.equal(out, plain)
    dz = to_nchw(out)
    y64 = y.double()
    mask = (y64.float() * sc[None, :, None, None] + sh[None, :, None, None]) > 0      # fp32 like the kernel
    gm = torch.where(mask, dz, torch.zeros_like(dz))
    xh = (y64 - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    sums = red.sum(dim=1).cpu().double()
    scale = float(gm.abs().sum(dim=(0, 2, 3)).max()) + 1.0
    assert float((sums[0] - gm.sum(dim=(0, 2, 3))).abs().max()) <= 2e-5 * scale
    assert float((sums[1] - (gm * xh).sum(dim=(0, 2, 3))).abs().max()) <= 6e-5 * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("B,H,W,C", [(2, 12, 20, 64), (1, 33, 17, 16), (3, 64, 64, 32), (2, 5, 7, 256)])
def test_upsample_backward_with_fused_bn_reduction(dtype, B, H, W, C):
    """dt_upsample2x_bwd_bn(_bf16): dx bit-identical to dt_upsample2x_bwd(_bf16); partial sums = BatchNorm-backward
    reduction of (dx, y) with the virtual-activation mask."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + H + C)
    dup = torch.randn((B, 2 * H, 2 * W, C), generator=g).to(dtype)
    y = (torch.randn((B, H, W, C), generator=g) * 1.3 + 0.1).to(dtype)
    mean = y.float().mean(dim=(0, 1, 2))
    invstd = 1.0 / torch.sqrt(y.float().var(dim=(0, 1, 2), unbiased=False) + 1e-5)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    sc, sh = gamma * invstd, beta - mean * gamma * invstd
    dx, red = ops.upsample2x_bwd_bn(dup.to(DEV), y.to(DEV), mean.to(DEV), invstd.to(DEV), sc.to(DEV), sh.to(DEV))
    if dtype == torch.float32:
        plain = torch.empty((B, H, W, C), dtype=dtype, device=DEV)
        from deadtrees_amd import _lib
        _lib.check(_lib.load().dt_upsample2x_bwd(dup.to(DEV).data_ptr(), plain.data_ptr(), 0, B, H, W, C,
                                                 torch.cuda.current_stream().cuda_stream), "dt_upsample2x_bwd")
        act = y * sc + sh
    else:
        plain = ops.upsample2x_bwd_bf16(dup.to(DEV))
        act = (y.float() * sc + sh).to(torch.bfloat16).float()
    assert torch.equal(dx, plain)
    d64 = dx.float().cpu().double()
    gm = torch.where(act > 0, d64, torch.zeros_like(d64))
    xh = (y.double() - mean.double()) * invstd.double()
    sums = red.sum(dim=1).cpu().double()
    scale = float(gm.abs().sum(dim=(0, 1, 2)).max()) + 1.0
    assert float((sums[0] - gm.sum(dim=(0, 1, 2))).abs().max()) <= 3e-5 * scale
    assert float((sums[1] - (gm * xh).sum(dim=(0, 1, 2))).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 40, 64, 64), (2, 9, 70, 128, 96), (1, 16, 16, 256, 256)])
def test_conv_gradient_join_with_fused_bn_backward_reduction(B, H, W, Cin, Cout):
    """dt_conv2d_bn_bwd on a gradient join (accumulate) with the mask taken from a stored activation: the joined tensor
    equals dt_conv2d(accumulate) bit for bit; the sums are the BatchNorm-backward reduction of the JOINED gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 5 + Cin + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5
    y = torch.randn((B, Cout, H, W), generator=g) * 1.5 + 0.2
    z = torch.relu(torch.randn((B, Cout, H, W), generator=g))          # stored block output (about half zeros)
    base = torch.randn((B, Cout, H, W), generator=g)
    mean = y.mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    xg, wg = nhwc(x), hwio(w)
    plain, _, _ = ops.conv2d(xg, wg, 3, 1, 1, out0=nhwc(base), accumulate=True)
    out, red = ops.conv2d_bn_bwd(xg, wg, nhwc(y), mean.to(DEV), invstd.to(DEV), act=nhwc(z), join_into=nhwc(base))
    assert torch.equal(out, plain)
    dz = to_nchw(out)
    gm = torch.where(z.double() > 0, dz, torch.zeros_like(dz))
    xh = (y.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    sums = red.sum(dim=1).cpu().double()
    scale = float(gm.abs().sum(dim=(0, 2, 3)).max()) + 1.0
    assert float((sums[0] - gm.sum(dim=(0, 2, 3))).abs().max()) <= 2e-5 * scale
    assert float((sums[1] - (gm * xh).sum(dim=(0, 2, 3))).abs().max()) <= 6e-5 * scale
    with pytest.raises(RuntimeError):          # a join needs the stored activation
        ops.conv2d_bn_bwd(xg, wg, nhwc(y), mean.to(DEV), invstd.to(DEV), act_scale=mean.to(DEV), act_shift=mean.to(DEV),
                          join_into=nhwc(base))


@pytest.mark.parametrize("acc", [False, True])
def test_maxpool_backward_with_fused_batchnorm_sums(acc):
    """dt_maxpool3x3s2_bwd_bn: the gradient is bit-identical to dt_maxpool3x3s2_bwd, the partial rows sum to the
    BatchNorm-backward reduction of the pooled layer over that gradient (mask from y * scale + shift) — fp32 and bf16"""
    import ctypes as C
    from deadtrees_amd import _lib
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator().manual_seed(31)
    B, Cc, H, W = 2, 64, 20, 28
    st = torch.cuda.current_stream().cuda_stream
    yraw = torch.randn((B, H, W, Cc), generator=g)
    sc, sh = 1 + 0.3 * torch.randn(Cc, generator=g), 0.2 * torch.randn(Cc, generator=g)
    mu, istd = 0.1 * torch.randn(Cc, generator=g), 1 + 0.2 * torch.rand(Cc, generator=g)
    x = F.relu(yraw * sc + sh)
    coef = [t.to(DEV) for t in (mu, istd, sc, sh)]
    for bf in (False, True):
        dt = torch.bfloat16 if bf else torch.float32
        xd = x.to(dt).to(DEV)
        yd = yraw.to(dt).to(DEV)
        pooled, am = (ops.maxpool3x3s2_bf16(xd) if bf else ops.maxpool3x3s2(xd))
        dout = torch.randn(pooled.shape, generator=g).to(dt).to(DEV)
        prev = torch.randn((B, H, W, Cc), generator=g).to(dt).to(DEV) if acc else None
        want = (ops.maxpool3x3s2_bwd_bf16 if bf else ops.maxpool3x3s2_bwd)(dout, am, H, W, dx=None if prev is None else prev.clone())
        rows_fn = lib.dt_maxpool3x3s2_bwd_bn_bf16_rows if bf else lib.dt_maxpool3x3s2_bwd_bn_rows
        fn = lib.dt_maxpool3x3s2_bwd_bn_bf16 if bf else lib.dt_maxpool3x3s2_bwd_bn
        P = rows_fn(B, H, W, Cc)
        assert P > 0 and rows_fn(B, H + 1, W, Cc) == 0
        red = torch.empty(lib.dt_bn_stats_floats(P, Cc), dtype=torch.float32, device=DEV)
        dx = prev.clone() if acc else torch.empty((B, H, W, Cc), dtype=dt, device=DEV)
        fuse = _lib.BnBwdFuse(yd.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr())
        _lib.check(fn(dout.data_ptr(), am.data_ptr(), dx.data_ptr(), 1 if acc else 0, C.byref(fuse), red.data_ptr(), B, H, W, Cc,
                      st), "maxpool_bwd_bn")
        assert torch.equal(dx, want)
        r = red[:2 * P * Cc].view(2, P, Cc).double().sum(1).cpu()
        y64, g64 = yd.double().cpu(), dx.double().cpu()
        act = (yd.float().cpu() * sc + sh)
        act = act.to(torch.bfloat16).float() if bf else act
        gm = torch.where(act > 0, g64, torch.zeros((), dtype=torch.float64))
        xhat = (y64 - mu.double()) * istd.double()
        tol = 1e-4 * float(gm.abs().sum(dim=(0, 1, 2)).max())
        np.testing.assert_allclose(r[0], gm.sum(dim=(0, 1, 2)), rtol=1e-4, atol=tol)
        np.testing.assert_allclose(r[1], (gm * xhat).sum(dim=(0, 1, 2)), rtol=1e-4, atol=3 * tol)
