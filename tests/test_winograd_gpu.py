"""Winograd F(2x2,3x3) fp32 convolution (conv_wino.hip) against torch CPU fp64 and against the direct kernel.

Tolerance: the transforms add/subtract inputs before multiplying (|B^T d B| <= 4 max|d|, products of (G g G^T) with
coefficients 1/4..1) and the output transform sums 9 of the 16 products, so the result is no longer an exact fp32 fma
chain: error ~ a few 1e-6 of max|y| per sqrt(K/1000) — bounded here at 1e-5 * max|y| (the direct kernel: 2e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from deadtrees_amd import ops
    return ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).double()


CASES = [  # B, H, W (stored size of source 0), C0, C1, mode0, Cout, split, transform
    (2, 32, 32, 64, 0, 0, 64, 0, False), (1, 34, 70, 512, 0, 0, 64, 0, True), (3, 17, 33, 32, 0, 0, 128, 0, False),
    (2, 16, 20, 64, 32, 1, 128, 0, True), (2, 16, 24, 128, 64, 1, 64, 0, False), (2, 40, 48, 64, 0, 0, 192, 64, False),
    (2, 16, 16, 16, 0, 0, 64, 0, False), (1, 5, 7, 16, 16, 0, 64, 0, True),
    # persistent workgroups over several rounds (490 and 735 tiles on 256 CUs), ragged edges, tile-boundary prefetch
    (5, 100, 100, 16, 0, 0, 128, 0, True), (5, 50, 50, 16, 16, 1, 192, 64, False),
    # maps of at most 8x8 pixels: four images share a 16x16-pixel tile (layer 4 of a 256x256 sub-tile); batch not a
    # multiple of 4, ragged maps, fused input transform, upsample + concat from 4x4, split outputs + join
    (6, 8, 8, 64, 0, 0, 64, 0, False), (5, 7, 6, 128, 0, 0, 128, 0, True), (3, 4, 4, 64, 32, 1, 64, 0, False),
    (7, 8, 8, 64, 0, 0, 192, 64, False), (70, 8, 8, 32, 0, 0, 128, 0, False)]


@pytest.mark.parametrize("B,H,W,C0,C1,mode0,Cout,split,tf", CASES)
def test_winograd_conv_matches_fp64_and_direct_kernel(B, H, W, C0, C1, mode0, Cout, split, tf):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H + C0 + Cout)
    x = torch.randn((B, C0, H, W), generator=g)
    Hin, Win = (H, W) if mode0 == 0 else (2 * H, 2 * W)
    s1 = torch.randn((B, C1, Hin, Win), generator=g) if C1 else None
    wt = torch.randn((Cout, C0 + C1, 3, 3), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    sc = 1 + 0.3 * torch.randn(C0, generator=g) if tf else None
    sh = 0.3 * torch.randn(C0, generator=g) + 0.2 if tf else None
    z = x.double()
    if tf:
        z = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]).double()   # fp32 affine like the kernel
    if mode0 == 1:
        z = F.interpolate(z, scale_factor=2, mode="nearest")
    xin = torch.cat([z, s1.double()], 1) if C1 else z
    ref = F.conv2d(xin, wt.double(), padding=1)
    w_hwio = wt.permute(2, 3, 1, 0).contiguous().to(DEV)
    u = ops.winograd_weights(w_hwio)
    kw = dict(src1=nhwc(s1) if C1 else None, mode0=mode0, in_scale=sc.to(DEV) if tf else None,
              in_shift=sh.to(DEV) if tf else None)
    scale = float(ref.abs().max())
    if split:
        base = torch.randn((B, Hin, Win, split), generator=g).to(DEV)
        o0, o1, _ = ops.conv2d_winograd(nhwc(x), u, split=split, out0=base.clone(), accumulate=True, **kw)
        got = torch.cat([nchw(o0) - nchw(base), nchw(o1)], 1)
        assert float((got - ref).abs().max()) <= 1e-5 * scale + 1e-6 * float(base.abs().max())
        return
    y, _, st = ops.conv2d_winograd(nhwc(x), u, want_stats=True, **kw)
    got = nchw(y)
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    np.testing.assert_allclose(st[0].double().sum(0).cpu(), ref.sum(dim=(0, 2, 3)), rtol=1e-4,
                               atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()))
    np.testing.assert_allclose(st[1].double().sum(0).cpu(), (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4)
    d, _, _ = ops.conv2d(nhwc(x), w_hwio, 3, 1, 1, **kw)
    assert float((nchw(d) - got).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("C1,mode0", [(0, 0), (64, 1)])
def test_winograd_fused_input_relu_keeps_nan_like_torch(C1, mode0):
    """the BatchNorm + ReLU applied while the Winograd kernels stage a raw producer output must behave like torch.relu /
    bn_act on non-finite values (v_max returned the non-NaN operand, so relu(NaN) became 0 and — in a
    source-1 chunk, whose ReLU floor is -inf — NaN became -inf): a NaN in source 0, a NaN scale and a NaN in the skip
    source all reach the outputs they touch, in the forward kernel and in the weight gradient; padding stays zero."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, H, W, C0, Cout = 2, 16, 16, 64, 64
    x = torch.randn((B, C0, H, W), generator=g)
    Hin, Win = (H, W) if mode0 == 0 else (2 * H, 2 * W)
    s1 = torch.randn((B, C1, Hin, Win), generator=g) if C1 else None
    wt = torch.randn((Cout, C0 + C1, 3, 3), generator=g) * 0.1
    sc, sh = 1 + 0.3 * torch.randn(C0, generator=g), 0.3 * torch.randn(C0, generator=g)
    x[0, 3, 5, 6] = float("nan")          # one raw value: relu(nan * sc + sh) = nan -> its 3x3 (6x6 up-sampled) footprint
    if C1:
        s1[1, 2, 9, 9] = float("nan")     # skip source: no ReLU there, but the floor select must not turn it into -inf
    z = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    if mode0 == 1:
        z = F.interpolate(z, scale_factor=2, mode="nearest")
    xin = torch.cat([z, s1], 1) if C1 else z
    ref = F.conv2d(xin.double(), wt.double(), padding=1)
    w_hwio = wt.permute(2, 3, 1, 0).contiguous().to(DEV)
    kw = dict(src1=nhwc(s1) if C1 else None, mode0=mode0, in_scale=sc.to(DEV), in_shift=sh.to(DEV))
    y, _, _ = ops.conv2d_winograd(nhwc(x), ops.winograd_weights(w_hwio), **kw)
    got = nchw(y)
    # a NaN input pixel reaches all 2x2 outputs of the Winograd tiles whose 4x4 patch holds it: a superset of the direct
    # convolution's 3x3 footprint, at most 4x as many pixels — never an Inf, never a silent 0
    assert bool(torch.isnan(got)[torch.isnan(ref)].all())
    assert int(torch.isnan(got).sum()) <= 4 * int(torch.isnan(ref).sum())
    assert not bool(torch.isinf(got).any())
    fin = ~torch.isnan(got)
    assert float((got[fin] - ref[fin]).abs().max()) <= 1e-5 * float(ref[fin].abs().max())
    # weight gradient of the same layer: the NaN pixel poisons the gradient rows of ITS input channel(s) only
    dy = torch.randn((B, Cout, Hin, Win), generator=g)
    dw = ops.conv2d_wgrad_winograd(nhwc(x), nhwc(dy), **kw).cpu()          # HWIO
    bad_rows = torch.zeros(C0 + C1, dtype=torch.bool)
    bad_rows[3] = True
    if C1:
        bad_rows[C0 + 2] = True
    assert bool(torch.isnan(dw[:, :, bad_rows]).all()) and not bool(torch.isnan(dw[:, :, ~bad_rows]).any())
    assert not bool(torch.isinf(dw).any())
    # a NaN BatchNorm coefficient (the channel's batch statistics overflowed) poisons every output pixel
    sc2 = sc.clone()
    sc2[7] = float("nan")
    y2, _, _ = ops.conv2d_winograd(nhwc(torch.randn((B, C0, H, W), generator=g)), ops.winograd_weights(w_hwio),
                                   src1=kw["src1"], mode0=mode0, in_scale=sc2.to(DEV), in_shift=sh.to(DEV))
    assert bool(torch.isnan(y2).all())


def test_winograd_weight_transform():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    Cin, Cout = 24, 64
    w = torch.randn((3, 3, Cin, Cout), generator=g, dtype=torch.float64)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    want = torch.einsum("ia,abco,jb->ijco", G, w, G)             # [4][4][Cin][Cout]
    u = ops.winograd_weights(w.float().to(DEV)).cpu().double()   # [16][Cin/8][2][Cout][4]
    got = u.permute(0, 1, 2, 4, 3).reshape(16, Cin, Cout).reshape(4, 4, Cin, Cout)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize("join,B,H,W", [(False, 3, 100, 84), (True, 3, 100, 84), (False, 9, 8, 8), (True, 6, 7, 8)])
def test_winograd_fused_bn_backward_sums_match_the_direct_kernel(join, B, H, W):
    """dt_conv2d_winograd_bn_bwd against dt_conv2d_bn_bwd: the gradient within the Winograd tolerance, the
    BatchNorm-backward partial sums (virtual activation / stored activation + gradient join) equal per channel; ragged
    map, several rounds of the persistent workgroups; small maps packed four images to a tile."""
    ops = _ops()
    g = torch.Generator().manual_seed(7 + join)
    Cin, Cout = 32, 128
    dy = torch.randn((B, H, W, Cin), generator=g).to(DEV)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) * 0.05).to(DEV)
    y = (torch.randn((B, H, W, Cout), generator=g) * 1.5 + 0.2).to(DEV)
    mean = y.mean(dim=(0, 1, 2)).contiguous()
    invstd = (1.0 / torch.sqrt(y.var(dim=(0, 1, 2), unbiased=False) + 1e-5)).contiguous()
    sc = (1 + 0.2 * torch.randn(Cout, generator=g)).to(DEV)
    sh = (0.2 * torch.randn(Cout, generator=g)).to(DEV)
    act = torch.relu(y * sc + sh) if join else None
    base = torch.randn((B, H, W, Cout), generator=g).to(DEV) if join else None
    kw = dict(act=act, join_into=None) if join else dict(act_scale=sc, act_shift=sh)
    if join:
        kw["join_into"] = base.clone()
    ref_out, ref_red = ops.conv2d_bn_bwd(dy, w, y, mean, invstd, **kw)
    if join:
        kw["join_into"] = base.clone()
    out, red = ops.conv2d_winograd_bn_bwd(dy, ops.winograd_weights(w), y, mean, invstd, **kw)
    scale = float(ref_out.abs().max())
    assert float((out - ref_out).abs().max()) <= 1e-5 * scale
    a, b = red.double().sum(1).cpu(), ref_red.double().sum(1).cpu()
    # a gradient within 1e-5 of zero may fall on the other side of nothing (the mask comes from y, not from the result)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-4 * float(b.abs().max()))


def test_winograd_weight_images_of_a_network_match_the_per_layer_transform():
    """dt_winograd_weight_images (all eligible layers, forward and data-gradient images, one launch each) against
    dt_winograd_weights layer by layer"""
    from deadtrees_amd.network.unet import UNetHIP
    ops = _ops()
    m = UNetHIP().to(DEV)
    m.reset_parameters(seed=1)
    eng, params = m.engine, m.flat_params.detach()
    eng.winograd = True
    u_all = eng._wino_images(params, "t_u", False)
    wd_all = torch.empty_like(params)
    eng._weight_images(params, wd_all, 0)
    ud_all = eng._wino_images(wd_all, "t_ud", True)
    n = 0
    for dgrad, buf, src in ((False, u_all, params), (True, ud_all, wd_all)):
        offs = eng._wino_table(params.device, dgrad)[4]
        for c in eng.spec.convs:
            if c.key not in offs:
                continue
            cin, cout = (c.cout, c.cin) if dgrad else (c.cin, c.cout)
            want = ops.winograd_weights(src[c.w_off:c.w_off + c.w_size].view(3, 3, cin, cout))
            off, size = offs[c.key]
            assert torch.equal(buf[off:off + size], want.reshape(-1)), (c.key, dgrad)
            n += 1
    assert n >= 50


WG_CASES = [  # B, H, W (stored size of source 0), C0, C1, mode0, Cout, transform
    (2, 32, 32, 64, 0, 0, 64, False), (1, 34, 70, 128, 0, 0, 64, True), (3, 17, 33, 64, 0, 0, 128, False),
    (2, 16, 20, 64, 64, 1, 128, True), (4, 64, 64, 64, 0, 0, 64, False), (2, 16, 16, 512, 0, 0, 512, False),
    # 32 output channels (dec3.conv1: up(64) + skip(64) -> 32): one co block, the two waves of a ci half split the chunk's K
    (2, 16, 20, 64, 64, 1, 32, True), (3, 17, 33, 64, 0, 0, 32, False), (4, 64, 64, 128, 0, 0, 32, False)]


@pytest.mark.parametrize("B,H,W,C0,C1,mode0,Cout,tf", WG_CASES)
def test_winograd_weight_gradient_matches_fp64_and_direct_kernel(B, H, W, C0, C1, mode0, Cout, tf):
    """dt_conv2d_wgrad_winograd vs torch CPU fp64 autograd and vs dt_conv2d_wgrad: odd maps (half-empty edge tiles),
    several splits with the two-stage reduction, upsample + concat + fused BatchNorm/ReLU on the input side"""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + H + C0 + Cout)
    x = torch.randn((B, C0, H, W), generator=g)
    Hin, Win = (H, W) if mode0 == 0 else (2 * H, 2 * W)
    s1 = torch.randn((B, C1, Hin, Win), generator=g) if C1 else None
    sc = 1 + 0.3 * torch.randn(C0, generator=g) if tf else None
    sh = 0.3 * torch.randn(C0, generator=g) + 0.2 if tf else None
    z = x.double()
    if tf:
        z = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]).double()
    if mode0 == 1:
        z = F.interpolate(z, scale_factor=2, mode="nearest")
    xin = torch.cat([z, s1.double()], 1) if C1 else z
    wt = (torch.randn((Cout, C0 + C1, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    y = F.conv2d(xin, wt, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    want = wt.grad.permute(2, 3, 1, 0)      # HWIO
    kw = dict(src1=nhwc(s1) if C1 else None, mode0=mode0, in_scale=sc.to(DEV) if tf else None,
              in_shift=sh.to(DEV) if tf else None)
    got = ops.conv2d_wgrad_winograd(nhwc(x), nhwc(dy), **kw).cpu().double()
    direct = ops.conv2d_wgrad(nhwc(x), nhwc(dy), 3, 1, 1, **kw).cpu().double()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale, float((got - want).abs().max()) / scale
    assert float((direct - want).abs().max()) <= 2e-5 * scale
    again = ops.conv2d_wgrad_winograd(nhwc(x), nhwc(dy), **kw).cpu().double()
    assert torch.equal(again, got)          # fixed-order reduction: run-to-run bit-identical


@pytest.mark.parametrize("B,H,W,C0,C1,mode0,Cout,residual", [(2, 32, 32, 64, 0, 0, 64, True), (1, 34, 70, 128, 0, 0, 128, False),
                                                            (2, 16, 24, 128, 64, 1, 64, False), (3, 17, 33, 64, 0, 0, 64, True),
                                                            (10, 8, 8, 128, 0, 0, 128, True), (5, 6, 8, 64, 0, 0, 64, False)])
def test_winograd_inference_epilogue_is_bit_identical_to_conv_plus_bn_act(B, H, W, C0, C1, mode0, Cout, residual):
    """dt_conv2d_winograd_affine (eval-mode BatchNorm + ReLU (+ residual) applied to the accumulators) == dt_conv2d_winograd
    followed by dt_bn_act, bit for bit: ragged edges, upsample + concat, residual add"""
    import ctypes as C
    from deadtrees_amd import _lib
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 100 + H + Cout)
    Hin, Win = (H, W) if mode0 == 0 else (2 * H, 2 * W)
    x = torch.randn((B, H, W, C0), generator=g).to(DEV)
    s1 = torch.randn((B, Hin, Win, C1), generator=g).to(DEV) if C1 else None
    w = (torch.randn((3, 3, C0 + C1, Cout), generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5).to(DEV)
    u = ops.winograd_weights(w)
    scale = (1 + 0.3 * torch.randn(Cout, generator=g)).to(DEV)
    shift = (0.3 * torch.randn(Cout, generator=g)).to(DEV)
    res = torch.randn((B, Hin, Win, Cout), generator=g).to(DEV) if residual else None
    y, _, _ = ops.conv2d_winograd(x, u, src1=s1, mode0=mode0)
    want = ops.bn_act(y, scale, shift, res=res)
    d = ops.conv_desc(B, Hin, Win, C0, C1, mode0, Cout, 3, 1, 1)
    got = torch.empty_like(y)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.dt_conv2d_winograd_affine(C.byref(d), p(x), p(s1), p(u), p(got), p(scale), p(shift), p(res), st),
               "dt_conv2d_winograd_affine")
    assert torch.equal(got, want)


def test_fused_inference_forward_is_bit_identical_to_the_unfused_one():
    """UNetHIP.predict_classes / eval forward with the BatchNorm + ReLU (+ residual) epilogues (default) against the same
    network with DT_FUSE_EVAL off: identical logits and class maps"""
    from deadtrees_amd.network.unet import UNetHIP
    m = UNetHIP(in_channels=3, classes=2)
    m.reset_parameters(seed=3)
    m.to(DEV).eval()
    # non-trivial running statistics
    g = torch.Generator().manual_seed(0)
    m.bn_state.copy_((torch.rand(m.bn_state.shape, generator=g) * 0.5 + 0.25).to(DEV))
    x = torch.randn((2, 3, 96, 160), generator=g).to(DEV)
    eng = m.engine
    assert eng._fuse_eval_opt
    with torch.no_grad():
        a = m(x).clone()
        ca = m.predict_classes(x, dtype="uint8").clone()
        eng._fuse_eval_opt = False
        eng._ws.pop("affine_key", None)
        b = m(x).clone()
        cb = m.predict_classes(x, dtype="uint8").clone()
        eng._fuse_eval_opt = True
    assert torch.equal(a, b) and torch.equal(ca, cb)


@pytest.mark.parametrize("B,H,W,Cy,cx,sk", [(2, 32, 48, 64, 128, 64), (1, 36, 20, 32, 64, 64), (3, 18, 34, 64, 64, 0),
                                           (2, 16, 16, 128, 256, 128)])
def test_winograd_data_gradient_with_fused_upsample_backward(B, H, W, Cy, cx, sk):
    """dt_conv2d_winograd_upsampled_dgrad (epilogue form 6 + a plain launch for the skip's channels) against the chain it
    replaces — dt_conv2d_winograd with split outputs, then dt_upsample2x_bwd_bn: the 2x2-summed gradient and the skip's
    gradient bit-identical (same output transform, same (a + b) + (c + d) order), the BatchNorm-backward sums equal up to the
    grouping of the partial rows; ragged maps (H, W not multiples of 16), with and without a skip"""
    import ctypes as C
    from deadtrees_amd import _lib
    ops = _ops()
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 7 + H + cx)
    st = torch.cuda.current_stream().cuda_stream
    dy = torch.randn((B, H, W, Cy), generator=g).to(DEV)
    wt = torch.randn((3, 3, cx + sk, Cy), generator=g) * 0.05            # forward HWIO: (cx + sk) -> Cy
    wd = ops.weight_flip_transpose(wt.to(DEV))                            # data-gradient image: Cy -> (cx + sk)
    u = ops.winograd_weights(wd)
    yl = torch.randn((B, H // 2, W // 2, cx), generator=g).to(DEV)
    mu, istd = (0.1 * torch.randn(cx, generator=g)).to(DEV), (1 + 0.2 * torch.rand(cx, generator=g)).to(DEV)
    sc, sh = (1 + 0.3 * torch.randn(cx, generator=g)).to(DEV), (0.2 * torch.randn(cx, generator=g)).to(DEV)
    # the chain
    dup, dskip_ref, _ = ops.conv2d_winograd(dy, u, split=cx if sk else 0)
    g_ref, red_ref = ops.upsample2x_bwd_bn(dup, yl, mu, istd, sc, sh)
    # one call
    d = _lib.ConvDesc(B, H, W, Cy, 0, 0, H, W, cx + sk, 3, 1, 1, cx, 0)
    assert lib.dt_conv2d_winograd_upsampled_dgrad_supported(C.byref(d))
    P = lib.dt_conv2d_winograd_upsampled_dgrad_rows(C.byref(d))
    red = torch.empty(lib.dt_bn_stats_floats(P, cx), dtype=torch.float32, device=DEV)
    gx = torch.empty((B, H // 2, W // 2, cx), dtype=torch.float32, device=DEV)
    dskip = torch.empty((B, H, W, sk), dtype=torch.float32, device=DEV) if sk else None
    fuse = _lib.BnBwdFuse(yl.data_ptr(), mu.data_ptr(), istd.data_ptr(), sc.data_ptr(), sh.data_ptr())
    _lib.check(lib.dt_conv2d_winograd_upsampled_dgrad(C.byref(d), dy.data_ptr(), u.data_ptr(), gx.data_ptr(),
                                                      dskip.data_ptr() if sk else None, red.data_ptr(), C.byref(fuse), 3, st),
               "dt_conv2d_winograd_upsampled_dgrad")
    assert torch.equal(gx, g_ref)
    if sk:
        assert torch.equal(dskip, dskip_ref)
    got = red[:2 * P * cx].view(2, P, cx).double().sum(1).cpu()
    want = red_ref.double().sum(1).cpu()
    tol = 1e-5 * float(want.abs().max())
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=tol)
