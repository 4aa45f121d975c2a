"""Seeded random-shape sweep of the convolution entry points (fp32 and bf16; forward with BatchNorm statistics,
fused input transform, virtual upsample + concat, split / accumulate outputs; weight gradients) against fp64 torch
on the CPU.  The network itself only uses a handful of shapes; this keeps the C ABI honest for ragged tiles,
channel counts that do not fill a chunk or an MFMA tile, tiny maps and single-pixel rows."""
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


def _cases(seed, n, bf16):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        k, s = [(3, 1), (3, 1), (3, 1), (3, 2), (1, 2), (1, 1)][rng.integers(0, 6)]
        if k == 1 and s == 1 and not bf16:
            continue                      # the fp32 path has no 1x1 stride-1 layer
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 41)), int(rng.integers(1, 73))
        if s == 2:
            H, W = 2 * ((H + 1) // 2), 2 * ((W + 1) // 2)
        step = 8 if bf16 else 4
        Cin = int(rng.integers(1, 25)) * step
        Cout = int(rng.integers(1, 25)) * step
        out.append((B, H, W, Cin, Cout, k, s, (k - 1) // 2))
    return out


def _ref(x, w, s, p):
    return F.conv2d(x.double(), w.double(), stride=s, padding=p)


@pytest.mark.parametrize("case", _cases(11, 40, False), ids=lambda c: "x".join(map(str, c)))
def test_fp32_conv_forward_stats_and_wgrad(case):
    ops = _ops()
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    ref = _ref(x, w, s, p)
    y, _, stats = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(2, 3, 1, 0).contiguous().to(DEV),
                             k, s, p, want_stats=True)
    got = y.cpu().permute(0, 3, 1, 2).double()
    tol = 2e-5 * max(1.0, (Cin * k * k / 1000.0) ** 0.5) * float(ref.abs().max() + 1e-6)
    assert float((got - ref).abs().max()) <= tol
    st = stats.sum(dim=1).cpu().double()
    np.testing.assert_allclose(st[0], ref.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3 * float(ref.abs().sum(dim=(0, 2, 3)).max() + 1))
    np.testing.assert_allclose(st[1], (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-6)
    # weight gradient of the same layer
    dy = torch.randn(ref.shape, generator=g)
    wt = w.double().clone().requires_grad_(True)
    F.conv2d(x.double(), wt, stride=s, padding=p).backward(dy.double())
    dw = ops.conv2d_wgrad(x.permute(0, 2, 3, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV), k, s, p)
    gotw = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((gotw - wt.grad).abs().max()) <= 3e-5 * float(wt.grad.abs().max() + 1e-6) * max(1.0, (B * H * W / 1000.0) ** 0.5)


@pytest.mark.parametrize("case", _cases(12, 40, True), ids=lambda c: "x".join(map(str, c)))
def test_bf16_conv_forward_stats_and_wgrad(case):
    ops = _ops()
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn((B, Cin, H, W), generator=g).to(BF)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(BF)
    ref = _ref(x, w, s, p)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = ops.pack_weights_bf16(w.float().permute(2, 3, 1, 0).contiguous().to(DEV))
    y, _, stats = ops.conv2d_bf16(xg, wp, k, s, p, Cout, want_stats=True)
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    tol = ref.abs() * 2.0 ** -8 + float(ref.abs().max()) * 2.0 ** -9 + 1e-9
    assert bool(((got - ref).abs() <= tol).all())
    st = stats.sum(dim=1).cpu().double()       # statistics come from the fp32 accumulators, before rounding
    np.testing.assert_allclose(st[0], ref.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3 * float(ref.abs().sum(dim=(0, 2, 3)).max() + 1))
    np.testing.assert_allclose(st[1], (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-6)
    if k == 1 and s == 1:
        return                                   # no 1x1 stride-1 weight gradient on the path
    dy = torch.randn(ref.shape, generator=g).to(BF)
    wt = w.double().clone().requires_grad_(True)
    F.conv2d(x.double(), wt, stride=s, padding=p).backward(dy.double())
    dw = ops.conv2d_wgrad_bf16(xg, dy.permute(0, 2, 3, 1).contiguous().to(DEV), k, s, p)
    gotw = dw.cpu().permute(3, 2, 0, 1).double()
    assert float((gotw - wt.grad).abs().max()) <= 3e-5 * float(wt.grad.abs().max() + 1e-6) * max(1.0, (B * H * W / 1000.0) ** 0.5)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_decoder_style_conv_random(seed, bf16):
    """virtual nearest upsample + concat + fused producer BatchNorm/ReLU, split outputs with a gradient join"""
    ops = _ops()
    rng = np.random.default_rng(100 + seed)
    step = 8 if bf16 else 4
    B, h, w_ = int(rng.integers(1, 3)), int(rng.integers(1, 17)), int(rng.integers(1, 25))
    C0 = int(rng.integers(1, 5)) * 32                      # concat needs whole chunks from source 0
    C1 = int(rng.integers(0, 9)) * step
    split = int(rng.integers(1, 3)) * 32 * (2 if bf16 and rng.integers(0, 2) else 1)
    Cout = split + int(rng.integers(1, 9)) * step
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((B, C0, h, w_), generator=g)
    skip = torch.randn((B, max(C1, step), 2 * h, 2 * w_), generator=g)[:, :C1]
    sc, sh = 1 + 0.3 * torch.randn(C0, generator=g), 0.3 * torch.randn(C0, generator=g) + 0.3
    wt = torch.randn((Cout, C0 + C1, 3, 3), generator=g) * 0.05
    base = torch.randn((B, split, 2 * h, 2 * w_), generator=g)
    if bf16:
        a, skip, wt, base = a.to(BF), skip.to(BF), wt.to(BF), base.to(BF)
        z = F.relu(a.float() * sc[None, :, None, None] + sh[None, :, None, None]).to(BF).double()
    else:
        z = F.relu(a.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    xin = torch.cat([F.interpolate(z, scale_factor=2, mode="nearest"), skip.double()], dim=1)
    ref = F.conv2d(xin, wt.double(), padding=1)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)  # noqa: E731
    kw = dict(src1=nh(skip) if C1 else None, mode0=1, in_scale=sc.to(DEV), in_shift=sh.to(DEV), split=split,
              out0=nh(base).clone(), accumulate=True)
    if bf16:
        wp = ops.pack_weights_bf16(wt.float().permute(2, 3, 1, 0).contiguous().to(DEV))
        o0, o1, _ = ops.conv2d_bf16(nh(a), wp, 3, 1, 1, Cout, **kw)
        tol0 = 2.0 ** -7
    else:
        o0, o1, _ = ops.conv2d(nh(a), wt.permute(2, 3, 1, 0).contiguous().to(DEV), 3, 1, 1, **kw)
        tol0 = 3e-5
    want0, want1 = ref[:, :split] + base.double(), ref[:, split:]
    scale = float(ref.abs().max() + base.double().abs().max())
    assert float((o0.float().cpu().permute(0, 3, 1, 2).double() - want0).abs().max()) <= tol0 * scale
    assert float((o1.float().cpu().permute(0, 3, 1, 2).double() - want1).abs().max()) <= tol0 * scale
