"""End-to-end parity of the HIP U-Net + fused losses against the CPU oracle (oracle/).  Needs an MI355X.

Tolerances: logits max-abs error <= 1e-4 * max|logit| (fp32, different summation order
than oneDNN), argmax maps bit-identical except where the top-2 logit margin is below that error bound,
losses rel 1e-5, parameter gradients rel-L2 <= 1e-4 per tensor (small tensors: 1e-3 of global scale).
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair(in_ch=3, classes=2, seed=0):
    from deadtrees_amd.network.unet import UNetHIP
    from oracle.unet_ref import make_oracle
    ref = make_oracle(in_ch, classes, seed=seed)
    m = UNetHIP(in_channels=in_ch, classes=classes)
    m.load_state_dict(ref.state_dict())
    return ref, m.to(DEV)


def _synth(B, H, W, C=3, K=2, seed=1234):
    from deadtrees_amd.data.synthetic import synth_batch
    return synth_batch(B, H, W, C, K, seed)


def test_state_dict_roundtrip_on_device():
    ref, m = _pair()
    sd_ref, sd = ref.state_dict(), m.state_dict()
    assert set(sd_ref.keys()) == set(sd.keys())
    for k in sd_ref:
        assert torch.equal(sd_ref[k].cpu(), sd[k].cpu()), k


@pytest.mark.parametrize("B,H,W,C,K", [(2, 64, 64, 3, 2), (1, 256, 256, 3, 2), (2, 96, 160, 4, 3)])
def test_forward_eval_parity_and_argmax(B, H, W, C, K):
    ref, m = _pair(C, K)
    img, _ = _synth(B, H, W, C, K)
    ref.eval()
    m.eval()
    with torch.no_grad():
        want = ref(img)
        want64 = ref.double()(img.double())
        got = m(img.to(DEV)).cpu()
    scale = float(want64.abs().max())
    err = float((got.double() - want64).abs().max())
    err_ref = float((want.double() - want64).abs().max())
    assert err <= 1e-4 * scale, (err, err_ref, scale)
    # argmax: identical wherever the fp64 top-2 margin exceeds the error bound of either side
    top2 = want64.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    safe = margin > 4 * max(err, err_ref)
    am_hip = m.predict_classes(img.to(DEV)).cpu()
    assert am_hip.dtype == torch.int64 and tuple(am_hip.shape) == (B, H, W)
    assert torch.equal(am_hip, got.argmax(dim=1))           # fused argmax == argmax of its own logits
    assert torch.equal(am_hip[safe], want64.argmax(dim=1)[safe])
    flips = int((am_hip != want.argmax(dim=1)).sum())
    assert flips <= int((~safe).sum())
    assert float(safe.float().mean()) > 0.99


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_parity_at_benchmark_tile_size_with_flip_report(mode):
    """BASELINE configs[1]'s tile size (512x512, B=2 so that the fp64 oracle finishes in seconds): logits against the
    fp64 oracle, and the argmax report SURVEY §7.3(4) asks for — flip count, the top-2 logit margin histogram, and the
    assertion that every flipped pixel is a near-tie (margin below the summed fp32 errors of both sides)."""
    B, H, W = 2, 512, 512
    ref, m = _pair(3, 2)
    img, _ = _synth(B, H, W, 3, 2)
    if mode == "eval":
        ref.eval()
        m.eval()
    else:
        ref.train()
        m.train()
    import copy
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        want = ref(img)
        want64 = ref64(img.double())
        got = m(img.to(DEV)).cpu()
    scale = float(want64.abs().max())
    err = float((got.double() - want64).abs().max())
    err_ref = float((want.double() - want64).abs().max())
    assert err <= 1e-4 * scale, (err, err_ref, scale)
    top2 = want64.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    am_hip, am64, am32 = got.argmax(dim=1), want64.argmax(dim=1), want.argmax(dim=1)
    flips64 = am_hip != am64
    flips32 = am_hip != am32
    edges = [0.0, 1e-6, 1e-5, 1e-4, 1e-3, float("inf")]
    hist = [int(((margin >= lo) & (margin < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
    fmax = float(margin[flips64].max()) if bool(flips64.any()) else 0.0
    from conftest import parity_report
    parity_report(f"[fp32 {mode} {B}x{H}x{W}, engine.winograd={m.engine.winograd}] max |logit err| HIP {err:.2e} / torch-CPU fp32 {err_ref:.2e} (max |logit| {scale:.3f}); "
          f"pixels {margin.numel()}, top-2 margin histogram [<1e-6, <1e-5, <1e-4, <1e-3, >=1e-3] = {hist}; "
          f"argmax flips vs fp64 oracle: {int(flips64.sum())} (largest margin among them {fmax:.2e}), "
          f"vs fp32 oracle: {int(flips32.sum())}; torch-CPU fp32 vs fp64 flips: {int((am32 != am64).sum())}")
    # every flipped pixel is a near-tie: its fp64 margin is below the two sides' summed error
    assert fmax <= 2 * err + 1e-12, (fmax, err)
    assert int(flips64.sum()) <= hist[0] + hist[1] + hist[2] + hist[3]
    if mode == "eval":
        am_fused = m.predict_classes(img.to(DEV)).cpu()
        assert torch.equal(am_fused, am_hip)


@pytest.mark.parametrize("C,K,names", [(3, 2, ("GDICE", "FOCAL")), (4, 3, ("DICE", "FOCAL", "BOUNDARY")),
                                       (3, 3, ("GWDICE", "FOCAL"))])
@pytest.mark.parametrize("winograd", [False, True])
def test_train_step_gradient_parity(C, K, names, winograd):
    """loss + every parameter gradient of one training step vs the oracle (RGB / RGBN, 2 / 3 classes, each dice
    flavour, boundary loss on device-built distance maps).

    End-to-end gradients of a ReLU/max-pool network are ill-conditioned in fp32 (an activation within
    rounding distance of zero flips its mask), so the yardstick is the fp64 oracle: the HIP result must
    be as close to it as the fp32 CPU oracle is (factor 4 per tensor, factor 2 over all parameters).  Exact per-kernel backward
    parity is covered in tests/test_kernels_gpu.py."""
    import copy
    from deadtrees_amd.loss.seg_loss import seg_loss
    from oracle.train_ref import loss_from_logits
    from deadtrees_amd.data.distmap import distmaps_on_device
    from oracle.losses_ref import dist_map, one_hot
    B, H, W = 2, 128, 128
    ref, m = _pair(C, K)
    # winograd False: every convolution an exact fp32 fma chain (direct kernels) -> the bounds the round-1 path was held
    # to.  True (the default engine): the 3x3 stride-1 layers run as Winograd F(2x2,3x3), ~1e-6 relative per layer
    # instead of ~3e-7, which flips a few more ReLU masks of this ill-conditioned end-to-end gradient (measured: up to
    # 4.3x per tensor and 3.2x overall the fp32 CPU oracle's own distance from fp64, i.e. 1.2 % vs 0.4 % of the gradient
    # norm): bounded at 10x / 5x instead of 4x / 2x.  The well-conditioned checks (per-kernel parity in
    # tests/test_winograd_gpu.py at 1e-5, frozen-BatchNorm backward at 1e-4, logits, loss) hold for both paths.
    m.engine.winograd = winograd
    per_tensor, overall = (10.0, 5.0) if winograd else (4.0, 2.0)
    ref64 = copy.deepcopy(ref).double()
    img, mask = _synth(B, H, W, C, K)
    dist = None
    if "BOUNDARY" in names:   # the reference loader's maps for the oracle; the HIP side builds its own on the device
        oh = one_hot(mask, K).numpy()
        dist = torch.from_numpy(np.stack([dist_map(oh[i]) for i in range(B)]).astype(np.float32))
    ref.train()
    ref64.train()
    m.train()
    logits_ref = ref(img)
    loss_ref, _ = loss_from_logits(logits_ref, mask, names, dist)
    loss_ref.backward()
    logits64 = ref64(img.double())
    loss64, _ = loss_from_logits(logits64, mask, names, None if dist is None else dist.double())
    loss64.backward()
    logits = m(img.to(DEV))
    ddist = distmaps_on_device(mask.to(DEV), K) if dist is not None else None
    if ddist is not None:
        assert torch.equal(ddist.cpu(), dist)
    loss, parts, err = seg_loss(logits, mask.to(DEV), ddist, names)
    loss.backward()
    assert int(err) == 0
    e_hip = float((logits.detach().cpu().double() - logits64.detach()).abs().max())
    e_ref = float((logits_ref.detach().double() - logits64.detach()).abs().max())
    assert e_hip <= max(3 * e_ref, 1e-4 * float(logits64.detach().abs().max())), (e_hip, e_ref)
    assert float(loss.detach()) == pytest.approx(float(loss64.detach()), rel=2e-5)
    grads = m.smp_grad_dict()
    g32 = {k: p.grad for k, p in ref.named_parameters()}
    g64 = {k: p.grad for k, p in ref64.named_parameters()}
    assert set(grads.keys()) == set(g64.keys())
    tot_hip = tot_ref = tot = 0.0
    for k, g in g64.items():
        n = float(g.norm()) + 1e-30
        eh = float((grads[k].double() - g).norm())
        er = float((g32[k].double() - g).norm())
        assert eh <= per_tensor * er + 1e-4 * n, (k, eh / n, er / n)   # per tensor: N x the fp32 oracle's own error
        tot_hip += eh ** 2
        tot_ref += er ** 2
        tot += n ** 2
    assert tot_hip ** 0.5 <= overall * tot_ref ** 0.5 + 1e-5 * tot ** 0.5, (tot_hip, tot_ref, tot)
    # BN running statistics were updated like torch's
    sd_ref, sd = ref.state_dict(), m.state_dict()
    for k in sd_ref:
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), sd_ref[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(sd_ref[k]) == 1
    # flat .grad delivered through autograd equals the engine's buffer
    assert torch.equal(m.flat_params.grad, m._grad_buffer())


CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "losses_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_fused_losses_vs_reference_golden(path):
    """values and d(loss)/d(logits) against vectors produced by the imported reference losses."""
    from deadtrees_amd.loss.seg_loss import seg_loss
    z = np.load(path)
    mask = torch.from_numpy(z["mask"]).to(DEV)
    dist = torch.from_numpy(z["distmap"]).to(DEV)
    for combo in ("GDICE+FOCAL", "DICE+FOCAL", "GDICE+BOUNDARY+FOCAL", "GWDICE+FOCAL"):
        names = tuple(combo.split("+"))
        lg = torch.from_numpy(z["logits"]).to(DEV).requires_grad_(True)
        total, parts, err = seg_loss(lg, mask, dist if "BOUNDARY" in names else None, names)
        total.backward()
        assert int(err) == 0
        assert float(total.detach()) == pytest.approx(float(z[f"loss[{combo}]"]), rel=1e-5, abs=1e-6)
        ref = z[f"dlogits[{combo}]"]
        np.testing.assert_allclose(lg.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max() + 1e-10)
        if "GWDICE" in names:   # incl. the cross-sample broadcast of gwdl.py:180-198 (B > 1 in every fixture)
            assert float(parts["dice_loss"]) == pytest.approx(float(z["gwdice"]), rel=1e-5, abs=1e-6)
        elif "GDICE" in names:
            assert float(parts["dice_loss"]) == pytest.approx(float(z["gdice"]), rel=1e-5, abs=1e-6)
        else:
            assert float(parts["dice_loss"]) == pytest.approx(float(z["dice"]), rel=1e-5, abs=1e-6)
        assert float(parts["focal_loss"]) == pytest.approx(float(z["focal"]), rel=1e-5, abs=1e-6)
        assert float(parts["ce_loss"]) == pytest.approx(float(z["ce"]), rel=1e-5, abs=1e-6)
        if "BOUNDARY" in names:
            assert float(parts["boundary_loss"]) == pytest.approx(float(z["boundary"]), rel=1e-5, abs=1e-5)


def test_fscore_matches_oracle_and_label_error_flag():
    from deadtrees_amd.loss.seg_loss import seg_loss
    from oracle import losses_ref as L
    g = torch.Generator().manual_seed(0)
    logits = torch.randn((3, 3, 32, 32), generator=g) * 3
    mask = torch.randint(0, 3, (3, 32, 32), generator=g)
    _, parts, err = seg_loss(logits.to(DEV), mask.to(DEV), None, ("GDICE",))
    p = logits.softmax(1)
    assert float(parts["dice"]) == pytest.approx(float(L.fscore(p, mask, ignore_channels=(0,))), rel=1e-6)
    assert float(parts["dice_with_bg"]) == pytest.approx(float(L.fscore(p, mask)), rel=1e-6)
    assert int(err) == 0
    bad = mask.clone()
    bad[0, 0, 0] = 7
    _, _, err = seg_loss(logits.to(DEV), bad.to(DEV), None, ("GDICE",))
    assert int(err) == 1


@pytest.mark.parametrize("B,H,W,C,K", [(2, 128, 128, 3, 2), (1, 256, 256, 4, 3)])
def test_bf16_inference_leg_against_fp32(B, H, W, C, K):
    """bf16 storage / fp32 accumulate forward vs the fp64 oracle and the fp32 HIP path.  Tolerance (bf16 has 8
    significant bits and ~50 roundings sit between image and logits): relative L2 error of the logits <= 2e-2,
    max error <= 5e-2 of max|logit|; class maps identical wherever the logit margin exceeds the max error."""
    ref, m = _pair(C, K)
    img, _ = _synth(B, H, W, C, K)
    m.eval()
    ref.eval()
    with torch.no_grad():
        want64 = ref.double()(img.double())
        l32 = m(img.to(DEV)).cpu()
        l16 = m.forward_bf16(img.to(DEV)).cpu()
    scale = float(want64.abs().max())
    err16 = float((l16.double() - want64).abs().max())
    rel_l2 = float((l16.double() - want64).norm() / want64.norm())
    print(f"bf16: rel L2 {rel_l2:.3e}, max err {err16 / scale:.3e} of max|logit|")
    assert rel_l2 <= 2e-2, rel_l2
    assert err16 <= 5e-2 * scale, (err16, scale)
    assert float((l16 - l32).abs().max()) <= 5e-2 * scale
    am16 = m.predict_classes(img.to(DEV), precision="bf16").cpu()
    assert torch.equal(am16, l16.argmax(dim=1))
    top2 = want64.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * err16
    assert torch.equal(am16[safe], want64.argmax(dim=1)[safe])
    assert float((am16 == want64.argmax(dim=1)).float().mean()) > 0.97
