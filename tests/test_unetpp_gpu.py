"""architecture "unet++" / "unetplusplus" (smp.UnetPlusPlus, reference deadtrees/network/segmodel.py:64-65) on the HIP
kernels, against oracle/unetpp_ref.py — whose dense decoder wiring is pinned by executing the reference's in-tree
efficientunetplusplus/decoder.py (tests/golden/unetpp_decoder.npz, tests/test_oracle_golden.py)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(C=3, K=2, seed=0):
    from deadtrees_amd.network.unet import UNetHIP
    from oracle.unetpp_ref import make_unetpp_oracle
    ref = make_unetpp_oracle(C, K, seed=seed)
    m = UNetHIP(in_channels=C, classes=K, decoder="unetplusplus")
    assert set(m.state_dict()) == set(ref.state_dict())           # smp key names incl. decoder.blocks.x_{d}_{l}.*
    assert m.spec.n_true_params == sum(p.numel() for p in ref.parameters())
    m.load_state_dict(ref.state_dict())
    return ref, m.to(DEV)


def test_channel_slice_kernel_is_torch_cat_and_its_backward():
    import ctypes as C
    from deadtrees_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    parts = [torch.randn((2, 5, 7, c), generator=g).to(DEV) for c in (64, 8, 128)]
    wide = torch.empty((2, 5, 7, 200), device=DEV)
    off = 0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for t in parts:
        _lib.check(lib.dt_channel_slice(C.c_void_p(t.data_ptr()), C.c_void_p(wide.data_ptr()), 70, t.shape[-1], 200, off, 1, 0, st), "slice")
        off += t.shape[-1]
    assert torch.equal(wide, torch.cat(parts, dim=-1))
    base = torch.randn((2, 5, 7, 8), generator=g).to(DEV)
    out = base.clone()
    _lib.check(lib.dt_channel_slice(C.c_void_p(wide.data_ptr()), C.c_void_p(out.data_ptr()), 70, 8, 200, 64, 0, 1, st), "slice")
    assert torch.equal(out, base + parts[1])
    _lib.check(lib.dt_channel_slice(C.c_void_p(wide.data_ptr()), C.c_void_p(out.data_ptr()), 70, 8, 200, 64, 0, 0, st), "slice")
    assert torch.equal(out, parts[1])


def test_channel_slice_bf16_and_accumulating_upsample_backward():
    """the bf16 pieces of the dense decoder: slice copies both ways, the accumulating forms (fp32 add, one rounding)"""
    from deadtrees_amd import _lib
    import ctypes as C
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    st = torch.cuda.current_stream().cuda_stream
    B, H, W = 2, 6, 10
    parts = [torch.randn((B, H, W, c), generator=g).to(torch.bfloat16).to(DEV) for c in (16, 8, 40)]
    wide = torch.empty((B, H, W, 64), dtype=torch.bfloat16, device=DEV)
    off = 0
    for t in parts:
        _lib.check(lib.dt_channel_slice_bf16(t.data_ptr(), wide.data_ptr(), B * H * W, t.shape[-1], 64, off, 1, 0, st), "slice")
        off += t.shape[-1]
    assert torch.equal(wide, torch.cat(parts, dim=-1))
    back = torch.empty_like(parts[2])
    _lib.check(lib.dt_channel_slice_bf16(wide.data_ptr(), back.data_ptr(), B * H * W, 40, 64, 24, 0, 0, st), "slice")
    assert torch.equal(back, parts[2])
    prev = torch.randn(parts[1].shape, generator=g).to(torch.bfloat16).to(DEV)
    acc = prev.clone()
    _lib.check(lib.dt_channel_slice_bf16(wide.data_ptr(), acc.data_ptr(), B * H * W, 8, 64, 16, 0, 1, st), "slice")
    assert torch.equal(acc, (prev.float() + parts[1].float()).to(torch.bfloat16))
    dup = torch.randn((B, 2 * H, 2 * W, 16), generator=g).to(torch.bfloat16).to(DEV)
    quad = dup.float().reshape(B, H, 2, W, 2, 16)
    quad = (quad[:, :, 0, :, 0] + quad[:, :, 0, :, 1]) + (quad[:, :, 1, :, 0] + quad[:, :, 1, :, 1])
    dx = torch.empty((B, H, W, 16), dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.dt_upsample2x_bwd_acc_bf16(dup.data_ptr(), dx.data_ptr(), 0, B, H, W, 16, st), "ups")
    assert torch.equal(dx, quad.to(torch.bfloat16))
    p2 = torch.randn(dx.shape, generator=g).to(torch.bfloat16).to(DEV)
    dx2 = p2.clone()
    _lib.check(lib.dt_upsample2x_bwd_acc_bf16(dup.data_ptr(), dx2.data_ptr(), 1, B, H, W, 16, st), "ups")
    assert torch.equal(dx2, (p2.float() + quad).to(torch.bfloat16))


@pytest.mark.parametrize("B,H,W,C,K", [(2, 64, 64, 3, 2), (1, 128, 160, 4, 3)])
def test_unetpp_forward_eval_parity_and_argmax(B, H, W, C, K):
    from deadtrees_amd.data.synthetic import synth_batch
    ref, m = _pair(C, K)
    img, _ = synth_batch(B, H, W, C, K, seed=5)
    ref.eval()
    m.eval()
    with torch.no_grad():
        want64 = copy.deepcopy(ref).double()(img.double())
        got = m(img.to(DEV)).cpu()
    scale, err = float(want64.abs().max()), float((got.double() - want64).abs().max())
    assert err <= 1e-4 * scale, (err, scale)
    top2 = want64.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4 * err
    am = m.predict_classes(img.to(DEV)).cpu()
    assert torch.equal(am, got.argmax(dim=1)) and torch.equal(am[safe], want64.argmax(dim=1)[safe])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_unetpp_training_step_gradients(mode):
    """loss + every parameter gradient of one step against the fp64 oracle.  eval = frozen BatchNorm (the
    well-conditioned case): each tensor within 1e-4 of its norm (measured 9.6e-6; a wiring error — a wrong member of a dense concat, a
    missing gradient contribution of one of a node's consumers — is O(0.1..1)); train = batch statistics: within 10x /
    5x (per tensor / overall) the fp32 CPU oracle's own distance from fp64, the yardstick of tests/test_model_gpu.py, plus
    2e-3 of the tensor's norm (single tensors where the CPU oracle happens to be exact to 1e-5 still see mask flips)."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.loss.seg_loss import seg_loss
    from oracle.train_ref import loss_from_logits
    ref, m = _pair(3, 2, seed=3)
    img, mask = synth_batch(2, 128, 128, 3, 2, seed=6)
    ref64, ref32 = copy.deepcopy(ref).double(), copy.deepcopy(ref)
    for mod in (ref64, ref32, m):
        mod.train(mode == "train")
    logits = m(img.to(DEV))
    loss, _, _ = seg_loss(logits, mask.to(DEV), None, ("GDICE", "FOCAL"))
    loss.backward()
    l64 = ref64(img.double())
    loss64, _ = loss_from_logits(l64, mask, ("GDICE", "FOCAL"))
    loss64.backward()
    loss_from_logits(ref32(img), mask, ("GDICE", "FOCAL"))[0].backward()
    e = float((logits.detach().cpu().double() - l64.detach()).abs().max())
    assert e <= 1e-4 * float(l64.detach().abs().max())
    assert float(loss.detach()) == pytest.approx(float(loss64.detach()), rel=2e-5)
    grads = m.smp_grad_dict()
    g32 = {k: p.grad for k, p in ref32.named_parameters()}
    assert set(grads) == {k for k, _ in ref64.named_parameters()}
    tot_h = tot_r = tot = 0.0
    worst = (0.0, "")
    for k, p in ref64.named_parameters():
        n = float(p.grad.norm()) + 1e-30
        eh = float((grads[k].double() - p.grad).norm())
        er = float((g32[k].double() - p.grad).norm())
        worst = max(worst, (eh / n, k))
        if mode == "eval":
            assert eh <= 1e-4 * n, (k, eh / n, er / n)
        else:
            assert eh <= 10.0 * er + 2e-3 * n, (k, eh / n, er / n)     # a flipped ReLU mask moves a tensor by ~1e-3
        tot_h += eh ** 2
        tot_r += er ** 2
        tot += n ** 2
    print(f"[unet++ {mode}] worst per-tensor gradient rel-L2 vs fp64 oracle {worst[0]:.2e} ({worst[1]}); overall "
          f"{(tot_h / tot) ** 0.5:.2e} (fp32 CPU oracle {(tot_r / tot) ** 0.5:.2e})")
    if mode == "train":
        assert tot_h ** 0.5 <= 5.0 * tot_r ** 0.5 + 1e-5 * tot ** 0.5
        sd_ref, sd = ref32.state_dict(), m.state_dict()
        for k in sd_ref:
            if k.endswith("running_mean") or k.endswith("running_var"):
                np.testing.assert_allclose(sd[k].cpu().numpy(), sd_ref[k].numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


def test_unetpp_trains_through_semsegment_and_hiptrainer():
    from deadtrees.network.segmodel import SemSegment
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.trainer import HipTrainer
    from deadtrees_amd.utils.config import default_network, default_training
    model = SemSegment(default_network(architecture="unet++"), default_training()).to(DEV)
    assert model.model.spec.decoder_kind == "unetplusplus" and len(model.model.spec.decoder) == 11
    img, mask = synth_batch(4, 64, 64, 3, 2, seed=9)
    img[:, 0] += 2.5 * mask.float()
    tr = HipTrainer(model.model, lr=3e-4)
    losses = [float(tr.step(img.to(DEV), mask.to(DEV))) for _ in range(12)]
    assert np.isfinite(losses).all() and min(losses[1:]) < losses[0], losses
    e0 = [float(tr.step(img.to(DEV), mask.to(DEV))) for _ in range(2)]
    assert np.isfinite(e0).all()
    # under AMP (round 3: bf16 dense decoder — teacher-forced parity in tests/test_bf16_e2e_gpu.py): training goes on, with
    # HIP-graph replay too; bf16 inference agrees with fp32 on the class maps
    trb = HipTrainer(model.model, lr=3e-4, precision="bf16", graph=True)
    lb = [float(trb.step(img.to(DEV), mask.to(DEV))) for _ in range(8)]
    assert np.isfinite(lb).all() and lb[-1] < losses[0], lb
    model.model.eval()
    a32 = model.model.predict_classes(img.to(DEV), dtype="uint8")
    a16 = model.model.predict_classes(img.to(DEV), dtype="uint8", precision="bf16")
    assert float((a32 == a16).float().mean()) > 0.97
