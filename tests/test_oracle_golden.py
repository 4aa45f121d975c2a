"""Oracle restatement vs. the golden vectors produced by the imported reference
(oracle/make_golden.py).  CPU only."""
import glob
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref as L
from oracle import train_ref as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "losses_*.npz")))


def test_have_cases():
    assert len(CASES) >= 5


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_loss_values(path):
    z = np.load(path)
    logits = torch.from_numpy(z["logits"])
    mask = torch.from_numpy(z["mask"])
    dist = torch.from_numpy(z["distmap"])
    K = logits.shape[1]
    p = logits.softmax(dim=1)
    oh = L.one_hot(mask, K)
    assert str(oh.dtype) == str(z["onehot_dtype"]) == "torch.int32"
    np.testing.assert_array_equal(oh.sum(dim=(0, 2, 3)).numpy(), z["onehot_sum"])
    fg, al = list(range(1, K)), list(range(K))
    # fp64 restatement vs the reference's fp32 result: tolerance = fp32 rounding of the reductions
    assert float(L.gdice(p, mask)) == pytest.approx(float(z["gdice"]), rel=2e-6, abs=2e-7)
    assert float(L.dice(p, mask, fg)) == pytest.approx(float(z["dice"]), rel=2e-6, abs=2e-7)
    assert float(L.focal(p, mask, al, 2.0)) == pytest.approx(float(z["focal"]), rel=2e-6, abs=2e-7)
    assert float(L.cross_entropy(p, mask, al)) == pytest.approx(float(z["ce"]), rel=2e-6, abs=2e-7)
    assert float(L.boundary(p, dist, fg)) == pytest.approx(float(z["boundary"]), rel=2e-6, abs=2e-6)
    assert float(L.gwdice(p.double(), mask)) == pytest.approx(float(z["gwdice"]), rel=2e-6, abs=2e-7)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
def test_distmap_restatement(path):
    z = np.load(path)
    mask = torch.from_numpy(z["mask"])
    K = z["logits"].shape[1]
    oh = L.one_hot(mask, K).numpy()
    d = np.stack([L.dist_map(oh[i]) for i in range(oh.shape[0])]).astype(np.float32)
    np.testing.assert_array_equal(d, z["distmap"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c) for c in CASES])
@pytest.mark.parametrize("combo", ["GDICE+FOCAL", "DICE+FOCAL", "GDICE+BOUNDARY+FOCAL", "GWDICE+FOCAL"])
def test_differentiable_restatement_grads(path, combo):
    """train_ref.loss_from_logits (used by the CPU baseline / gradient oracle) vs autograd through
    the imported reference losses."""
    z = np.load(path)
    logits = torch.from_numpy(z["logits"]).clone().requires_grad_(True)
    mask = torch.from_numpy(z["mask"])
    dist = torch.from_numpy(z["distmap"])
    names = tuple(combo.split("+"))
    loss, _ = T.loss_from_logits(logits, mask, names, dist if "BOUNDARY" in names else None)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(z[f"loss[{combo}]"]), rel=1e-6, abs=1e-7)
    ref = z[f"dlogits[{combo}]"]
    np.testing.assert_allclose(logits.grad.numpy(), ref, rtol=1e-5, atol=1e-9 + 1e-6 * np.abs(ref).max())


def test_blocks_known_answer(golden_dir):
    z = np.load(os.path.join(golden_dir, "blocks.npz"))
    # reference tests/test_tiler.py:56-77 toy
    np.testing.assert_array_equal(L.make_blocks(z["toy"], 2), z["toy_blocks"])
    np.testing.assert_array_equal(L.unmake_blocks(z["toy_blocks"][:, 0], 2, 4, 4), z["toy_merged"])
    np.testing.assert_array_equal(z["toy_merged"], z["toy"][0])
    rng = np.random.default_rng(int(z["big_seed"]))
    big = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)
    blocks = L.make_blocks(big, 256)
    assert list(blocks.shape) == list(z["big_blocks_shape"])
    assert hashlib.sha256(blocks.tobytes()).hexdigest() == str(z["big_blocks_sha256"])
    merged = L.unmake_blocks(blocks[:, 1], 256, 512, 512)
    assert hashlib.sha256(merged.tobytes()).hexdigest() == str(z["big_merged_sha256"])


def test_unet_ref_shapes_and_keys():
    from oracle.unet_ref import make_oracle
    m = make_oracle(3, 2)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 24_436_514  # SURVEY A.2 (K=2, Cin=3)
    sd = m.state_dict()
    for k in ["encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer4.2.bn2.num_batches_tracked", "decoder.blocks.0.conv1.0.weight",
              "decoder.blocks.4.conv2.1.bias", "segmentation_head.0.weight", "segmentation_head.0.bias"]:
        assert k in sd, k
    assert tuple(sd["decoder.blocks.0.conv1.0.weight"].shape) == (256, 768, 3, 3)
    assert tuple(sd["decoder.blocks.4.conv1.0.weight"].shape) == (16, 32, 3, 3)
    m.eval()
    with torch.no_grad():
        y = m(torch.randn(1, 3, 64, 64))
    assert tuple(y.shape) == (1, 2, 64, 64)
    # K=1 gives the well-known smp parameter count
    from oracle.unet_ref import UNetR34Ref
    assert sum(p.numel() for p in UNetR34Ref(3, 1).parameters()) == 24_436_369


def test_resunet_decoder_oracle_against_the_executed_reference(golden_dir):
    """oracle/resunet_ref.ResUnetDecoderRef against tests/golden/resunet_decoder.npz (the reference's own
    ResUnetDecoder, executed by oracle/make_golden_resunet.py): output, every parameter gradient, every feature
    gradient, BatchNorm running statistics — and, with the identity convolutions zeroed, the PLAIN U-Net decoder of
    oracle/unet_ref.py, which pins its wiring (feature order, nearest x2, cat([x, skip]), channel arithmetic)."""
    import torch
    from oracle.resunet_ref import ResUnetDecoderRef
    from oracle.unet_ref import UnetDecoder
    z = np.load(os.path.join(golden_dir, "resunet_decoder.npz"))
    enc_ch, dec_ch = tuple(int(v) for v in z["enc_ch"]), tuple(int(v) for v in z["dec_ch"])
    dec = ResUnetDecoderRef(enc_ch, dec_ch)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    assert set(sd) == set(dec.state_dict())                  # same parameter / buffer names as the reference module
    dec.load_state_dict(sd)
    dec.train()
    feats = [None] + [torch.from_numpy(z[f"feat{i}"]).requires_grad_(True) for i in range(1, 6)]
    out = dec(*feats)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=1e-5, atol=1e-5)
    (out * torch.from_numpy(z["gout"])).sum().backward()
    for i in range(1, 6):
        np.testing.assert_allclose(feats[i].grad.numpy(), z[f"dfeat{i}"], rtol=1e-4, atol=1e-5)
    for k, p in dec.named_parameters():
        ref = z[f"grad:{k}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    for k, v in dec.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), z[f"after:{k}"], rtol=1e-5, atol=1e-6)
    # plain U-Net decoder = the same weights without the 1x1 residual branch
    plain = UnetDecoder(enc_ch, dec_ch)
    psd = {k: v for k, v in sd.items() if "identity_conv" not in k}
    assert set(psd) == set(plain.state_dict())
    plain.load_state_dict(psd)
    plain.train()
    with torch.no_grad():
        got = plain(*[None if f is None else f.detach() for f in feats])
    np.testing.assert_allclose(got.numpy(), z["out_plain"], rtol=1e-5, atol=1e-5)
    plain.eval()
    with torch.no_grad():
        got = plain(*[None if f is None else f.detach() for f in feats])
    np.testing.assert_allclose(got.numpy(), z["out_plain_eval"], rtol=1e-5, atol=1e-5)


def test_unetplusplus_decoder_oracle_against_the_executed_reference_wiring(golden_dir):
    """oracle/unetpp_ref.UnetPlusPlusDecoderRef against tests/golden/unetpp_decoder.npz: the reference's own dense decoder
    class (network/extra/efficientunetplusplus/decoder.py — smp's UnetPlusPlusDecoder constructor and forward loop),
    executed by oracle/make_golden_unetpp.py with smp's plain decoder block: same state_dict names (x_{depth}_{layer}),
    output, every parameter and feature gradient, BatchNorm running statistics."""
    import torch
    from oracle.unetpp_ref import UnetPlusPlusDecoderRef
    z = np.load(os.path.join(golden_dir, "unetpp_decoder.npz"))
    enc_ch, dec_ch = tuple(int(v) for v in z["enc_ch"]), tuple(int(v) for v in z["dec_ch"])
    dec = UnetPlusPlusDecoderRef(enc_ch, dec_ch)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    assert set(sd) == set(dec.state_dict()) and len(dec.blocks) == 11
    dec.load_state_dict(sd)
    dec.train()
    feats = [None] + [torch.from_numpy(z[f"feat{i}"]).requires_grad_(True) for i in range(1, 6)]
    out = dec(*feats)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=1e-5, atol=1e-5)
    (out * torch.from_numpy(z["gout"])).sum().backward()
    for i in range(1, 6):
        np.testing.assert_allclose(feats[i].grad.numpy(), z[f"dfeat{i}"], rtol=1e-4, atol=1e-5)
    for k, p in dec.named_parameters():
        ref = z[f"grad:{k}"]
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())))
    for k, v in dec.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), z[f"after:{k}"], rtol=1e-5, atol=1e-6)
