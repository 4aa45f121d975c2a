"""The rounding-aware bf16 oracle (oracle/unet_bf16_ref.py) checked against itself on the CPU:

* teacher forcing closes the loop: tensors recorded from a free run with fp32 accumulation, fed as the forced inputs
  of an fp64-accumulating oracle, agree per unit to accumulation accuracy (<= 5e-4 relative L2) — the bound the GPU
  test tests/test_bf16_e2e_gpu.py then applies to the HIP path;
* free running it is chaotic in the way documented there: a 1e-6 input perturbation moves the logits by percents,
  which is why no end-to-end bound tighter than the oracle's own floor can be asked of any implementation;
* with every rounding removed (bf16 -> identity) its forward/backward arithmetic IS the plain fp32 oracle's: this
  pins the hand-written backward schedule of the oracle against torch autograd."""
import copy

import pytest
import torch

from oracle import unet_bf16_ref as R
from oracle.train_ref import loss_from_logits
from oracle.unet_ref import make_oracle


def _batch(B=2, H=64, W=64, C=3, K=2, seed=3):
    from deadtrees_amd.data.synthetic import synth_batch
    return synth_batch(B, H, W, C, K, seed=seed)


def _run(o, img, mask):
    lg = o.forward(img).clone().requires_grad_(True)
    loss, _ = loss_from_logits(lg, mask, ("GDICE", "FOCAL"))
    loss.backward()
    return lg.detach(), float(loss.detach()), o.backward(lg.grad)


def test_teacher_forced_oracle_agrees_with_its_own_free_run():
    ref = make_oracle(3, 2, seed=0).train()
    img, mask = _batch()
    a = R.Bf16TrainOracle(copy.deepcopy(ref), torch.float32, record=True, update_running=False)
    _, _, ga = _run(a, img, mask)
    forced = dict(a.rec)
    forced.update({f"grad:{k}": g for k, g in ga.items()})
    b = R.Bf16TrainOracle(copy.deepcopy(ref), torch.float64, forced=forced, update_running=False)
    _, _, gb = _run(b, img, mask)
    assert not b.unforced and len(b.errs) > 300
    for k, (rel, mx) in b.errs.items():
        assert (mx <= 1e-5) if ".bn." in k else (rel <= 5e-4), (k, rel, mx)
    for k in ga:
        assert float((ga[k] - gb[k]).norm()) <= 1e-5 * float(gb[k].norm()) + 1e-12, k


def test_free_running_oracle_is_chaotic_under_tiny_perturbations():
    ref = make_oracle(3, 2, seed=0).train()
    img, mask = _batch(2, 128, 128)
    g = torch.Generator().manual_seed(1)
    l1, _, _ = _run(R.Bf16TrainOracle(copy.deepcopy(ref), torch.float32, update_running=False), img, mask)
    l2, _, _ = _run(R.Bf16TrainOracle(copy.deepcopy(ref), torch.float32, update_running=False),
                    img * (1 + 1e-6 * torch.randn(img.shape, generator=g)), mask)
    rel = float((l1 - l2).norm() / l1.norm())
    assert 5e-3 < rel < 0.5, rel          # percents, from a 1e-6 perturbation: rounding flips cascade


def test_without_rounding_the_oracle_equals_plain_autograd(monkeypatch):
    monkeypatch.setattr(R, "rbf", lambda x: x)
    ref = make_oracle(3, 2, seed=0).train()
    img, mask = _batch()
    plain = copy.deepcopy(ref).double()
    lp = plain(img.double())
    loss, _ = loss_from_logits(lp, mask, ("GDICE", "FOCAL"))
    loss.backward()
    lo, loss_o, go = _run(R.Bf16TrainOracle(copy.deepcopy(ref), torch.float64, update_running=False), img, mask)
    # the oracle keeps its BatchNorm coefficients and elementwise affine maps in fp32 like the kernels: 1e-5, not 1e-12
    assert float((lo.double() - lp.detach()).abs().max()) <= 2e-5 * float(lp.detach().abs().max())
    assert loss_o == pytest.approx(float(loss.detach()), rel=1e-5)
    for k, p in plain.named_parameters():
        e = float((go[k].double() - p.grad).norm())
        assert e <= 2e-3 * float(p.grad.norm()) + 1e-6 * max(float(q.grad.norm()) for q in plain.parameters()), (k, e)
