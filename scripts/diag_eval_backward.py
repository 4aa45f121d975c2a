"""Why does the RGBN / 3-class eval-mode (frozen-BatchNorm) backward sit above 1e-4 per tensor while RGB / 2-class meets it?
(tests/test_reference_surface_gpu.py::test_eval_mode_backward_matches_oracle)

Separates the candidates by running the SAME comparison against the fp64 oracle for every (in_channels, classes)
combination, on the Winograd engine and on the exact-fma direct kernels, with torch's own fp32 CPU result as the
yardstick, and — the conditioning probe — with the fp32 CPU oracle evaluated on an input perturbed by one fp32 ulp
(what any change of summation order amounts to).  Prints one line per configuration and the five worst tensors.

usage: python scripts/diag_eval_backward.py [size=128]
"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd.data.synthetic import synth_batch
from deadtrees_amd.loss.seg_loss import seg_loss
from deadtrees_amd.network.unet import UNetHIP
from oracle.train_ref import loss_from_logits
from oracle.unet_ref import make_oracle

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out_path = os.environ.get("DT_PARITY_REPORT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                               "gpurun_out", "diag_eval_backward.txt")
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def grads_of(model, img, mask, dtype):
    model.zero_grad()
    lg = model(img.to(dtype))
    loss, _ = loss_from_logits(lg, mask, ("GDICE", "FOCAL"))
    loss.backward()
    return {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}, lg.detach().double()


for C, K in ((3, 2), (4, 2), (3, 3), (4, 3)):
    ref = make_oracle(C, K, seed=2).eval()
    ref64 = copy.deepcopy(ref).double().eval()
    img, mask = synth_batch(2, S, S, C, K, seed=8)
    g64, l64 = grads_of(ref64, img, mask, torch.float64)
    g32, l32 = grads_of(ref, img, mask, torch.float32)
    # conditioning probe: the same fp32 CPU network on an input moved by ~1 fp32 ulp (relative 6e-8)
    gen = torch.Generator().manual_seed(3)
    img_p = img * (1.0 + 6e-8 * (2 * torch.rand(img.shape, generator=gen) - 1))
    g32p, _ = grads_of(ref, img_p, mask, torch.float32)
    res = {}
    for wino in (True, False):
        m = UNetHIP(in_channels=C, classes=K)
        m.load_state_dict(ref.state_dict())
        m.to("cuda").eval()
        m.engine.winograd = wino
        m.engine.overlap_wgrad = False
        lg = m(img.cuda())
        loss, _, _ = seg_loss(lg, mask.cuda(), None, ("GDICE", "FOCAL"))
        loss.backward()
        torch.cuda.synchronize()
        res[wino] = ({k: v.double() for k, v in m.smp_grad_dict().items()}, lg.detach().cpu().double())
    rows = []
    for k, g in g64.items():
        n = float(g.norm()) + 1e-30
        rows.append((float((res[True][0][k] - g).norm()) / n, float((res[False][0][k] - g).norm()) / n,
                     float((g32[k] - g).norm()) / n, float((g32p[k] - g32[k]).norm()) / n, k))
    rows.sort(reverse=True)
    lsc = float(l64.abs().max())
    say(f"[C={C} K={K} {S}x{S}] logits vs fp64 (of max|logit|): HIP winograd {float((res[True][1] - l64).abs().max()) / lsc:.2e}, "
        f"HIP direct {float((res[False][1] - l64).abs().max()) / lsc:.2e}, torch-CPU fp32 {float((l32 - l64).abs().max()) / lsc:.2e}; "
        f"tensors above 1e-4: winograd {sum(r[0] > 1e-4 for r in rows)}, direct {sum(r[1] > 1e-4 for r in rows)}, "
        f"torch-CPU fp32 {sum(r[2] > 1e-4 for r in rows)}, fp32-CPU under a 1-ulp input perturbation {sum(r[3] > 1e-4 for r in rows)} "
        f"of {len(rows)}")
    for r in rows[:5]:
        say("    winograd %.2e  direct %.2e  torch-CPU-fp32 %.2e  1-ulp-perturbed-fp32 %.2e  %s" % r)
try:
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "a") as f:
        f.write("\n".join(lines) + "\n")
except OSError:
    pass
