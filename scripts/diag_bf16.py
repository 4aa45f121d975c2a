import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deadtrees_amd.data.synthetic import synth_batch
from deadtrees_amd.loss.seg_loss import seg_loss
from deadtrees_amd.network.unet import UNetHIP
from oracle.unet_ref import make_oracle
DEV = "cuda"
for (B, S, rand_bn) in [(2, 128, True), (2, 128, False), (4, 256, False)]:
    ref = make_oracle(3, 2, seed=0, randomize_bn=rand_bn)
    img, mask = synth_batch(B, S, S, 3, 2, seed=3)
    img, mask = img.to(DEV), mask.to(DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        m = UNetHIP(); m.load_state_dict(ref.state_dict()); m.to(DEV).train(); m.precision = prec
        logits = m(img)
        loss, _, _ = seg_loss(logits, mask, None, ("GDICE", "FOCAL"))
        loss.backward()
        res[prec] = (float(loss.detach()), m._grad_buffer().clone(), logits.detach().clone())
    l32, g32, lg32 = res["fp32"]; l16, g16, lg16 = res["bf16"]
    cos = float((g16.double() * g32.double()).sum() / (g16.double().norm() * g32.double().norm()))
    print(f"B={B} S={S} rand_bn={rand_bn}: loss {l32:.5f} vs {l16:.5f}; logits rel L2 {float((lg16-lg32).norm()/lg32.norm()):.3e}; "
          f"grad cos {cos:.4f}; grad rel L2 {float((g16-g32).norm()/g32.norm()):.3e}; |g32| {float(g32.norm()):.3e} |g16| {float(g16.norm()):.3e}")
