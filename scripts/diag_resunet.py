"""per-tensor gradient error of the resunet / unet path vs the fp64 oracle's backward for the SAME upstream gradient"""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deadtrees_amd.data.synthetic import synth_batch
from deadtrees_amd.loss.seg_loss import seg_loss
from deadtrees_amd.network.unet import UNetHIP
from oracle.resunet_ref import make_resunet_oracle
from oracle.unet_ref import make_oracle
mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
kind = sys.argv[2] if len(sys.argv) > 2 else "resunet"
ref = make_resunet_oracle(3, 2, seed=3) if kind == "resunet" else make_oracle(3, 2, seed=3)
m = UNetHIP(decoder=kind); m.load_state_dict(ref.state_dict()); m.to("cuda")
img, mask = synth_batch(2, 128, 128, 3, 2, seed=6)
ref64, ref32 = copy.deepcopy(ref).double(), copy.deepcopy(ref)
for mod in (ref64, ref32, m): mod.train(mode == "train")
logits = m(img.cuda()); logits.retain_grad()
loss, _, _ = seg_loss(logits, mask.cuda(), None, ("GDICE", "FOCAL")); loss.backward()
dl = logits.grad.detach().cpu()
l64 = ref64(img.double()); l64.backward(dl.double())
l32 = ref32(img); l32.backward(dl)
g = m.smp_grad_dict(); g32 = {k: p.grad for k, p in ref32.named_parameters()}
print("head bias: hip", g["segmentation_head.0.bias"].tolist(), "exact", dl.double().sum(dim=(0, 2, 3)).tolist())
rows = []
for k, p in ref64.named_parameters():
    n = float(p.grad.norm()) + 1e-30
    rows.append((float((g[k].double()-p.grad).norm())/n, float((g32[k].double()-p.grad).norm())/n, k))
for e, e32, k in rows:
    if "blocks.4" in k or "blocks.3" in k or "head" in k: print(f"{e:.2e} {e32:.2e} {k}")
print("--- worst")
for e, e32, k in sorted(rows, reverse=True)[:6]: print(f"{e:.2e} {e32:.2e} {k}")
