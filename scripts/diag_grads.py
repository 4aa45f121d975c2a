import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deadtrees_amd.network.unet import UNetHIP
from deadtrees_amd.loss.seg_loss import seg_loss
from deadtrees_amd.data.synthetic import synth_batch
from oracle.unet_ref import make_oracle
from oracle.train_ref import loss_from_logits
B,H,W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[2])
ref = make_oracle(3,2,seed=0); m = UNetHIP(); m.load_state_dict(ref.state_dict()); m.to("cuda").train()
ref64 = copy.deepcopy(ref).double().train(); ref.train()
img, mask = synth_batch(B,H,W)
l32 = ref(img); loss32,_ = loss_from_logits(l32, mask); loss32.backward()
l64 = ref64(img.double()); loss64,_ = loss_from_logits(l64, mask); loss64.backward()
lg = m(img.cuda()); loss,_,_ = seg_loss(lg, mask.cuda()); loss.backward()
print("loss", float(loss), float(loss32), float(loss64))
print("logit err hip", float((lg.detach().cpu().double()-l64.detach()).abs().max()), "ref32", float((l32.detach().double()-l64.detach()).abs().max()))
g = m.smp_grad_dict(); g32 = {k:p.grad for k,p in ref.named_parameters()}; g64 = {k:p.grad for k,p in ref64.named_parameters()}
rows=[]
for k in g64:
    n = float(g64[k].norm())+1e-30
    rows.append((float((g[k].double()-g64[k]).norm())/n, float((g32[k].double()-g64[k]).norm())/n, k, n))
rows.sort(reverse=True)
for r in rows[:25]: print("hip %.3e  ref32 %.3e  %s  |g|=%.3e" % r)
