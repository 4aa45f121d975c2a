"""Run the bf16 training step twice from the same state and report the first traced tensors whose bits differ.
Usage (GPU box): python scripts/diag_determinism.py [B] [S] [dma mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd import _lib
from deadtrees_amd.data.synthetic import synth_batch
from deadtrees_amd.loss.seg_loss import loss_backward, loss_forward
from deadtrees_amd.network.unet import UNetHIP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_lib.load().dt_set_option(b"bf16_dma", dma)


class Sums(dict):
    def __setitem__(self, k, v):
        iv = v.contiguous().view(torch.int16 if v.element_size() == 2 else torch.int32)
        super().__setitem__(k, (int(iv.to(torch.int64).sum()), tuple(v.shape), str(v.dtype)))


def run(img, mask, reps):
    out = []
    for _ in range(reps):
        m = UNetHIP()
        m.reset_parameters(seed=0)
        m = m.to("cuda")
        m.precision = "bf16"
        eng = m.engine
        eng.trace = Sums()
        params, grads = m.flat_params.detach(), m._grad_buffer()
        with torch.no_grad():
            logits = eng.forward_bf16_train(img, params, m.bn_state)
            parts, err, saved = loss_forward(logits, mask, None, {"losses": ["GDICE", "FOCAL"], "alpha": 0.0})
            dl = loss_backward(saved)
            eng.backward_bf16(dl, params, grads)
        torch.cuda.synchronize()
        tr = dict(eng.trace)
        tr["~logits"] = (int(logits.view(torch.int32).to(torch.int64).sum()), tuple(logits.shape), "f32")
        tr["~grads"] = (int(grads.view(torch.int32).to(torch.int64).sum()), tuple(grads.shape), "f32")
        tr["~loss"] = (float(parts[7]), (), "f32")
        out.append(tr)
        eng.trace = None
        del m, eng
    return out


img, mask = synth_batch(B, S, S, 3, 2, seed=4321)
img, mask = img.cuda(), mask.cuda()
runs = run(img, mask, 3)
keys = list(runs[0].keys())
bad = [k for k in keys if any(r[k] != runs[0][k] for r in runs[1:])]
print(f"B={B} S={S} dma={dma}: {len(keys)} traced tensors, {len(bad)} differ between runs")
for k in bad[:40]:
    print("  ", k, [r[k][0] for r in runs], runs[0][k][1:])

# ---- whole trainer steps: loss / gradient / parameter / BatchNorm-state / Adam-moment checksums after every step
from deadtrees_amd.trainer import HipTrainer


def cs(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())


def steps(n):
    m = UNetHIP()
    m.reset_parameters(seed=0)
    m = m.to("cuda")
    tr = HipTrainer(m, precision="bf16")
    rows = []
    for _ in range(n):
        loss = tr.step(img, mask)
        torch.cuda.synchronize()
        opt = tr.opt
        row = {"loss": cs(loss.reshape(1)), "grads": cs(m._grad_buffer()), "params": cs(m.flat_params), "bn": cs(m.bn_state),
               "norm": cs(tr.last["grad_norm"].reshape(1))}
        for name in ("m", "v", "exp_avg", "exp_avg_sq"):
            if hasattr(opt, name) and torch.is_tensor(getattr(opt, name)):
                row[name] = cs(getattr(opt, name))
        rows.append(row)
    return rows


a, b = steps(3), steps(3)
for i, (ra, rb) in enumerate(zip(a, b)):
    print("step", i, {k: ("same" if ra[k] == rb[k] else "DIFF") for k in ra})


def steps_nosync(n):
    m = UNetHIP()
    m.reset_parameters(seed=0)
    m = m.to("cuda")
    tr = HipTrainer(m, precision="bf16")
    losses = [tr.step(img, mask).clone() for _ in range(n)]
    torch.cuda.synchronize()
    return [cs(l.reshape(1)) for l in losses], cs(m.flat_params), cs(m.bn_state)


for rep in range(3):
    print("no host sync between steps:", steps_nosync(4))
