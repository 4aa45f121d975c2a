"""Time dt_signed_distmap on a training batch of label tiles (B x 512 x 512, K classes) against scipy on the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deadtrees_amd import ops
from deadtrees_amd.data.distmap import distmaps_for_batch
from deadtrees_amd.data.synthetic import synth_batch

B, K, S = int(os.environ.get("B", "32")), int(os.environ.get("K", "2")), int(os.environ.get("S", "512"))
_, mask = synth_batch(B, S, S, 3, K, seed=0)
m = mask.cuda()
ops.signed_distmap(m, K)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    d, _ = ops.signed_distmap(m, K)
e1.record()
torch.cuda.synchronize()
gpu_ms = e0.elapsed_time(e1) / 10
n = min(B, 4)
t0 = time.perf_counter()
ref = distmaps_for_batch(mask[:n], K)
cpu_ms = (time.perf_counter() - t0) * 1e3 / n * B
print({"B": B, "K": K, "size": S, "gpu_ms_per_batch": round(gpu_ms, 3), "scipy_ms_per_batch_1core": round(cpu_ms, 1),
       "equal": bool((d[:n].cpu() == ref).all()), "fg_frac": float((mask > 0).float().mean())})
