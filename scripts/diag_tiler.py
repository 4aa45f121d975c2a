"""where the tiler-inclusive milliseconds go: one 2048x2048 RGBN raster through infer_tile, phase by phase (host clock, device
synchronised between phases) — scripts/diag_tiler.py on the GPU box"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deadtrees_amd import ops
from deadtrees_amd.data.synthetic import MEAN, STD
from deadtrees_amd.deployment.tiler import infer_tile
from deadtrees_amd.network.unet import UNetHIP

dev = torch.device("cuda:0")
m = UNetHIP(in_channels=3, classes=2)
m.reset_parameters(seed=0)
m.to(dev).eval()


class Inf:
    in_channels = 3

    def run_blocks(self, raster, d, first, count):
        return m.predict_classes(ops.split_normalize_u8(raster, d, first, count, MEAN, STD, 3), dtype="uint8", nhwc=True)


ortho = np.random.default_rng(7).integers(0, 256, (4, 2048, 2048), dtype=np.uint8)
for _ in range(3):
    infer_tile(Inf(), ortho, subtile=256, batch_size=64, device="cuda:0")
torch.cuda.synchronize()


def clock(fn, n=10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r


t_all, _ = clock(lambda: infer_tile(Inf(), ortho, subtile=256, batch_size=64, device="cuda:0"))
t_h2d, x = clock(lambda: torch.from_numpy(np.ascontiguousarray(ortho[:3])).to(dev, non_blocking=True))
t_fwd, maps = clock(lambda: Inf().run_blocks(x, 256, 0, 64))
t_merge, merged = clock(lambda: maps.view(8, 8, 256, 256).permute(0, 2, 1, 3).reshape(2048, 2048)[:2048, :2048].contiguous())
t_d2h, _ = clock(lambda: merged.cpu().numpy())
pin = torch.from_numpy(ortho[:3].copy()).pin_memory()
t_h2d_pin, _ = clock(lambda: pin.to(dev, non_blocking=True))
print(f"infer_tile {t_all:.3f} ms = H2D {t_h2d:.3f} (pinned source: {t_h2d_pin:.3f}) + split/normalise/forward/argmax {t_fwd:.3f} "
      f"+ merge {t_merge:.3f} + D2H {t_d2h:.3f}  (sum {t_h2d + t_fwd + t_merge + t_d2h:.3f})")
