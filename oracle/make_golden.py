"""Generate tests/golden/*.npz by running the IMPORTED reference (build container only).

TEST INFRASTRUCTURE.  Usage (in the build container, where /root/reference exists):

    PYTHONPATH=/root/reference python -m oracle.make_golden

Imports deadtrees.loss.{losses,gdl} and deadtrees.utils.data_handling from the reference (plain
python modules needing only torch/numpy/scipy/pandas), evaluates them on seeded inputs and stores
INPUTS and OUTPUTS as data.  No reference source text is stored.  The fixtures travel to the GPU
box; the reference does not.

G1 losses   : cases (seed,B,K,H,W); logits f32, mask i64 (incl. an all-background sample and a
              class-missing case) -> one-hot checksum, GDICE, DICE, GWDICE, FOCAL(g=2), CE, BOUNDARY (int32
              truncated distance maps, as the loader produces them) and d(loss)/d(logits) from
              autograd through the reference callables for GDICE+FOCAL, DICE+FOCAL, +BOUNDARY, GWDICE+FOCAL.
G2 blocks   : make/unmake_blocks_vectorized on the toy of reference tests/test_tiler.py:57-65 and
              on a random (4,512,512) uint8 array split into 256-blocks.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _labels(g, B, K, H, W, force_bg=True, drop_class=None):
    grid = max(H // 8, 1)
    coarse = (torch.rand((B, grid, grid), generator=g) < 0.25)
    cls = torch.randint(1, K, (B, grid, grid), generator=g)
    lab = (coarse * cls).repeat_interleave(H // grid, 1).repeat_interleave(W // grid, 2)
    if drop_class is not None:
        lab[lab == drop_class] = 0
    if force_bg:
        lab[0] = 0
    return lab.to(torch.int64)


def main():
    sys.path.insert(0, "/root/reference")
    from deadtrees.loss.gdl import GeneralizedDiceLoss
    from deadtrees.loss.gwdl import GeneralizedWassersteinDiceLoss
    from deadtrees.loss.losses import (BoundaryLoss, CrossEntropy, DiceLoss, FocalLoss,
                                       class2one_hot, one_hot2dist)
    from deadtrees.utils.data_handling import make_blocks_vectorized, unmake_blocks_vectorized

    os.makedirs(OUT, exist_ok=True)
    cases = [(0, 2, 2, 32, 32, None), (1, 2, 3, 32, 32, None), (2, 4, 2, 64, 64, None),
             (3, 2, 3, 32, 32, 2), (4, 3, 2, 32, 64, None)]
    for seed, B, K, H, W, drop in cases:
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn((B, K, H, W), generator=g) * 2.0
        mask = _labels(g, B, K, H, W, force_bg=True, drop_class=drop)
        y = class2one_hot(mask, K)
        dist = torch.from_numpy(
            np.stack([one_hot2dist(y[i].numpy(), resolution=[1, 1]) for i in range(B)])
        )
        assert dist.dtype == torch.int32  # the reference's truncation quirk
        distf = dist.float()
        idc_all, idc_fg = list(range(K)), list(range(1, K))
        out = {"logits": logits.numpy(), "mask": mask.numpy(), "distmap": distf.numpy(),
               "onehot_sum": y.sum(dim=(0, 2, 3)).numpy().astype(np.int64),
               "onehot_dtype": str(y.dtype)}
        p = logits.softmax(dim=1)
        out["gdice"] = GeneralizedDiceLoss()(p, y).item()
        out["dice"] = DiceLoss(idc=idc_fg)(p, y).item()
        out["focal"] = FocalLoss(idc=idc_all, gamma=2)(p, y).item()
        out["ce"] = CrossEntropy(idc=idc_all)(p, y).item()
        out["boundary"] = BoundaryLoss(idc=idc_fg)(p, distf).item()
        gw_m = np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 0.5], [1.0, 0.5, 0.0]])[:K, :K]  # segmodel.py:119-121
        gwdl = GeneralizedWassersteinDiceLoss(dist_matrix=gw_m)
        out["gwdice"] = gwdl(p, torch.argmax(y, dim=1)).item()                      # the call of segmodel.py:176
        combos = {"GDICE+FOCAL": ("g", "f"), "DICE+FOCAL": ("d", "f"),
                  "GDICE+BOUNDARY+FOCAL": ("g", "b", "f"), "GWDICE+FOCAL": ("w", "f")}
        for name, parts in combos.items():
            lg = logits.clone().requires_grad_(True)
            pp = lg.softmax(dim=1)
            tot = 0
            if "g" in parts:
                tot = tot + GeneralizedDiceLoss()(pp, y)
            if "w" in parts:
                tot = tot + gwdl(pp, torch.argmax(y, dim=1))
            if "d" in parts:
                tot = tot + DiceLoss(idc=idc_fg)(pp, y)
            if "b" in parts:
                tot = tot + BoundaryLoss(idc=idc_fg)(pp, distf)
            if "f" in parts:
                tot = tot + FocalLoss(idc=idc_all, gamma=2)(pp, y)
            tot.backward()
            out[f"loss[{name}]"] = tot.item()
            out[f"dlogits[{name}]"] = lg.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"losses_s{seed}_b{B}k{K}_{H}x{W}.npz"), **out)

    # G2: block split / merge
    toy = np.array([np.arange(16).reshape(4, 4)] * 3)
    toy_blocks = make_blocks_vectorized(toy, 2)
    toy_merged = unmake_blocks_vectorized(toy_blocks[:, 0], 2, 4, 4)
    rng = np.random.default_rng(7)
    big = rng.integers(0, 256, (4, 512, 512), dtype=np.uint8)
    big_blocks = make_blocks_vectorized(big, 256)
    big_merged = unmake_blocks_vectorized(big_blocks[:, 1], 256, 512, 512)
    np.savez_compressed(
        os.path.join(OUT, "blocks.npz"), toy=toy, toy_blocks=toy_blocks, toy_merged=toy_merged,
        big_seed=7, big_blocks_sha256=hashlib.sha256(big_blocks.tobytes()).hexdigest(),
        big_blocks_shape=np.array(big_blocks.shape),
        big_merged_sha256=hashlib.sha256(big_merged.tobytes()).hexdigest())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
