"""Device signed distance maps (dt_signed_distmap) — bit-exact against the reference's loader pass.

Checkers: (1) the committed golden vectors (tests/golden/losses_*.npz hold `distmap` arrays produced by the
reference's own one_hot2dist, oracle/make_golden.py), (2) oracle.losses_ref.dist_map (scipy EDT, the reference's
dependency) on seeded masks incl. the edge cases: absent class, class covering the whole tile (scipy's phantom
background pixel at (-1, 0)), single pixels, ragged (non-square, non-multiple-of-64) tiles, 3 classes.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "losses_*.npz")))


def _oracle(mask: np.ndarray, K: int) -> np.ndarray:
    from oracle.losses_ref import dist_map
    out = np.empty((mask.shape[0], K) + mask.shape[1:], np.float32)
    for i in range(mask.shape[0]):
        oh = (mask[i][None] == np.arange(K)[:, None, None]).astype(np.int32)
        out[i] = dist_map(oh).astype(np.float32)
    return out


def _blobs(seed, B, K, H, W, n=6):
    rng = np.random.default_rng(seed)
    m = np.zeros((B, H, W), np.int64)
    yy, xx = np.mgrid[:H, :W]
    for b in range(B):
        for _ in range(n):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, max(2, min(H, W) // 6))
            m[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rng.integers(1, K)
    return m


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_distmap_matches_reference_golden(path):
    from deadtrees_amd import ops
    z = np.load(path)
    mask = torch.from_numpy(z["mask"]).to(DEV)
    K = z["logits"].shape[1]
    d, err = ops.signed_distmap(mask, K)
    assert int(err) == 0
    np.testing.assert_array_equal(d.cpu().numpy(), z["distmap"])


@pytest.mark.parametrize("B,K,H,W", [(2, 2, 256, 256), (2, 3, 512, 512), (3, 2, 96, 160), (1, 3, 37, 301),
                                     (2, 2, 1, 70), (2, 2, 70, 1)])
def test_distmap_matches_scipy_oracle(B, K, H, W):
    from deadtrees_amd import ops
    m = _blobs(B * 100 + K, B, K, H, W)
    d, err = ops.signed_distmap(torch.from_numpy(m).to(DEV), K)
    assert int(err) == 0
    np.testing.assert_array_equal(d.cpu().numpy(), _oracle(m, K))


def test_distmap_edge_cases():
    from deadtrees_amd import ops
    H = W = 128
    m = np.zeros((6, H, W), np.int64)
    m[1] = 1                                  # class 1 covers everything, class 0 absent
    m[2, 5, 7] = 1                            # a single pixel
    m[3, :, : W // 2] = 1                     # half plane
    m[4, 0, 0] = 1; m[4, H - 1, W - 1] = 1    # corners
    m[5, ::2, ::2] = 1                        # checkerboard-ish
    d, err = ops.signed_distmap(torch.from_numpy(m).to(DEV), 2)
    assert int(err) == 0
    ref = _oracle(m, 2)
    np.testing.assert_array_equal(d.cpu().numpy(), ref)
    assert (ref[0, 1] == 0).all() and ref[1, 1, 0, 0] == 0 and ref[1, 1, H - 1, W - 1] < -100  # what is being pinned


def test_distmap_flags_bad_labels():
    from deadtrees_amd import ops
    m = torch.zeros((1, 64, 64), dtype=torch.int64, device=DEV)
    m[0, 3, 3] = 2
    _, err = ops.signed_distmap(m, 2)
    assert int(err) == 1


def test_boundary_training_uses_device_distmap():
    """SemSegment-style step with BOUNDARY and no loader-provided distmap: maps come from the HIP kernel and the
    loss equals the oracle's with scipy maps."""
    from deadtrees_amd.data.synthetic import synth_batch
    from deadtrees_amd.data.distmap import distmaps_on_device
    from deadtrees_amd.loss.seg_loss import seg_loss
    from oracle import losses_ref as L
    g = torch.Generator().manual_seed(3)
    _, mask = synth_batch(2, 128, 128, 3, 2, seed=11)
    logits = torch.randn((2, 2, 128, 128), generator=g)
    dist = distmaps_on_device(mask.to(DEV), 2)
    _, parts, err = seg_loss(logits.to(DEV), mask.to(DEV), dist, ("GDICE", "BOUNDARY"))
    p = torch.softmax(logits.double(), 1)
    ref = float(L.boundary(p, torch.from_numpy(_oracle(mask.numpy(), 2)), [1]))
    assert float(parts["boundary_loss"]) == pytest.approx(ref, rel=1e-5, abs=1e-6)
