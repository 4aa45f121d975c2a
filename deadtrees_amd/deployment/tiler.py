"""Block split / merge of the reference tiler (deadtrees/utils/data_handling.py:9-34, used by
deployment/tiler.py:142-170) and a tile-queue inference driver that shards sub-tile batches over ranks.

The reference cuts a zero-padded 2048x2048 tile into NON-overlapping sub-tiles and pastes predictions back
(no overlap stitching exists in the reference — SURVEY fact 8); reassembly must be bit-exact.
GeoTIFF I/O (rioxarray) is out of scope: arrays in, arrays out.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


def divisible_without_remainder(a, b):
    if b == 0:
        return False
    return True if a % b == 0 else False


def make_blocks_vectorized(x: np.ndarray, d: int) -> np.ndarray:
    """[C,M,N] -> [(M/d)*(N/d), C, d, d] (row-major over blocks)"""
    p, m, n = x.shape
    return x.reshape(p, m // d, d, n // d, d).transpose(1, 3, 0, 2, 4).reshape(-1, p, d, d)


def unmake_blocks_vectorized(x, d: int, m: int, n: int) -> np.ndarray:
    """sequence of [b,d,d] batches -> [m,n]"""
    return np.concatenate(x).reshape(m // d, n // d, d, d).transpose(0, 2, 1, 3).reshape(m, n)


class Tiler:
    """array-level equivalent of reference deployment/tiler.py:59-170"""

    def __init__(self, tile_size: int = 2048, subtile_size: int = 256):
        if not divisible_without_remainder(tile_size, subtile_size):
            raise ValueError(f"Tile size not divisible by subtile size: {tile_size}, {subtile_size}")
        self.tile_size, self.subtile_size = tile_size, subtile_size
        self._source: Optional[np.ndarray] = None
        self._shape = None
        self._batch_shape = None

    def load_array(self, arr_chw_u8: np.ndarray):
        c, h, w = arr_chw_u8.shape
        if h > self.tile_size or w > self.tile_size:
            raise ValueError("tile larger than tile_size")
        self._shape = (h, w)
        pad = np.zeros((c, self.tile_size, self.tile_size), dtype=arr_chw_u8.dtype)
        pad[:, :h, :w] = arr_chw_u8
        self._source = pad

    def get_batches(self, batch_size: int = 64) -> List[np.ndarray]:
        subtiles = make_blocks_vectorized(self._source, self.subtile_size)
        self._batch_shape = len(subtiles)
        n = max(len(subtiles) // batch_size, 1)
        return np.array_split(subtiles, n, axis=0)

    def put_batches(self, batches) -> np.ndarray:
        merged = unmake_blocks_vectorized(batches, self.subtile_size, self.tile_size, self.tile_size)
        h, w = self._shape
        return merged[:h, :w]


def infer_tile(inference, arr_chw_u8: np.ndarray, subtile: int = 256, batch_size: int = 64, rank: int = 0,
               world: int = 1, device: str = "cuda", group=None) -> np.ndarray:
    """whole-tile inference; with world > 1 batches j = rank (mod world) are processed locally and the uint8
    class maps are all-gathered (no other collective: tiles are independent units)."""
    t = Tiler(max(arr_chw_u8.shape[1], arr_chw_u8.shape[2]) if arr_chw_u8.shape[1] % subtile == 0 and
              arr_chw_u8.shape[1] == arr_chw_u8.shape[2] else 2048, subtile)
    t.load_array(arr_chw_u8)
    batches = t.get_batches(batch_size)
    outs: List[Optional[np.ndarray]] = [None] * len(batches)
    for j, b in enumerate(batches):
        if j % world != rank:
            continue
        u8 = torch.from_numpy(np.ascontiguousarray(b.transpose(0, 2, 3, 1)))     # [B,d,d,C] uint8
        outs[j] = inference.run_u8(u8, device=device).cpu().numpy()
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, [(j, o) for j, o in enumerate(outs) if o is not None], group=group)
        for lst in gathered:
            for j, o in lst:
                outs[j] = o
    return t.put_batches(outs)
