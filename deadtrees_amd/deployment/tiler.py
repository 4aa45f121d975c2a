"""``Tiler`` — the reference's whole-tile splitter/merger (deadtrees/deployment/tiler.py:22-170) with the same
constructor, ``load_file`` / ``get_batches`` / ``put_batches`` / ``write_file`` contract, plus an array-level entry
(``load_array``) and a rank-sharded tile-queue driver (``infer_tile``) for the MI355X path.

Contract kept from the reference (scripts/inference.py:80-115 runs against it unchanged):

* ``Tiler(infile=None, tile_shape=(2048, 2048), subtile_shape=(256, 256))``; non-square sub-tiles -> ``ValueError``;
* the source raster is zero-padded to ``tile_shape`` (tiler.py:108-114) and cut into NON-overlapping sub-tiles with
  ``make_blocks_vectorized`` (utils/data_handling.py:9-19) — there is no overlap stitching in the reference;
* ``get_batches() -> ndarray [n_used, C, d, d]``: only the sub-tiles that intersect the valid raster
  (``_subtiles_to_use``, tiler.py:121-134) — all-padding sub-tiles never reach the network;
* ``put_batches(ndarray [n_used, d, d])``: the skipped sub-tiles are zero-filled, blocks are merged with
  ``unmake_blocks_vectorized`` (data_handling.py:22-34) into the padded ``_outdata`` and cropped to the raster size.

GeoTIFF I/O goes through rioxarray exactly like the reference when that package is importable; it is absent from
this image, so ``load_file`` / ``write_file`` raise an ``ImportError`` that says so, and ``load_array`` /
``result`` are the array-level way in and out (what the tests and the bench use).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Tuple, Union

import numpy as np
import torch


def divisible_without_remainder(a, b):
    if b == 0:
        return False
    return True if a % b == 0 else False


def make_blocks_vectorized(x: np.ndarray, d: int) -> np.ndarray:
    """[C,M,N] -> [(M/d)*(N/d), C, d, d] (row-major over blocks) — utils/data_handling.py:9-19"""
    p, m, n = x.shape
    return x.reshape(p, m // d, d, n // d, d).transpose(1, 3, 0, 2, 4).reshape(-1, p, d, d)


def unmake_blocks_vectorized(x, d: int, m: int, n: int) -> np.ndarray:
    """blocks [k,d,d] (or a sequence of such batches) -> [m,n] — utils/data_handling.py:22-34"""
    return np.concatenate(x).reshape(m // d, n // d, d, d).transpose(0, 2, 1, 3).reshape(m, n)


@dataclass
class TileInfo:
    size: Tuple[int, int]
    subtiles: Tuple[int, int]


def _rioxarray():
    try:
        import rioxarray  # type: ignore
        return rioxarray
    except Exception as e:  # noqa: BLE001
        raise ImportError("GeoTIFF I/O needs the `rioxarray` package (reference deployment/tiler.py:15), which is not "
                          "installed here: use Tiler.load_array(array) / Tiler.result for in-memory rasters") from e


def inspect_tile(infile, tile_shape: Tuple[int, int] = (8192, 8192),
                 subtile_shape: Tuple[int, int] = (512, 512)) -> TileInfo:
    """reference tiler.py:34-56; ``infile``: a path (rioxarray), or anything with ``.shape`` [(C,)H,W]"""
    if hasattr(infile, "shape"):
        shape = tuple(int(v) for v in infile.shape[-2:])
    else:
        with _rioxarray().open_rasterio(infile).sel(band=1, drop=True) as da:
            shape = tuple(da.shape)
    if not divisible_without_remainder(tile_shape[0], subtile_shape[0]):
        raise ValueError(f"Shapes unaligned (v): {tile_shape[0], subtile_shape[0]}")
    if not divisible_without_remainder(tile_shape[1], subtile_shape[1]):
        raise ValueError(f"Shapes unaligned (h): {tile_shape[1], subtile_shape[1]}")
    subtiles = (math.ceil(shape[0] / subtile_shape[0]), math.ceil(shape[1] / subtile_shape[1]))
    return TileInfo(size=shape, subtiles=subtiles)


class Tiler:
    def __init__(self, infile: Optional[Union[str, Path]] = None, tile_shape: Optional[Tuple[int, int]] = (2048, 2048),
                 subtile_shape: Optional[Tuple[int, int]] = (256, 256)) -> None:
        self._infile = infile
        self._tile_shape = tuple(tile_shape)
        self._subtile_shape = tuple(subtile_shape)
        if subtile_shape[0] != subtile_shape[1]:
            raise ValueError("Subtile required to have matching x/y dims")
        self._source = None
        self._target = None
        self._indata: Optional[np.ndarray] = None
        self._outdata: Optional[np.ndarray] = None
        self._batch_shape = None
        self._subtiles_to_use: Optional[np.ndarray] = None
        self._tile_info: Optional[TileInfo] = None

    # ------------------------------------------------------------------ sources
    def _set_shapes(self, tile_shape, subtile_shape):
        self._tile_shape = tuple(tile_shape) if tile_shape else self._tile_shape
        if subtile_shape:
            if subtile_shape[0] != subtile_shape[1]:
                raise ValueError("Subtile required to have matching x/y dims")
        self._subtile_shape = tuple(subtile_shape) if subtile_shape else self._subtile_shape

    def _stage(self, sv: np.ndarray):
        """pad to the tile shape, allocate the output plane, mark the sub-tiles that hold data (tiler.py:106-134)"""
        if sv.shape[1] > self._tile_shape[0] or sv.shape[2] > self._tile_shape[1]:
            raise ValueError(f"raster {sv.shape[1:]} larger than the tile shape {self._tile_shape}")
        if tuple(self._tile_shape) != tuple(self._tile_info.size):
            self._indata = np.zeros((sv.shape[0], *self._tile_shape), dtype=sv.dtype)
            self._indata[:, 0:sv.shape[1], 0:sv.shape[2]] = sv
        else:
            self._indata = sv
        self._outdata = np.zeros(self._tile_shape, dtype="uint8")
        mask = np.zeros((self._tile_shape[0] // self._subtile_shape[0], self._tile_shape[1] // self._subtile_shape[1]),
                        dtype=bool)
        mask[0:self._tile_info.subtiles[0], 0:self._tile_info.subtiles[1]] = 1
        self._subtiles_to_use = mask.ravel()
        self._batch_shape = None

    def load_file(self, infile: Union[str, Path], tile_shape: Optional[Tuple[int, int]] = None,
                  subtile_shape: Optional[Tuple[int, int]] = None) -> None:
        """reference tiler.py:82-134 (needs rioxarray)"""
        rio = _rioxarray()
        self._infile = infile
        self._set_shapes(tile_shape, subtile_shape)
        self._tile_info = inspect_tile(self._infile, self._tile_shape, self._subtile_shape)
        self._source = rio.open_rasterio(self._infile, chunks={"band": 4, "x": 256, "y": 256})
        self._target = self._source.sel(band=1, drop=True).astype("uint8").copy(deep=True)
        self._stage(self._source.values)

    def load_array(self, arr_chw: np.ndarray, tile_shape: Optional[Tuple[int, int]] = None,
                   subtile_shape: Optional[Tuple[int, int]] = None) -> None:
        """the in-memory twin of ``load_file``: a [C,H,W] raster (what ``rioxarray.open_rasterio(f).values`` holds)"""
        arr_chw = np.asarray(arr_chw)
        if arr_chw.ndim != 3:
            raise ValueError(f"expected a [C,H,W] raster, got shape {arr_chw.shape}")
        self._infile = None
        self._set_shapes(tile_shape, subtile_shape)
        self._tile_info = inspect_tile(arr_chw, self._tile_shape, self._subtile_shape)
        self._source, self._target = arr_chw, None
        self._stage(arr_chw)

    # ------------------------------------------------------------------ batches
    def get_batches(self) -> np.ndarray:
        if self._indata is None:
            raise RuntimeError("Tiler: load_file / load_array first")
        subtiles = make_blocks_vectorized(self._indata, self._subtile_shape[0])
        self._batch_shape = self._batch_shape or subtiles.shape
        return subtiles[self._subtiles_to_use]

    def put_batches(self, batches: np.ndarray) -> None:
        batches = np.asarray(batches)
        n_used = int(self._subtiles_to_use.sum())
        d = self._subtile_shape[0]
        if batches.shape[0] != n_used or tuple(batches.shape[1:]) != (d, d):
            raise ValueError(f"expected {n_used} sub-tile maps of {d}x{d}, got {batches.shape}")
        expanded = np.zeros((self._subtiles_to_use.size, d, d), dtype=batches.dtype)   # skipped sub-tiles stay zero
        expanded[self._subtiles_to_use] = batches
        self._outdata = unmake_blocks_vectorized(expanded, d, self._tile_shape[0], self._tile_shape[1])
        if self._target is not None:   # geo-registered copy of the valid part (tiler.py:166-170)
            self._target = self._target.load()
            self._target.loc[:] = self._outdata[0:self._tile_info.size[0], 0:self._tile_info.size[1]]

    @property
    def result(self) -> np.ndarray:
        """the merged class map cropped to the raster size (what ``write_file`` stores)"""
        return self._outdata[0:self._tile_info.size[0], 0:self._tile_info.size[1]]

    def write_file(self, outfile: Union[str, Path]) -> None:
        """reference tiler.py:136-142 (LZW-compressed tiled GeoTIFF through rioxarray)"""
        if self._target is None:
            _rioxarray()
            raise RuntimeError("Tiler.write_file: no geo-registered source (load_file was not used); read Tiler.result")
        self._target[:] = self._outdata[0:self._tile_info.size[0], 0:self._tile_info.size[1]]
        self._target.rio.to_raster(outfile, compress="LZW", tiled=True)


def infer_rasters(inference, rasters, subtile: int = 256, batch_size: int = 64, rank: int = 0, world: int = 1,
                  device: str = "cuda", tile_shape: Optional[Tuple[int, int]] = None, skip_blank: bool = True):
    """the directory loop of scripts/inference.py:71-115 over in-memory rasters (GeoTIFF I/O needs rioxarray, absent
    here): ``rasters`` yields ``array`` or ``(key, array)``; rank r of ``world`` takes rasters r, r + world, ... (tiles are
    independent: no collective).  Yields ``(key, class_map)`` in input order of the rank's share; rasters whose band 1
    holds only 0 / 255 (``is_valid_tile``, :60-62) are skipped like the reference does — ``(key, None)`` — without a
    forward pass (device reduction over the uploaded raster, ``ops.band_has_data``)."""
    for i, item in enumerate(rasters):
        if i % world != rank:
            continue
        key, arr = item if isinstance(item, tuple) else (i, item)
        yield key, infer_tile(inference, arr, subtile=subtile, batch_size=batch_size, device=device, tile_shape=tile_shape,
                              skip_blank=skip_blank)


def infer_tile(inference, arr_chw_u8: np.ndarray, subtile: int = 256, batch_size: int = 64, rank: int = 0,
               world: int = 1, device: str = "cuda", group=None, tile_shape: Optional[Tuple[int, int]] = None,
               on_device: Optional[bool] = None, skip_blank: bool = False) -> Optional[np.ndarray]:
    """whole-tile inference of scripts/inference.py:80-115 on the MI355X path: split -> (uint8 H2D, normalise on the
    device) -> forward + fused argmax -> uint8 D2H -> merge.  With world > 1 the sub-tile batches j = rank (mod world)
    are processed locally and the uint8 class maps are all-gathered (no other collective: tiles are independent).
    ``on_device`` (default: single rank + uint8 raster + HIP device) does the block split / merge on the GPU as well:
    one H2D copy of the raster, one D2H copy of the merged map (``_infer_tile_on_device``; same result, tested)."""
    if tile_shape is None:
        h, w = arr_chw_u8.shape[1], arr_chw_u8.shape[2]
        if h <= 2048 and w <= 2048 and 2048 % subtile == 0:
            tile_shape = (2048, 2048)           # the reference's tile (tiler.py:63)
        else:
            tile_shape = (-(-h // subtile) * subtile, -(-w // subtile) * subtile)
    if on_device is None:
        on_device = (world == 1 and str(device).startswith("cuda") and torch.cuda.is_available()
                     and arr_chw_u8.dtype == np.uint8)
    if on_device:
        if world != 1:
            raise ValueError("infer_tile: the on-device split / merge is the single-rank form")
        return _infer_tile_on_device(inference, arr_chw_u8, subtile, batch_size, device, tile_shape, skip_blank)
    if skip_blank and bool(np.isin(arr_chw_u8[0], [0, 255]).all()):    # scripts/inference.py:60-62 is_valid_tile
        return None
    t = Tiler(tile_shape=tile_shape, subtile_shape=(subtile, subtile))
    t.load_array(arr_chw_u8)
    used = t.get_batches()
    batches = np.array_split(used, math.ceil(len(used) / batch_size), axis=0)
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
    t.put_batches(np.concatenate(outs, axis=0))
    return t.result


def _infer_tile_on_device(inference, arr_chw_u8: np.ndarray, subtile: int, batch_size: int, device: str,
                          tile_shape: Tuple[int, int], skip_blank: bool = False) -> Optional[np.ndarray]:
    """single-rank form of ``infer_tile`` with the block split / merge on the device: ONE uint8 H2D copy of the raster, the
    sub-tiles of ``Tiler.get_batches`` (same ones, same row-major order: the [0:ceil(h/d), 0:ceil(w/d)] blocks of the
    zero-padded tile, reference tiler.py:121-134 + utils/data_handling.py:9-20) as a strided view -> NHWC uint8 batches
    -> ``run_u8`` -> class maps merged like ``unmake_blocks_vectorized`` -> ONE uint8 D2H copy of the cropped map."""
    C, h, w = arr_chw_u8.shape
    d = subtile
    if h > tile_shape[0] or w > tile_shape[1]:
        raise ValueError(f"raster {(h, w)} larger than the tile shape {tuple(tile_shape)}")
    if tile_shape[0] % d or tile_shape[1] % d:
        raise ValueError(f"Shapes unaligned: {tuple(tile_shape)} / {d}")
    nby, nbx = -(-h // d), -(-w // d)
    nch = int(getattr(inference, "in_channels", C) or C)
    if 0 < nch < C:                 # band planes the network never reads (N of an RGBN raster under an RGB model) stay on the host
        arr_chw_u8, C = arr_chw_u8[:nch], nch
    x = torch.from_numpy(np.ascontiguousarray(arr_chw_u8)).to(device, non_blocking=True)
    if skip_blank:
        from .. import ops
        if int(ops.band_has_data(x[0])) == 0:       # is_valid_tile on the uploaded raster: nothing but 0 / 255 in band 1
            return None
    if hasattr(inference, "run_blocks"):
        # round 3: split + zero padding + Normalize + NHWC in one gather per batch (dt_split_normalize_u8), no ATen passes
        outs = [inference.run_blocks(x, d, j, min(batch_size, nby * nbx - j)) for j in range(0, nby * nbx, batch_size)]
    else:
        if (nby * d, nbx * d) != (h, w):
            xp = torch.zeros((C, nby * d, nbx * d), dtype=torch.uint8, device=x.device)
            xp[:, :h, :w] = x
            x = xp
        blocks = x.view(C, nby, d, nbx, d).permute(1, 3, 2, 4, 0).contiguous().view(nby * nbx, d, d, C)
        outs = [inference.run_u8(blocks[j:j + batch_size], device=device) for j in range(0, nby * nbx, batch_size)]
    maps = (outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)).to(torch.uint8)
    merged = maps.view(nby, nbx, d, d).permute(0, 2, 1, 3).reshape(nby * d, nbx * d)
    return merged[:h, :w].contiguous().cpu().numpy()
