"""``Tiler`` against the reference's contract (deadtrees/deployment/tiler.py:22-170), written after the reference's
own tests/test_tiler.py: same assertions, on synthetic rasters of the reference's CURRENT geometry (2048x2048 tiles,
256x256 sub-tiles, 4 bands — tiler.py:62-64) because the reference's GeoTIFF fixtures are DVC pointers and rioxarray
is not installed here.  The block known-answer vectors of tests/test_tiler.py:56-77 are checked in
tests/test_host_logic.py::test_blocks_match_reference_known_answer against tests/golden/blocks.npz."""
from math import prod

import numpy as np
import pytest

from deadtrees.deployment.tiler import Tiler, divisible_without_remainder, inspect_tile

# (raster size, sub-tiles holding data) — the three situations of the reference's examples: full tile,
# ragged right edge, ragged bottom edge
EXAMPLES = [((2048, 2048), (8, 8)), ((2048, 1700), (8, 7)), ((662, 2048), (3, 8))]


def _raster(size, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (4, *size), dtype=np.uint8)


@pytest.fixture
def tiler():
    return Tiler()


@pytest.mark.parametrize("a,b,result", [(10, 2, True), (5, 4, False), (2, 0, False)])
def test_divisible_without_remainder(a, b, result):
    assert divisible_without_remainder(a, b) == result


@pytest.mark.parametrize("size,subtiles", EXAMPLES)
def test_tiler_inspect_tile(size, subtiles):
    info = inspect_tile(_raster(size), tile_shape=(2048, 2048), subtile_shape=(256, 256))
    assert info.size == size and info.subtiles == subtiles


def test_tiler_inspect_tile_subtile_not_divisible():
    with pytest.raises(ValueError):
        inspect_tile(_raster((2048, 2048)), tile_shape=(2048, 2048), subtile_shape=(512, 211))


def test_tiler_catch_bad_subtile_dims():
    with pytest.raises(ValueError):
        Tiler(None, tile_shape=(2048, 2048), subtile_shape=(256, 250))


@pytest.mark.parametrize("size,subtiles", EXAMPLES)
def test_tiler_subtiles_to_use_and_get_batches(tiler, size, subtiles):
    src = _raster(size)
    tiler.load_array(src)
    assert sum(tiler._subtiles_to_use) == prod(subtiles)
    batches = tiler.get_batches()
    assert batches.shape == (prod(subtiles), 4, 256, 256) and batches.dtype == np.uint8
    # row-major over the used sub-tiles; padding is zero
    np.testing.assert_array_equal(batches[0], src[:, :256, :256])
    j = subtiles[1] - 1                       # last used sub-tile of the first row: ragged on the right
    w = size[1] - 256 * j
    np.testing.assert_array_equal(batches[j][:, :, :w], src[:, :256, 256 * j:])
    assert not batches[j][:, :, w:].any()


@pytest.mark.parametrize("size,subtiles", EXAMPLES)
def test_tiler_put_batches(tiler, size, subtiles):
    tiler.load_array(_raster(size))
    batches = tiler.get_batches()
    pred = np.random.default_rng(1).choice(a=[1, 0], size=(len(batches), 256, 256), p=[0.1, 0.9])   # single layer
    assert tiler.put_batches(pred) is None
    assert tiler._outdata.shape == (2048, 2048)
    assert tiler.result.shape == size
    # the merged map restricted to sub-tile (r, c) is the prediction of that sub-tile; skipped sub-tiles are zero
    k = 0
    for r in range(8):
        for c in range(8):
            blk = tiler._outdata[256 * r:256 * (r + 1), 256 * c:256 * (c + 1)]
            if r < subtiles[0] and c < subtiles[1]:
                np.testing.assert_array_equal(blk, pred[k])
                k += 1
            else:
                assert not blk.any()
    with pytest.raises(ValueError):
        tiler.put_batches(pred[:-1])


def test_tiler_roundtrip_is_identity(tiler):
    src = _raster((1500, 1111), seed=3)
    tiler.load_array(src)
    tiler.put_batches(tiler.get_batches()[:, 1])
    np.testing.assert_array_equal(tiler.result, src[1])


def test_tiler_other_geometry_and_file_io_without_rioxarray(tmp_path):
    t = Tiler(tile_shape=(1024, 512), subtile_shape=(128, 128))
    t.load_array(_raster((1000, 300)))
    assert t.get_batches().shape == (8 * 3, 4, 128, 128)
    try:
        import rioxarray  # noqa: F401
    except Exception:
        with pytest.raises(ImportError, match="rioxarray"):
            t.load_file(tmp_path / "ortho.tif")
        with pytest.raises(ImportError, match="rioxarray"):
            t.write_file(tmp_path / "out.tif")
    with pytest.raises(ValueError):
        Tiler(tile_shape=(512, 512)).load_array(_raster((600, 100)))     # raster larger than the tile
