"""Host logic of the full-domain tiler: tile tables cover the domain with at least the requested
overlap, x origins are quad-aligned, and the NumPy specification of the blend is a partition of unity."""
import numpy as np
import pytest

from oracle import tiler_ref as OT
from sbgm_danra_amd.tiling import axis_origins


@pytest.mark.parametrize("L,tile,ov,align", [(589, 256, 64, 1), (792, 256, 64, 4), (256, 256, 64, 1), (300, 256, 64, 4),
                                             (1024, 256, 64, 4), (2048, 256, 0, 4), (257, 256, 64, 1)])
def test_axis_origins_cover_with_overlap(L, tile, ov, align):
    o = axis_origins(L, tile, ov, align)
    assert o[0] == 0 and o[-1] == L - tile and all(b > a for a, b in zip(o, o[1:]))
    assert all(v % align == 0 for v in o)
    assert all(a + tile - b >= min(ov, tile - (b - a)) and a + tile >= b for a, b in zip(o, o[1:]))   # no gaps
    if len(o) > 2:
        assert all(a + tile - b >= ov - align for a, b in zip(o, o[1:]))


def test_axis_origins_errors():
    with pytest.raises(ValueError):
        axis_origins(100, 256, 64)
    with pytest.raises(ValueError):
        axis_origins(789, 256, 64, align=4)          # 533 is not a multiple of 4: the tiler pads the domain instead
    with pytest.raises(ValueError):
        axis_origins(589, 256, 256)


def test_blend_is_a_partition_of_unity():
    Hd, Wd, tile, R = 589, 792, 256, 64
    org = [(y, x) for y in axis_origins(Hd, tile, R) for x in axis_origins(Wd, tile, R, 4)]
    assert len(org) == 12
    rng = np.random.default_rng(0)
    dom = rng.standard_normal((2, Hd, Wd)).astype(np.float32)
    back = OT.stitch(OT.extract(dom, org, tile), org, Hd, Wd, R)
    assert np.abs(back - dom).max() <= 1e-6 * np.abs(dom).max()
    # tiles that disagree are blended monotonically across the overlap
    tiles = np.stack([np.full((1, tile, tile), float(i), np.float32) for i in range(len(org))])
    out = OT.stitch(tiles, org, Hd, Wd, R)[0]
    row = out[10, :]                                   # first tile row: tiles 0..3 left to right
    assert row[0] == 0.0 and row[-1] == 3.0 and np.all(np.diff(row) >= -1e-6)
