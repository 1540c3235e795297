"""CPU side of the input-preparation row: the oracle's restatement of Pillow's 8-bit resampling against the
fixtures the reference's own `_resize_and_pad` produced (tests/golden/letterbox_*.npz, oracle/make_golden.py --letterbox) and,
where Pillow is importable, against Pillow itself on more sizes; the product's host-side coefficient tables (yolo_dual_amd/
data.py, an independent vectorised implementation) against the oracle's."""
import numpy as np
import pytest

from oracle import pil_ops as P
from tests.util import Golden, names

SIZES = [(960, 720, 640), (720, 960, 640), (1920, 1080, 640), (641, 479, 640), (50, 50, 256), (3000, 200, 640), (7, 300, 64)]


@pytest.mark.parametrize("name", names("letterbox_"))
def test_oracle_matches_reference_fixture(name):
    g = Golden(name)
    w, h, S, nc = [int(v) for v in g.flat["meta"]]
    img, mask = g.flat["img"], np.clip(g.flat["mask"], 0, nc - 1).astype(np.uint8)
    oi, om = P.resize_and_pad(img, mask, S)
    assert np.array_equal(oi, g.flat["out_img"])            # bit-exact float32
    assert np.array_equal(om, g.flat["out_mask"])


@pytest.mark.parametrize("w,h,S", SIZES)
def test_oracle_matches_pillow(w, h, S):
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(w + 7 * h)
    img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    mask = rs.randint(0, 12, size=(h, w)).astype(np.uint8)
    nw, nh, _pl, _pt = P.letterbox_geometry(w, h, S)
    assert np.array_equal(P.resize_bilinear_u8(img, nw, nh), np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)))
    assert np.array_equal(P.resize_nearest_u8(mask, nw, nh), np.array(Image.fromarray(mask).resize((nw, nh), Image.NEAREST)))


@pytest.mark.parametrize("n_in,n_out", [(960, 640), (37, 23), (1920, 640), (333, 164), (641, 640), (50, 256), (3000, 640), (7, 640),
                                        (640, 7), (479, 478)])
def test_product_tables_equal_the_oracles(n_in, n_out):
    from yolo_dual_amd.data import _bilinear_tables, _nearest_table
    b1, k1, ks1 = P.bilinear_coeffs(n_in, n_out)
    b2, k2, ks2 = _bilinear_tables(n_in, n_out)
    assert ks1 == ks2 and np.array_equal(b1, b2) and np.array_equal(k1, k2)
    assert int(k2.sum(1).min()) >= (1 << 22) - 8 and int(k2.sum(1).max()) <= (1 << 22) + 8      # normalised fixed point
    assert np.array_equal(P.nearest_table(n_in, n_out), _nearest_table(n_in, n_out))


def test_geometry_follows_the_reference():
    from yolo_dual_amd.data import letterbox_geometry
    for (w, h, S) in SIZES + [(640, 640, 640), (100, 37, 64)]:
        assert letterbox_geometry(w, h, S) == P.letterbox_geometry(w, h, S)
    assert letterbox_geometry(960, 720, 640) == (640, 480, 0, 80)
    with pytest.raises(ValueError):
        letterbox_geometry(10000, 1, 64)
