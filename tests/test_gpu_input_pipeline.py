"""GPU parity of the input-preparation kernels (csrc/input.hip through yolo_dual_amd.data.LetterboxGPU) — bit-exact against
the fixtures produced by the reference's `_resize_and_pad` with Pillow, and against the CPU oracle at the benchmark's 640²."""
import numpy as np
import pytest
import torch

from oracle import pil_ops as P
from tests.util import Golden, names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", names("letterbox_"))
def test_letterbox_matches_reference_fixture(name):
    from yolo_dual_amd.data import LetterboxGPU
    g = Golden(name)
    w, h, S, nc = [int(v) for v in g.flat["meta"]]
    lb = LetterboxGPU(S, num_classes=nc)
    img, mask = lb(g.flat["img"], g.flat["mask"])
    assert img.dtype == torch.float32 and mask.dtype == torch.int64
    assert torch.equal(img.cpu(), g.t("out_img"))
    assert torch.equal(mask.cpu(), g.t("out_mask"))


@pytest.mark.parametrize("w,h", [(960, 720), (720, 960), (1920, 1080), (640, 640), (641, 479), (320, 240), (3000, 200)])
def test_letterbox_full_size_against_oracle(w, h):
    from yolo_dual_amd.data import LetterboxGPU
    rs = np.random.RandomState(w * 3 + h)
    img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    mask = rs.randint(0, 14, size=(h, w)).astype(np.uint8)
    lb = LetterboxGPU(640, num_classes=12)
    gi, gm = lb(img, mask)
    oi, om = P.resize_and_pad(img, np.clip(mask, 0, 11).astype(np.uint8), 640)
    assert torch.equal(gi.cpu(), torch.from_numpy(oi))
    assert torch.equal(gm.cpu(), torch.from_numpy(om))
    # size-independent properties: the canvas outside the pasted box is grey / background, values are k/255
    nw, nh, pl, pt = P.letterbox_geometry(w, h, 640)
    outside = torch.ones(640, 640, dtype=torch.bool)
    outside[pt:pt + nh, pl:pl + nw] = False
    assert bool((gi.cpu()[:, outside] == np.float32(128) / np.float32(255)).all()) and bool((gm.cpu()[outside] == 0).all())
    gc = gi.cpu()
    assert torch.equal((gc * 255).round() / 255, gc)


def test_batch_collation_and_errors():
    from yolo_dual_amd.data import LetterboxGPU
    rs = np.random.RandomState(5)
    imgs = [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for (w, h) in [(200, 100), (64, 64), (90, 300)]]
    masks = [rs.randint(0, 12, size=i.shape[:2]).astype(np.uint8) for i in imgs]
    lb = LetterboxGPU(128)
    bi, bm = lb.batch(imgs, masks)
    assert tuple(bi.shape) == (3, 3, 128, 128) and tuple(bm.shape) == (3, 128, 128)
    for i in range(3):
        oi, om = P.resize_and_pad(imgs[i], masks[i], 128)
        assert torch.equal(bi[i].cpu(), torch.from_numpy(oi)) and torch.equal(bm[i].cpu(), torch.from_numpy(om))
    with pytest.raises(TypeError):
        lb(imgs[0].astype(np.float32))
    with pytest.raises(ValueError):
        lb(imgs[0], masks[1])
    # the letterboxed batch feeds the model like any other (N,3,S,S) float tensor
    import yolo_dual_amd as ydl
    m = ydl.ResNet18Seg({"nc": 12}).cuda().train()
    out = m(bi)
    assert out.shape[0] == 3 and torch.isfinite(out).all()
