"""GPU parity of the DCNv3 operator against the reference's own pure-PyTorch core (golden vectors generated from
models/ops_dcnv3/.../dcnv3_func.py:148-189 at test.py's shapes, seed-free numpy data) with the tolerances the
reference's test script uses (models/ops_dcnv3/test.py:85,134: rtol=1e-2, atol=1e-3) — we hold f32 to 100x tighter."""
import pytest
import torch

from tests.util import Golden, names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", names("dcnv3_"))
def test_dcnv3_forward_backward(name):
    from yolo_dual_amd.dcnv3 import dcnv3_core
    g = Golden(name)
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).cuda().requires_grad_(True) for k in ("inp", "off", "msk"))
    out = dcnv3_core(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(g.flat["offset_scale"]))
    ref = g.t("out")
    assert out.shape == ref.shape
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-5), float((out.cpu() - ref).abs().max())
    gup = g.t("gup").cuda() if g.has("gup") else torch.ones_like(out)
    (out * gup).sum().backward()
    for t, k in ((inp, "ginp"), (off, "goff"), (msk, "gmsk")):
        r = g.t(k)
        assert torch.allclose(t.grad.cpu(), r, rtol=1e-3, atol=1e-5), (k, float((t.grad.cpu() - r).abs().max()))


def test_dcnv3_bf16_runs_close():
    from yolo_dual_amd.dcnv3 import dcnv3_core
    g = Golden("dcnv3_D32")
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).cuda().bfloat16() for k in ("inp", "off", "msk"))
    out = dcnv3_core(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(g.flat["offset_scale"]))
    ref = g.t("out")
    # reference tolerance for reduced precision: rtol=1e-2, atol=1e-3 (test.py:85)
    assert torch.allclose(out.float().cpu(), ref, rtol=5e-2, atol=1e-3)
