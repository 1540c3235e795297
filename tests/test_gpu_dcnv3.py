"""GPU parity of the DCNv3 operator against the reference's own pure-PyTorch core (golden vectors generated from
models/ops_dcnv3/.../dcnv3_func.py:148-189 at test.py's shapes, seed-free numpy data) with the tolerances the
reference's test script uses (models/ops_dcnv3/test.py:85,134: rtol=1e-2, atol=1e-3) — we hold f32 to 100x tighter."""
import pytest
import torch

from tests.util import Golden, names

pytestmark = pytest.mark.gpu
BF16_GRAD_TOL = 0.2        # relative L2 of a bf16 gradient against the f32 fixture (measured: see the assertion messages)


@pytest.mark.parametrize("name", names("dcnv3_"))
def test_dcnv3_forward_backward(name):
    from yolo_dual_amd.dcnv3 import dcnv3_core
    g = Golden(name)
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).cuda().requires_grad_(True) for k in ("inp", "off", "msk"))
    out = dcnv3_core(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(g.flat["offset_scale"]))
    ref = g.t("out")
    assert out.shape == ref.shape
    # absolute floors relative to the fixture's magnitude (the tiled-backward fixture has O(1) inputs, the test.py ones O(0.01))
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=max(1e-5, 1e-5 * float(ref.abs().max()))), float((out.cpu() - ref).abs().max())
    gup = g.t("gup").cuda() if g.has("gup") else torch.ones_like(out)
    (out * gup).sum().backward()
    for t, k in ((inp, "ginp"), (off, "goff"), (msk, "gmsk")):
        r = g.t(k)
        assert torch.allclose(t.grad.cpu(), r, rtol=1e-3, atol=max(1e-5, 2e-5 * float(r.abs().max()))), (k, float((t.grad.cpu() - r).abs().max()))


def test_dcnv3_bf16_runs_close():
    from yolo_dual_amd.dcnv3 import dcnv3_core
    g = Golden("dcnv3_D32")
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).cuda().bfloat16() for k in ("inp", "off", "msk"))
    out = dcnv3_core(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(g.flat["offset_scale"]))
    ref = g.t("out")
    # reference tolerance for reduced precision: rtol=1e-2, atol=1e-3 (test.py:85)
    assert torch.allclose(out.float().cpu(), ref, rtol=5e-2, atol=1e-3)


def test_dcnv3_fp16_matches_f32():
    """the op also takes fp16 storage like the reference (AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69,147): C ABI call
    with YDL_F16 against the f32 golden, reference tolerance for reduced precision (test.py:85: rtol 1e-2, atol 1e-3)"""
    import ctypes
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd.tape import _p, _stream
    g = Golden("dcnv3_D32")
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).cuda().half().contiguous() for k in ("inp", "off", "msk"))
    N, H, W, C = inp.shape
    out = torch.empty_like(inp)
    L.call("ydl_dcnv3_fwd", L.YDL_F16, _p(inp), _p(off), _p(msk), _p(out), kh, kw, sh, sw, ph, pw, dh, dw, G, D,
           ctypes.c_float(float(g.flat["offset_scale"])), N, H, W, H, W, _stream())
    torch.cuda.synchronize()
    assert torch.allclose(out.float().cpu(), g.t("out"), rtol=1e-2, atol=1e-3)
    gin = torch.zeros(inp.shape, dtype=torch.float32, device="cuda")
    goff = torch.empty(off.shape, dtype=torch.float32, device="cuda")
    gmsk = torch.empty(msk.shape, dtype=torch.float32, device="cuda")
    go = torch.ones_like(inp)
    L.call("ydl_dcnv3_bwd", L.YDL_F16, _p(inp), _p(off), _p(msk), _p(go), _p(gin), _p(goff), _p(gmsk), kh, kw, sh, sw, ph, pw,
           dh, dw, G, D, ctypes.c_float(float(g.flat["offset_scale"])), N, H, W, H, W, _stream())
    torch.cuda.synchronize()
    assert torch.allclose(gin.cpu(), g.t("ginp"), rtol=2e-2, atol=1e-3)
    assert torch.allclose(gmsk.cpu(), g.t("gmsk"), rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", names("dcnmod_"))
def test_dcnv3_module(name, mode):
    """the DCNv3 module (modules/dcnv3.py:50-136: input_proj / depth-wise Conv+BN+SiLU / offset + mask Linear / softmax over the
    K*K points / sampling op / output_proj) against the fixture the reference's own class produced"""
    import yolo_dual_amd as ydl
    from tests.test_gpu_blocks import _load, _run
    ydl.set_compute_dtype(mode)
    try:
        g = Golden(name)
        C, k, s, pad, G = [int(v) for v in g.flat["meta"]]
        m = _load(ydl.DCNv3(channels=C, kernel_size=k, stride=s, pad=pad, group=G), g)
        if mode == "bf16":
            # sampling positions come from bf16-rounded offsets: outputs are compared in relative L2
            from tests.util import l2_err
            x = g.t("x0").cuda().requires_grad_(True)
            out = m(x)
            assert l2_err(out.detach().cpu(), g.t("out")) < 6e-2
            (out * g.t("gup").cuda()).sum().backward()
            assert l2_err(x.grad.cpu(), g.t("gx0")) < 0.2
            grads = g.group("grad") if any(k_.startswith("grad/") for k_ in g.flat) else {}
            gscale = max([float(v.abs().max()) for v in grads.values()] + [0.0])
            named = dict(m.named_parameters())
            errs = {kk: l2_err(named[kk].grad.detach().float().cpu(), v) for kk, v in grads.items()
                    if float(v.abs().max()) >= 1e-4 * gscale}
            bad = {k_: round(e, 4) for k_, e in errs.items() if not e < BF16_GRAD_TOL}
            assert not bad, (bad, {k_: round(e, 4) for k_, e in errs.items()})
            return
        _run(m, g, mode)
    finally:
        ydl.set_compute_dtype("bf16")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", names("c3_dcnv3_"))
def test_c3_dcnv3(name, mode):
    """C3_DCNV3 -> Bottleneck_DCNV3 -> DCNV3_YoLo -> DCNv3 ("common and yolo.py":2-38), fixture from the reference's classes"""
    import yolo_dual_amd as ydl
    from tests.util import l2_err, rel_err
    ydl.set_compute_dtype(mode)
    try:
        g = Golden(name)
        c1, c2, n = [int(v) for v in g.flat["meta"]]
        m = ydl.C3_DCNV3(c1, c2, n, "noshortcut" not in name)
        m.load_state_dict(g.group("sd"))
        m = m.cuda().train()
        x = g.t("x0").cuda().requires_grad_(True)
        out = m(x)
        ref = g.t("out")
        if mode == "bf16":
            # throughput mode (what bench.py --workload cfg5dcn runs): the backward of the whole block — DCNv3 sampling gradients,
            # depth-wise conv weight gradient, offset / mask Linear gradients, group-softmax backward — against the f32 fixture in
            # relative L2 (sampling positions come from bf16-rounded offsets; a wrong-but-finite backward gives O(1))
            assert l2_err(out.detach().cpu(), ref) < 8e-2
            (out * g.t("gup").cuda()).sum().backward()
            errs = {"x": l2_err(x.grad.cpu(), g.t("gx0"))}
            grads = g.group("grad")
            gscale = max(float(v.abs().max()) for v in grads.values())
            named = dict(m.named_parameters())
            for kk, v in grads.items():
                if float(v.abs().max()) < 1e-4 * gscale:      # mathematically zero in f32: only rounding noise to compare
                    continue
                errs[kk] = l2_err(named[kk].grad.detach().float().cpu(), v)
            # What bf16 can reproduce here (tools/dcn_bf16_debug.py): the sampling positions are bf16-rounded offsets (spacing 2^-6 px
            # at |offset| in [2, 4)), and d(sample)/d(position) jumps where a position crosses an integer; the few percent of samples
            # whose rounded position lands on the other side of a grid line get an O(1) different offset gradient.  One DCNv3 deep
            # (module fixture, n = 1 with the shortcut) that is 0.05-0.18 relative L2 on what sits behind the offset branch; every
            # further DCNv3 in series WITHOUT a shortcut around it compounds it (the fixture's random offset weights make that branch
            # strong; the module itself starts them at zero): n = 2 / no shortcut measures 0.2-0.3 behind the last DCNv3 and
            # 0.5-0.9 behind both, the same at 64 channels x 40x40 x batch 4 (bf16 vs f32 of this path), i.e. not a small-fixture
            # artefact.  Bounds: BF16_GRAD_TOL behind at most one DCNv3; further upstream a gradient must still POINT the right way
            # (cosine > 0.5 with a norm within 2x) — a wrong-but-finite backward gives a cosine around zero.
            deep = "noshortcut" in name and n > 1
            tight = {k_: e for k_, e in errs.items() if not deep or k_.startswith(("cv2.", "cv3.")) or (k_.startswith("m.1.") and ".offset" not in k_
                                                                                                        and ".dw_conv" not in k_ and ".cv1" not in k_
                                                                                                        and ".conv." not in k_)}
            bad = {k_: round(e, 4) for k_, e in tight.items() if not e < BF16_GRAD_TOL}
            assert not bad, (bad, {k_: round(e, 4) for k_, e in errs.items()})
            for k_ in errs:
                if k_ in tight:
                    continue
                a = (x.grad.cpu() if k_ == "x" else named[k_].grad.detach().float().cpu()).double().flatten()
                b = (g.t("gx0") if k_ == "x" else grads[k_]).double().flatten()
                cos = float((a @ b) / (a.norm() * b.norm()))
                ratio = float(a.norm() / b.norm())
                assert cos > 0.5 and 0.5 < ratio < 2.0, (k_, cos, ratio)
            return
        assert rel_err(out.detach().cpu(), ref) < 1e-4, rel_err(out.detach().cpu(), ref)
        (out * g.t("gup").cuda()).sum().backward()
        assert rel_err(x.grad.cpu(), g.t("gx0")) < 5e-4
        grads = g.group("grad")
        gscale = max(float(v.abs().max()) for v in grads.values())
        named = dict(m.named_parameters())
        for kk, v in grads.items():
            got = named[kk].grad.detach().cpu()
            if float(v.abs().max()) < 1e-4 * gscale:          # mathematically zero gradients (constant in front of conv + BN)
                assert float(got.abs().max()) < 1e-3 * gscale, kk
                continue
            assert rel_err(got, v) < 2e-3, (kk, rel_err(got, v))
        after = m.state_dict()
        for kk, v in g.group("sd_after").items():
            if v.dtype.is_floating_point:
                assert rel_err(after[kk].cpu(), v) < 1e-4, kk
    finally:
        ydl.set_compute_dtype("bf16")

