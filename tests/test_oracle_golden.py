"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the reference's own classes
(oracle/make_golden.py).  CPU only.  Tolerances: same ATen op sequence => essentially exact; we allow 1e-6
relative for re-associated sums."""
import numpy as np
import pytest
import torch
import yaml

from oracle import ref_cpu as R
from tests.util import Golden, names, rel_err

TOL = 2e-6


def _sd(g, requires_grad=True):
    sd = g.group("sd") if not g.has("sd_keys") else g.rebuild_sd()
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k and requires_grad:
            v.requires_grad_(True)
    return sd


def _check(g, sd, out, xs, prefix_strip=""):
    assert out.shape == g.t("out").shape
    assert rel_err(out, g.t("out")) < TOL, g.name
    (out * g.t("gup")).sum().backward()
    for i, x in enumerate(xs):
        if g.has(f"gx{i}"):
            assert rel_err(x.grad, g.t(f"gx{i}")) < 5e-6, (g.name, "gx", i)
    if g.has("grad_names"):
        for k, n in zip(g.strs("grad_names"), g.flat["grad_norms"]):
            assert abs(float(sd[k].grad.double().norm()) - n) <= 1e-5 * max(n, 1e-6), (g.name, k)
    else:
        for k, v in g.group("grad").items():
            assert rel_err(sd[k].grad, v) < 1e-5, (g.name, k)
    for k, v in g.group("sd_after").items():
        if v.dtype.is_floating_point:
            assert rel_err(sd[k].detach(), v) < TOL, (g.name, k)
        else:
            assert torch.equal(sd[k], v), (g.name, k)


def _x(g, i=0):
    return g.t(f"x{i}").clone().requires_grad_(True)


def _wrap(sd, pre):
    """fixtures of a bare module have keys without a module prefix; functional oracle wants 'pre.key'."""
    return {f"{pre}.{k}": v for k, v in sd.items()}, (lambda d: {k[len(pre) + 1:]: v for k, v in d.items()})


@pytest.mark.parametrize("name", names("v5_conv_"))
def test_conv(name):
    g = Golden(name)
    c1, c2, k, s, p, act = [int(v) for v in g.flat["meta"]]
    sd = _sd(g)
    wsd, unwrap = _wrap(sd, "m")
    x = _x(g)
    out = R.conv_bn_act(wsd, "m", x, s=s, p=None if p < 0 else p, act="silu" if act else "none")
    _check(g, sd, out, [x])


@pytest.mark.parametrize("name", names("v5_c3_"))
def test_c3_script(name):
    g = Golden(name)
    c1, c2, n = [int(v) for v in g.flat["meta"]]
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    add = (c1 == c2) and "noshortcut" not in name
    _check(g, sd, R.c3_script(wsd, "m", x, n, add), [x])


@pytest.mark.parametrize("name", names("v5_sppf") + ["cm_sppf"])
def test_sppf(name):
    g = Golden(name)
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    _check(g, sd, R.sppf(wsd, "m", x, 5), [x])


@pytest.mark.parametrize("name", names("v5_concat_"))
def test_concat(name):
    g = Golden(name)
    xs = [_x(g, 0), _x(g, 1)]
    out = R.concat_align(xs)
    assert torch.equal(out, g.t("out"))          # index arithmetic: bit-exact
    (out * g.t("gup")).sum().backward()
    for i, x in enumerate(xs):
        assert rel_err(x.grad, g.t(f"gx{i}")) < TOL


@pytest.mark.parametrize("name", names("v5_upsample_"))
def test_upsample(name):
    g = Golden(name)
    x = _x(g)
    out = R.upsample_nearest(x, int(name[-1]))
    assert torch.equal(out, g.t("out"))
    (out * g.t("gup")).sum().backward()
    assert rel_err(x.grad, g.t("gx0")) < TOL


@pytest.mark.parametrize("name", ["cm_bottleneck", "cm_c3_n1", "cm_c3_n2"])
def test_common_blocks(name):
    g = Golden(name)
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    if name == "cm_bottleneck":
        out = R.bottleneck(wsd, "m", x, True)
    else:
        out = R.c3_common(wsd, "m", x, int(g.flat["meta"][2]), True)
    _check(g, sd, out, [x])


@pytest.mark.parametrize("name", names("v8_c2f_"))
def test_c2f(name):
    g = Golden(name)
    c1, c2, n = [int(v) for v in g.flat["meta"]]
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    _check(g, sd, R.c2f(wsd, "m", x, n, c1 == c2), [x])


def test_c3k2_gam():
    g = Golden("v9_c3k2")
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    _check(g, sd, R.c3k2(wsd, "m", x, 1, True), [x])
    g = Golden("v9_gam")
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    _check(g, sd, R.gam(wsd, "m", x), [x])


@pytest.mark.parametrize("name", ["r18_basic", "r18_basic_down", "r50_bneck", "r50_bneck_down"])
def test_resnet_blocks(name):
    g = Golden(name)
    stride = int(g.flat["meta"][2])
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    fn = R.basic_block if name.startswith("r18") else R.bottleneck_block
    _check(g, sd, fn(wsd, "m", x, stride), [x])


def test_resnet_stem():
    g = Golden("r18_stem")
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    y = R.conv_bn_act(wsd, "m.0", x, s=2, p=3)
    out = torch.nn.functional.max_pool2d(y, 3, 2, 1)
    _check(g, sd, out, [x])


def test_segment_head():
    g = Golden("seghead")
    sd = _sd(g)
    wsd, _ = _wrap(sd, "head")
    xs = [_x(g, i) for i in range(3)]
    _check(g, sd, R.segment_head(wsd, "head", xs), xs)


@pytest.mark.parametrize("name", names("loss_"))
def test_losses(name):
    g = Golden(name)
    logits = g.t("logits").clone().requires_grad_(True)
    pred = logits.softmax(1) if int(g.flat["softmax_in"]) else logits
    cw = g.t("cw")
    cw = cw if cw.numel() else None
    kind = "jaccard" if "jaccard" in name else "dice"
    total, ce, ov = R.seg_loss(pred, g.t("target"), cw, kind, float(g.flat["ls"]))
    items = g.flat["items"]
    # north_star tolerance: loss within 1e-4 relative; the oracle itself is held to 1e-6
    assert abs(float(total) - items[0]) <= 1e-6 * abs(items[0])
    assert abs(float(ce) - items[1]) <= 1e-6 * abs(items[1])
    assert abs(float(ov) - items[2]) <= 1e-6 * abs(items[2])
    total.backward()
    assert rel_err(logits.grad, g.t("glogits")) < 1e-5


@pytest.mark.parametrize("name", names("dcnv3_"))
def test_dcnv3(name):
    g = Golden(name)
    kh, kw, sh, sw, ph, pw, dh, dw, G, D = [int(v) for v in g.flat["meta"]]
    inp, off, msk = (g.t(k).clone().requires_grad_(True) for k in ("inp", "off", "msk"))
    out = R.dcnv3_core(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(g.flat["offset_scale"]))
    ref = g.t("out")
    # models/ops_dcnv3/test.py:85 float tolerance: rtol=1e-2, atol=1e-3; the restatement does much better
    # (absolute floor relative to the fixture's magnitude: the tile fixture has O(1) inputs, the test.py ones O(0.01))
    assert torch.allclose(out, ref, rtol=1e-4, atol=max(1e-6, 1e-5 * float(ref.abs().max()))), float((out - ref).abs().max())
    (out * (g.t("gup") if g.has("gup") else torch.ones_like(out))).sum().backward()
    for t, k in ((inp, "ginp"), (off, "goff"), (msk, "gmsk")):
        r = g.t(k)
        assert torch.allclose(t.grad, r, rtol=1e-3, atol=max(1e-5, 1e-5 * float(r.abs().max()))), (k, float((t.grad - r).abs().max()))


def test_miou():
    g = Golden("miou")
    cm = R.confusion_matrix(g.t("pred"), g.t("target"), 12, 11)
    assert torch.equal(cm, g.t("matrix"))
    miou, ious = R.miou_from_confusion(cm, 11)
    assert abs(miou - float(g.flat["miou"])) < 1e-12
    assert np.allclose(ious, g.flat["ious"], atol=1e-12)


def test_sgd_ema():
    g = Golden("optim_sgd_ema")
    lr, mom, wd = [float(v) for v in g.flat["hyp"]]
    sd = {k: v.clone() for k, v in g.group("sd").items()}
    ema = {k: v.clone() for k, v in sd.items()}
    x = g.t("x0")
    bufs = {}
    pnames = [k for k in sd if k.endswith("conv.weight") or k.endswith("bn.weight") or k.endswith("bn.bias")]
    for st in range(int(g.flat["steps"])):
        ps = {k: sd[k].clone().requires_grad_(True) for k in pnames}
        run = dict(sd)
        run.update(ps)
        y = R.conv_bn_act(run, "1", R.conv_bn_act(run, "0", x))
        y.square().mean().backward()
        if st == 0:
            for k, v in g.group("g0").items():
                assert rel_err(ps[k].grad, v) < 1e-5
        for k in pnames:
            decay = wd if k.endswith("conv.weight") else 0.0      # smart_optimizer groups, torch_utils.py:318-346
            p = sd[k]
            bufs[k] = R.sgd_nesterov_step(p, ps[k].grad, bufs.get(k), lr, mom, decay)
        for k in ("0.bn.running_mean", "0.bn.running_var", "1.bn.running_mean", "1.bn.running_var",
                  "0.bn.num_batches_tracked", "1.bn.num_batches_tracked"):
            sd[k] = run[k]
        d = R.ema_decay(st + 1)
        for k, v in ema.items():
            if v.dtype.is_floating_point:
                R.ema_update(v, sd[k], d)
    for k, v in g.group("sd_after").items():
        if v.dtype.is_floating_point:
            assert rel_err(sd[k], v) < 1e-5, k
    for k, v in g.group("ema_after").items():
        if v.dtype.is_floating_point:
            assert rel_err(ema[k], v) < 1e-5, k


# ------------------------------------------------------------------ whole models
def _model_check(g, fwd, make_sd, loss_kw, steps=2, lr=0.01):
    from oracle.fill import fill_state_dict
    sd = make_sd()
    fill_state_dict(sd, 1234, bn_stats=False)
    pnames = g.strs("param_names")
    x, tgt = g.t("x"), g.t("target")
    bufs = {}
    for st in range(steps):
        ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
        run = dict(sd)
        run.update(ps)
        out = fwd(run, x)
        total, ce, ov = R.seg_loss(out, tgt, **loss_kw)
        total.backward()
        if st == 0:
            assert list(out.shape) == [int(v) for v in g.flat["out_shape"]]
            vals = out.detach().flatten()[g.t("out_idx")]
            assert rel_err(vals, g.t("out_vals")) < 1e-5          # logits within 1e-4 rel (north_star); oracle 1e-5
            none = sorted(k for k in pnames if ps[k].grad is None)
            assert none == sorted(g.strs("grad_none"))
            for k, n in zip(g.strs("grad_names"), g.flat["grad_norms"]):
                assert abs(float(ps[k].grad.double().norm()) - n) <= 2e-4 * max(n, 1e-8), (k, n)
        items = g.flat[f"loss_items_{st}"]
        for a, b in zip((total, ce, ov), items):
            assert abs(float(a) - b) <= 1e-5 * abs(b), (st, float(a), b)
        for k in sd:
            if k not in ps:
                sd[k] = run[k]
        for k in pnames:
            if ps[k].grad is not None:
                bufs[k] = R.sgd_nesterov_step(sd[k], ps[k].grad, bufs.get(k), lr, 0.937, 0.0)
    for k, s, a in zip(g.strs("final_keys"), g.flat["final_sums"], g.flat["final_abs"]):
        assert abs(float(sd[k].double().sum()) - s) <= 1e-4 * max(a, 1e-6), k


def _yaml_sd(cfg_path, swap):
    """state_dict skeleton (names + shapes) of a yaml script model, built by the oracle's own shape logic."""
    from tests.model_shapes import script_model_state_shapes
    cfg = yaml.safe_load(open(cfg_path))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = swap.get(l[2], l[2])
    shapes = script_model_state_shapes(cfg)
    return cfg, lambda: {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
                         for k, s in shapes.items()}


CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)


def test_model_yolov5seg():
    import os
    g = Golden("model_yolov5seg_64")
    cfg, mk = _yaml_sd(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov5_seg.yaml"),
                       {"C3_DCN": "C3"})
    _model_check(g, lambda sd, x: R.script_model_forward(sd, cfg, x, (64, 64)), mk, dict(class_weights=CW, kind="dice"))


def test_model_yolov8seg():
    import os
    g = Golden("model_yolov8seg_64")
    cfg, mk = _yaml_sd(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov8_seg.yaml"),
                       {"C2f_DCN": "C2f"})
    _model_check(g, lambda sd, x: R.script_model_forward(sd, cfg, x, (64, 64), family="v8"), mk,
                 dict(class_weights=CW, kind="jaccard"))


def test_model_resnet18seg():
    from tests.model_shapes import resnet_seg_state_shapes
    g = Golden("model_resnet18seg_64")
    shapes = resnet_seg_state_shapes("basic", 12)
    mk = lambda: {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
                  for k, s in shapes.items()}
    _model_check(g, lambda sd, x: R.resnet_seg_forward(sd, x, "basic"), mk, dict(class_weights=None, kind="dice"))


def test_model_resnet50seg():
    """BASELINE config 3 (segment/train.py ResNet50 + SegmentHead)"""
    from tests.model_shapes import resnet_seg_state_shapes
    g = Golden("model_resnet50seg_64")
    shapes = resnet_seg_state_shapes("bottleneck", 12)
    mk = lambda: {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
                  for k, s in shapes.items()}
    _model_check(g, lambda sd, x: R.resnet_seg_forward(sd, x, "bottleneck", out_size=(640, 640)), mk,
                 dict(class_weights=None, kind="dice"))


def test_model_resnet50_yaml():
    """the yaml-driven ResNet50 + UNet-lite head (unet-lite/Resnet50/seg_diceloss_Resnet50.py:438-710, resnet50.yaml): ReLU Convs,
    argument-casting builder (C3 [512, False] -> n = 0), absolute head indices"""
    import os
    from tests.model_shapes import resnet50_yaml_state_shapes
    g = Golden("model_resnet50yaml_64")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "resnet50_seg.yaml")))
    shapes, alias = resnet50_yaml_state_shapes(cfg)

    def mk():
        sd = {}
        for k, s_ in shapes.items():
            sd[k] = sd[alias[k]] if k in alias else (torch.zeros(s_) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
        return sd
    _model_check(g, lambda sd, x: R.resnet50_yaml_forward(sd, cfg, x, (64, 64)), mk, dict(class_weights=CW, kind="dice"))


def test_model_yolov9seg():
    """BASELINE config 5 family (C3k2 + GAM + SPPF backbone); the fixture was generated with `GAM []` because the
    reference cannot build its own yaml's `GAM [512]` (GAM(c1, 512) is a TypeError)"""
    import os
    g = Golden("model_yolov9seg_64")
    cfg, mk = _yaml_sd(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov9_seg.yaml"), {})
    _model_check(g, lambda sd, x: R.script_model_forward(sd, cfg, x, (64, 64), family="v9"), mk,
                 dict(class_weights=CW, kind="dice"))


@pytest.mark.parametrize("name", names("dcnmod_"))
def test_dcnv3_module(name):
    """the DCNv3 module (modules/dcnv3.py:50-136) — fixture from the reference's own class with its pure-PyTorch core"""
    g = Golden(name)
    C, k, s, pad, G = [int(v) for v in g.flat["meta"]]
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    out = R.dcnv3_module(wsd, "m", x, k, s, pad, G)
    # the restated core gathers explicitly where the reference uses grid_sample: sums re-associate
    assert rel_err(out, g.t("out")) < 2e-5, g.name
    (out * g.t("gup")).sum().backward()
    assert rel_err(x.grad, g.t("gx0")) < 5e-5
    for kk, v in g.group("grad").items():
        assert rel_err(sd[kk].grad, v) < 1e-4, (g.name, kk, rel_err(sd[kk].grad, v))
    for kk, v in g.group("sd_after").items():
        if v.dtype.is_floating_point:
            assert rel_err(sd[kk].detach(), v) < TOL, (g.name, kk)


@pytest.mark.parametrize("name", names("c3_dcnv3_"))
def test_c3_dcnv3(name):
    """C3_DCNV3 / Bottleneck_DCNV3 / DCNV3_YoLo ("common and yolo.py":2-38)"""
    g = Golden(name)
    c1, c2, n = [int(v) for v in g.flat["meta"]]
    sd = _sd(g)
    wsd, _ = _wrap(sd, "m")
    x = _x(g)
    out = R.c3_dcnv3(wsd, "m", x, n, shortcut="noshortcut" not in name)
    assert rel_err(out, g.t("out")) < 2e-5, g.name
    (out * g.t("gup")).sum().backward()
    assert rel_err(x.grad, g.t("gx0")) < 1e-4
    grads = g.group("grad")
    gscale = max(float(v.abs().max()) for v in grads.values())
    for kk, v in grads.items():
        if float(v.abs().max()) < 1e-4 * gscale:
            # mathematically zero (a per-channel constant in front of conv + train-mode BN, e.g. output_proj.bias): rounding noise
            assert float(sd[kk].grad.abs().max()) < 1e-4 * gscale, (g.name, kk)
            continue
        assert rel_err(sd[kk].grad, v) < 2e-4, (g.name, kk, rel_err(sd[kk].grad, v))
