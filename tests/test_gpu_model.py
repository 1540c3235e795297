"""GPU parity of whole models against the reference's own 2-step training trajectories (tests/golden/model_*.npz),
plus size-independent properties at the benchmark's full size."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle.fill import fill_state_dict
from tests.util import Golden, l2_err, rel_err

pytestmark = pytest.mark.gpu
CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
CFG = os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg")


def _cfg(name, swap):
    cfg = yaml.safe_load(open(os.path.join(CFG, name)))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = swap.get(l[2], l[2])
    return cfg


def _train_check(g, model, crit, mode, steps=2, lr=0.01, f32_grad_tol=2e-3, later_loss_tol=None, f32_out_tol=1e-4,
                 bf16_out_tol=0.35, bf16_grad_tol=0.15):
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype(mode)
    # bf16 at this toy size (64x64, batch 2: the deepest BatchNorms see 8 values per channel): loss 5e-2, sampled logits
    # ``bf16_out_tol`` max-relative, every gradient NORM within ``bf16_grad_tol`` of the reference's — about twice what was measured
    # on MI355X (YOLOv5Seg: logits 0.17, gradient norms median 0.010 / max 0.062; the ill-conditioned yaml ResNet50, whose f32
    # run is itself 1.6e-2 off on single norms: logits 0.77, norms median 0.16 / max 0.38); per-layer bf16 accuracy at a realistic
    # size is test_bf16_tracks_f32 below
    tol = dict(f32=dict(out=f32_out_tol, loss=1e-4, gn=f32_grad_tol, fin=f32_grad_tol),
               bf16=dict(out=bf16_out_tol, loss=5e-2, gn=bf16_grad_tol, fin=None))[mode]
    sd = model.state_dict()
    fill_state_dict(sd, 1234, bn_stats=False)
    model.load_state_dict(sd)
    model = model.cuda().train()
    # the reference run used torch.optim.SGD(model.parameters(), lr, momentum=0.937, nesterov=True): one group, no decay
    opt = ydl.FlatSGDEMA(model, lr=lr, momentum=0.937, weight_decay=0.0, ema=False)
    x, tgt = g.t("x").cuda(), g.t("target").cuda()
    for st in range(steps):
        opt.zero_grad()
        out = model(x)
        total, items = crit(out, tgt)
        total.backward()
        if st == 0:
            assert list(out.shape) == [int(v) for v in g.flat["out_shape"]]
            vals = out.detach().flatten()[g.t("out_idx").cuda()].cpu()
            e = rel_err(vals, g.t("out_vals"))
            if os.environ.get("YDL_TEST_PRINT"):
                named_ = dict(model.named_parameters())
                gn_ = [abs(float(named_[k].grad.double().norm()) - n) / max(n, 1e-7) for k, n in zip(g.strs("grad_names"), g.flat["grad_norms"])]
                print(f"[train_check {g.name} {mode}] logits rel {e:.3e}; grad-norm rel err median {float(np.median(gn_)):.3e} max {max(gn_):.3e}")
            assert tol["out"] is None or e < tol["out"], ("logits", e)
            named = dict(model.named_parameters())
            none = sorted(k for k, p in named.items() if not getattr(p, "_ydl_touched", False))
            assert none == sorted(g.strs("grad_none")), "set of parameters without gradient differs (SURVEY T4)"
            bad = []
            for k, n in zip(g.strs("grad_names"), g.flat["grad_norms"]):
                got = float(named[k].grad.double().norm())
                if tol["gn"] is not None and abs(got - n) > tol["gn"] * max(n, 1e-7):
                    bad.append((k, got, n))
            assert not bad, bad[:5]
        ref = g.flat[f"loss_items_{st}"]
        ltol = tol["loss"] if (st == 0 or later_loss_tol is None or mode != "f32") else later_loss_tol
        for a, b in zip(items, ref):
            assert abs(a - b) <= ltol * abs(b), (st, items, ref)
        opt.step()
    sd = model.state_dict()
    bad = []
    for k, s, a in zip(g.strs("final_keys"), g.flat["final_sums"], g.flat["final_abs"]):
        got = float(sd[k].double().sum())
        if tol["fin"] is not None and abs(got - s) > tol["fin"] * max(a, 1e-6):
            bad.append((k, got, s, a))
    assert not bad, bad[:5]
    ydl.set_compute_dtype("bf16")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_yolov5seg_trajectory(mode):
    import yolo_dual_amd as ydl
    g = Golden("model_yolov5seg_64")
    m = ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml", {"C3_DCN": "C3"}))
    m.img_size = [64, 64]
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, CW, "dice"), mode)


@pytest.mark.parametrize("mode", ["f32"])
def test_yolov8seg_trajectory(mode):
    import yolo_dual_amd as ydl
    g = Golden("model_yolov8seg_64")
    m = ydl.YOLOv8Seg(_cfg("yolov8_seg.yaml", {"C2f_DCN": "C2f"}))
    m.img_size = [64, 64]
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, CW, "jaccard"), mode)


@pytest.mark.parametrize("mode", ["f32"])
def test_resnet18seg_trajectory(mode):
    import yolo_dual_amd as ydl
    g = Golden("model_resnet18seg_64")
    m = ydl.ResNet18Seg({"nc": 12})
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, None, "dice"), mode)


@pytest.mark.parametrize("mode", ["f32"])
def test_resnet50seg_trajectory(mode):
    """BASELINE config 3: ResNet50 (SiLU Conv blocks, ReLU residual joins) + multi-scale SegmentHead"""
    import yolo_dual_amd as ydl
    g = Golden("model_resnet50seg_64")
    m = ydl.ResNet50Seg({"nc": 12})
    # logits and loss are held to 1e-4; the gradient NORMS of this 53-conv net on a 64x64 input (BatchNorm over 32 values in
    # layer3) are only reproducible to ~1 % in fp32: the CPU oracle in f32 vs f64 differs by 1.6 % (median, element-wise),
    # this path in f32 vs the f64 oracle by 1.1 % (tools/r50_debug.py) — hence 1e-2 here instead of 2e-3
    # (the loss after the first update inherits that gradient noise: 2.3e-4 measured, bound 1e-3; step 0 — identical
    # weights and inputs — stays at 1e-4)
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, None, "dice"), mode, f32_grad_tol=1e-2, later_loss_tol=1e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_resnet50_yaml_trajectory(mode):
    """the yaml-driven ResNet50 + UNet-lite head (unet-lite/Resnet50/seg_diceloss_Resnet50.py:438-710 with resnet50.yaml): ReLU Conv
    blocks, argument-casting builder, C3 without residual — 2-step trajectory of the reference's own class"""
    import yolo_dual_amd as ydl
    g = Golden("model_resnet50yaml_64")
    m = ydl.ResNet50SegYaml(_cfg("resnet50_seg.yaml", {}))
    m.img_size = [64, 64]
    # 53 convolutions on a 64x64 input: the deepest BatchNorms normalise over 8 values (2x2 pixels, batch 2), and fp32 itself is
    # only reproducible to 3.4e-4 on the output probabilities here (CPU oracle in f32 vs the same oracle in f64; the f32 oracle
    # equals the reference fixture bit for bit) — this path lands at 3.2e-4.  Hence 1e-3 on the outputs; the loss, an average over
    # all pixels, still holds 1e-4.  Gradient norms: the f32 oracle with every weight scaled by (1 + 1e-7) moves the layer1 / layer2
    # BatchNorm gradient norms of this fixture by up to 1.5e-2 (backbone.1.layer.0.conv2.bn.bias), the f64 oracle by 1.3e-2 — ReLU
    # masks flip under BatchNorms that see 8..128 values.  The HIP path differs by up to 1.6e-2 on the same entries: bound 3e-2.
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, CW, "dice"), mode, f32_grad_tol=3e-2, later_loss_tol=1e-3, f32_out_tol=1e-3,
                 bf16_out_tol=1.6, bf16_grad_tol=0.8)


@pytest.mark.parametrize("mode", ["f32"])
def test_yolov9seg_trajectory(mode):
    """BASELINE config 5 family without the DCN swap: C3k2 + GAM + SPPF backbone, v9 head"""
    import yolo_dual_amd as ydl
    g = Golden("model_yolov9seg_64")
    m = ydl.YOLOv9Seg(_cfg("yolov9_seg.yaml", {}))
    m.img_size = [64, 64]
    _train_check(g, m, ydl.SegmentationLoss(12, 0.0, CW, "dice"), mode)


def test_bf16_tracks_f32():
    """throughput mode vs parity mode of the same kernels at a realistic size (256x256, batch 4; the f32 path is pinned to the
    reference at 1e-4): probabilities, loss and EVERY parameter gradient, per layer.  Measured on MI355X (tools/bf16_layer_err.py
    prints the table): relative L2 of the bf16 gradient from the f32 one is 0.003-0.06 at the last layer, 0.05-0.13 through the
    head down to the SPPF's second conv, and 0.24-0.32 for everything behind the SPPF's max-pools (bf16 rounding re-routes
    arg-max elements there), flat from backbone.9.cv1 to the stem — it does not compound per layer.  Bounds: 1.5x the measured
    worst of each group.  (Even the last layer's weight gradient is at 5 %: BatchNorm's backward removes the mean and the
    x-hat component of dz, the remainder is small against the bf16-rounded operands it is computed from.)"""
    import yolo_dual_amd as ydl
    from tests.util import l2_err
    from yolo_dual_amd import config
    res = {}
    planes = {}
    # third leg (round 5): the bf16 run again, but its SPPF max-pool BACKWARD routes by the f32 run's arg-max planes — the test of the
    # attribution above: if the backbone's 0.24-0.32 is re-routed arg-max elements, it must fall to the head's level with them forced
    for mode in ("f32", "bf16", "bf16_forced"):
        ydl.set_compute_dtype("bf16" if mode.startswith("bf16") else "f32")
        if mode == "f32":
            config.set_sppf_argmax_hook(lambda idx: planes.setdefault("f32", [t.clone() for t in idx]))
        elif mode == "bf16_forced":
            def force(idx):
                assert all(a.shape == b.shape for a, b in zip(idx, planes["f32"]))
                planes["changed"] = [float((a != b).float().mean()) for a, b in zip(idx, planes["f32"])]
                for a, b in zip(idx, planes["f32"]):
                    a.copy_(b)
            config.set_sppf_argmax_hook(force)
        else:
            config.set_sppf_argmax_hook(None)
        m = ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml", {"C3_DCN": "C3"}))
        m.img_size = [256, 256]
        sd = m.state_dict()
        fill_state_dict(sd, 99, bn_stats=False)
        m.load_state_dict(sd)
        m = m.cuda().train()
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
        gen = torch.Generator("cuda").manual_seed(5)
        x = torch.rand(4, 3, 256, 256, device="cuda", generator=gen)
        t = torch.randint(0, 12, (4, 256, 256), device="cuda", generator=gen)
        out = m(x)
        total, items = crit(out, t)
        total.backward()
        res[mode] = (out.detach().float().cpu(), items,
                     {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if getattr(p, "_ydl_touched", False)})
    config.set_sppf_argmax_hook(None)
    ydl.set_compute_dtype("bf16")
    assert sorted(res["bf16"][2]) == sorted(res["f32"][2])
    errs_forced = {k: l2_err(res["bf16_forced"][2][k], res["f32"][2][k]) for k in res["f32"][2]}
    back_forced = [e for k, e in errs_forced.items() if k.startswith("backbone.") and not k.startswith("backbone.9.cv2.")]
    print(f"[bf16 vs f32] arg-max elements that differ per pool stage: {planes['changed']}; backbone gradient error with the f32 "
          f"planes forced: median {float(np.median(back_forced)):.3f}, worst {max(back_forced):.3f}")
    # measured on MI355X: 2.5 / 2.2 / 1.4 % of the arg-max elements of the three pool stages differ between the two modes; with the f32
    # planes forced the backbone's error falls from median 0.265 / worst 0.32 to median 0.125 / worst 0.156 — the level of the head
    # below the SPPF (0.05-0.13): the re-routed elements account for the jump at the pools, the rest is ordinary bf16 rounding
    assert max(back_forced) < 0.2 and float(np.median(back_forced)) < 0.16, \
        sorted(((e, k) for k, e in errs_forced.items() if k.startswith("backbone.")), reverse=True)[:5]
    assert l2_err(res["bf16"][0], res["f32"][0]) < 0.09                           # measured 0.062
    assert abs(res["bf16"][1][0] - res["f32"][1][0]) <= 1e-3 * abs(res["f32"][1][0])      # measured 1e-5
    errs = {k: l2_err(res["bf16"][2][k], res["f32"][2][k]) for k in res["f32"][2]}

    def bound(k):
        if k.startswith("head.17."):
            return 0.09                                                            # measured <= 0.055
        if k.startswith("head.") or k.startswith("backbone.9.cv2."):
            return 0.2                                                             # measured <= 0.133
        return 0.42                                                                # behind the max-pools: 1.3 x the measured 0.32
    bad = {k: (e, bound(k)) for k, e in errs.items() if e > bound(k)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:6]
    back = [e for k, e in errs.items() if k.startswith("backbone.") and not k.startswith("backbone.9.cv2.")]
    assert float(np.median(back)) < 0.35, float(np.median(back))                  # measured 0.265


@pytest.mark.parametrize("shape", [(2, 64, 96, 20, 24), (2, 64, 64, 320, 328)])
def test_wgrad_transposed_read_matches_scalar_read(shape):
    """A/B the ds_read_b64_tr_b16 operand paths of the bf16 wgrad kernels (64x64-tile kernel; 128-wide pipelined kernel,
    selected for M >= 200k pixels) against the scalar-LDS-read kernel."""
    N, C1, C2, H, W = shape
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    ydl.set_compute_dtype("bf16")
    torch.manual_seed(0)
    res = []
    config.set_bn_sums(False)        # both runs see bit-identical dy (the atomically added BN sums differ in their last bits run to run)
    for tr in (1, 0):
        L.debug_set(0, tr)
        m = ydl.Conv(C1, C2, 3, 1).cuda().train()
        torch.manual_seed(1)
        with torch.no_grad():
            m.conv.weight.normal_(0, 0.05)
        x = torch.randn(N, C1, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(2)).requires_grad_(True)
        out = m(x)
        out.square().sum().backward()
        res.append(m.conv.weight.grad.detach().clone())
    L.debug_set(0, 1)
    config.set_bn_sums(True)
    assert rel_err(res[0].cpu(), res[1].cpu()) < 1e-4       # same products, different split-K / atomic order only


@pytest.mark.parametrize("mode,c1,c2", [("bf16", 64, 64), ("bf16", 128, 128), ("bf16", 64, 128), ("bf16", 256, 256),
                                        ("bf16", 128, 256), ("bf16", 256, 64), ("f32", 32, 64), ("f32", 64, 128),
                                        ("f32", 128, 128), ("f32", 128, 256)])
def test_pointwise_streaming_kernel_matches_tiled_kernel(mode, c1, c2):
    """A/B the weight-stationary streaming kernel (1x1 convs, K row <= 512 B, M >= 65536 pixels; forward with BN partials
    and dgrad) against the tiled implicit-GEMM kernel on a ragged pixel count: same products, different summation order."""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    ydl.set_compute_dtype(mode)
    res = []
    for pw in (1, 0):
        L.debug_set(1, pw)
        torch.manual_seed(1)
        m = ydl.Conv(c1, c2, 1, 1).cuda().train()
        with torch.no_grad():
            m.conv.weight.normal_(0, 0.1)
            m.bn.weight.uniform_(0.5, 1.5)
            m.bn.bias.normal_(0, 0.2)
        x = (torch.randn(2, c1, 200, 173, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) + 0.3).requires_grad_(True)
        out = m(x)
        (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(3))).sum().backward()
        res.append([t.detach().float().cpu() for t in (out, x.grad, m.bn.running_mean, m.bn.running_var, m.conv.weight.grad,
                                                        m.bn.weight.grad)])
    L.debug_set(1, 1)
    tol = 1e-5 if mode == "f32" else 2e-2        # bf16: the two kernels round the same f32 sums, but BN statistics
    names_ = ("out", "dx", "running_mean", "running_var", "dw", "dgamma")    # differing in the last bit move bf16 outputs
    for a, b, nm in zip(res[0], res[1], names_):
        assert l2_err(a, b) < tol, (nm, l2_err(a, b))
    assert l2_err(res[0][2], res[1][2]) < 1e-5 and l2_err(res[0][3], res[1][3]) < 1e-5     # statistics come from f32 accumulators


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pointwise_streaming_kernel_in_c3_block(mode):
    """the same A/B through a C3 block: fused cv1|cv2 pair writing channel slices (ld != C), dgrad accumulation into a shared
    input gradient"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    ydl.set_compute_dtype(mode)
    c = 64 if mode == "f32" else 128
    res = []
    for pw in (1, 0):
        L.debug_set(1, pw)
        torch.manual_seed(4)
        m = ydl.C3(c, c, 1).cuda().train()
        x = torch.randn(2, c, 184, 180, device="cuda", generator=torch.Generator("cuda").manual_seed(5)).requires_grad_(True)
        out = m(x)
        (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(6))).sum().backward()
        res.append([out.detach().float().cpu(), x.grad.float().cpu()] + [p.grad.float().cpu().flatten() for p in m.parameters()])
    L.debug_set(1, 1)
    tol = 2e-5 if mode == "f32" else 5e-2
    for a, b in zip(res[0], res[1]):
        assert l2_err(a, b) < tol, l2_err(a, b)


@pytest.mark.parametrize("family", ["YOLOv5Seg", "YOLOv8Seg", "YOLOv9Seg"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_commuted_concat_conv_matches_materialised_concat(mode, family):
    """conv1x1(cat(a, up(b))) evaluated as conv_a(a) + up(conv_b(b)) (virtual concat, column-block weight gradients) == the plain path
    (up-sample b, concat, one conv) on whole models: prediction, loss, every gradient.  ``up`` is the Concat's own bilinear
    auto-align (seg_diceloss_yolov5.py:484-507); the yolov8 / yolov9 heads (explicit nn.Upsample rows, Concat with an earlier
    backbone layer) go through the same switch and must not change either."""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import config
    ydl.set_compute_dtype(mode)
    cfg = {"YOLOv5Seg": _cfg("yolov5_seg.yaml", {"C3_DCN": "C3"}), "YOLOv8Seg": _cfg("yolov8_seg.yaml", {"C2f_DCN": "C2f"}),
           "YOLOv9Seg": _cfg("yolov9_seg.yaml", {})}[family]
    res = []
    # the deterministic forms on both sides, so that the backbone (identical in the two arms) contributes nothing and the difference
    # is the head algebra alone: throughput mode's atomically added BN sums differ in their last bits from run to run, which at
    # this toy size (2 x 128 x 128: BatchNorms over 32 values) already moves the early layers' bf16 gradients by 20 % between two
    # runs of the SAME arm
    config.set_deterministic(True)
    for on in (True, False):
        config.set_commute_concat(on)
        try:
            m = getattr(ydl, family)(cfg)
            m.img_size = [128, 128]
            sd = m.state_dict()
            fill_state_dict(sd, 5)
            m.load_state_dict(sd)
            m = m.cuda().train()
            opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.9, weight_decay=0.0)
            crit = ydl.SegmentationLoss(12, 0.0, CW, "dice", sync=False)
            gen = torch.Generator("cuda").manual_seed(0)
            x = torch.rand(2, 3, 128, 128, device="cuda", generator=gen)
            t = torch.randint(0, 12, (2, 128, 128), device="cuda", generator=gen)
            opt.zero_grad()
            out = m(x)
            total, _ = crit(out, t)
            total.backward()
            named = dict(m.named_parameters())
            grads = {k: p.grad.detach().float().cpu().clone() for k, p in named.items() if getattr(p, "_ydl_touched", False)}
            res.append((out.detach().cpu(), float(total), grads,
                        {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items() if "running" in k}))
        finally:
            config.set_commute_concat(True)
    config.set_deterministic(None)
    (o1, l1, g1, r1), (o0, l0, g0, r0) = res
    assert sorted(g1) == sorted(g0)                       # same dead-parameter set
    to, tg = (1e-5, 2e-4) if mode == "f32" else (3e-2, 0.15)
    assert l2_err(o1, o0) < to and abs(l1 - l0) <= to * abs(l0)
    for k in r0:
        assert l2_err(r1[k], r0[k]) < to, k
    bad = {k: l2_err(g1[k], g0[k]) for k in g0 if l2_err(g1[k], g0[k]) >= tg}
    assert not bad, bad


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_resize_last_order_of_the_commuted_concat_conv(mode):
    """replica-sums mode CAN evaluate conv1x1(cat(a, bilinear_up(b))) as conv_a(a) first and ``y += bilinear_up(conv_b(b))`` last, with
    the BatchNorm statistics of the sum taken by that streaming pass (ydl_resize_acc_sums; config.set_resize_last, off by default: it
    measured no faster); the default order is: resize initialises y, conv_a accumulates with statistics.  Whole yolov5 model, both orders in sums mode: prediction, loss,
    running statistics and every gradient agree (f32: to the atomics' arrival-order noise; bf16: one rounding of y moves)."""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    ydl.set_compute_dtype(mode)
    cfg = _cfg("yolov5_seg.yaml", {"C3_DCN": "C3"})
    res = []
    config.set_deterministic(False)                       # f32 too runs the replica-sums path then
    try:
        assert config.bn_sums(mode)
        for on in (True, False):
            config.set_resize_last(on)
            m = ydl.YOLOv5Seg(cfg)
            m.img_size = [128, 128]
            sd = m.state_dict()
            fill_state_dict(sd, 5)
            m.load_state_dict(sd)
            m = m.cuda().train()
            crit = ydl.SegmentationLoss(12, 0.0, CW, "dice", sync=False)
            gen = torch.Generator("cuda").manual_seed(0)
            x = torch.rand(4, 3, 128, 128, device="cuda", generator=gen)
            t = torch.randint(0, 12, (4, 128, 128), device="cuda", generator=gen)
            n0 = L.launch_count()
            L.profile_begin()
            out = m(x)
            names = [r["name"] for r in L.profile_end()]
            assert ("ydl_resize_acc_sums" in names) == on, names
            total, _ = crit(out, t)
            total.backward()
            named = dict(m.named_parameters())
            grads = {k: p.grad.detach().float().cpu().clone() for k, p in named.items() if getattr(p, "_ydl_touched", False)}
            res.append((out.detach().cpu(), float(total), grads,
                        {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items() if "running" in k}))
    finally:
        config.set_resize_last(False)
        config.set_deterministic(None)
        ydl.set_compute_dtype("bf16")
    (o1, l1, g1, r1), (o0, l0, g0, r0) = res
    assert sorted(g1) == sorted(g0)
    # (bf16: the atomically added BN sums already move the early layers' gradients by 20 % between two runs of the SAME arm at this size)
    to, tg = (2e-5, 5e-4) if mode == "f32" else (4e-2, 0.5)
    assert l2_err(o1, o0) < to and abs(l1 - l0) <= to * abs(l0), (l2_err(o1, o0), l1, l0)
    for k in r0:
        assert l2_err(r1[k], r0[k]) < to, k
    bad = {k: l2_err(g1[k], g0[k]) for k in g0 if l2_err(g1[k], g0[k]) >= tg}
    assert not bad, bad


@pytest.mark.parametrize("mode,c1,k,st,p", [("f32", 3, 6, 2, 2), ("bf16", 3, 6, 2, 2), ("f32", 1, 4, 2, 0), ("f32", 4, 4, 2, 2),
                                            ("f32", 3, 6, 3, 3)])
def test_stem_space_to_depth_matches_plain_conv(mode, c1, k, st, p):
    """a stem conv with k and p multiples of its stride, on the raw region input, runs as the stride-1 conv over the
    space-to-depth form (ydl_nchw_to_s2d / ydl_weight_prep_s2d / ydl_wgrad_unpack_s2d): same output, statistics and weight
    gradient as the plain strided conv (which the golden tests pin to the reference)"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import config
    ydl.set_compute_dtype(mode)
    res = []
    for on in (True, False):
        config.set_stem_s2d(on)
        try:
            torch.manual_seed(7)
            m = ydl.Conv(c1, 24, k, st, p).cuda().train()
            with torch.no_grad():
                m.conv.weight.normal_(0, 0.2)
            x = torch.randn(2, c1, 36, 48, device="cuda", generator=torch.Generator("cuda").manual_seed(8))
            out = m(x)
            (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(9))).sum().backward()
            res.append([t.detach().float().cpu() for t in (out, m.conv.weight.grad, m.bn.running_mean, m.bn.running_var,
                                                            m.bn.weight.grad, m.bn.bias.grad)])
        finally:
            config.set_stem_s2d(True)
    tol = 2e-5 if mode == "f32" else 3e-2
    for a, b, nm in zip(res[0], res[1], ("out", "dw", "running_mean", "running_var", "dgamma", "dbeta")):
        assert a.shape == b.shape and l2_err(a, b) < tol, (nm, l2_err(a, b))


@pytest.mark.parametrize("mode,k,st,p,hw", [("f32", 3, 2, 1, (37, 50)), ("bf16", 3, 2, 1, (37, 50)), ("f32", 6, 2, 2, (36, 42)),
                                            ("f32", 3, 3, 1, (31, 29)), ("f32", 1, 2, 0, (20, 21))])
def test_strided_dgrad_single_launch_matches_per_class_launches(mode, k, st, p, hw):
    """stride-s input gradient: all s*s output-parity classes in one launch (class table in the kernel arguments) == one
    launch per class; odd sizes give the classes different pixel counts, k < s leaves classes without taps (zeros)"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    ydl.set_compute_dtype(mode)
    res = []
    for merged in (1, 0):
        L.debug_set(2, merged)
        torch.manual_seed(11)
        m = ydl.Conv(16, 24, k, st, p).cuda().train()
        x = torch.randn(2, 16, *hw, device="cuda", generator=torch.Generator("cuda").manual_seed(12)).requires_grad_(True)
        out = m(x)
        (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(13))).sum().backward()
        res.append(x.grad.detach().float().cpu())
    L.debug_set(2, 1)
    assert torch.equal(res[0], res[1])          # same tiles, same K order: bit-identical


def test_full_size_properties():
    """BASELINE config 2 at its full per-GPU size (640x640, bs=16): size-independent checks — probabilities sum to 1,
    finite loss, every live parameter gets a finite non-zero gradient, and the dead-parameter set equals the one of the
    small golden run."""
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("bf16")
    g = Golden("model_yolov5seg_64")
    m = ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml", {"C3_DCN": "C3"})).cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
    gen = torch.Generator("cuda").manual_seed(0)
    x = torch.rand(16, 3, 640, 640, device="cuda", generator=gen)
    t = torch.randint(0, 12, (16, 640, 640), device="cuda", generator=gen)
    opt.zero_grad()
    out = m(x)
    assert out.shape == (16, 12, 640, 640)
    s = out.sum(1)
    assert float((s - 1).abs().max()) < 1e-4
    total, items = crit(out, t)
    assert np.isfinite(items).all()
    total.backward()
    named = dict(m.named_parameters())
    none = sorted(k for k, p in named.items() if not getattr(p, "_ydl_touched", False))
    assert none == sorted(g.strs("grad_none"))
    for k, p in named.items():
        if getattr(p, "_ydl_touched", False):
            gn = float(p.grad.float().norm())
            assert np.isfinite(gn) and gn > 0, k
    before = float(items[0])
    opt.step()
    for _ in range(3):
        opt.zero_grad()
        total, items = crit(m(x), t)
        total.backward()
        opt.step()
    assert items[0] < before, "loss did not decrease on a fixed batch"


def test_graph_capture_matches_eager():
    """the HIP-graph replay of the training step follows the eager trajectory (same kernels, same order)"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import config
    from yolo_dual_amd.graph import GraphedTrainStep
    ydl.set_compute_dtype("bf16")
    # bit-level comparison: the deterministic forms (slab weight gradients, partial-row BN statistics); throughput mode adds
    # weight gradients and BN sums with f32 atomics, whose arrival order differs from run to run
    config.set_deterministic(True)
    losses = {}
    for mode in ("eager", "graph"):
        m = ydl.YOLOv5Seg(_cfg("yolov5_seg.yaml", {"C3_DCN": "C3"}))
        m.img_size = [128, 128]
        sd = m.state_dict()
        fill_state_dict(sd, 5, bn_stats=False)
        m.load_state_dict(sd)
        m = m.cuda().train()
        opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice", sync=False)
        gen = torch.Generator("cuda").manual_seed(3)
        x = torch.rand(4, 3, 128, 128, device="cuda", generator=gen)
        t = torch.randint(0, 12, (4, 128, 128), device="cuda", generator=gen)
        rec = []
        if mode == "eager":
            for _ in range(6):
                opt.zero_grad()
                total, items = crit(m(x), t)
                total.backward()
                opt.step()
                rec.append(float(items[0]))
        else:
            g = GraphedTrainStep(m, crit, opt, x, t, warmup=2)      # 2 eager warm-up steps inside
            rec = [None, None]
            for _ in range(4):
                items = g.step()
                rec.append(float(items[0]))
            nbt = int(m.state_dict()["backbone.0.bn.num_batches_tracked"])
            assert nbt == 6, nbt
            assert opt.updates == 6
        losses[mode] = rec
    # same kernels on the same data in the same order: the replayed trajectory is the eager one to the last bit of the
    # printed loss (a 2e-2 tolerance here once hid a replay that ran with broken cross-stream ordering)
    config.set_deterministic(None)
    for a, b in zip(losses["eager"][2:], losses["graph"][2:]):
        assert abs(a - b) <= 1e-6 * abs(a), (losses)


def test_config4_size_yolov8_jaccard():
    """BASELINE config 4 at its full per-GPU size (YOLOv8 C2f backbone, Jaccard loss, 1024x1024, bs=8 — 2^21 stem output
    pixels, the largest pixel count of any config): size-independent checks — probabilities sum to 1, finite loss, every
    live parameter gets a finite non-zero gradient, the loss goes down on a fixed batch"""
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("bf16")
    m = ydl.YOLOv8Seg(_cfg("yolov8_seg.yaml", {"C2f_DCN": "C2f"}))
    m.img_size = [1024, 1024]
    m = m.cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "jaccard")
    gen = torch.Generator("cuda").manual_seed(0)
    x = torch.rand(8, 3, 1024, 1024, device="cuda", generator=gen)
    t = torch.randint(0, 12, (8, 1024, 1024), device="cuda", generator=gen)
    losses = []
    for st in range(3):
        opt.zero_grad()
        out = m(x)
        if st == 0:
            assert out.shape == (8, 12, 1024, 1024)
            assert float((out.sum(1) - 1).abs().max()) < 1e-4
        total, items = crit(out, t)
        assert np.isfinite(items).all()
        total.backward()
        if st == 0:
            for k, p in m.named_parameters():
                if getattr(p, "_ydl_touched", False):
                    gn = float(p.grad.float().norm())
                    assert np.isfinite(gn) and gn > 0, k
        opt.step()
        losses.append(float(items[0]))
    assert losses[-1] < losses[0], losses


def test_config3_size_resnet50():
    """BASELINE config 3 at its full per-GPU size (ResNet50 + multi-scale SegmentHead, 640x640, bs=32: 3.3 M stem output
    pixels, 25 600 BN partial rows): finite loss, finite non-zero gradients on the live parameters (layer4 stays without
    gradient), loss goes down on a fixed batch"""
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("bf16")
    m = ydl.ResNet50Seg({"nc": 12}).cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, None, "dice")
    gen = torch.Generator("cuda").manual_seed(0)
    x = torch.rand(32, 3, 640, 640, device="cuda", generator=gen)
    t = torch.randint(0, 12, (32, 640, 640), device="cuda", generator=gen)
    losses = []
    for st in range(3):
        opt.zero_grad()
        out = m(x)
        assert out.shape == (32, 12, 640, 640)
        total, items = crit(out, t)
        assert np.isfinite(items).all()
        total.backward()
        if st == 0:
            named = dict(m.named_parameters())
            live = [k for k, p in named.items() if getattr(p, "_ydl_touched", False)]
            assert live and not any(k.startswith("backbone.layer4") for k in live)
            for k in live:
                gn = float(named[k].grad.float().norm())
                assert np.isfinite(gn) and gn > 0, k
        opt.step()
        losses.append(float(items[0]))
    assert losses[-1] < losses[0], losses


def test_config5_size_yolov9():
    """BASELINE config 5 family at its full per-GPU size (YOLOv9 C3k2 + GAM + SPPF backbone, 640x640, bs=16; the DCN
    variants of its blocks are parity-unpinned and not built): same size-independent checks"""
    import yolo_dual_amd as ydl
    ydl.set_compute_dtype("bf16")
    m = ydl.YOLOv9Seg(_cfg("yolov9_seg.yaml", {})).cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
    gen = torch.Generator("cuda").manual_seed(0)
    x = torch.rand(16, 3, 640, 640, device="cuda", generator=gen)
    t = torch.randint(0, 12, (16, 640, 640), device="cuda", generator=gen)
    losses = []
    for st in range(3):
        opt.zero_grad()
        out = m(x)
        assert out.shape == (16, 12, 640, 640)
        total, items = crit(out, t)
        assert np.isfinite(items).all()
        total.backward()
        if st == 0:
            for k, p in m.named_parameters():
                if getattr(p, "_ydl_touched", False):
                    gn = float(p.grad.float().norm())
                    assert np.isfinite(gn) and gn > 0, k
        opt.step()
        losses.append(float(items[0]))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("cfg_name,cls", [("yolov5_seg.yaml", "YOLOv5Seg"), ("yolov8_seg.yaml", "YOLOv8Seg")])
def test_bn_backward_reduce_fused_into_the_last_dgrad(cfg_name, cls):
    """throughput mode: the BatchNorm backward's reduce pass runs in the epilogue of the dgrad that writes the layer's output gradient
    last (ydl_conv_dgrad_bnred + ydl_bn_act_bwd_apply_sums, Tape._try_bnred) instead of as its own launch.  Same gradients as the
    two-launch form to the run-to-run noise of that form itself (f32 atomics in both), on a whole model with shortcuts, sibling
    pairs, concat slices and residual aliasing; and the fused entry points did run."""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    ydl.set_compute_dtype("bf16")
    real_call = L.call
    counts = {}

    def counting(name, *a):
        counts[name] = counts.get(name, 0) + 1
        return real_call(name, *a)
    grads = {}
    try:
        for tag, fuse in (("two_a", False), ("two_b", False), ("fused", True)):
            config.set_bn_bwd_fuse(fuse)
            m = getattr(ydl, cls)(_cfg(cfg_name, {"C3_DCN": "C3", "C2f_DCN": "C2f"}))
            m.img_size = [256, 256]
            sd = m.state_dict()
            fill_state_dict(sd, 5, bn_stats=False)
            m.load_state_dict(sd)
            m = m.cuda().train()
            opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
            crit = ydl.SegmentationLoss(12, 0.0, CW, "dice", sync=False)
            gen = torch.Generator("cuda").manual_seed(3)
            x = torch.rand(8, 3, 256, 256, device="cuda", generator=gen)
            t = torch.randint(0, 12, (8, 256, 256), device="cuda", generator=gen)
            opt.zero_grad()
            counts.clear()
            L.call = counting
            try:
                total, _items = crit(m(x), t)
                total.backward()
            finally:
                L.call = real_call
            torch.cuda.synchronize()
            grads[tag] = (opt.grads_arena.detach().float().cpu().clone(), dict(counts))
        n_fused = grads["fused"][1].get("ydl_conv_dgrad_bnred", 0)
        n_apply = grads["fused"][1].get("ydl_bn_act_bwd_apply_sums", 0)
        n_full = grads["fused"][1].get("ydl_bn_act_bwd_sums", 0)
        print(f"[bnred {cls}] fused dgrads {n_fused}, apply-only BN backward {n_apply}, two-launch BN backward {n_full}")
        assert n_fused >= 5 and n_apply >= n_fused, grads["fused"][1]
        assert grads["two_a"][1].get("ydl_conv_dgrad_bnred", 0) == 0
        noise = l2_err(grads["two_b"][0], grads["two_a"][0])
        got = l2_err(grads["fused"][0], grads["two_a"][0])
        print(f"[bnred {cls}] gradient arena: fused vs two-launch {got:.2e}, two-launch vs itself {noise:.2e}")
        assert got <= 3 * noise + 1e-4, (got, noise)
    finally:
        L.call = real_call
        config.set_bn_bwd_fuse(False)
        ydl.set_compute_dtype("bf16")


def test_bn_backward_reduce_fused_in_blocks():
    """the same comparison where it is sharp: single blocks (one to three BatchNorm layers between the seeded gradient and the input),
    so that bf16 rounding chaos cannot hide an error — input gradient and every parameter gradient of the fused form against the
    two-launch form, and the fused entry point must have run"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    ydl.set_compute_dtype("bf16")
    real_call = L.call
    counts = {}

    def counting(name, *a):
        counts[name] = counts.get(name, 0) + 1
        return real_call(name, *a)
    try:
        for make, shape in ((lambda: ydl.C3(128, 128, 1), (8, 128, 80, 80)), (lambda: ydl.C3(128, 128, 2, False), (4, 128, 96, 96)),
                            (lambda: ydl.Bottleneck(128, 128), (8, 128, 64, 64)), (lambda: ydl.C2f(128, 128, 1), (8, 128, 48, 48))):
            res = []
            for fuse in (False, True):
                config.set_bn_bwd_fuse(fuse)
                m = make()
                sd = m.state_dict()
                fill_state_dict(sd, 11, bn_stats=True)
                m.load_state_dict(sd)
                m = m.cuda().train()
                opt = ydl.FlatSGDEMA(m, lr=0.01)
                x = torch.randn(*shape, device="cuda", generator=torch.Generator("cuda").manual_seed(2)).requires_grad_(True)
                opt.zero_grad()
                counts.clear()
                L.call = counting
                try:
                    out = m(x)
                    (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(3))).sum().backward()
                finally:
                    L.call = real_call
                res.append(([x.grad.float().cpu()] + [p.grad.detach().float().clone().cpu() for p in m.parameters()], dict(counts)))
            assert res[1][1].get("ydl_conv_dgrad_bnred", 0) >= 1, (shape, res[1][1])
            assert res[0][1].get("ydl_conv_dgrad_bnred", 0) == 0
            worst = max(l2_err(b, a) for a, b in zip(res[0][0], res[1][0]))
            print(f"[bnred block {shape}] fused dgrads {res[1][1].get('ydl_conv_dgrad_bnred', 0)}, worst gradient difference {worst:.2e}")
            assert worst < 1.5e-2, (shape, worst)
    finally:
        L.call = real_call
        config.set_bn_bwd_fuse(False)
        ydl.set_compute_dtype("bf16")
