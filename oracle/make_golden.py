#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes on CPU (build container only).

TEST INFRASTRUCTURE.  This script *reads* /root/reference at run time (it execs the nn.Module / loss line
ranges listed in SURVEY.md §8c inside a scratch namespace with stubs for thop/LOGGER/check_yaml) and writes
only data — inputs, weights, expected outputs and gradients — into tests/golden/.  No reference source is
copied into this repository.  It never runs on the GPU box (/root/reference does not exist there).

    python oracle/make_golden.py            # regenerate everything
"""
from __future__ import annotations

import logging
import math
import os
import sys
import types
from copy import deepcopy
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.fill import fill_state_dict, rs_tensor  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load_ref(rel: str, ranges, extra=None) -> dict:
    """exec the given 1-based inclusive line ranges of a reference file in a stub namespace."""
    path = os.path.join(REF, rel)
    lines = open(path, encoding="utf-8").read().split("\n")
    src = "\n".join("\n".join(lines[a - 1:b]) for a, b in ranges)
    thop = types.SimpleNamespace(profile=lambda *a, **k: (0.0, 0.0))
    import warnings
    ns = dict(torch=torch, nn=nn, F=F, yaml=yaml, math=math, deepcopy=deepcopy, np=np, warnings=warnings,
              LOGGER=logging.getLogger("ref"), check_yaml=lambda x: x, thop=thop,
              Optional=Optional, Dict=Dict, List=List, Tuple=Tuple, Union=Union, Any=Any)
    if extra:
        ns.update(extra)
    exec(compile(src, path, "exec"), ns)
    return ns


def save(name: str, **arrs):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        elif torch.is_tensor(v):
            flat[k] = v.detach().cpu().numpy()
        else:
            flat[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    print("wrote", name, {k: tuple(v.shape) for k, v in list(flat.items())[:4]}, "...")


def run_module(mod: nn.Module, inputs, seed: int, as_list: bool = False, store_weights: bool = True):
    """Fill params deterministically, run train-mode fwd + bwd with a fixed upstream gradient.
    ``store_weights=False`` (big modules): weights are re-created in the test by oracle.fill with ``fill_seed``
    from the stored key/shape list, and parameter gradients are kept as L2 norms + 64 sampled values."""
    fill_state_dict(mod.state_dict(), seed)
    sd0 = {k: v.clone() for k, v in mod.state_dict().items()}
    mod.train()
    xs = [t.clone().requires_grad_(True) for t in inputs]
    out = mod(xs) if as_list else mod(*xs)
    gup = rs_tensor(seed + 77, out.shape)
    (out * gup).sum().backward()
    grads = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
    sd1 = {k: v.clone() for k, v in mod.state_dict().items() if "running" in k or "tracked" in k}
    if store_weights:
        res = dict(sd=sd0, out=out, gup=gup, grad=grads, sd_after=sd1)
    else:
        keys = list(sd0.keys())
        res = dict(out=out, gup=gup, sd_after=sd1, fill_seed=np.array(seed), sd_keys=np.array(keys),
                   sd_shapes=np.array([list(sd0[k].shape) + [0] * (4 - sd0[k].dim()) for k in keys]),
                   sd_ndim=np.array([sd0[k].dim() for k in keys]),
                   sd_isfloat=np.array([int(sd0[k].dtype.is_floating_point) for k in keys]),
                   grad_names=np.array(list(grads.keys())),
                   grad_norms=np.array([float(g.double().norm()) for g in grads.values()]),
                   grad_head=np.stack([g.flatten()[:64].numpy() if g.numel() >= 64 else
                                       np.pad(g.flatten().numpy(), (0, 64 - g.numel())) for g in grads.values()]))
    for i, t in enumerate(xs):
        res[f"x{i}"] = t
        if t.grad is not None:
            res[f"gx{i}"] = t.grad
    return res


# --------------------------------------------------------------------------------------
def gen_blocks_v5():
    ns = load_ref("unet-lite/yolo5-seg/seg_diceloss_yolov5.py", [(381, 750)])
    Conv, C3, SPPF, Concat = ns["Conv"], ns["C3"], ns["SPPF"], ns["Concat"]
    # Conv: k in {1,3,6}, s in {1,2}, act on/off
    for name, (c1, c2, k, s, p, act), hw in [
        ("conv_k1s1", (16, 24, 1, 1, None, True), 16),
        ("conv_k3s1", (16, 16, 3, 1, None, True), 16),
        ("conv_k3s2", (8, 32, 3, 2, None, True), 20),
        ("conv_k6s2", (3, 16, 6, 2, 2, True), 32),
        ("conv_k1s1_noact", (16, 12, 1, 1, None, False), 16),
    ]:
        m = Conv(c1, c2, k, s, p, 1, act)
        save("v5_" + name, meta=np.array([c1, c2, k, s, -1 if p is None else p, int(act)]),
             **run_module(m, [rs_tensor(1, (2, c1, hw, hw))], seed=10))
    for n in (0, 1, 2):
        for c1, c2 in ((16, 16), (24, 16)):
            m = C3(c1, c2, n)
            save(f"v5_c3_n{n}_{c1}_{c2}", meta=np.array([c1, c2, n]),
                 **run_module(m, [rs_tensor(2, (2, c1, 16, 16))], seed=20 + n))
    m = C3(16, 16, 1, False)
    save("v5_c3_n1_noshortcut", meta=np.array([16, 16, 1]), **run_module(m, [rs_tensor(2, (2, 16, 16, 16))], seed=29))
    m = SPPF(32, 32, 5)
    save("v5_sppf", meta=np.array([32, 32, 5]), **run_module(m, [rs_tensor(3, (2, 32, 20, 20))], seed=30))
    m = SPPF(16, 24, 5)
    save("v5_sppf_small", meta=np.array([16, 24, 5]), **run_module(m, [rs_tensor(3, (2, 16, 6, 6))], seed=31))
    # Concat with auto-align: up (10->20), down (40->20), mixed
    cat = Concat(1)
    for name, shapes in [("up", [(2, 8, 20, 20), (2, 16, 5, 5)]), ("down", [(2, 8, 10, 10), (2, 8, 40, 40)]),
                         ("same", [(2, 8, 12, 12), (2, 24, 12, 12)]), ("odd", [(2, 8, 20, 20), (2, 8, 7, 9)])]:
        xs = [rs_tensor(40 + i, s).requires_grad_(True) for i, s in enumerate(shapes)]
        out = cat(xs)
        gup = rs_tensor(49, out.shape)
        (out * gup).sum().backward()
        save("v5_concat_" + name, out=out, gup=gup, **{f"x{i}": x for i, x in enumerate(xs)},
             **{f"gx{i}": x.grad for i, x in enumerate(xs)})
    for sc in (2, 4):
        x = rs_tensor(50, (2, 8, 10, 10)).requires_grad_(True)
        out = nn.Upsample(scale_factor=float(sc), mode="nearest")(x)
        gup = rs_tensor(51, out.shape)
        (out * gup).sum().backward()
        save(f"v5_upsample_x{sc}", x0=x, out=out, gup=gup, gx0=x.grad)
    # bilinear resize to explicit size, both conventions (model tail / SegmentHead)
    for ac in (False, True):
        for name, (hin, hout) in (("up", ((7, 9), (20, 24))), ("down", ((24, 20), (10, 7)))):
            x = rs_tensor(52, (2, 6) + hin).requires_grad_(True)
            out = F.interpolate(x, size=hout, mode="bilinear", align_corners=ac)
            gup = rs_tensor(53, out.shape)
            (out * gup).sum().backward()
            save(f"bilinear_{name}_ac{int(ac)}", x0=x, out=out, gup=gup, gx0=x.grad)
    return ns


def gen_losses(ns5):
    ns8 = load_ref("yolov8/seg_jaccardloss_yolov8.py", [(366, 400), (755, 816)])
    ns18 = load_ref("unet-lite/Resnet18/seg_diceloss_resnet18.py", [(458, 505)])
    cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)  # weight.yaml:3-14
    cases = [
        ("dice_w", ns5["SegmentationLoss"], dict(num_classes=12, class_weights=cw), (2, 12, 16, 16), (2, 16, 16), False),
        ("dice_w_ls", ns5["SegmentationLoss"], dict(num_classes=12, label_smoothing=0.1, class_weights=cw), (2, 12, 16, 16), (2, 16, 16), False),
        ("dice_w_softmaxin", ns5["SegmentationLoss"], dict(num_classes=12, class_weights=cw), (2, 12, 16, 16), (2, 16, 16), True),
        ("dice_w_resize", ns5["SegmentationLoss"], dict(num_classes=12, class_weights=cw), (2, 12, 16, 16), (2, 8, 8), False),
        ("dice_w_resize_odd", ns5["SegmentationLoss"], dict(num_classes=12, class_weights=cw), (2, 12, 20, 12), (2, 7, 9), False),
        ("dice_unw", ns18["SegmentationLoss"], dict(num_classes=12), (2, 12, 16, 16), (2, 16, 16), False),
        ("dice_unw_ls", ns18["SegmentationLoss"], dict(num_classes=5, label_smoothing=0.1), (3, 5, 9, 11), (3, 9, 11), False),
        ("jaccard_w", ns8["SegmentationLoss"], dict(num_classes=12, class_weights=cw), (2, 12, 16, 16), (2, 16, 16), False),
        ("jaccard_w_ls", ns8["SegmentationLoss"], dict(num_classes=12, label_smoothing=0.1, class_weights=cw), (2, 12, 16, 16), (2, 16, 16), True),
    ]
    for name, cls, kw, ps, ts, softmax_in in cases:
        crit = cls(**kw)
        logits = (rs_tensor(60, ps) * 3.0).requires_grad_(True)
        pred = logits.softmax(1) if softmax_in else logits
        rs = np.random.RandomState(61)
        tgt = torch.from_numpy(rs.randint(0, ps[1], size=ts).astype(np.int64))
        total, items = crit(pred, tgt)
        total.backward()
        save("loss_" + name, logits=logits, target=tgt, total=total, items=np.array(items, dtype=np.float64),
             glogits=logits.grad, softmax_in=np.array(int(softmax_in)),
             cw=kw.get("class_weights", torch.zeros(0)), ls=np.array(kw.get("label_smoothing", 0.0)))


def gen_blocks_common():
    ns = load_ref("models/common.py", [(38, 64), (115, 125), (161, 172), (223, 238), (310, 317)])
    m = ns["Bottleneck"](16, 16, True)
    save("cm_bottleneck", **run_module(m, [rs_tensor(4, (2, 16, 12, 12))], seed=70))
    for n in (1, 2):
        m = ns["C3"](16, 24, n)
        save(f"cm_c3_n{n}", meta=np.array([16, 24, n]), **run_module(m, [rs_tensor(4, (2, 16, 12, 12))], seed=71 + n))
    m = ns["SPPF"](16, 16, 5)
    save("cm_sppf", **run_module(m, [rs_tensor(4, (2, 16, 12, 12))], seed=75))


def gen_blocks_v8():
    ns = load_ref("yolov8/seg_jaccardloss_yolov8.py", [(366, 414)])
    for n in (0, 1, 2):
        for c1, c2 in ((16, 16), (24, 16)):
            m = ns["C2f"](c1, c2, n)
            save(f"v8_c2f_n{n}_{c1}_{c2}", meta=np.array([c1, c2, n]),
                 **run_module(m, [rs_tensor(5, (2, c1, 16, 16))], seed=80 + n))


def gen_blocks_v9():
    ns = load_ref("unet-lite/yolo9-seg/seg_diceloss_yolov9.py", [(84, 90), (413, 510)])
    m = ns["C3k2"](16, 16, 1)
    save("v9_c3k2", meta=np.array([16, 16, 1]), **run_module(m, [rs_tensor(6, (2, 16, 12, 12))], seed=90))
    m = ns["GAM"](32)
    save("v9_gam", meta=np.array([32]), **run_module(m, [rs_tensor(6, (2, 32, 10, 10))], seed=91))


def gen_resnet():
    ns18 = load_ref("unet-lite/Resnet18/seg_diceloss_resnet18.py", [(191, 505)])
    ns50 = load_ref("unet-lite/Resnet50/seg_jaccardloss_Resnet50.py", [(175, 498)])
    BB, BN = ns18["BasicBlock"], ns50["BottleneckBlock"]
    m = BB(16, 16, 1, None)
    save("r18_basic", meta=np.array([16, 16, 1]), **run_module(m, [rs_tensor(7, (2, 16, 12, 12))], seed=100))
    m = BB(16, 32, 2, ns18["Conv"](16, 32, 1, 2, 0, act=False))
    save("r18_basic_down", meta=np.array([16, 32, 2]), **run_module(m, [rs_tensor(7, (2, 16, 12, 12))], seed=101))
    m = BN(32, 8, 1, None)
    save("r50_bneck", meta=np.array([32, 8, 1]), **run_module(m, [rs_tensor(7, (2, 32, 12, 12))], seed=102))
    m = BN(16, 8, 2, ns50["Conv"](16, 32, 1, 2, 0, act=False))
    save("r50_bneck_down", meta=np.array([16, 8, 2]), **run_module(m, [rs_tensor(7, (2, 16, 12, 12))], seed=103))
    # stem: 7x7/s2 conv + maxpool 3/s2
    stem = nn.Sequential(ns18["Conv"](3, 16, 7, 2, 3), nn.MaxPool2d(3, 2, 1))
    save("r18_stem", **run_module(stem, [rs_tensor(7, (2, 3, 36, 36))], seed=104))
    # SegmentHead, 3 scales
    hd = ns18["SegmentHead"](num_classes=12, in_channels=[16, 24, 32])
    feats = [rs_tensor(8, (2, 16, 16, 16)), rs_tensor(9, (2, 24, 8, 8)), rs_tensor(10, (2, 32, 4, 4))]
    save("seghead", **run_module(hd, feats, seed=105, as_list=True, store_weights=False))
    return ns18, ns50


def _model_fixture(name, model, loss_fn, x, tgt, lr=0.01, steps=2):
    """Whole-model fixture: weights are NOT stored (filled by oracle.fill with a fixed seed); we keep the
    output at sampled positions, per-parameter gradient norms, the grad-None list and post-step checksums."""
    fill_state_dict(model.state_dict(), 1234, bn_stats=False)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.937, nesterov=True)
    rec = {}
    rs = np.random.RandomState(5)
    for st in range(steps):
        opt.zero_grad(set_to_none=True)
        out = model(x)
        total, items = loss_fn(out, tgt)
        total.backward()
        if st == 0:
            flat = out.detach().flatten()
            idx = torch.from_numpy(rs.randint(0, flat.numel(), size=256).astype(np.int64))
            rec["out_idx"] = idx
            rec["out_vals"] = flat[idx]
            rec["out_shape"] = np.array(out.shape)
            rec["out_mean"] = out.detach().double().mean()
            names = [k for k, _ in model.named_parameters()]
            rec["grad_none"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
            rec["grad_names"] = np.array([k for k, p in model.named_parameters() if p.grad is not None])
            rec["grad_norms"] = np.array([float(p.grad.double().norm()) for k, p in model.named_parameters() if p.grad is not None])
            rec["param_names"] = np.array(names)
        rec[f"loss_items_{st}"] = np.array(items, dtype=np.float64)
        opt.step()
    sd = model.state_dict()
    keys = [k for k in sd if sd[k].dtype.is_floating_point]
    rec["final_keys"] = np.array(keys)
    rec["final_sums"] = np.array([float(sd[k].double().sum()) for k in keys])
    rec["final_abs"] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    save(name, x=x, target=tgt, **rec)


def gen_models(ns5, ns18):
    cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
    S = 64
    x = rs_tensor(200, (2, 3, S, S)).abs().clamp(0, 1)
    tgt = torch.from_numpy(np.random.RandomState(201).randint(0, 12, size=(2, S, S)).astype(np.int64))
    # YOLOv5Seg with C3_DCN -> C3 (BASELINE config 2)
    cfg = yaml.safe_load(open(os.path.join(REF, "unet-lite/yolo5-seg/yolov5_seg.yaml")))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            if l[2] == "C3_DCN":
                l[2] = "C3"
    torch.manual_seed(0)
    m = ns5["YOLOv5Seg"](cfg)
    m.img_size = [S, S]
    _model_fixture("model_yolov5seg_64", m, ns5["SegmentationLoss"](12, 0.0, cw), x, tgt)
    # YOLOv8Seg with C2f_DCN -> C2f (config 4)
    ns8 = load_ref("yolov8/seg_jaccardloss_yolov8.py", [(366, 816)])
    cfg8 = yaml.safe_load(open(os.path.join(REF, "yolov8/yolov8.yaml")))
    for sec in ("backbone", "head"):
        for l in cfg8[sec]:
            if l[2] == "C2f_DCN":
                l[2] = "C2f"
    m8 = ns8["YOLOv8Seg"](cfg8)
    m8.img_size = [S, S]
    _model_fixture("model_yolov8seg_64", m8, ns8["SegmentationLoss"](12, 0.0, cw), x, tgt)
    # ResNet18Seg (config 1)
    m18 = ns18["ResNet18Seg"]({"nc": 12})
    _model_fixture("model_resnet18seg_64", m18, ns18["SegmentationLoss"](12, 0.0), x, tgt)


def gen_models_more():
    """ResNet50 + SegmentHead (BASELINE config 3, segment/train.py) and YOLOv9Seg (config 5; yaml `GAM [512]` cannot be
    built by the reference itself — GAM(c1, 512) — so the fixture uses `GAM []`, like the module's own signature)"""
    cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
    S = 64
    x = rs_tensor(200, (2, 3, S, S)).abs().clamp(0, 1)
    tgt = torch.from_numpy(np.random.RandomState(201).randint(0, 12, size=(2, S, S)).astype(np.int64))
    nst = load_ref("segment/train.py", [(50, 338)])
    torch.manual_seed(0)
    m50 = nst["ResNet50Seg"]({"nc": 12}) if "cfg" in nst["ResNet50Seg"].__init__.__code__.co_varnames else nst["ResNet50Seg"](12)
    _model_fixture("model_resnet50seg_64", m50, nst["SegmentationLoss"](12, 0.0), x, tgt)
    ns9 = load_ref("unet-lite/yolo9-seg/seg_diceloss_yolov9.py", [(84, 90), (413, 880)])
    cfg9 = yaml.safe_load(open(os.path.join(REF, "unet-lite/yolo9-seg/yolov9_seg.yaml")))
    for sec in ("backbone", "head"):
        for l in cfg9[sec]:
            if l[2] == "GAM":
                l[3] = []
    m9 = ns9["YOLOv9Seg"](cfg9)
    m9.img_size = [S, S]
    _model_fixture("model_yolov9seg_64", m9, ns9["SegmentationLoss"](12, 0.0, cw), x, tgt)


def gen_resnet50_yaml():
    """the yaml-driven ResNet50 + UNet-lite head (unet-lite/Resnet50/seg_diceloss_Resnet50.py:382-790 with resnet50.yaml):
    2-step training trajectory like the other whole-model fixtures"""
    ns = load_ref("unet-lite/Resnet50/seg_diceloss_Resnet50.py", [(60, 86), (382, 790)])
    cfg = yaml.safe_load(open(os.path.join(REF, "unet-lite/Resnet50/resnet50.yaml")))
    cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
    S = 64
    x = rs_tensor(200, (2, 3, S, S)).abs().clamp(0, 1)
    tgt = torch.from_numpy(np.random.RandomState(201).randint(0, 12, size=(2, S, S)).astype(np.int64))
    m = ns["ResNet50Seg"](cfg)
    m.img_size = [S, S]
    _model_fixture("model_resnet50yaml_64", m, ns["SegmentationLoss"](12, 0.0, cw), x, tgt)


def gen_dcnv3():
    ns = load_ref("models/ops_dcnv3/build/lib.linux-x86_64-cpython-38/functions/dcnv3_func.py", [(92, 189)],
                  extra=dict(DCNv3=None))
    core = ns["dcnv3_core_pytorch"]
    # models/ops_dcnv3/test.py:19-39 shapes: N=2, M=4 groups, H=W=8, K=3, offset_scale=2, pad=1
    N, M, H, K, pad = 2, 4, 8, 3, 1
    P = K * K
    for D in (1, 16, 30, 32, 64, 71, 1025):          # test.py:257-260
        rs = np.random.RandomState(300 + D)
        inp = torch.from_numpy(rs.rand(N, H, H, M * D).astype(np.float32) * 0.01).requires_grad_(True)
        off = torch.from_numpy(rs.rand(N, H, H, M * P * 2).astype(np.float32) * 10).requires_grad_(True)
        msk = torch.from_numpy(rs.rand(N, H, H, M, P).astype(np.float32) + 1e-5)
        msk = (msk / msk.sum(-1, keepdim=True)).reshape(N, H, H, M * P).requires_grad_(True)
        out = core(inp, off, msk, K, K, 1, 1, pad, pad, 1, 1, M, D, 2.0)
        out.sum().backward()   # test.py:123 uses sum() as the scalar
        save(f"dcnv3_D{D}", inp=inp, off=off, msk=msk, out=out, ginp=inp.grad, goff=off.grad, gmsk=msk.grad,
             meta=np.array([K, K, 1, 1, pad, pad, 1, 1, M, D]), offset_scale=np.array(2.0))
    # strided / dilated variant
    rs = np.random.RandomState(399)
    D, Hi, Ho = 8, 11, 5
    inp = torch.from_numpy(rs.randn(N, Hi, Hi, M * D).astype(np.float32)).requires_grad_(True)
    off = torch.from_numpy(rs.randn(N, Ho, Ho, M * P * 2).astype(np.float32)).requires_grad_(True)
    msk = torch.from_numpy(rs.rand(N, Ho, Ho, M * P).astype(np.float32)).requires_grad_(True)
    out = core(inp, off, msk, K, K, 2, 2, 1, 1, 2, 2, M, D, 1.0)
    gup = rs_tensor(398, out.shape)
    (out * gup).sum().backward()
    save("dcnv3_s2d2", inp=inp, off=off, msk=msk, out=out, gup=gup, ginp=inp.grad, goff=off.grad, gmsk=msk.grad,
         meta=np.array([K, K, 2, 2, 1, 1, 2, 2, M, D]), offset_scale=np.array(1.0))


def gen_dcnv3_tile():
    """a larger, odd-sized case (2 groups of 128 channels, 19 x 13, O(1) inputs): offsets mostly within 1.5 pixels, 5 % of them up to
    9 pixels away (many samples leave the image), random upstream gradient"""
    nsf = load_ref("models/ops_dcnv3/build/lib.linux-x86_64-cpython-38/functions/dcnv3_func.py", [(92, 189)], extra=dict(DCNv3=None))
    core = nsf["dcnv3_core_pytorch"]
    rs = np.random.RandomState(397)
    N, M, H, W, D, K = 2, 2, 19, 13, 128, 3
    P = K * K
    inp = torch.from_numpy(rs.randn(N, H, W, M * D).astype(np.float32)).requires_grad_(True)
    o = (rs.rand(N, H, W, M * P * 2).astype(np.float32) - 0.5) * 3.0
    far = rs.rand(*o.shape) < 0.05
    o[far] *= 6.0
    off = torch.from_numpy(o).requires_grad_(True)
    mk = rs.rand(N, H, W, M, P).astype(np.float32) + 1e-3
    msk = torch.from_numpy((mk / mk.sum(-1, keepdims=True)).reshape(N, H, W, M * P)).requires_grad_(True)
    out = core(inp, off, msk, K, K, 1, 1, 1, 1, 1, 1, M, D, 1.0)
    gup = rs_tensor(396, out.shape)
    (out * gup).sum().backward()
    save("dcnv3_D128_19x13", inp=inp, off=off, msk=msk, out=out, gup=gup, ginp=inp.grad, goff=off.grad, gmsk=msk.grad,
         meta=np.array([K, K, 1, 1, 1, 1, 1, 1, M, D]), offset_scale=np.array(1.0))


def gen_dcnv3_module():
    """the DCNv3 *module* and its YOLO wiring, from the reference's own classes (modules/dcnv3.py:27-136 with the module's
    relative import replaced by a shim whose ``apply`` is the reference's pure-PyTorch core; "common and yolo.py":2-38)"""
    nsf = load_ref("models/ops_dcnv3/build/lib.linux-x86_64-cpython-38/functions/dcnv3_func.py", [(92, 189)], extra=dict(DCNv3=None))
    core = nsf["dcnv3_core_pytorch"]

    class _Fn:                      # DCNv3Function.apply(input, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, G, Gc, scale, im2col_step)
        @staticmethod
        def apply(*a):
            return core(*a[:14])
    from torch.nn.init import constant_, xavier_uniform_
    nsm = load_ref("models/ops_dcnv3/build/lib.linux-x86_64-cpython-38/modules/dcnv3.py", [(17, 136)],
                   extra=dict(DCNv3Function=_Fn, dcnv3_core_pytorch=core, xavier_uniform_=xavier_uniform_, constant_=constant_))
    DCNv3 = nsm["DCNv3"]
    nsc = load_ref("models/common.py", [(38, 64)])
    nsy = load_ref("models/ops_dcnv3/common and yolo.py", [(2, 38)], extra=dict(Conv=nsc["Conv"], DCNv3=DCNv3))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DCNv3(channels=32, kernel_size=3, stride=1, pad=1, group=4)
        save("dcnmod_c32_g4", meta=np.array([32, 3, 1, 1, 4]), **run_module(m, [rs_tensor(701, (2, 12, 12, 32))], seed=700))
        m = DCNv3(channels=24, kernel_size=3, stride=1, pad=1, group=1)
        save("dcnmod_c24_g1", meta=np.array([24, 3, 1, 1, 1]), **run_module(m, [rs_tensor(703, (2, 9, 11, 24))], seed=702))
        m = nsy["C3_DCNV3"](32, 32, 1)
        save("c3_dcnv3_n1", meta=np.array([32, 32, 1]), **run_module(m, [rs_tensor(705, (2, 32, 12, 12))], seed=704))
        m = nsy["C3_DCNV3"](24, 32, 2, False)
        save("c3_dcnv3_n2_noshortcut", meta=np.array([24, 32, 2]), **run_module(m, [rs_tensor(707, (2, 24, 10, 10))], seed=706))


def gen_miou():
    ns = load_ref("unet-lite/yolo5-seg/val_diceloss.py", [(37, 75)])
    rs = np.random.RandomState(500)
    pred = rs.randint(0, 12, size=(2, 24, 24))
    tgt = rs.randint(0, 12, size=(2, 24, 24))
    cm = ns["SegmentationConfusionMatrix"](12, ignore_index=11)
    cm.process_batch(pred, tgt)
    miou, ious = cm.compute_iou()
    save("miou", pred=pred, target=tgt, matrix=cm.matrix, miou=np.array(miou), ious=np.array(ious))


def gen_optim():
    """smart_optimizer SGD-nesterov groups + ModelEMA (utils/torch_utils.py:318-346, 404-428) on a tiny net."""
    ns = load_ref("utils/torch_utils.py", [(318, 346), (404, 428)],
                  extra=dict(colorstr=lambda *a: "", de_parallel=lambda m: m, copy_attr=None))
    nsb = load_ref("unet-lite/yolo5-seg/seg_diceloss_yolov5.py", [(381, 413)])
    net = nn.Sequential(nsb["Conv"](4, 8, 3, 1), nsb["Conv"](8, 4, 1, 1))
    fill_state_dict(net.state_dict(), 600)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    opt = ns["smart_optimizer"](net, "SGD", 0.01, 0.937, 5e-4)
    ema = ns["ModelEMA"](net)
    x = rs_tensor(601, (2, 4, 8, 8))
    for st in range(3):
        opt.zero_grad()
        net(x).square().mean().backward()
        if st == 0:
            g0 = {k: p.grad.clone() for k, p in net.named_parameters()}
        opt.step()
        ema.update(net)
    save("optim_sgd_ema", x0=x, sd=sd0, g0=g0, sd_after=dict(net.state_dict()), ema_after=dict(ema.ema.state_dict()),
         hyp=np.array([0.01, 0.937, 5e-4]), steps=np.array(3))


def gen_letterbox():
    """per-sample input preparation: the reference's own `_resize_and_pad` method (unet-lite/yolo5-seg/
    seg_diceloss_yolov5.py:320-349, exec'd inside a shim class) on Pillow images, then the conversion of __getitem__ (:315-316)"""
    from PIL import Image
    path = os.path.join(REF, "unet-lite/yolo5-seg/seg_diceloss_yolov5.py")
    lines = open(path, encoding="utf-8").read().split("\n")
    ns = dict(Image=Image, np=np, torch=torch)
    exec(compile("class _DS:\n" + "\n".join(lines[320 - 1:349]), path, "exec"), ns)
    for i, (w, h, S) in enumerate([(100, 37, 64), (37, 100, 64), (50, 50, 96), (333, 517, 128), (129, 96, 128), (96, 96, 96)]):
        rs = np.random.RandomState(700 + i)
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        mask = rs.randint(0, 16, size=(h, w)).astype(np.uint8)            # labels above 11 exercise the clip of :303
        ds = ns["_DS"]()
        ds.img_size = S
        mclip = np.clip(mask.astype(np.int64), 0, 11)
        pi, pm = ds._resize_and_pad(Image.fromarray(img).convert("RGB"), Image.fromarray(mclip.astype(np.uint8)))
        out_i = torch.from_numpy(np.array(pi)).permute(2, 0, 1).float() / 255.0
        out_m = torch.from_numpy(np.array(pm)).long()
        save(f"letterbox_{w}x{h}_{S}", img=img, mask=mask, out_img=out_i, out_mask=out_m, meta=np.array([w, h, S, 12]))


if __name__ == "__main__":
    torch.set_num_threads(4)
    if "--letterbox" in sys.argv:
        gen_letterbox()
        sys.exit(0)
    if "--models-more" in sys.argv:
        gen_models_more()
        sys.exit(0)
    if "--r50yaml" in sys.argv:
        gen_resnet50_yaml()
        sys.exit(0)
    if "--dcnv3-tile" in sys.argv:
        gen_dcnv3_tile()
        sys.exit(0)
    if "--dcnv3" in sys.argv:
        gen_dcnv3()
        gen_dcnv3_tile()
        gen_dcnv3_module()
        sys.exit(0)
    ns5 = gen_blocks_v5()
    gen_losses(ns5)
    gen_blocks_common()
    gen_blocks_v8()
    gen_blocks_v9()
    ns18, ns50 = gen_resnet()
    gen_dcnv3()
    gen_dcnv3_tile()
    gen_dcnv3_module()
    gen_miou()
    gen_optim()
    gen_models(ns5, ns18)
    gen_models_more()
    gen_resnet50_yaml()
    gen_letterbox()
