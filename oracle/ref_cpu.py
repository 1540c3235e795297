"""CPU oracle: plain-torch fp32 restatement of the reference's segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``yolo_dual_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only
as the checker / the reported CPU baseline.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against
golden vectors produced by running the reference's own classes in the build container
(``oracle/make_golden.py`` execs the reference line ranges listed in SURVEY.md §8c and stores
inputs, weights, outputs and gradients under ``tests/golden/``).

The restatement is *functional*: parameters live in a flat ``dict`` keyed by the reference's
``state_dict`` names (``cv1.conv.weight``, ``cv1.bn.running_mean`` ...), so a golden fixture's
weights feed it directly.  Citations are file:line into the reference repository.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

BN_EPS = 1e-5       # torch default; the seg scripts never call initialize_weights
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def autopad(k: int, p: Optional[int] = None) -> int:
    """unet-lite/yolo5-seg/seg_diceloss_yolov5.py:381-385, models/common.py:38-44."""
    return k // 2 if p is None else p


def _act(x: torch.Tensor, act: str) -> torch.Tensor:
    if act == "silu":
        return F.silu(x)
    if act == "relu":
        return F.relu(x)
    if act == "none":
        return x
    raise ValueError(act)


def conv_bn_act(sd: SD, pre: str, x: torch.Tensor, s: int = 1, p: Optional[int] = None,
                act: str = "silu", train: bool = True) -> torch.Tensor:
    """``act(bn(conv(x)))`` — seg_diceloss_yolov5.py:388-409 (SiLU), Resnet50/seg_diceloss_Resnet50.py:389-402
    (ReLU variant).  Kernel size is read off the weight; bias-free conv; train-mode BN updates the
    running statistics in ``sd`` in place (momentum 0.1, unbiased variance) like nn.BatchNorm2d."""
    w = sd[pre + ".conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride=s, padding=autopad(k, p))
    rm, rv = sd[pre + ".bn.running_mean"], sd[pre + ".bn.running_var"]
    if train and (pre + ".bn.num_batches_tracked") in sd:
        sd[pre + ".bn.num_batches_tracked"] += 1
    y = F.batch_norm(y, rm, rv, sd[pre + ".bn.weight"], sd[pre + ".bn.bias"], train, BN_MOMENTUM, BN_EPS)
    return _act(y, act)


def c3_script(sd: SD, pre: str, x: torch.Tensor, n: int, add: bool, act: str = "silu",
              train: bool = True) -> torch.Tensor:
    """Seg-script C3: cv3(cat(m(cv1 x), cv2 x)) (+x) with m = n plain 3x3 Convs, outer residual —
    seg_diceloss_yolov5.py:416-428 (trap T2: differs from models/common.py C3)."""
    a = conv_bn_act(sd, pre + ".cv1", x, act=act, train=train)
    for i in range(int(n)):
        a = conv_bn_act(sd, f"{pre}.m.{i}", a, act=act, train=train)
    b = conv_bn_act(sd, pre + ".cv2", x, act=act, train=train)
    y = conv_bn_act(sd, pre + ".cv3", torch.cat((a, b), 1), act=act, train=train)
    return y + x if add else y


def c3k2(sd: SD, pre: str, x: torch.Tensor, n: int, add: bool, train: bool = True) -> torch.Tensor:
    """yolo9 C3k2 = script C3 + crop-align of the two branches — seg_diceloss_yolov9.py:451-472."""
    a = conv_bn_act(sd, pre + ".cv1", x, train=train)
    for i in range(int(n)):
        a = conv_bn_act(sd, f"{pre}.m.{i}", a, train=train)
    b = conv_bn_act(sd, pre + ".cv2", x, train=train)
    if a.shape[2:] != b.shape[2:]:
        h, w = min(a.shape[2], b.shape[2]), min(a.shape[3], b.shape[3])
        a, b = a[:, :, :h, :w], b[:, :, :h, :w]
    y = conv_bn_act(sd, pre + ".cv3", torch.cat((a, b), 1), train=train)
    return y + x if add else y


def bottleneck(sd: SD, pre: str, x: torch.Tensor, add: bool, train: bool = True) -> torch.Tensor:
    """models/common.py:115-125: x + cv2(cv1 x), cv1 1x1, cv2 3x3."""
    y = conv_bn_act(sd, pre + ".cv2", conv_bn_act(sd, pre + ".cv1", x, train=train), train=train)
    return x + y if add else y


def c3_common(sd: SD, pre: str, x: torch.Tensor, n: int, shortcut: bool, train: bool = True) -> torch.Tensor:
    """models/common.py:161-172: cv3(cat(m(cv1 x), cv2 x)), m = n Bottlenecks (e=1.0), no outer residual."""
    a = conv_bn_act(sd, pre + ".cv1", x, train=train)
    for i in range(int(n)):
        a = bottleneck(sd, f"{pre}.m.{i}", a, shortcut, train=train)   # c1 == c2 == c_ inside C3
    b = conv_bn_act(sd, pre + ".cv2", x, train=train)
    return conv_bn_act(sd, pre + ".cv3", torch.cat((a, b), 1), train=train)


def c2f(sd: SD, pre: str, x: torch.Tensor, n: int, add: bool, train: bool = True) -> torch.Tensor:
    """yolov8/seg_jaccardloss_yolov8.py:401-414: cv1 -> chunk(2) -> n chained 3x3 on the last -> cv2 (+x)."""
    y = list(conv_bn_act(sd, pre + ".cv1", x, train=train).chunk(2, 1))
    for i in range(int(n)):
        y.append(conv_bn_act(sd, f"{pre}.m.{i}", y[-1], train=train))
    out = conv_bn_act(sd, pre + ".cv2", torch.cat(y, 1), train=train)
    return out + x if add else out


def sppf(sd: SD, pre: str, x: torch.Tensor, k: int = 5, act: str = "silu", train: bool = True) -> torch.Tensor:
    """seg_diceloss_yolov5.py:468-481 / models/common.py:223-238."""
    x = conv_bn_act(sd, pre + ".cv1", x, act=act, train=train)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_bn_act(sd, pre + ".cv2", torch.cat((x, y1, y2, y3), 1), act=act, train=train)


def concat_align(xs: Sequence[torch.Tensor], dim: int = 1) -> torch.Tensor:
    """Auto-aligning Concat — seg_diceloss_yolov5.py:484-507: every input whose HxW differs from the
    first is bilinearly resized (align_corners=False; up OR down) before torch.cat."""
    if len(xs) == 1:
        return xs[0]
    tgt = xs[0].shape[2:]
    ys = [t if t.shape[2:] == tgt else F.interpolate(t, size=tgt, mode="bilinear", align_corners=False)
          for t in xs]
    return torch.cat(ys, dim)


def upsample_nearest(x: torch.Tensor, scale: float) -> torch.Tensor:
    """nn.Upsample(scale_factor=scale, mode='nearest') as built at seg_diceloss_yolov5.py:588-609."""
    return F.interpolate(x, scale_factor=float(scale), mode="nearest")


def gam(sd: SD, pre: str, x: torch.Tensor, train: bool = True) -> torch.Tensor:
    """yolo9 GAM — seg_diceloss_yolov9.py:475-510.  conv1 runs twice (BN stats update twice)."""
    h, w = x.shape[2:]
    y1 = conv_bn_act(sd, pre + ".conv1", x, train=train)
    y1 = conv_bn_act(sd, pre + ".conv2", F.adaptive_avg_pool2d(y1, 1), act="none", train=train)
    y2 = conv_bn_act(sd, pre + ".conv1", x, train=train)
    y2 = conv_bn_act(sd, pre + ".conv3", F.adaptive_max_pool2d(y2, 1), act="none", train=train)
    y = torch.sigmoid(y1 + y2)
    y = F.interpolate(y, size=(h, w), mode="bilinear", align_corners=False)
    return x * y


# --------------------------------------------------------------------------------------
# ResNet backbones + multi-scale SegmentHead
# --------------------------------------------------------------------------------------
def basic_block(sd: SD, pre: str, x: torch.Tensor, stride: int, train: bool = True) -> torch.Tensor:
    """Resnet18/seg_diceloss_resnet18.py:216-240: conv1 3x3(s)+SiLU, conv2 3x3 no act, ReLU(out+identity)."""
    out = conv_bn_act(sd, pre + ".conv1", x, s=stride, p=1, train=train)
    out = conv_bn_act(sd, pre + ".conv2", out, p=1, act="none", train=train)
    idt = x
    if (pre + ".downsample.conv.weight") in sd:
        idt = conv_bn_act(sd, pre + ".downsample", x, s=stride, p=0, act="none", train=train)
    return F.relu(out + idt)


def bottleneck_block(sd: SD, pre: str, x: torch.Tensor, stride: int, train: bool = True) -> torch.Tensor:
    """segment/train.py:74-100 / Resnet50/seg_jaccardloss_Resnet50.py:199-225."""
    out = conv_bn_act(sd, pre + ".conv1", x, p=0, train=train)
    out = conv_bn_act(sd, pre + ".conv2", out, s=stride, p=1, train=train)
    out = conv_bn_act(sd, pre + ".conv3", out, p=0, act="none", train=train)
    idt = x
    if (pre + ".downsample.conv.weight") in sd:
        idt = conv_bn_act(sd, pre + ".downsample", x, s=stride, p=0, act="none", train=train)
    return F.relu(out + idt)


def resnet_backbone(sd: SD, pre: str, x: torch.Tensor, blocks: Sequence[int], kind: str,
                    train: bool = True) -> List[torch.Tensor]:
    """Stem 7x7/s2 + maxpool3/s2, then layer1..3 (layer4 is built by the reference but never run) —
    Resnet18:243-297, segment/train.py:103-156."""
    x = conv_bn_act(sd, pre + ".stem.0", x, s=2, p=3, train=train)
    x = F.max_pool2d(x, 3, 2, 1)
    blk = basic_block if kind == "basic" else bottleneck_block
    feats = []
    for li, nb in enumerate(blocks[:3]):
        for bi in range(nb):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = blk(sd, f"{pre}.layer{li + 1}.{bi}", x, stride, train=train)
        feats.append(x)
    return feats


def segment_head(sd: SD, pre: str, feats: Sequence[torch.Tensor], train: bool = True) -> torch.Tensor:
    """segment/train.py:159-210 / Resnet18:300-349: lateral 1x1 -> 128, bilinear(align_corners=True) up by
    2**i (F.interpolate fallback to the exact size), cat, 3x3 -> 256, 1x1 -> nc (no act)."""
    tgt = feats[0].shape[2:]
    outs = []
    for i, f in enumerate(feats):
        f = conv_bn_act(sd, f"{pre}.lateral_convs.{i}", f, train=train)
        if f.shape[2:] != tgt:
            f = F.interpolate(f, scale_factor=float(2 ** i), mode="bilinear", align_corners=True)
            if f.shape[2:] != tgt:
                f = F.interpolate(f, size=tgt, mode="bilinear", align_corners=True)
        outs.append(f)
    y = conv_bn_act(sd, pre + ".final_conv.0", torch.cat(outs, 1), train=train)
    return conv_bn_act(sd, pre + ".final_conv.1", y, act="none", train=train)


def resnet_seg_forward(sd: SD, x: torch.Tensor, kind: str, train: bool = True, out_size=None) -> torch.Tensor:
    """ResNet18Seg._forward_once — Resnet18:389-403: backbone, head, bilinear(align_corners=False) to the input size.
    ``out_size``: segment/train.py's SegmentHead.forward (:208-209) instead ends with a hard-coded
    ``F.interpolate(size=(640, 640))`` whatever the input size (ResNet50Seg, BASELINE config 3): pass (640, 640)."""
    blocks = (2, 2, 2, 2) if kind == "basic" else (3, 4, 6, 3)
    feats = resnet_backbone(sd, "backbone", x, blocks, kind, train=train)
    out = segment_head(sd, "head", feats, train=train)
    size = tuple(out_size) if out_size is not None else tuple(x.shape[2:])
    if tuple(out.shape[2:]) != size or out_size is not None:
        out = F.interpolate(out, size=size, mode="bilinear", align_corners=False)
    return out


# --------------------------------------------------------------------------------------
# yaml-driven script models (YOLOv5Seg / YOLOv8Seg / YOLOv9Seg)
# --------------------------------------------------------------------------------------
def upsample_spec(args, family: str = "v5"):
    """How the script builders turn a yaml ``nn.Upsample [size, scale, mode]`` row into nn.Upsample kwargs.
    The yaml writes ``None`` (a YAML *string*, not null) for size:
    * v5 / v9 builders (seg_diceloss_yolov5.py:588-609, seg_diceloss_yolov9.py:692-717): int("None") raises,
      size stays None, ``scale_factor=float(scale)`` is used;
    * v8 builder (yolov8/seg_jaccardloss_yolov8.py:583-660): a non-numeric size falls through to its
      "final fallback" ``size=(256, 256)`` and the scale factor is dropped (trap T10, found by the golden
      whole-model fixture): every Upsample of YOLOv8Seg emits 256x256."""
    size_arg = args[0] if len(args) >= 1 else None
    scale_arg = args[1] if len(args) >= 2 else 2.0
    if family == "v8" and len(args) < 3:
        size_arg, scale_arg = None, 2.0
    size = None
    if size_arg is not None:
        if isinstance(size_arg, (list, tuple)):
            size = tuple(int(v) for v in size_arg[-2:])
        else:
            try:
                size = (int(size_arg), int(size_arg))
            except (TypeError, ValueError):
                size = (256, 256) if family == "v8" else None
    if size is not None:
        return ("size", size)
    return ("scale", float(scale_arg) if isinstance(scale_arg, (int, float)) else 2.0)


def script_model_forward(sd: SD, cfg: dict, x: torch.Tensor, img_size: Tuple[int, int] = (640, 640),
                         train: bool = True, act: str = "silu", family: str = "v5") -> torch.Tensor:
    """YOLOv5Seg._forward_once with the builder's quirks — seg_diceloss_yolov5.py:537-659:
    * ``Conv(c1, *args)``, ``C3(c1, *args)``: the yaml ``number`` column and the multiples are ignored (T3),
      so ``C3 [512, False]`` binds n=False=0;
    * head ``from`` indices are absolute into backbone_outs+head_outs (T4);
    * every layer output is kept; the result is bilinearly resized to ``img_size`` (T7)."""
    outs: List[torch.Tensor] = []

    def run(kind: str, pre: str, inp, args):
        if kind == "Conv":
            k = args[1] if len(args) > 1 else 1
            s = args[2] if len(args) > 2 else 1
            p = args[3] if len(args) > 3 else None
            a = "none" if (len(args) > 5 and not args[5]) else act
            return conv_bn_act(sd, pre, inp, s=s, p=p, act=a, train=train)
        if kind in ("C3", "C3k2", "C2f"):
            c2 = args[0]
            n = int(args[1]) if len(args) > 1 else 1
            shortcut = args[2] if len(args) > 2 else True
            add = bool(shortcut) and inp.shape[1] == c2
            if kind == "C3":
                return c3_script(sd, pre, inp, n, add, act=act, train=train)
            if kind == "C3k2":
                return c3k2(sd, pre, inp, n, add, train=train)
            return c2f(sd, pre, inp, n, add, train=train)
        if kind == "SPPF":
            return sppf(sd, pre, inp, args[1] if len(args) > 1 else 5, act=act, train=train)
        if kind == "C3_DCNV3":            # C3_DCNV3(c1, c2, n=1, shortcut=True, g=1) — "common and yolo.py":27-38
            return c3_dcnv3(sd, pre, inp, int(args[1]) if len(args) > 1 else 1, bool(args[2]) if len(args) > 2 else True,
                            int(args[3]) if len(args) > 3 else 1, train=train)
        if kind == "GAM":
            return gam(sd, pre, inp, train=train)
        if kind in ("Upsample", "nn.Upsample"):
            how, val = upsample_spec(args, family)
            if how == "size":
                return F.interpolate(inp, size=val, mode="nearest")
            return upsample_nearest(inp, val)
        if kind == "Concat":
            return concat_align(inp, args[0] if args else 1)
        if kind == "nn.Softmax":
            return torch.softmax(inp, args[0] if args else 1)
        raise NotImplementedError(kind)

    for i, (frm, _num, kind, args) in enumerate(cfg["backbone"]):
        inp = x if frm == -1 else outs[frm]
        x = run(kind, f"backbone.{i}", inp, args)
        outs.append(x)
    for i, (frm, _num, kind, args) in enumerate(cfg["head"]):
        inp = [outs[f] for f in frm] if isinstance(frm, list) else outs[frm]
        x = run(kind, f"head.{i}", inp, args)
        outs.append(x)
    if tuple(x.shape[2:]) != tuple(img_size):
        x = F.interpolate(x, size=tuple(img_size), mode="bilinear", align_corners=False)
    return x


def resnet50_yaml_forward(sd: SD, cfg: dict, x: torch.Tensor, img_size: Tuple[int, int] = (640, 640), train: bool = True) -> torch.Tensor:
    """the yaml-driven ``ResNet50Seg`` of unet-lite/Resnet50/seg_diceloss_Resnet50.py:539-710 (resnet50.yaml): ReLU ``Conv``
    (:389-402), torchvision-style bottlenecks (:405-435), ResNetStem (:438-448), ResNet50Layer (:451-470), a C3 without
    residual (:522-535); the builder casts its yaml arguments (``C3 [512, False]`` -> n = 0) and the head's ``from`` is absolute."""
    outs: List[torch.Tensor] = []

    def conv(pre, inp, s=1, p=None, act=True):
        return conv_bn_act(sd, pre, inp, s=s, p=p, act="relu" if act else "none", train=train)

    for i, (frm, _n, kind, args) in enumerate(cfg["backbone"]):
        inp = x if frm == -1 else outs[frm]
        pre = f"backbone.{i}"
        if kind == "ResNetStem":
            y = F.max_pool2d(conv(pre + ".stem.0", inp, s=2, p=3), 3, 2, 1)
        elif kind == "ResNet50Layer":
            nb, stride = int(args[1]), int(args[2]) if len(args) >= 3 else 1
            y = inp
            for b in range(nb):
                bp = f"{pre}.layer.{b}"
                st = stride if b == 0 else 1
                o = conv(bp + ".conv1", y, 1, 0)
                o = conv(bp + ".conv2", o, st, 1)
                o = conv(bp + ".conv3", o, 1, 0, act=False)
                # the down-sampling Conv is registered on the layer and on its first block (same tensors under two names,
                # :458-463); the layer-level name is the one named_parameters() reports
                idt = conv(pre + ".downsample", y, st, 0, act=False) if (b == 0 and (pre + ".downsample.conv.weight") in sd) else y
                y = F.relu(o + idt)
        else:
            raise NotImplementedError(kind)
        outs.append(y)
        x = y
    for i, (frm, _n, kind, args) in enumerate(cfg["head"]):
        inp = [outs[f] for f in frm] if isinstance(frm, list) else outs[frm]
        pre = f"head.{i}"
        if kind == "Conv":
            k = int(args[1]) if len(args) >= 2 else 1
            s_ = int(args[2]) if len(args) >= 3 else 1
            y = conv(pre, inp, s_, None, act=args[5] if len(args) >= 6 else True)
        elif kind == "C3":
            n = int(args[1]) if len(args) >= 2 else 1
            y = c3_script(sd, pre, inp, n, False, act="relu", train=train)
        elif kind == "SPPF":
            y = sppf(sd, pre, inp, int(args[1]) if len(args) >= 2 else 5, act="relu", train=train)
        elif kind == "Upsample":
            y = F.interpolate(inp, scale_factor=float(args[1]) if len(args) >= 2 else 2.0, mode="nearest")
        elif kind == "Concat":
            y = concat_align(inp, 1)
        elif kind == "nn.Softmax":
            y = torch.softmax(inp, 1)
        else:
            raise NotImplementedError(kind)
        outs.append(y)
        x = y
    if tuple(x.shape[2:]) != tuple(img_size):
        x = F.interpolate(x, size=tuple(img_size), mode="bilinear", align_corners=False)
    return x


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def seg_loss(pred: torch.Tensor, target: torch.Tensor, class_weights: Optional[torch.Tensor] = None,
             kind: str = "dice", label_smoothing: float = 0.0, eps: float = 1e-6
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CE(weight, label_smoothing) + 0.5 * (1 - mean_{b,c} ratio) — seg_diceloss_yolov5.py:712-750 (dice),
    yolov8/seg_jaccardloss_yolov8.py:774-815 (jaccard), segment/train.py:289-337 (unweighted).
    ``pred`` is whatever the model emits (for the yaml models that is already a softmax: trap T5);
    it is log-softmaxed by CE and softmaxed again for the overlap term, exactly as the reference does.
    Returns (total, ce, overlap_loss)."""
    nc = pred.shape[1]
    if pred.shape[2:] != target.shape[1:]:
        target = F.interpolate(target.unsqueeze(1).float(), size=pred.shape[2:], mode="nearest").squeeze(1).long()
    ce = F.cross_entropy(pred, target, weight=class_weights, label_smoothing=label_smoothing)
    p = torch.softmax(pred, 1)
    onehot = torch.zeros_like(p).scatter_(1, target.unsqueeze(1), 1.0)
    if class_weights is not None:
        p = p * class_weights.view(1, nc, 1, 1)
    inter = (p * onehot).sum((2, 3))
    psum = p.sum((2, 3))
    tsum = onehot.sum((2, 3))
    if kind == "dice":
        ratio = (2.0 * inter + eps) / (psum + tsum + eps)
    elif kind == "jaccard":
        ratio = (inter + eps) / (psum + tsum - inter + eps)
    else:
        raise ValueError(kind)
    ov = 1.0 - ratio.mean()
    return ce + 0.5 * ov, ce, ov


# --------------------------------------------------------------------------------------
# optimizer / EMA (utils/torch_utils.py:318-346, 404-428)
# --------------------------------------------------------------------------------------
def sgd_nesterov_step(p: torch.Tensor, g: torch.Tensor, buf: Optional[torch.Tensor], lr: float, momentum: float,
                      weight_decay: float) -> torch.Tensor:
    """One torch.optim.SGD(nesterov=True, dampening=0) update, in place; returns the momentum buffer."""
    g = g + weight_decay * p if weight_decay != 0.0 else g
    buf = g.clone() if buf is None else buf.mul_(momentum).add_(g)
    p.add_(g + momentum * buf, alpha=-lr)
    return buf


def ema_decay(updates: int, decay: float = 0.9999, tau: float = 2000.0) -> float:
    return decay * (1.0 - math.exp(-updates / tau))


def ema_update(ema: torch.Tensor, model: torch.Tensor, d: float) -> None:
    ema.mul_(d).add_(model, alpha=1.0 - d)


# --------------------------------------------------------------------------------------
# DCNv3 core (functions/dcnv3_func.py:92-189) — grid_sample based restatement
# --------------------------------------------------------------------------------------
def dcnv3_core(inp: torch.Tensor, offset: torch.Tensor, mask: torch.Tensor, kh: int, kw: int, sh: int, sw: int,
               ph: int, pw: int, dh: int, dw: int, group: int, gc: int, offset_scale: float) -> torch.Tensor:
    """NHWC deformable sampling.  Direct (gather-free) restatement: for output (n,ho,wo), group g, point
    (i over kw outer, j over kh inner — cuh:253-254), sample location in the *unpadded* image is
    ``w = wo*sw - pw + (dw*(kw-1))//2 ... `` derived below, bilinear with zero padding, times mask."""
    N, H, W, _ = inp.shape
    _, Ho, Wo, _ = offset.shape
    P = kh * kw
    dev, dt = inp.device, inp.dtype
    # centre of the kernel window in padded coordinates (pixel units), cuh:236-244
    base_h = (dh * (kh - 1)) // 2 + torch.arange(Ho, device=dev, dtype=dt) * sh
    base_w = (dw * (kw - 1)) // 2 + torch.arange(Wo, device=dev, dtype=dt) * sw
    # kernel point grid: outer over w (i), inner over h (j) — matches _generate_dilation_grids ordering
    gi = (-((dw * (kw - 1)) // 2) + torch.arange(kw, device=dev, dtype=dt) * dw)
    gj = (-((dh * (kh - 1)) // 2) + torch.arange(kh, device=dev, dtype=dt) * dh)
    pt_w = gi.view(kw, 1).expand(kw, kh).reshape(P)
    pt_h = gj.view(1, kh).expand(kw, kh).reshape(P)
    off = offset.view(N, Ho, Wo, group, P, 2)
    # location in padded image coordinates where pixel centres are at integer+0.5 (grid_sample align_corners=False)
    loc_w = base_w.view(1, 1, Wo, 1, 1) + (pt_w.view(1, 1, 1, 1, P) + off[..., 0]) * offset_scale
    loc_h = base_h.view(1, Ho, 1, 1, 1) + (pt_h.view(1, 1, 1, 1, P) + off[..., 1]) * offset_scale
    # to unpadded pixel-index coordinates: x_pad_idx = loc (since +0.5 centre then -0.5), minus pad
    fw = loc_w - pw
    fh = loc_h - ph
    w0 = torch.floor(fw)
    h0 = torch.floor(fh)
    lw, lh = fw - w0, fh - h0
    x = inp.view(N, H * W, group, gc)
    out = torch.zeros(N, Ho, Wo, group, gc, device=dev, dtype=dt)
    m = mask.view(N, Ho, Wo, group, P)
    nidx = torch.arange(N, device=dev).view(N, 1, 1, 1, 1)
    gidx = torch.arange(group, device=dev).view(1, 1, 1, group, 1)
    for (oh, ow, wgt) in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
        hh = (h0 + oh).long()
        ww = (w0 + ow).long()
        valid = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
        lin = hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)
        v = x[nidx, lin, gidx]                        # (N,Ho,Wo,G,P,gc)
        out = out + (v * (wgt * m * valid.to(dt)).unsqueeze(-1)).sum(4)
    return out.reshape(N, Ho, Wo, group * gc)


# --------------------------------------------------------------------------------------
# DCNv3 module and its YOLO wiring (modules/dcnv3.py:50-136, "common and yolo.py":2-38)
# --------------------------------------------------------------------------------------
def linear_nhwc(sd: SD, pre: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear on the last dimension of an (N, H, W, C) tensor — modules/dcnv3.py:92-100"""
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def dwconv_bn_act(sd: SD, pre: str, x: torch.Tensor, train: bool = True) -> torch.Tensor:
    """``Conv(c, c, k, g=c)`` of modules/dcnv3.py:89 (:27-39): depth-wise conv -> BN -> SiLU, NCHW in/out"""
    w = sd[pre + ".conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride=1, padding=k // 2, groups=w.shape[0])
    if train and (pre + ".bn.num_batches_tracked") in sd:
        sd[pre + ".bn.num_batches_tracked"] += 1
    y = F.batch_norm(y, sd[pre + ".bn.running_mean"], sd[pre + ".bn.running_var"], sd[pre + ".bn.weight"], sd[pre + ".bn.bias"],
                     train, BN_MOMENTUM, BN_EPS)
    return F.silu(y)


def dcnv3_module(sd: SD, pre: str, inp: torch.Tensor, k: int, stride: int, pad: int, group: int, offset_scale: float = 1.0,
                 train: bool = True) -> torch.Tensor:
    """DCNv3.forward — modules/dcnv3.py:109-136.  ``inp`` is (N, H, W, C); dilation is forced to 1 (:82)."""
    N, H, W, C = inp.shape
    x = linear_nhwc(sd, pre + ".input_proj", inp)
    x1 = dwconv_bn_act(sd, pre + ".dw_conv", inp.permute(0, 3, 1, 2), train=train).permute(0, 2, 3, 1)
    offset = linear_nhwc(sd, pre + ".offset", x1)
    mask = linear_nhwc(sd, pre + ".mask", x1).reshape(N, H, W, group, -1)
    mask = F.softmax(mask, -1).reshape(N, H, W, -1)
    y = dcnv3_core(x, offset, mask, k, k, stride, stride, pad, pad, 1, 1, group, C // group, offset_scale)
    return linear_nhwc(sd, pre + ".output_proj", y)


def dcnv3_yolo(sd: SD, pre: str, x: torch.Tensor, k: int = 3, s: int = 1, g: int = 1, train: bool = True) -> torch.Tensor:
    """DCNV3_YoLo — "common and yolo.py":2-14: Conv(inc, ouc, 1) then DCNv3(ouc, kernel_size=k, stride=s, group=g) (pad stays 1)"""
    x = conv_bn_act(sd, pre + ".conv", x, train=train)
    y = dcnv3_module(sd, pre + ".dcnv3", x.permute(0, 2, 3, 1), k, s, 1, g, train=train)
    return y.permute(0, 3, 1, 2)


def bottleneck_dcnv3(sd: SD, pre: str, x: torch.Tensor, add: bool, g: int = 1, train: bool = True) -> torch.Tensor:
    """"common and yolo.py":16-25"""
    y = dcnv3_yolo(sd, pre + ".cv2", conv_bn_act(sd, pre + ".cv1", x, train=train), 3, 1, g, train=train)
    return x + y if add else y


def c3_dcnv3(sd: SD, pre: str, x: torch.Tensor, n: int, shortcut: bool = True, g: int = 1, train: bool = True) -> torch.Tensor:
    """"common and yolo.py":27-38"""
    p = (pre + ".") if pre else ""
    a = conv_bn_act(sd, p + "cv1", x, train=train)
    for i in range(int(n)):
        a = bottleneck_dcnv3(sd, f"{p}m.{i}", a, shortcut, g, train=train)
    b = conv_bn_act(sd, p + "cv2", x, train=train)
    return conv_bn_act(sd, p + "cv3", torch.cat((a, b), 1), train=train)


# --------------------------------------------------------------------------------------
# mIoU evaluator (val_diceloss.py:37-75): confusion matrix, ignore index nc-1, 0/0 -> 0
# --------------------------------------------------------------------------------------
def confusion_matrix(pred_cls: torch.Tensor, target: torch.Tensor, nc: int,
                     ignore_index: Optional[int] = 11) -> torch.Tensor:
    """val_diceloss.py:44-58: pixels whose *target* is the ignore class are dropped, the rest are counted at
    [t, p] when both indices are in range (the per-pixel python loop there == bincount(nc*t + p),
    Resnet50/test.py:431-434)."""
    t, p = target.flatten().long(), pred_cls.flatten().long()
    k = (t >= 0) & (t < nc) & (p >= 0) & (p < nc)
    if ignore_index is not None:
        k &= t != ignore_index
    return torch.bincount(nc * t[k] + p[k], minlength=nc * nc).view(nc, nc)


def miou_from_confusion(cm: torch.Tensor, ignore_index: Optional[int] = 11) -> Tuple[float, List[float]]:
    """val_diceloss.py:60-75: IoU_c = TP/(TP+FP+FN) over the full matrix, class ``ignore_index`` skipped,
    0/0 -> 0, mean over the remaining classes."""
    cm = cm.double()
    ious = []
    for c in range(cm.shape[0]):
        if c == ignore_index:
            continue
        tp = cm[c, c]
        union = cm[:, c].sum() + cm[c, :].sum() - tp
        ious.append(float(tp / union) if union != 0 else 0.0)
    return float(sum(ious) / len(ious)), ious
