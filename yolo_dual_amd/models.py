"""Segmentation models behind the reference's constructors and yaml surface.

* ``YOLOv5Seg`` / ``YOLOv8Seg`` / ``YOLOv9Seg`` — yaml-driven script builders with every quirk of the reference
  reproduced (SURVEY T3/T4/T7 and the v8 Upsample trap found by the golden fixture):
  unet-lite/yolo5-seg/seg_diceloss_yolov5.py:511-681, yolov8/seg_jaccardloss_yolov8.py:502-720,
  unet-lite/yolo9-seg/seg_diceloss_yolov9.py:587-779.
* ``ResNet18Seg`` / ``ResNet50Seg`` — unet-lite/Resnet18/seg_diceloss_resnet18.py:243-403, segment/train.py:103-286.
* ``parse_model`` / ``SegYoloModel`` — models/yolo.py:299-382 + BaseModel._forward_once :114-125 semantics
  (depth/width gains, ``n`` insertion, ``.i .f .type .np`` tags, save-list routing).

Each model's forward is ONE taped region: the whole network (fwd and bwd) runs on the HIP kernels behind a single
autograd node; parameter names/shapes match the reference so ``state_dict``s interchange.
"""
from __future__ import annotations

import logging
import math
import os
from copy import deepcopy
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import yaml

from . import _lib as L
from .modules import (GAM, BasicBlock, Bottleneck, Bottleneck_DCNV3, BottleneckBlock, C2f, C3, C3_DCNV3, C3Common, C3k2, Concat,
                      Conv, MaxPool2d, SegmentHead, SPPF, Upsample, YdlModule, run_region)
from .tape import Tape, Var

LOGGER = logging.getLogger("yolo_dual_amd")
CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg")


def check_yaml(path: str) -> str:
    if os.path.isfile(path):
        return path
    cand = os.path.join(CFG_DIR, path)
    if os.path.isfile(cand):
        return cand
    raise FileNotFoundError(path)


class Softmax(nn.Module):
    """holder for the yaml's ``nn.Softmax`` row (executed by the model's taped region)"""

    def __init__(self, dim=1):
        super().__init__()
        self.dim = dim
        if dim != 1:
            raise NotImplementedError("Softmax over the channel dimension only")


def _kaiming_init(model: nn.Module, nonlinearity: str) -> None:
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity=nonlinearity)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _upsample_from_yaml(args, family: str) -> Upsample:
    """The yaml writes ``[None, 2, 'nearest']`` where ``None`` is a YAML *string*.
    v5/v9 builders: int('None') fails -> size None -> scale_factor=float(scale) (seg_diceloss_yolov5.py:588-609,
    seg_diceloss_yolov9.py:692-717).  v8 builder: the same failure falls through to size=(256, 256) and drops the
    scale (yolov8/seg_jaccardloss_yolov8.py:583-660) — every Upsample of YOLOv8Seg emits 256x256."""
    size_arg = args[0] if len(args) >= 1 else None
    scale_arg = args[1] if len(args) >= 2 else 2.0
    mode = args[2] if len(args) >= 3 else "nearest"
    if family == "v8" and len(args) < 3:
        size_arg, scale_arg, mode = None, 2.0, "nearest"
    size = None
    if size_arg is not None:
        if isinstance(size_arg, (list, tuple)):
            size = tuple(int(v) for v in size_arg[-2:])
        else:
            try:
                size = (int(size_arg), int(size_arg))
            except (TypeError, ValueError):
                size = (256, 256) if family == "v8" else None
    ac = False if mode in ("bilinear", "bicubic") else None
    if size is not None:
        return Upsample(size=size, mode=mode, align_corners=ac)
    sf = float(scale_arg) if isinstance(scale_arg, (int, float)) else 2.0
    return Upsample(scale_factor=sf, mode=mode, align_corners=ac)


class _YamlSegModel(YdlModule):
    family = "v5"
    backbone_modules: Dict[str, type] = {}
    head_modules: Dict[str, type] = {}
    init_nonlinearity = "leaky_relu"

    def __init__(self, cfg, num_classes: Optional[int] = None):
        super().__init__()
        if isinstance(cfg, str):
            with open(check_yaml(cfg), "r") as f:
                self.yaml = yaml.safe_load(f)
        else:
            self.yaml = cfg
        self.num_classes = self.yaml["nc"] if num_classes is None else num_classes
        self.yaml["nc"] = self.num_classes
        self.img_size = [640, 640]
        self.stride = torch.tensor([2, 4, 8, 16, 32])
        self.backbone, self.backbone_out_chs = self._build_backbone(self.yaml["backbone"])
        self.head, self.head_out_chs = self._build_head(self.yaml["head"], self.backbone_out_chs)
        _kaiming_init(self, self.init_nonlinearity)
        self._dead_head = self._find_dead_head_layers()
        self._log_model_info()

    def _find_dead_head_layers(self):
        """head layers whose output never reaches the model output (absolute ``from`` indices leave whole
        sub-chains of the yaml head unused).  They are still executed — their BatchNorm running statistics are part
        of the reference's state — but on the side stream, concurrently with the live path."""
        nb, head = len(self.yaml["backbone"]), self.yaml["head"]
        live = {nb + len(head) - 1}
        for i in range(len(head) - 1, -1, -1):
            a = nb + i
            if a not in live:
                continue
            f = head[i][0]
            for src in (f if isinstance(f, list) else [f]):
                live.add(src if src >= 0 else a + src)
        return {i for i in range(len(head)) if nb + i not in live}

    # builders: ``Module(c1, *args)``; the yaml ``number`` column and the multiples are ignored (T3)
    def _make(self, table, module: str, c1: int, args, where: str):
        if module in ("C3_DCN", "C2f_DCN"):
            raise NotImplementedError(
                f"{module} wraps torchvision.ops.DeformConv2d, which is not part of the reference repository "
                "(parity unpinned, SURVEY §8c); substitute C3/C2f as the benchmark configs do")
        if module in ("Upsample", "nn.Upsample"):
            return _upsample_from_yaml(args, self.family), c1
        if module == "Concat":
            return Concat(*args), c1
        if module == "nn.Softmax":
            return Softmax(*(args if args else [1])), c1
        if module == "GAM":
            return GAM(c1, *args), c1
        if module not in table:
            raise NotImplementedError(f"unknown {where} module: {module}")
        return table[module](c1, *args), args[0]

    def _build_backbone(self, cfg):
        backbone, out_chs, prev = nn.ModuleList(), [], 3
        for i, (from_, _num, module, args) in enumerate(cfg):
            c1 = prev if from_ == -1 else out_chs[from_]
            layer, c2 = self._make(self.backbone_modules, module, c1, args, "backbone")
            backbone.append(layer)
            out_chs.append(c2)
            prev = c2
            LOGGER.debug("backbone %d: %s %d->%d", i, module, c1, c2)
        return backbone, out_chs

    def _build_head(self, cfg, backbone_out_chs):
        head, all_chs = nn.ModuleList(), list(backbone_out_chs)
        for i, (from_, _num, module, args) in enumerate(cfg):
            c1 = sum(all_chs[f] for f in from_) if isinstance(from_, list) else all_chs[from_]   # absolute (T4)
            layer, c2 = self._make(self.head_modules, module, c1, args, "head")
            head.append(layer)
            all_chs.append(c2)
            LOGGER.debug("head %d: %s %d->%d", i, module, c1, c2)
        return head, all_chs

    def forward(self, x: torch.Tensor, augment: bool = False, profile: bool = False) -> torch.Tensor:
        return self._forward_once(x, profile)

    def _forward_once(self, x: torch.Tensor, profile: bool = False) -> torch.Tensor:
        return run_region(self, [x])

    def _fwd(self, tape: Tape, x: Var):
        outs: List[Var] = []
        for layer in self.backbone:
            x = layer._fwd(tape, x)
            outs.append(x)
        n_head = len(self.head)
        from . import config as _cfg
        from .tape import side_stream, stream_wait
        use_side = bool(self._dead_head) and _cfg.overlap_wgrad()
        main = torch.cuda.current_stream()
        side = side_stream(tape.device) if use_side else None
        fork_mark = -1                   # launch counter at the last fork: nothing new to wait for if unchanged
        for i, (layer, (from_, _num, _module, _args)) in enumerate(zip(self.head, self.yaml["head"])):
            inp = [outs[f] for f in from_] if isinstance(from_, list) else outs[from_]
            if use_side and i in self._dead_head:
                # inputs come from layers already enqueued on main: one fork per RUN of consecutive dead layers.
                # (A fork per layer put two event waits back to back on the side stream whenever a dead layer launched
                # nothing — a virtual concat — and the captured HIP graph then replayed with broken ordering.)
                if (i == 0 or (i - 1) not in self._dead_head) and L.launch_count() != fork_mark:
                    stream_wait(side, main)
                    fork_mark = L.launch_count()
                with torch.cuda.stream(side):
                    outs.append(layer._fwd(tape, inp) if not isinstance(layer, Softmax) else tape.softmax(inp))
                tape._side_fwd = True
                continue
            if isinstance(layer, Softmax):
                H, W = self.img_size
                if i == n_head - 1 and (inp.LH, inp.LW) == (H, W):
                    p = tape.softmax_nchw(inp)          # final layer, already at img_size: write NCHW f32 directly
                    tape.ext = (inp, p)
                    if use_side:
                        stream_wait(main, side)          # the dead branch has finished before the region returns
                    return p
                x = tape.softmax(inp)
            else:
                x = layer._fwd(tape, inp)
            outs.append(x)
        H, W = self.img_size
        if (x.LH, x.LW) != (H, W):                          # T7: always resized to the hard-coded img_size
            x = tape.resize(x, H, W, L.RESIZE_BILINEAR)
        if use_side:
            stream_wait(main, side)
        return x

    def _seed_external(self, tape: Tape, gout: torch.Tensor) -> None:
        inp, p = tape.ext
        tape.ext = None
        tape.softmax_nchw_backward(inp, p, gout)

    def _initialize_weights(self) -> None:
        _kaiming_init(self, self.init_nonlinearity)

    def fuse(self):
        """models/yolo.py:140-148: fold every Conv's BatchNorm into its convolution (inference only)"""
        for m in self.modules():
            if isinstance(m, Conv) and not m.depthwise:
                m.fuse()
        return self

    def _log_model_info(self) -> None:
        n = sum(p.numel() for p in self.parameters())
        LOGGER.info("model: %s parameters, %d classes", f"{n:,}", self.num_classes)


class YOLOv5Seg(_YamlSegModel):
    family = "v5"
    backbone_modules = {"Conv": Conv, "C3": C3, "SPPF": SPPF, "C3_DCNV3": C3_DCNV3}
    head_modules = {"Conv": Conv, "C3": C3, "C3_DCNV3": C3_DCNV3}


class YOLOv8Seg(_YamlSegModel):
    family = "v8"
    backbone_modules = {"Conv": Conv, "C2f": C2f, "SPPF": SPPF, "C3_DCNV3": C3_DCNV3}
    head_modules = {"Conv": Conv, "C2f": C2f, "C3_DCNV3": C3_DCNV3}


class YOLOv9Seg(_YamlSegModel):
    family = "v9"
    backbone_modules = {"Conv": Conv, "C3k2": C3k2, "SPPF": SPPF, "C3_DCNV3": C3_DCNV3}
    head_modules = {"Conv": Conv, "C2f": C2f, "C3": C3, "C3_DCNV3": C3_DCNV3}


# ----------------------------------------------------------------------------------------------------------
# ResNet18 / ResNet50 + multi-scale SegmentHead
# ----------------------------------------------------------------------------------------------------------
class _ResNet(YdlModule):
    block = BasicBlock
    layers = (2, 2, 2, 2)

    def __init__(self):
        super().__init__()
        self.in_channels = 64
        self.stem = nn.Sequential(Conv(3, 64, 7, 2, 3), MaxPool2d(3, 2, 1))
        self.layer1 = self._make_layer(64, self.layers[0], stride=1)
        self.layer2 = self._make_layer(128, self.layers[1], stride=2)
        self.layer3 = self._make_layer(256, self.layers[2], stride=2)
        self.layer4 = self._make_layer(512, self.layers[3], stride=2)    # built, never run (reference does the same)
        e = self.block.expansion
        self.feat_channels = [64 * e, 128 * e, 256 * e]

    def _make_layer(self, mid: int, num_blocks: int, stride: int = 1) -> nn.Sequential:
        e = self.block.expansion
        downsample = None
        if stride != 1 or self.in_channels != mid * e:
            downsample = Conv(self.in_channels, mid * e, 1, stride, 0, act=False)
        blocks = [self.block(self.in_channels, mid, stride, downsample)]
        self.in_channels = mid * e
        for _ in range(1, num_blocks):
            blocks.append(self.block(self.in_channels, mid))
        return nn.Sequential(*blocks)

    def _fwd(self, tape: Tape, x: Var) -> List[Var]:
        if x.C != 3:
            raise ValueError(f"ResNet expects a 3-channel image, got {x.C} channels")
        x = self.stem[1]._fwd(tape, self.stem[0]._fwd(tape, x))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk._fwd(tape, x)
            feats.append(x)
        return feats


class ResNet18(_ResNet):
    block, layers = BasicBlock, (2, 2, 2, 2)


class ResNet50(_ResNet):
    block, layers = BottleneckBlock, (3, 4, 6, 3)


class _ResNetSeg(YdlModule):
    backbone_cls = ResNet18

    def __init__(self, cfg, num_classes: Optional[int] = None):
        super().__init__()
        if isinstance(cfg, str):
            with open(check_yaml(cfg), "r") as f:
                self.yaml = yaml.safe_load(f)
        else:
            self.yaml = cfg
        self.num_classes = self.yaml["nc"] if num_classes is None else num_classes
        self.yaml["nc"] = self.num_classes
        self.img_size = [640, 640]
        self.stride = torch.tensor([8, 16, 32])
        self.backbone = self.backbone_cls()
        self.head = SegmentHead(num_classes=self.num_classes, in_channels=self.backbone.feat_channels)
        _kaiming_init(self, "relu")

    def forward(self, x: torch.Tensor, augment: bool = False, profile: bool = False) -> torch.Tensor:
        if augment:
            raise NotImplementedError("test-time augmentation is outside the training hot path")
        return run_region(self, [x])

    fixed_out = None        # (H, W) when the reference hard-codes the output size

    def _fwd(self, tape: Tape, x: Var) -> Var:
        out = self.head._fwd(tape, self.backbone._fwd(tape, x))
        H, W = self.fixed_out if self.fixed_out is not None else (x.H, x.W)
        if (out.H, out.W) != (H, W):
            out = tape.resize(out, H, W, L.RESIZE_BILINEAR)
        return out


class ResNet18Seg(_ResNetSeg):
    """unet-lite/Resnet18/seg_diceloss_resnet18.py:352-403: output resized to the input size"""
    backbone_cls = ResNet18


class ResNet50Seg(_ResNetSeg):
    """segment/train.py:213-250 with its SegmentHead (:159-210), whose forward ends with a hard-coded
    ``F.interpolate(size=(640, 640), bilinear, align_corners=False)`` whatever the input size"""
    backbone_cls = ResNet50
    fixed_out = (640, 640)


# ----------------------------------------------------------------------------------------------------------
# yaml-driven ResNet50 + UNet-lite head (unet-lite/Resnet50/seg_diceloss_Resnet50.py:389-710, resnet50.yaml)
# ----------------------------------------------------------------------------------------------------------
class ConvReLU(Conv):
    """that script's ``Conv(c1, c2, k=1, s=1, p=None, g=1, act=True)``: Conv2d -> BatchNorm2d -> **ReLU** (:389-402)"""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__(c1, c2, k, s, p, g, nn.ReLU(inplace=True) if act is True else (act if isinstance(act, nn.Module) else False))


class BottleneckBlockReLU(BottleneckBlock):
    """:405-435 — the torchvision bottleneck with ReLU ``Conv`` blocks"""

    def __init__(self, in_channels, mid_channels, stride=1, downsample=None):
        YdlModule.__init__(self)
        self.conv1 = ConvReLU(in_channels, mid_channels, 1, 1, 0, act=True)
        self.conv2 = ConvReLU(mid_channels, mid_channels, 3, stride, 1, act=True)
        self.conv3 = ConvReLU(mid_channels, mid_channels * self.expansion, 1, 1, 0, act=False)
        self.downsample = downsample
        self.act = nn.ReLU(inplace=True)


class ResNetStem(YdlModule):
    """:438-448: 7x7/s2 Conv + 3x3/s2 max-pool"""

    def __init__(self, out_channels: int):
        super().__init__()
        self.stem = nn.Sequential(ConvReLU(3, out_channels, 7, 2, 3), MaxPool2d(3, 2, 1))

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return self.stem[1]._fwd(tape, self.stem[0]._fwd(tape, x))


class ResNet50Layer(YdlModule):
    """:451-470: ``num_blocks`` bottlenecks; the down-sampling Conv is registered on the layer AND on its first block (the
    reference's state_dict therefore lists it under both names — kept, so that checkpoints interchange)"""

    def __init__(self, in_channels: int, out_channels: int, num_blocks: int, stride: int = 1):
        super().__init__()
        mid = out_channels // BottleneckBlockReLU.expansion
        self.downsample = None
        if stride != 1 or in_channels != out_channels:
            self.downsample = ConvReLU(in_channels, out_channels, 1, stride, 0, act=False)
        blocks = [BottleneckBlockReLU(in_channels, mid, stride, self.downsample)]
        for _ in range(1, num_blocks):
            blocks.append(BottleneckBlockReLU(out_channels, mid))
        self.layer = nn.Sequential(*blocks)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        for blk in self.layer:
            x = blk._fwd(tape, x)
        return x


class C3Plain(YdlModule):
    """that script's C3 (:522-535): cv3(cat(m(cv1 x), cv2 x)) with ReLU Convs and NO residual whatever ``shortcut`` says"""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = self.c_ = int(c2 * e)
        self.cv1 = ConvReLU(c1, self.c, 1, 1)
        self.cv2 = ConvReLU(c1, self.c, 1, 1)
        self.cv3 = ConvReLU(2 * self.c, c2, 1, 1)
        self.m = nn.Sequential(*(ConvReLU(self.c, self.c, 3, 1, g=g) for _ in range(n)))

    def _fwd(self, tape: Tape, x: Var) -> Var:
        from .modules import _csp_forward
        return _csp_forward(self, tape, x, False)


class SPPFReLU(SPPF):
    def __init__(self, c1, c2, k=5):
        YdlModule.__init__(self)
        c_ = c1 // 2
        self.cv1 = ConvReLU(c1, c_, 1, 1)
        self.cv2 = ConvReLU(c_ * 4, c2, 1, 1)
        self.k, self.c_ = k, c_


def parse_none(value):
    """:76-85: the yaml's ``None`` / ``none`` strings become Python None (recursively)"""
    if isinstance(value, str) and value.lower() == "none":
        return None
    if isinstance(value, list):
        return [parse_none(v) for v in value]
    if isinstance(value, dict):
        return {k: parse_none(v) for k, v in value.items()}
    return value


class ResNet50SegYaml(_YamlSegModel):
    """``ResNet50Seg`` of unet-lite/Resnet50/seg_diceloss_Resnet50.py:539-710: ResNetStem / ResNet50Layer backbone rows and a
    UNet-lite head whose builder CASTS its yaml arguments (``C3 [512, False]`` -> n = int(False) = 0, no residual; Upsample
    ``[None, 2, 'nearest']`` -> scale_factor 2.0), ReLU everywhere, the head's ``from`` absolute (resnet50.yaml:23-38)."""
    family = "r50"
    init_nonlinearity = "relu"

    def __init__(self, cfg, num_classes: Optional[int] = None):
        if not isinstance(cfg, str):
            cfg = parse_none(cfg)
        super().__init__(cfg, num_classes)
        if isinstance(self.yaml, dict):
            self.yaml = parse_none(self.yaml)
        self.stride = torch.tensor([4, 8, 16, 32])

    def _build_backbone(self, cfg):
        cfg = parse_none(cfg)
        backbone, out_chs, prev = nn.ModuleList(), [], 3
        for from_, _num, module, args in cfg:
            c1 = prev if from_ == -1 else out_chs[from_]
            if module == "ResNetStem":
                out_ch = int(args[0])
                layer = ResNetStem(out_ch)
            elif module == "ResNet50Layer":
                out_ch = int(args[0])
                layer = ResNet50Layer(c1, out_ch, int(args[1]), int(args[2]) if len(args) >= 3 else 1)
            else:
                raise NotImplementedError(f"Backbone unknown module: {module}")
            backbone.append(layer)
            out_chs.append(out_ch)
            prev = out_ch
        return backbone, out_chs

    def _build_head(self, cfg, backbone_out_chs):
        cfg = parse_none(cfg)
        head, all_chs = nn.ModuleList(), list(backbone_out_chs)
        for from_, _num, module, args in cfg:
            c1 = sum(all_chs[f] for f in from_) if isinstance(from_, list) else all_chs[from_]
            if module == "Conv":
                out_ch = int(args[0])
                k = int(args[1]) if len(args) >= 2 else 1
                s_ = int(args[2]) if len(args) >= 3 else 1
                p_ = int(args[3]) if (len(args) >= 4 and args[3] is not None) else None
                g = int(args[4]) if len(args) >= 5 else 1
                act = args[5] if len(args) >= 6 else True
                layer = ConvReLU(c1, out_ch, k, s_, p_, g, act)
            elif module == "SPPF":
                out_ch = int(args[0])
                layer = SPPFReLU(c1, out_ch, int(args[1]) if len(args) >= 2 else 5)
            elif module == "Upsample":
                size = args[0] if len(args) >= 1 else None
                sf = args[1] if len(args) >= 2 else 2.0
                mode = args[2] if len(args) >= 3 else "nearest"
                if sf is not None:
                    sf = float(sf)
                if size is not None and sf is not None:
                    size = None
                layer = Upsample(size=size, scale_factor=sf, mode=mode, align_corners=False if mode != "nearest" else None)
                out_ch = c1
            elif module == "Concat":
                layer = Concat(int(args[0]) if len(args) >= 1 else 1)
                out_ch = c1
            elif module == "C3":
                out_ch = int(args[0])
                n = int(args[1]) if len(args) >= 2 else 1
                layer = C3Plain(c1, out_ch, n, bool(args[2]) if len(args) >= 3 else False, int(args[3]) if len(args) >= 4 else 1,
                                float(args[4]) if len(args) >= 5 else 0.5)
            elif module == "nn.Softmax":
                layer = Softmax(int(args[0]) if len(args) >= 1 else 1)
                out_ch = c1
            else:
                raise NotImplementedError(f"head unknown module: {module}")
            head.append(layer)
            all_chs.append(out_ch)
        return head, all_chs


# ----------------------------------------------------------------------------------------------------------
# models/yolo.py surface: parse_model + save-list forward
# ----------------------------------------------------------------------------------------------------------
def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


_PARSE_TABLE = {"Conv": Conv, "Bottleneck": Bottleneck, "C3": C3Common, "SPPF": SPPF, "Concat": Concat,
                "nn.Upsample": Upsample, "Upsample": Upsample, "C3_DCNV3": C3_DCNV3, "Bottleneck_DCNV3": Bottleneck_DCNV3}


def parse_model(d: dict, ch: List[int]):
    """models/yolo.py:299-382 for the block set of this path: resolves module names, applies depth/width gains
    (``n = max(round(n*gd), 1)``, ``c2 = make_divisible(c2*gw, 8)``), inserts ``n`` for C3, and tags every layer
    with ``.i .f .type .np``.  Returns (nn.Sequential, sorted save-list)."""
    gd, gw = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0)
    no = d.get("nc", 0)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        if m not in _PARSE_TABLE:
            raise NotImplementedError(f"parse_model: module {m} is outside the segmentation hot path")
        cls = _PARSE_TABLE[m]
        args = [None if a == "None" else a for a in args]
        n = n_ = max(round(n * gd), 1) if n > 1 else n
        if cls in (Conv, Bottleneck, C3Common, SPPF, C3_DCNV3, Bottleneck_DCNV3):
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if cls in (C3Common, C3_DCNV3):          # models/yolo.py:327-329 + the C3_DCNV3 wiring note ("common and yolo.py")
                args.insert(2, n)
                n = 1
        elif cls is Concat:
            c2 = sum(ch[x] for x in f)
        elif cls is Upsample:
            c2 = ch[f]
            args = [args[0], args[1], args[2] if len(args) > 2 else "nearest"]
        m_ = nn.Sequential(*(cls(*args) for _ in range(n))) if n > 1 else cls(*args)
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, f"models.common.{m}"
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class SegYoloModel(YdlModule):
    """BaseModel of models/yolo.py:114-125 restricted to this block set: ``model`` + ``save`` routing by ``m.f``."""

    def __init__(self, cfg, ch: int = 3, nc: Optional[int] = None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            with open(check_yaml(cfg), encoding="ascii", errors="ignore") as f:
                self.yaml = yaml.safe_load(f)
        if nc is not None:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(self.yaml, [ch])
        _kaiming_init(self, "leaky_relu")

    def forward(self, x, augment=False, profile=False, visualize=False):
        return run_region(self, [x])

    def _fwd(self, tape: Tape, x: Var) -> Var:
        y: List[Optional[Var]] = []
        for m in self.model:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            if isinstance(m, nn.Sequential):
                for sub in m:
                    x = sub._fwd(tape, x)
            else:
                x = m._fwd(tape, x)
            y.append(x if m.i in self.save else None)
        return x
