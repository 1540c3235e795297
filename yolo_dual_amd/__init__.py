"""yolo_dual_amd — MI355X-native (gfx950) segmentation training hot path behind the reference's module API.

Public surface mirrors the reference's seg scripts and models/common.py (names, constructor signatures, state_dict
layout).  All compute runs in hand-written HIP kernels (libydl_hip.so, C ABI in include/ydl.h); there is no CPU or
ATen fallback — importing is cheap, the first GPU call loads the library and fails loudly if it was not built."""
from . import config
from .config import compute_dtype, set_compute_dtype
from .checkpoint import (fuse_conv_and_bn, intersect_dicts, load_checkpoint, load_weights, save_checkpoint, smart_resume,
                         strip_optimizer)
from .data import LetterboxGPU, letterbox_geometry
from .evaluate import ConfusionMatrix
from .loss import JaccardSegmentationLoss, SegmentationLoss
from .models import (ResNet18, ResNet18Seg, ResNet50, ResNet50Seg, ResNet50SegYaml, SegYoloModel, YOLOv5Seg, YOLOv8Seg,
                     YOLOv9Seg, parse_model)
from .modules import (GAM, C2f, C3, C3k2, C3_DCNV3, BasicBlock, Bottleneck, Bottleneck_DCNV3, BottleneckBlock, C3Common, Concat,
                      Conv, DCNv3, DCNV3_YoLo, Linear, MaxPool2d, SegmentHead, SPPF, Upsample, autopad)
from .optim import FlatSGDEMA, smart_optimizer

__all__ = [n for n in dir() if not n.startswith("_")]
