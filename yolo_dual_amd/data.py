"""GPU half of the reference's per-sample input preparation.

Mirrors ``JSONSegmentDataset._resize_and_pad`` and the format conversion of ``__getitem__``
(unet-lite/yolo5-seg/seg_diceloss_yolov5.py:309-349; Resnet18/seg_diceloss_resnet18.py:84-149): aspect-preserving
``Image.resize(BILINEAR)`` of the image / ``Image.resize(NEAREST)`` of the label map, paste on a 128-grey / 0 canvas of
``img_size``, ``/255`` and HWC→CHW float32, labels as int64.  Decoding files, JSON parsing and the random augmentations stay on
the host (out of scope); what is here starts from the decoded uint8 arrays and is bit-exact with Pillow (12.2.0 checked):
the coefficient and index tables are built here in double precision the way Pillow builds them, the HIP kernels do the 22-bit
fixed-point arithmetic (``csrc/input.hip``).  No CPU fallback: tensors are moved to the GPU, the kernels run there."""
from __future__ import annotations

import functools
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .tape import _p, _stream

_PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c: PRECISION_BITS for 8 bits per channel


@functools.lru_cache(maxsize=256)
def _bilinear_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle ("bilinear") filter over the whole input range,
    all output samples at once; the tap weights of one sample are summed tap by tap (Pillow's order) before normalising."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale                                   # filter support 1.0, stretched when down-scaling
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    cnt = xmax - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    arg = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    w = np.where((arg < 1.0) & (taps < cnt[:, None]), 1.0 - arg, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                             # sequential accumulation, like `ww += w` in the C loop
        ww = ww + w[:, t]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    coef = np.trunc(0.5 + k * float(1 << _PRECISION_BITS)).astype(np.int32)
    coef[taps.repeat(out_size, 0) >= cnt[:, None]] = 0
    bounds = np.stack([xmin, cnt], axis=1).astype(np.int32)
    if int((bounds[:, 0] + bounds[:, 1]).max()) > in_size:
        raise AssertionError("resample taps outside the source")
    return np.ascontiguousarray(bounds), np.ascontiguousarray(coef), ksize


@functools.lru_cache(maxsize=256)
def _nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """Geometry.c ImagingScaleAffine: xo = a/2, then `xo += a` per output sample — np.cumsum adds sequentially in double too"""
    a = in_size / out_size
    steps = np.full(out_size, a, dtype=np.float64)
    steps[0] = a * 0.5
    xo = np.cumsum(steps)
    return np.clip(np.trunc(xo).astype(np.int64), 0, in_size - 1).astype(np.int32)


def letterbox_geometry(w: int, h: int, img_size: int) -> Tuple[int, int, int, int]:
    """``(new_w, new_h, pad_left, pad_top)`` of _resize_and_pad (:327-339)"""
    scale = min(img_size / w, img_size / h)
    new_w, new_h = int(w * scale), int(h * scale)
    if new_w < 1 or new_h < 1:
        raise ValueError(f"image {w}x{h} collapses to {new_w}x{new_h} at img_size {img_size}")
    return new_w, new_h, (img_size - new_w) // 2, (img_size - new_h) // 2


class LetterboxGPU:
    """``lb = LetterboxGPU(640, num_classes=12); img, mask = lb(img_u8_hwc, mask_u8_hw)`` — the arrays a
    ``JSONSegmentDataset.__getitem__`` holds after decoding (and augmenting) go in, what it returns comes out, on the GPU."""

    def __init__(self, img_size: int = 640, num_classes: int = 12, device=None, fill: int = 128):
        if not torch.cuda.is_available():
            raise RuntimeError("LetterboxGPU runs on the GPU only (yolo_dual_amd has no CPU fallback)")
        self.img_size = int(img_size)
        self.num_classes = int(num_classes)
        self.fill = int(fill)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._dev_tables = {}

    def _tab(self, kind: str, n_in: int, n_out: int):
        key = (kind, n_in, n_out)
        t = self._dev_tables.get(key)
        if t is None:
            if kind == "bil":
                b, k, ks = _bilinear_tables(n_in, n_out)
                t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), ks)
            else:
                t = torch.from_numpy(_nearest_table(n_in, n_out)).to(self.device)
            if len(self._dev_tables) > 512:
                self._dev_tables.clear()
            self._dev_tables[key] = t
        return t

    @staticmethod
    def _u8(a, ndim: int) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        if t.dtype != torch.uint8 or t.dim() != ndim:
            raise TypeError(f"expected a uint8 array with {ndim} dimensions, got {t.dtype} {tuple(t.shape)}")
        return t

    def __call__(self, img, mask=None, out_img: Optional[torch.Tensor] = None, out_mask: Optional[torch.Tensor] = None):
        S = self.img_size
        img = self._u8(img, 3)
        if img.shape[2] != 3:
            raise ValueError("image must be [H][W][3] RGB")
        h, w = int(img.shape[0]), int(img.shape[1])
        new_w, new_h, pl, pt = letterbox_geometry(w, h, S)
        img = img.to(self.device, non_blocking=True).contiguous()
        st = _stream()
        if out_img is None:
            out_img = torch.empty((3, S, S), dtype=torch.float32, device=self.device)
        elif tuple(out_img.shape) != (3, S, S) or out_img.dtype != torch.float32 or not out_img.is_contiguous():
            raise ValueError("out_img must be a contiguous float32 [3][S][S] tensor")
        xb = xk = yb = yk = None
        xks = yks = 0
        tmp = None
        if new_w != w:
            xb, xk, xks = self._tab("bil", w, new_w)
            tmp = torch.empty((h, new_w, 3), dtype=torch.uint8, device=self.device)
        if new_h != h:
            yb, yk, yks = self._tab("bil", h, new_h)
        L.call("ydl_letterbox_image", _p(img), h, w, _p(tmp) if tmp is not None else None, _p(out_img), S, new_w, new_h, pl, pt,
               _p(xb) if xb is not None else None, _p(xk) if xk is not None else None, xks,
               _p(yb) if yb is not None else None, _p(yk) if yk is not None else None, yks, self.fill, st)
        if mask is None:
            return out_img, None
        mask = self._u8(mask, 2)
        if (int(mask.shape[0]), int(mask.shape[1])) != (h, w):
            raise ValueError("mask and image sizes differ")
        mask = mask.to(self.device, non_blocking=True).contiguous()
        if out_mask is None:
            out_mask = torch.empty((S, S), dtype=torch.int64, device=self.device)
        elif tuple(out_mask.shape) != (S, S) or out_mask.dtype != torch.int64 or not out_mask.is_contiguous():
            raise ValueError("out_mask must be a contiguous int64 [S][S] tensor")
        L.call("ydl_letterbox_mask", _p(mask), h, w, _p(out_mask), S, new_w, new_h, pl, pt,
               _p(self._tab("near", w, new_w)), _p(self._tab("near", h, new_h)), self.num_classes - 1, st)
        return out_img, out_mask

    def batch(self, imgs: Sequence, masks: Optional[Sequence] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """what the DataLoader's default collate makes of the per-sample tensors: (N,3,S,S) float32 and (N,S,S) int64"""
        n, S = len(imgs), self.img_size
        out_i = torch.empty((n, 3, S, S), dtype=torch.float32, device=self.device)
        out_m = torch.empty((n, S, S), dtype=torch.int64, device=self.device) if masks is not None else None
        for i in range(n):
            self(imgs[i], masks[i] if masks is not None else None, out_i[i], out_m[i] if out_m is not None else None)
        return out_i, out_m
