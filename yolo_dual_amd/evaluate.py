"""Validation metric of the seg scripts: argmax + confusion matrix + mIoU (unet-lite/yolo5-seg/val_diceloss.py:37-75).
The reference walks every pixel in a python loop (:56-58); here one HIP kernel does argmax over the class axis and a
block-local LDS histogram, then adds it to a device-resident int64 matrix."""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _lib as L
from .tape import _p, _stream


class ConfusionMatrix:
    def __init__(self, num_classes: int, ignore_index: int = 11, device="cuda"):
        self.num_classes = num_classes
        self.ignore_index = -1 if ignore_index is None else ignore_index
        self.matrix = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=device)

    def process_batch(self, preds: torch.Tensor, targets: torch.Tensor) -> None:
        """preds: (N,C,H,W) scores/probabilities (argmax taken on the GPU); targets: (N,H,W) int64."""
        if preds.dim() != 4 or preds.size(1) != self.num_classes:
            raise ValueError("preds must be (N, num_classes, H, W)")
        preds = preds.float()
        targets = targets.contiguous().long()
        N, C, H, W = preds.shape
        sn, sc, sh, sw = preds.stride()
        L.call("ydl_confusion_matrix", _p(preds), sn, sc, sh, sw, _p(targets), N, C, H, W, self.ignore_index,
               _p(self.matrix), _stream())

    def compute_iou(self) -> Tuple[float, List[float]]:
        m = self.matrix.cpu().double()
        ious = []
        for c in range(self.num_classes):
            if c == self.ignore_index:
                continue
            tp = m[c, c]
            union = m[:, c].sum() + m[c, :].sum() - tp
            ious.append(float(tp / union) if union != 0 else 0.0)
        return float(sum(ious) / len(ious)), ious
