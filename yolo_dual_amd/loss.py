"""SegmentationLoss = CrossEntropy(weight, label_smoothing) + 0.5 * (Dice | Jaccard), one fused HIP pass each way.

Interface of the reference classes (seg_diceloss_yolov5.py:693-750 weighted Dice, yolov8/seg_jaccardloss_yolov8.py:
755-815 weighted Jaccard, segment/train.py:289-337 unweighted Dice): ``forward(pred, target) -> (total, [total, ce,
overlap])``.  The reference's three ``.item()`` host syncs are kept lazily: the returned list holds floats read from
ONE device->host copy of the 3-float loss vector (``sync=False`` returns 0-dim tensors instead and never syncs)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import config
from .tape import _p, _stream


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, cw, kind, ls, eps, lazy):
        if pred.device.type != "cuda":
            raise RuntimeError("SegmentationLoss runs on the GPU only (no CPU fallback for the HIP kernels)")
        if pred.dtype != torch.float32:
            pred = pred.float()
        N, C, H, W = pred.shape
        target = target.contiguous()
        if target.dtype != torch.int64:
            target = target.long()
        Ht, Wt = target.shape[1:]
        ws = torch.empty(L.lib().ydl_seg_loss_ws_floats(N, C), dtype=torch.float32, device=pred.device)
        losses = torch.empty(3, dtype=torch.float32, device=pred.device)
        if lazy is not None and (Ht, Wt) != (H, W):
            lazy = None
        ctx.lazy = lazy
        if lazy is not None:
            low, (rh, rw) = lazy.low, lazy.rep
            sn, sc, sh, sw = low.stride()
            L.call("ydl_seg_loss_rep_fwd", _p(low), sn, sc, sh, sw, _p(target), _p(cw), kind, ls, eps,
                   N, C, H // rh, W // rw, rh, rw, _p(ws), _p(losses), _stream())
            ctx.save_for_backward(target, ws)
            ctx.shape = tuple(pred.shape)
        else:
            sn, sc, sh, sw = pred.stride()
            L.call("ydl_seg_loss_fwd", _p(pred), sn, sc, sh, sw, _p(target), Ht, Wt, _p(cw), kind, ls, eps,
                   N, C, H, W, _p(ws), _p(losses), _stream())
            ctx.save_for_backward(pred, target, ws)
        ctx.cw, ctx.kind, ctx.ls, ctx.eps = cw, kind, ls, eps
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, gtotal, _glosses):
        g = gtotal.contiguous().float()
        lazy = ctx.lazy
        if lazy is not None:
            target, ws = ctx.saved_tensors
            low, (rh, rw) = lazy.low, lazy.rep
            N, C, h, w = low.shape
            dlow = torch.empty_like(low)
            sn, sc, sh, sw = low.stride()
            L.call("ydl_seg_loss_rep_bwd", _p(low), sn, sc, sh, sw, _p(target), _p(ctx.cw), ctx.kind, ctx.ls, ctx.eps,
                   N, C, h, w, rh, rw, _p(ws), _p(g), _p(dlow), _stream())
            # the replica-summed gradient travels through the side channel; autograd gets a zero placeholder
            lazy.dlow = dlow if lazy.dlow is None else lazy.dlow + dlow      # (a repeated backward of this loss adds up)
            lazy.dummy = g.new_zeros(1).expand(ctx.shape)
            return lazy.dummy, None, None, None, None, None, None
        pred, target, ws = ctx.saved_tensors
        N, C, H, W = pred.shape
        Ht, Wt = target.shape[1:]
        dpred = torch.empty_like(pred)
        sn, sc, sh, sw = pred.stride()
        if dpred.stride() != pred.stride():
            dpred = torch.empty_strided(pred.shape, pred.stride(), dtype=pred.dtype, device=pred.device)
        L.call("ydl_seg_loss_bwd", _p(pred), sn, sc, sh, sw, _p(target), Ht, Wt, _p(ctx.cw), ctx.kind, ctx.ls, ctx.eps,
               N, C, H, W, _p(ws), _p(g), _p(dpred), _stream())
        return dpred, None, None, None, None, None, None


class SegmentationLoss(nn.Module):
    """``SegmentationLoss(num_classes=12, label_smoothing=0.0, class_weights=None, kind='dice')``."""

    def __init__(self, num_classes: int = 12, label_smoothing: float = 0.0, class_weights=None, kind: str = "dice",
                 sync: bool = True):
        super().__init__()
        self.num_classes = num_classes
        self.label_smoothing = float(label_smoothing)
        self.kind = {"dice": L.LOSS_DICE, "jaccard": L.LOSS_JACCARD}[kind]
        self.sync = sync
        self.use_replicated = config.replicated_loss()   # False forces the full-resolution kernels (tests compare the two)
        self.class_weights = None if class_weights is None else torch.as_tensor(class_weights, dtype=torch.float32)

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, List[float]]:
        if pred.size(0) != target.size(0):
            raise ValueError(f"batch size mismatch: output {pred.size(0)} vs labels {target.size(0)}")
        if pred.size(1) != self.num_classes:
            raise ValueError(f"pred has {pred.size(1)} channels, loss was built for {self.num_classes} classes")
        cw = self.class_weights
        if cw is not None and cw.device != pred.device:
            cw = self.class_weights = cw.to(pred.device)
        # a prediction that is an exact nearest replication (lazy Upsample -> Conv 1x1 -> Softmax tail) and has not been
        # edited in place since the model produced it: loss and gradient are evaluated per stored pixel
        lazy = getattr(pred, "_ydl_lazy", None) if self.use_replicated else None
        if lazy is not None and (lazy.version != pred._version or pred.dtype != torch.float32
                                 or not (pred.requires_grad and torch.is_grad_enabled())):
            lazy = None
        if lazy is not None:
            # ONE consumer may use the side channel; a second loss on the same prediction takes the dense kernels, whose
            # gradient autograd adds to the first one's (the region folds the side-channel part into the dense sum)
            if getattr(lazy, "claimed", False):
                lazy = None
            else:
                lazy.claimed = True
        total, losses = _SegLossFn.apply(pred, target, cw, self.kind, self.label_smoothing, 1e-6, lazy)
        if self.sync:
            return total, losses.tolist()
        return total, [losses[0], losses[1], losses[2]]


class JaccardSegmentationLoss(SegmentationLoss):
    def __init__(self, num_classes: int = 12, label_smoothing: float = 0.0, class_weights=None, sync: bool = True):
        super().__init__(num_classes, label_smoothing, class_weights, "jaccard", sync)
