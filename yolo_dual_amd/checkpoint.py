"""Checkpoint interop with the reference's trainers.

The reference saves ``{'model': ema.ema (a pickled nn.Module), 'optimizer': ..., 'epoch': ..., 'best_fitness': ...}``
(unet-lite/yolo5-seg/seg_diceloss_yolov5.py:1204-1212) and loads by name+shape intersection (:944-952,
utils/general.py:255-257).  Here a checkpoint holds the same four keys with ``'model'`` a *state_dict* — every file is read
with ``torch.load(..., weights_only=True)``: nothing is ever unpickled into code.  A reference checkpoint that pickles the
whole module is refused with an explanation (export ``ckpt['model'].float().state_dict()`` on a machine that trusts it)."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

__all__ = ["intersect_dicts", "load_checkpoint", "load_weights", "save_checkpoint", "strip_optimizer", "smart_resume",
           "fuse_conv_and_bn"]


def intersect_dicts(da: Dict[str, torch.Tensor], db: Dict[str, torch.Tensor], exclude=()) -> Dict[str, torch.Tensor]:
    """utils/general.py:255-257: matching keys and shapes, omitting ``exclude`` substrings, values from ``da``"""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}


def load_checkpoint(path: str) -> dict:
    """read a checkpoint with the weights-only loader; returns a dict with at least ``'model'`` = state_dict"""
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        raise RuntimeError(
            f"{path}: refused by torch.load(weights_only=True) ({type(e).__name__}).  The reference's checkpoints pickle whole "
            "nn.Module objects; this loader never executes pickled code.  Re-save the file as "
            "{'model': ckpt['model'].float().state_dict(), ...} in an environment that trusts it.") from e
    if isinstance(ckpt, dict) and "model" in ckpt:
        if not isinstance(ckpt["model"], dict):
            raise RuntimeError(f"{path}: 'model' must be a state_dict")
        return ckpt
    if isinstance(ckpt, dict) and all(torch.is_tensor(v) for v in ckpt.values()):
        return {"model": ckpt}                     # a bare state_dict
    raise RuntimeError(f"{path}: not a checkpoint dict with a 'model' state_dict")


def load_weights(model: nn.Module, ckpt, exclude=()) -> Tuple[int, int]:
    """seg_diceloss_yolov5.py:944-952: ``csd = intersect_dicts(ckpt_sd.float(), model.state_dict()); load_state_dict(strict=False)``.
    ``ckpt``: a path, a checkpoint dict or a state_dict.  Returns (matched, total)."""
    if isinstance(ckpt, (str, os.PathLike)):
        ckpt = load_checkpoint(ckpt)
    sd = ckpt["model"] if (isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict)) else ckpt
    sd = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    msd = model.state_dict()
    csd = intersect_dicts(sd, msd, exclude=exclude)
    model.load_state_dict(csd, strict=False)
    from . import config
    config.bump_weight_epoch()                     # parameter memory changed behind the compute-layout weight copies
    return len(csd), len(msd)


def save_checkpoint(path: str, model_state: Dict[str, torch.Tensor], optimizer=None, epoch: int = -1,
                    best_fitness: Optional[float] = None, ema_state: Optional[Dict[str, torch.Tensor]] = None,
                    updates: Optional[int] = None) -> None:
    """the reference's checkpoint dict (seg_diceloss_yolov5.py:1204-1212) with state_dicts in place of pickled modules"""
    ck = {"model": {k: v.detach().cpu().clone() for k, v in model_state.items()},
          "optimizer": optimizer.state_dict() if optimizer is not None else None,
          "epoch": int(epoch), "best_fitness": None if best_fitness is None else float(best_fitness)}
    if ema_state is not None:
        ck["ema"] = {k: v.detach().cpu().clone() for k, v in ema_state.items()}
        ck["updates"] = updates
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)


def strip_optimizer(f: str = "best.pt", s: str = "") -> float:
    """utils/general.py:1004-1018: replace model by EMA if present, drop optimizer / best_fitness / ema / updates, epoch = -1,
    weights to FP16; returns the file size in MB"""
    x = load_checkpoint(f)
    if x.get("ema"):
        x["model"] = x["ema"]
    for k in ("optimizer", "best_fitness", "ema", "updates"):
        x[k] = None
    x["epoch"] = -1
    x["model"] = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in x["model"].items()}
    torch.save(x, s or f)
    return os.path.getsize(s or f) / 1e6


def smart_resume(ckpt: dict, optimizer, ema=None, weights: str = "last.pt", epochs: int = 300, resume: bool = True):
    """utils/torch_utils.py:361-378.  ``ema``: the FlatSGDEMA (its EMA shadow is restored from ckpt['ema'] when present)"""
    best_fitness = 0.0
    start_epoch = ckpt["epoch"] + 1
    if ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        best_fitness = ckpt["best_fitness"] if ckpt.get("best_fitness") is not None else 0.0
    if ema is not None and ckpt.get("ema"):
        ema.load_ema_state_dict(ckpt["ema"])
        if ckpt.get("updates") is not None:
            ema.updates = ckpt["updates"]
    if resume:
        assert start_epoch > 0, (f"{weights} training to {epochs} epochs is finished, nothing to resume.\n"
                                 f"Start a new training without --resume, i.e. 'python train_seg.py --weights {weights}'")
    if epochs < start_epoch:
        epochs += ckpt["epoch"]
    return best_fitness, start_epoch, epochs


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils/torch_utils.py:248-269 as tensors: (weight OIHW, bias) of the convolution that equals eval-mode bn(conv(x))"""
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.eps + bn.running_var.detach().float())
    wf = (w.reshape(w.shape[0], -1) * scale[:, None]).reshape(w.shape)
    b_conv = torch.zeros(w.shape[0], device=w.device) if conv.bias is None else conv.bias.detach().float()
    bf = scale * b_conv + bn.bias.detach().float() - bn.weight.detach().float() * bn.running_mean.detach().float() / torch.sqrt(
        bn.running_var.detach().float() + bn.eps)
    return wf, bf
