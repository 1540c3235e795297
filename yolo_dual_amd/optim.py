"""Flat-arena SGD(nesterov) + EMA: the optimizer step of the hot path as ONE fused HIP kernel per parameter run.

Semantics restate ``smart_optimizer(model, 'SGD', lr, momentum, decay)`` (utils/torch_utils.py:318-346: three groups —
biases / BN weights without decay, other weights with decay; torch.optim.SGD(nesterov=True)) and ``ModelEMA``
(utils/torch_utils.py:404-428: every float entry of the state_dict, decay ``d = 0.9999 (1 - exp(-n/2000))``).

MI355X-first layout: parameters, gradients, momentum and the EMA shadow each live in one contiguous f32 arena
``[weights with decay | BN weights | biases | float buffers]``.  ``p.data`` / ``p.grad`` of every parameter are views
of the arenas (conv weights keep their KRSC physical layout), so
  * the wgrad / BN-backward kernels write gradients straight into the arena,
  * ``zero_grad`` is one memset, the step is one streaming kernel per run, and
  * data-parallel all-reduce works on a few large contiguous ranges (yolo_dual_amd.parallel).
Parameters whose gradient was not produced in the current step (the reference's dead head layers) are
skipped exactly like torch.optim.SGD skips ``grad is None``: no weight decay, no momentum update."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import config
from .tape import _p, _stream, zero_


def _is_bn(m: nn.Module) -> bool:
    return "Norm" in type(m).__name__ or isinstance(m, nn.modules.batchnorm._BatchNorm)


class FlatSGDEMA(torch.optim.Optimizer):
    def __init__(self, model: nn.Module, lr: float = 0.01, momentum: float = 0.937, weight_decay: float = 5e-4,
                 ema: bool = True, ema_decay: float = 0.9999, ema_tau: float = 2000.0, ema_updates: int = 0):
        g0, g1, g2 = [], [], []          # decay weights, BN weights, biases   (names kept for the arena order)
        seen = set()
        for mod in model.modules():
            for pname, p in mod.named_parameters(recurse=False):
                if id(p) in seen:
                    continue
                seen.add(id(p))
                if pname == "bias":
                    g2.append(p)
                elif pname == "weight" and _is_bn(mod):
                    g1.append(p)
                else:
                    g0.append(p)
        # float buffers, all running means first, then all running variances (module order inside each): sibling
        # convolutions keep adjacent statistics, which lets CSP blocks run cv1|cv2 as one fused conv
        named = [(k, b) for k, b in model.named_buffers() if b.dtype.is_floating_point]
        bufs = ([b for k, b in named if k.endswith("running_mean")] + [b for k, b in named if k.endswith("running_var")] +
                [b for k, b in named if not (k.endswith("running_mean") or k.endswith("running_var"))])
        self.model = model
        self._groups3 = (g0, g1, g2)
        self._bufs = bufs
        dev = next(model.parameters()).device
        sizes = [sum(p.numel() for p in g) for g in (g0, g1, g2)]
        self.n_params = sum(sizes)
        self.n_total = self.n_params + sum(b.numel() for b in bufs)
        self.params_arena = torch.empty(self.n_total, dtype=torch.float32, device=dev)
        self.grads_arena = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.mom_arena = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.ema_arena = torch.empty(self.n_total, dtype=torch.float32, device=dev) if ema else None
        self._slots: List[Tuple[nn.Parameter, int, int, int]] = []      # (param, offset, numel, group)
        off = 0
        for gi, g in enumerate((g0, g1, g2)):
            for p in g:
                n = p.numel()
                self._rehome(p, off, n)
                self._slots.append((p, off, n, gi))
                off += n
        self._buf_slots = []
        for b in bufs:
            n = b.numel()
            view = self.params_arena[off:off + n].view(b.shape)
            view.copy_(b.data)
            b.data = view
            self._buf_slots.append((b, off, n))
            off += n
        if ema:
            self.ema_arena.copy_(self.params_arena)
        self.ema_decay, self.ema_tau, self.updates = ema_decay, ema_tau, ema_updates
        self._has_buf: Dict[int, bool] = {}
        # torch.optim.Optimizer plumbing (lr schedulers read/write param_groups[*]['lr']); group order follows
        # smart_optimizer: biases, decay weights, BN weights
        defaults = dict(lr=lr, momentum=momentum, nesterov=True, weight_decay=0.0)
        super().__init__([{"params": g2 or [torch.nn.Parameter(torch.zeros(0, device=dev))]},
                          {"params": g0, "weight_decay": weight_decay},
                          {"params": g1, "weight_decay": 0.0}], defaults)
        config.bump_weight_epoch()

    # ------------------------------------------------------------------
    def _rehome(self, p: nn.Parameter, off: int, n: int) -> None:
        flat = self.params_arena[off:off + n]
        gflat = self.grads_arena[off:off + n]
        if p.dim() == 4:
            O, I, kh, kw = p.shape
            view = flat.view(O, kh, kw, I).permute(0, 3, 1, 2)          # KRSC physical, OIHW logical
            gview = gflat.view(O, kh, kw, I).permute(0, 3, 1, 2)
        else:
            view, gview = flat.view(p.shape), gflat.view(p.shape)
        view.copy_(p.data)
        p.data = view
        p.grad = gview
        p._ydl_touched = False

    def reattach(self) -> None:
        """(re)bind ``p.grad`` to the arena (needed if foreign code set grads to None)"""
        for p, off, n, _g in self._slots:
            if p.grad is None or p.grad.data_ptr() != self.grads_arena.data_ptr() + 4 * off:
                gflat = self.grads_arena[off:off + n]
                p.grad = gflat.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2) if p.dim() == 4 \
                    else gflat.view(p.shape)

    def zero_grad(self, set_to_none: bool = False) -> None:   # grads stay attached to the arena
        zero_(self.grads_arena)
        self.reattach()
        for p, *_ in self._slots:
            p._ydl_touched = False

    def live_ranges(self) -> List[Tuple[int, int]]:
        """contiguous arena ranges [a, b) of parameters that received a gradient this step"""
        out: List[Tuple[int, int]] = []
        for p, off, n, _g in self._slots:
            if getattr(p, "_ydl_touched", False):
                if out and out[-1][1] == off:
                    out[-1] = (out[-1][0], off + n)
                else:
                    out.append((off, off + n))
        return out

    # ------------------------------------------------------------------ graph-capturable step
    def ensure_hyper(self) -> None:
        """create the device hyper-parameter vector and its pinned staging ring.  Callers that route allocations to a private pool
        (HIP-graph capture, launch-list recording) call this BEFORE they do: a persistent tensor must not land on an address the
        captured / recorded step uses as scratch"""
        if getattr(self, "_hyper_ring", None) is None:
            # ring of pinned staging buffers, each guarded by the event of its last H2D copy: the host never rewrites a
            # buffer whose copy may still be queued behind a graph replay (one reused buffer raced with the next step's write)
            self._hyper_ring = [(torch.empty(7, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self._hyper_slot = 0
            self._hyper_used = [False] * 8
            self._hyper_dev = torch.empty(7, dtype=torch.float32, device=self.params_arena.device)

    def prepare_step(self, grad_scale: float = 1.0) -> None:
        """host half of a step: advance the EMA counter and push {lr, momentum, wd, grad_scale, ema decay} to the device
        vector the captured kernels read (call OUTSIDE the graph, before replaying it)"""
        lr_bias, lr_w, lr_bn = (g["lr"] for g in self.param_groups)
        d = 0.0
        if self.ema_arena is not None:
            self.updates += 1
            d = self.ema_decay * (1.0 - math.exp(-self.updates / self.ema_tau))
        self.ensure_hyper()
        i = self._hyper_slot
        self._hyper_slot = (i + 1) % len(self._hyper_ring)
        h, ev = self._hyper_ring[i]
        if self._hyper_used[i]:
            ev.synchronize()              # normally long complete (8 steps ago)
        h[0], h[1], h[2] = lr_w, lr_bn, lr_bias
        h[3], h[4] = self.param_groups[1]["momentum"], self.param_groups[1]["weight_decay"]
        h[5], h[6] = grad_scale, d
        self._hyper_dev.copy_(h, non_blocking=True)
        ev.record()
        self._hyper_used[i] = True

    def _runs(self, commit: bool) -> List[List]:
        """maximal runs of consecutive slots with identical (touched, group, first-step) state; ``commit`` marks the touched
        parameters as having a momentum buffer from now on"""
        runs: List[List] = []
        for p, off, n, gi in self._slots:
            touched = bool(getattr(p, "_ydl_touched", False))
            first = touched and not self._has_buf.get(id(p), False)
            key = (touched, gi, first)
            if runs and runs[-1][0] == key and runs[-1][2] == off:
                runs[-1][2] = off + n
            else:
                runs.append([key, off, off + n])
            if touched and commit:
                self._has_buf[id(p)] = True
        return runs

    def _run_rows(self, runs) -> tuple:
        """rows {offset, n_decay, n_params, n_total, lr index, flags} of ydl_sgd_ema_step_multi"""
        rows = []
        for (touched, gi, first), a, b in runs:
            n = b - a
            if touched:
                rows.append((a, n if gi == 0 else 0, n, n, gi, (1 if gi == 0 else 0) | (2 if first else 0)))
            elif self.ema_arena is not None:
                rows.append((a, 0, 0, n, 0, 0))
        if self.ema_arena is not None and self.n_total > self.n_params:
            rows.append((self.n_params, 0, 0, self.n_total - self.n_params, 0, 0))
        return tuple(rows)

    def ensure_runs_table(self) -> None:
        """device table of the CURRENT run structure (as the last backward left the touched flags), created now: call after the
        warm-up steps and before entering a private pool, next to ``ensure_hyper``"""
        sig = self._run_rows(self._runs(commit=False))
        if sig:
            self._runs_dev = self._runs_table(sig)

    def _runs_table(self, sig: tuple) -> tuple:
        """(signature, device table) for a run structure.  Tables are cached per signature and never freed: a recorded launch list or
        a captured graph holds the raw device pointer of the table it was recorded with, and a later eager step with another
        structure (a freeze change, a different touched set) must not hand that block back to the allocator"""
        cache = self.__dict__.setdefault("_runs_cache", {})
        tab = cache.get(sig)
        if tab is None:
            n_tot = self.n_total
            for off, _nd, _np, n, _gi, _fl in sig:      # rows index the arenas unchecked in the kernel: validate them here
                if off < 0 or n < 0 or off + n > n_tot:
                    raise RuntimeError(f"optimizer run table row ({off}, {n}) outside the arenas ({n_tot})")
            tab = (sig, torch.tensor(sig, dtype=torch.int64).to(self.params_arena.device))
            cache[sig] = tab
        return tab

    @torch.no_grad()
    def step_device_hyper(self) -> None:
        """device half: the same kernels as ``step`` but reading hyper-parameters from ``_hyper_dev`` (capturable)"""
        st = _stream()
        pa, ga, ma, ea = self.params_arena, self.grads_arena, self.mom_arena, self.ema_arena
        use_ema = 1 if ea is not None else 0
        runs = self._runs(commit=True)
        hp = _p(self._hyper_dev)
        # one launch for all runs.  The table is a persistent device tensor keyed by the run structure (constant once every live
        # parameter has its momentum buffer); inside a recording / capture pass a NEW table must not be allocated (private pool),
        # so a miss there takes the per-run launches below
        rows = self._run_rows(runs)
        sig = rows
        tab = getattr(self, "_runs_dev", None)
        if (tab is None or tab[0] != sig) and rows and L.recorder() is None and not torch.cuda.is_current_stream_capturing():
            tab = self._runs_table(sig)
            self._runs_dev = tab
        if tab is not None and tab[0] == sig:
            if rows:
                L.call("ydl_sgd_ema_step_multi", _p(pa), _p(ga), _p(ma), _p(ea) if ea is not None else None, _p(tab[1]), len(rows),
                       max(r[3] for r in rows), hp, use_ema, st)
            config.bump_weight_epoch()
            return
        for (touched, gi, first), a, b in runs:
            n = b - a
            eptr = _p(ea[a:b]) if ea is not None else None
            if touched:
                L.call("ydl_sgd_ema_step_dev", _p(pa[a:b]), _p(ga[a:b]), _p(ma[a:b]), eptr, n if gi == 0 else 0, n, n,
                       hp, gi, 1 if gi == 0 else 0, 1 if first else 0, use_ema, st)
            elif ea is not None:
                L.call("ydl_sgd_ema_step_dev", _p(pa[a:b]), _p(ga[a:b]), _p(ma[a:b]), eptr, 0, 0, n, hp, 0, 0, 0, 1, st)
        if ea is not None and self.n_total > self.n_params:
            a, b = self.n_params, self.n_total
            L.call("ydl_sgd_ema_step_dev", _p(pa[a:b]), _p(ga), _p(ma), _p(ea[a:b]), 0, 0, b - a, hp, 0, 0, 0, 1, st)
        config.bump_weight_epoch()

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        lr_bias, lr_w, lr_bn = (g["lr"] for g in self.param_groups)
        mom = self.param_groups[1]["momentum"]
        wd = self.param_groups[1]["weight_decay"]
        lrs = (lr_w, lr_bn, lr_bias)
        d = -1.0
        if self.ema_arena is not None:
            self.updates += 1
            d = self.ema_decay * (1.0 - math.exp(-self.updates / self.ema_tau))
        st = _stream()
        # maximal runs of consecutive slots with identical (touched, group, first-step) state
        runs: List[List] = []
        for p, off, n, gi in self._slots:
            touched = bool(getattr(p, "_ydl_touched", False))
            first = touched and not self._has_buf.get(id(p), False)
            key = (touched, gi, first)
            if runs and runs[-1][0] == key and runs[-1][2] == off:
                runs[-1][2] = off + n
            else:
                runs.append([key, off, off + n])
            if touched:
                self._has_buf[id(p)] = True
        pa, ga, ma = self.params_arena, self.grads_arena, self.mom_arena
        ea = self.ema_arena
        for (touched, gi, first), a, b in runs:
            n = b - a
            eptr = _p(ea[a:b]) if ea is not None else None
            if touched:
                nd = n if gi == 0 else 0
                L.call("ydl_sgd_ema_step", _p(pa[a:b]), _p(ga[a:b]), _p(ma[a:b]), eptr, nd, n, n,
                       lrs[gi], lrs[gi], mom, wd if gi == 0 else 0.0, grad_scale, 1 if first else 0, d, st)
            elif ea is not None:
                L.call("ydl_sgd_ema_step", _p(pa[a:b]), _p(ga[a:b]), _p(ma[a:b]), eptr, 0, 0, n,
                       0.0, 0.0, mom, 0.0, 1.0, 0, d, st)
        if ea is not None and self.n_total > self.n_params:
            a, b = self.n_params, self.n_total
            L.call("ydl_sgd_ema_step", _p(pa[a:b]), _p(ga), _p(ma), _p(ea[a:b]), 0, 0, b - a,
                   0.0, 0.0, mom, 0.0, 1.0, 0, d, st)
        config.bump_weight_epoch()
        return None

    # ------------------------------------------------------------------ checkpoint state (seg_diceloss_yolov5.py:1204-1212)
    def state_dict(self):
        """what ``'optimizer': optimizer.state_dict()`` of a checkpoint holds here: hyper-parameters per group, the momentum
        arena, which parameters already own a momentum buffer, and the EMA update counter (plain tensors / numbers only, so
        the file loads with ``torch.load(..., weights_only=True)``)"""
        return {"format": "ydl-flat-sgd-ema-1",
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "momentum": self.mom_arena.detach().cpu().clone(),
                "has_momentum": torch.tensor([bool(self._has_buf.get(id(p), False)) for p, *_ in self._slots]),
                "updates": int(self.updates)}

    def load_state_dict(self, sd) -> None:
        if sd.get("format") != "ydl-flat-sgd-ema-1":
            raise ValueError("optimizer state was not written by yolo_dual_amd.FlatSGDEMA")
        if sd["momentum"].numel() != self.mom_arena.numel():
            raise ValueError("optimizer state belongs to a different model (arena size differs)")
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            g.update(gs)
        self.mom_arena.copy_(sd["momentum"].to(self.mom_arena.device))
        for (p, *_), h in zip(self._slots, sd["has_momentum"].tolist()):
            self._has_buf[id(p)] = bool(h)
        self.updates = int(sd.get("updates", 0))

    def load_ema_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """restore the EMA shadow from a ``state_dict`` (smart_resume: ``ema.ema.load_state_dict(ckpt['ema']...)``)"""
        if self.ema_arena is None:
            raise RuntimeError("EMA disabled")
        cur = self.ema_state_dict()
        by_ptr = {}
        for p, off, n, _g in self._slots:
            by_ptr[p.data_ptr()] = (off, n)
        for b, off, n in self._buf_slots:
            by_ptr[b.data_ptr()] = (off, n)
        for k, v in self.model.state_dict().items():
            slot = by_ptr.get(v.data_ptr())
            if slot is None or not v.dtype.is_floating_point or k not in sd or tuple(sd[k].shape) != tuple(cur[k].shape):
                continue
            off, n = slot
            src = sd[k].float().to(self.ema_arena.device)
            if v.dim() == 4:
                src = src.permute(0, 2, 3, 1).contiguous()
            self.ema_arena[off:off + n].copy_(src.reshape(-1))

    # ------------------------------------------------------------------ EMA access (ModelEMA.ema equivalent)
    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        """state_dict of the EMA shadow (same keys as model.state_dict(); integer buffers are copied as is)."""
        if self.ema_arena is None:
            raise RuntimeError("EMA disabled")
        by_ptr = {}
        for p, off, n, _g in self._slots:
            by_ptr[p.data_ptr()] = (off, n)
        for b, off, n in self._buf_slots:
            by_ptr[b.data_ptr()] = (off, n)
        out = {}
        for k, v in self.model.state_dict().items():
            slot = by_ptr.get(v.data_ptr())
            if slot is None or not v.dtype.is_floating_point:
                out[k] = v.clone()
                continue
            off, n = slot
            flat = self.ema_arena[off:off + n]
            if v.dim() == 4:
                O, I, kh, kw = v.shape
                out[k] = flat.view(O, kh, kw, I).permute(0, 3, 1, 2).clone()
            else:
                out[k] = flat.view(v.shape).clone()
        return out


def smart_optimizer(model: nn.Module, name: str = "SGD", lr: float = 0.001, momentum: float = 0.9,
                    decay: float = 1e-5, ema: bool = True) -> FlatSGDEMA:
    """utils/torch_utils.py:318-346 signature; only the SGD-nesterov branch (what the seg scripts use) has a HIP path."""
    if name != "SGD":
        raise NotImplementedError(f"optimizer {name}: the fused HIP step implements the scripts' SGD(nesterov) only")
    return FlatSGDEMA(model, lr=lr, momentum=momentum, weight_decay=decay, ema=ema)
