"""nn.Module facade with the reference's names, constructor signatures and state_dict layout; all compute runs
through yolo_dual_amd.tape on the HIP kernels.

Both spellings of each block are served:
  * seg-script blocks  — unet-lite/yolo5-seg/seg_diceloss_yolov5.py:388-507, yolov8/seg_jaccardloss_yolov8.py:401-414,
    unet-lite/yolo9-seg/seg_diceloss_yolov9.py:451-510, segment/train.py:50-210
  * stock YOLOv5 blocks — models/common.py:38-64 (Conv), :115-125 (Bottleneck), :161-172 (C3), :223-238 (SPPF),
    :310-317 (Concat)
Every module accepts either a tape ``Var`` (inside a model: one taped region for the whole network) or plain
(N,C,H,W) f32 tensors (stand-alone use: the call becomes its own taped region behind one autograd node).
"""
from __future__ import annotations

import ctypes

import math
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import _lib as L
from . import config
from .tape import Tape, Var, round_up, _p, _stream, zero_

__all__ = ["autopad", "Conv", "C3", "C3Common", "Bottleneck", "C2f", "C3k2", "GAM", "SPPF", "Concat", "Upsample",
           "BasicBlock", "BottleneckBlock", "SegmentHead", "run_region", "Linear", "DCNv3", "DCNV3_YoLo", "Bottleneck_DCNV3",
           "C3_DCNV3"]


def autopad(k, p=None, d=1):
    """models/common.py:38-44 / seg_diceloss_yolov5.py:381-385."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


# ----------------------------------------------------------------------------------------------------------
# taped region behind ONE torch.autograd node
# ----------------------------------------------------------------------------------------------------------
class _Region(torch.autograd.Function):
    """forward(fn, n_in, *tensors): tensors = n_in external inputs followed by the parameters the region may touch
    (they are inputs only so that autograd schedules this node; their gradients are written into ``p.grad`` by the
    kernels and ``None`` is returned for them)."""

    @staticmethod
    def forward(ctx, fn, n_in, *tensors):
        ins = tensors[:n_in]
        dev = ins[0].device
        if dev.type != "cuda":
            raise RuntimeError("yolo_dual_amd runs on the GPU only: there is no CPU fallback for the HIP kernels "
                               "(move the module and its inputs to cuda)")
        record = any(ctx.needs_input_grad[2:])
        tape = Tape(config.compute_dtype(), dev, fn.training, record)
        tape._slab_hint = getattr(fn, "_ydl_slab_need", 0)
        refresh_weights(fn, tape)
        vs = [tape.input_nchw(t, lazy=not ctx.needs_input_grad[2 + i]) for i, t in enumerate(ins)]
        for i, v in enumerate(vs):
            v.need = bool(ctx.needs_input_grad[2 + i])
        res = fn._fwd(tape, vs if fn.takes_list else vs[0]) if n_in else None
        if isinstance(res, torch.Tensor):           # region already produced its external output (softmax head)
            out_t, out_v = res, None
        else:
            out_v = tape.materialize(res)           # a lazily up-sampled result becomes real at the region edge
            out_t = tape.export_nchw(out_v)
        ctx.tape, ctx.vs, ctx.out_v, ctx.fn = tape, vs, out_v, fn
        ctx.n_extra = len(tensors) - n_in
        return out_t

    @staticmethod
    def backward(ctx, gout):
        tape: Tape = ctx.tape
        if tape is None:
            raise RuntimeError("second backward through a taped region: its activations were released by the first one "
                               "(retain_graph is not supported; run the forward again)")
        if ctx.out_v is not None:
            tape.seed_grad_nchw(ctx.out_v, gout)
        else:
            ctx.fn._seed_external(tape, gout)
        tape.run_backward()
        gins = [tape.grad_nchw(v) if v.need else None for v in ctx.vs]
        if tape._slab_total:
            ctx.fn._ydl_slab_need = tape._slab_total + tape._slab_total // 8 + 1024      # next step: one slab, one memset
        ctx.tape = None
        return (None, None, *gins, *([None] * ctx.n_extra))


def refresh_weights(fn: nn.Module, tape: Tape) -> None:
    """one batched launch re-deriving the compute-layout weights of every stale Conv under ``fn``"""
    convs = getattr(fn, "_ydl_convs", None)
    if convs is None:
        convs = [m for m in fn.modules() if isinstance(m, Conv) and not m.depthwise]
        fn._ydl_convs = convs
        fn._ydl_csp = [m for m in fn.modules() if isinstance(getattr(m, "cv1", None), Conv) and isinstance(getattr(m, "cv2", None), Conv)]
    # sibling pairs that ran fused last time are prepared as ONE matrix (their masters are adjacent), their members not at all
    pairs = [b._pair for b in fn._ydl_csp
             if getattr(b, "_pair", None) is not None and b._pair._wcache.get("key") is not None and b._pair.still_valid()]
    if pairs:
        members = {id(m) for pr in pairs for m in (pr.a, pr.b)}
        units = [m for m in convs if id(m) not in members] + pairs
    else:
        units = convs
    stale = [m for m in units if m._wcache.get("key") != m._wkey(tape)]
    if len(stale) < 2:
        return
    rows, keep = [], []
    for m in stale:
        master = m._master_krsc()
        w, wt = m._wbuffers(tape, master)
        keep.append(master)
        rows.append([master.data_ptr(), w.data_ptr(), wt.data_ptr(), m.c2, m.k * m.k, m.c1, 0, 0])
    sig = tuple(r[0] for r in rows) + tuple(r[1] for r in rows) + (tape.dname,)
    cache = getattr(fn, "_ydl_wdesc", None)
    if cache is None or cache[0] != sig:
        desc = torch.tensor(rows, dtype=torch.int64).to(tape.device)
        fn._ydl_wdesc = (sig, desc)
    desc = fn._ydl_wdesc[1]
    L.call("ydl_weight_prep_batched", tape.dt, _p(desc), len(rows), _stream())
    for m in stale:
        m._wcache["key"] = m._wkey(tape)


def run_region(fn: nn.Module, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
    if torch.is_grad_enabled():
        allp = getattr(fn, "_ydl_params", None)          # walking the module tree costs ~0.5 ms per call on this model
        if allp is None or allp[0] != len(fn._parameters) + sum(1 for _ in fn.children()):
            allp = fn._ydl_params = (len(fn._parameters) + sum(1 for _ in fn.children()), list(fn.parameters()))
        params = [p for p in allp[1] if p.requires_grad]
    else:
        params = []
    out = _Region.apply(fn, len(inputs), *inputs, *params)
    lazy = getattr(getattr(out.grad_fn, "tape", None), "lazy_out", None)
    if lazy is not None:                # replicated output: let SegmentationLoss work at the stored resolution
        lazy.version = out._version
        out._ydl_lazy = lazy
    return out


class YdlModule(nn.Module):
    """base: dispatch between taped (Var) and stand-alone (torch.Tensor) calls"""
    takes_list = False

    def forward(self, x, *a, **kw):
        if isinstance(x, Var):
            return self._fwd(x.tape, x)
        if isinstance(x, (list, tuple)) and x and isinstance(x[0], Var):
            return self._fwd(x[0].tape, x)
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        return run_region(self, xs)

    def _fwd(self, tape: Tape, x):
        raise NotImplementedError

    def _seed_external(self, tape: Tape, gout: torch.Tensor) -> None:
        raise NotImplementedError


# ----------------------------------------------------------------------------------------------------------
# Conv = Conv2d(bias=False) -> BatchNorm2d -> SiLU
# ----------------------------------------------------------------------------------------------------------
def _launch_wgrad(tape: Tape, gp, x_ptr, dy_ptr, dw_ptr, st, fuse=None) -> bool:
    """dW += dy^T * im2col(x): f32 atomics (throughput mode) or the deterministic slab form (parity mode / config).
    ``fuse`` = (wt_ptr, dx_ptr, lddx, accumulate) of the SAME layer's input gradient: where the one-pass kernel applies (the
    HBM-bound 128 -> 128 1x1 layers, ydl_conv_bwd_pw) both gradients come from one launch and the call returns True."""
    if (fuse is not None and not config.deterministic(tape.dname) and config.fuse_pw_backward()
            and L.lib().ydl_conv_bwd_pw_supported(gp, tape.dt)):
        wt_ptr, dx_ptr, lddx, acc = fuse
        L.call("ydl_conv_bwd_pw", gp, tape.dt, x_ptr, dy_ptr, wt_ptr, dx_ptr, lddx, acc, dw_ptr, st)
        return True
    if config.deterministic(tape.dname):
        nbytes = L.lib().ydl_conv_wgrad_ws_bytes(gp, tape.dt)
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=tape.device)
        L.call("ydl_conv_wgrad_det", gp, tape.dt, x_ptr, dy_ptr, dw_ptr, _p(ws), st)
        tape._keep.append(ws)          # may be consumed on the side stream: lives until the streams are joined
    else:
        L.call("ydl_conv_wgrad", gp, tape.dt, x_ptr, dy_ptr, dw_ptr, st)
    return False


class _BNHolder(nn.BatchNorm2d):
    """Parameter/buffer holder with nn.BatchNorm2d's state_dict layout.  ``num_batches_tracked`` is advanced on the
    host and flushed into the buffer whenever the state is read, so the hot loop launches no extra kernel."""

    def __init__(self, c):
        super().__init__(c)
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(_BNHolder._flush_hook)

    @staticmethod
    def _flush_hook(module, prefix, keep_vars):
        module.flush()

    def flush(self):
        if self._nbt_pending:
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0

    def forward(self, x):  # pragma: no cover - never used: BN is fused into the conv epilogue + apply kernels
        raise RuntimeError("BatchNorm is fused into yolo_dual_amd.Conv; call the Conv module")


def _act_code(act) -> int:
    if act is True or isinstance(act, nn.SiLU):
        return L.ACT_SILU
    if act is False or act is None or isinstance(act, nn.Identity):
        return L.ACT_NONE
    if isinstance(act, nn.ReLU):
        return L.ACT_RELU
    raise NotImplementedError(f"activation {act!r} is not supported by the HIP path (SiLU/ReLU/Identity)")


class Conv(YdlModule):
    """``Conv(c1, c2, k=1, s=1, p=None, g=1, act=True)`` (seg scripts) and
    ``Conv(c1, c2, k=1, s=1, p=None, g=1, d=1, act=True)`` (models/common.py): the 7th positional argument is taken
    as ``act`` when it is a bool/Module and as the dilation when it is an int > 0 that is not a bool."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d_or_act=True, act=None):
        super().__init__()
        if act is None:
            if isinstance(d_or_act, bool) or isinstance(d_or_act, nn.Module) or d_or_act is None:
                act, d = d_or_act, 1
            else:
                d, act = int(d_or_act), True
        else:
            d = int(d_or_act) if not isinstance(d_or_act, bool) else 1
        if not all(isinstance(v, int) and not isinstance(v, bool) for v in (c1, c2, k, s, g)):
            raise TypeError(f"Conv arguments must be int: c1={c1}({type(c1)}), c2={c2}({type(c2)})")
        if g <= 0 or c1 % g != 0:
            raise ValueError(f"groups g={g} must be positive and divide c1={c1}")
        self.depthwise = g > 1 and g == c1 == c2
        if d != 1 or (g != 1 and not self.depthwise):
            raise NotImplementedError("the HIP path implements groups=1 (implicit GEMM) and groups=c1=c2 (depth-wise), dilation=1")
        if self.depthwise and (s != 1 or k not in (1, 3, 5, 7) or autopad(k, p) != k // 2):
            raise NotImplementedError("depth-wise Conv: stride 1, k in {1,3,5,7}, 'same' padding (the DCNv3 dw_conv branch)")
        self.c1, self.c2, self.k, self.s = c1, c2, k, s
        self.p = autopad(k, p)
        self.conv = nn.Conv2d(c1, c2, k, s, self.p, groups=g, bias=False)
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.bn = _BNHolder(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())
        self.act_code = _act_code(self.act)
        self._wcache = {}

    # -- parameters in compute layout -------------------------------------------------------------------
    def _master_krsc(self) -> torch.Tensor:
        w = self.conv.weight.detach().permute(0, 2, 3, 1)
        if not w.is_contiguous():                      # someone re-assigned .data in OIHW order: re-home it
            self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
            w = self.conv.weight.detach().permute(0, 2, 3, 1)
            if not w.is_contiguous():
                w = w.contiguous()
        return w

    def _wkey(self, tape: Tape):
        wp = self.conv.weight
        return (tape.dname, wp.data_ptr(), wp._version, config.weight_epoch())

    def mark_step(self, tape: Tape) -> None:
        if tape.train:
            self.bn._nbt_pending += 1

    def _wbuffers(self, tape: Tape, master: torch.Tensor):
        kk = self.k * self.k
        dev = master.device
        w = self._wcache.get("w")
        if w is None or self._wcache.get("dname") != tape.dname or w.device != dev:
            w = torch.empty((self.c2, kk, round_up(self.c1, 8)), dtype=tape.tdt, device=dev)
            wt = torch.empty((self.c1, kk, round_up(self.c2, 8)), dtype=tape.tdt, device=dev)
            self._wcache.update(w=w, wt=wt, dname=tape.dname, key=None)
        return self._wcache["w"], self._wcache["wt"]

    def compute_weights(self, tape: Tape):
        """compute-dtype copies w [Cout][taps][Cin_p] / wt [Cin][taps][Cout_p] of the f32 KRSC master weight, refreshed
        when the parameter changed (torch version counter, or the epoch the fused optimizer bumps)"""
        key = self._wkey(tape)
        if self._wcache.get("key") == key:
            return self._wcache["w"], self._wcache["wt"]
        master = self._master_krsc()
        w, wt = self._wbuffers(tape, master)
        L.call("ydl_weight_prep", tape.dt, _p(master), _p(w), _p(wt), self.c2, self.k * self.k, self.c1, _stream())
        self._wcache["key"] = key
        return w, wt

    def coeffs(self, device):
        cp = round_up(self.c2, 8)
        buf = torch.empty((4, cp), dtype=torch.float32, device=device)
        if cp != self.c2:
            zero_(buf)
        return {"mean": buf[0], "invstd": buf[1], "scale": buf[2], "shift": buf[3]}

    def _grad_of(self, p: nn.Parameter) -> torch.Tensor:
        if p.grad is None:
            p.grad = torch.zeros_like(p)       # preserves the KRSC (channels_last) strides of the weight
        return p.grad

    def grad_slot(self, tape: Tape, which: str):
        """gradient storage of gamma / beta; ``touch_bn`` must be called AFTER the kernel writing it is enqueued (the
        data-parallel hook may launch the bucket's all-reduce from inside mark_touched)"""
        p = self.bn.weight if which == "gamma" else self.bn.bias
        return self._grad_of(p), 1

    def touch_bn(self) -> None:
        if self.bn.weight.requires_grad:
            config.mark_touched(self.bn.weight)
        if self.bn.bias.requires_grad:
            config.mark_touched(self.bn.bias)

    def trainable(self):
        """(weight, BN weight, BN bias) ``requires_grad`` flags: a frozen parameter (``--freeze``, seg_diceloss_yolov5.py:955-959)
        gets no gradient kernel, is never marked touched and is therefore skipped by the optimizer like torch.optim.SGD skips
        ``grad is None``"""
        return (self.conv.weight.requires_grad, self.bn.weight.requires_grad, self.bn.bias.requires_grad)

    def splittable(self) -> bool:
        """the weight gradient can be written per input-channel block (``wgrad(col0=...)``): dense KRSC storage"""
        return self.c1 % 8 == 0 and self.conv.weight.detach().permute(0, 2, 3, 1).is_contiguous()

    def wgrad(self, tape: Tape, gp, x: Var, dy: Var, st, col0: int = 0, final: bool = True, fuse=None) -> bool:
        """``col0``: first input channel of the block this call covers (gp.ldw = total padded Cin then); ``final``: the
        last launch writing this parameter's gradient (only then may the data-parallel hook see it); ``fuse``: see
        ``_launch_wgrad`` — returns True when the input gradient was produced by the same launch"""
        p = self.conv.weight
        g = self._grad_of(p)
        gk = g.permute(0, 2, 3, 1)
        kk = self.k * self.k
        cin_p = round_up(self.c1, 8)
        if cin_p == self.c1 and gk.is_contiguous():
            done = _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), ctypes.c_void_p(gk.data_ptr() + 4 * col0), st, fuse)
            if final:
                config.mark_touched(p)
            return done
        assert col0 == 0 and final
        tmp = zero_(torch.empty((self.c2, kk, cin_p), dtype=torch.float32, device=g.device), st)
        _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), _p(tmp), st)
        if gk.is_contiguous():
            L.call("ydl_wgrad_unpad", _p(tmp), _p(gk), self.c2, kk, self.c1, 1, st)
        else:                                           # exotic grad layout: let torch place it (cold path)
            g.add_(tmp[:, :, :self.c1].view(self.c2, self.k, self.k, self.c1).permute(0, 3, 1, 2))
        config.mark_touched(p)
        return False

    # -- forward ----------------------------------------------------------------------------------------
    # -- depth-wise variant (weight [C,1,k,k]: the KRSC physical layout is [C][k*k]) -------------------------
    def master_dw(self) -> torch.Tensor:
        w = self.conv.weight.detach().permute(0, 2, 3, 1)
        return w if w.is_contiguous() else w.contiguous()

    def grad_dw(self) -> torch.Tensor:
        g = self._grad_of(self.conv.weight).permute(0, 2, 3, 1)
        if not g.is_contiguous():
            raise RuntimeError("depth-wise weight gradient must be KRSC-contiguous")
        return g

    def _fwd(self, tape: Tape, x: Var, out: Optional[Var] = None, res: Optional[Var] = None,
             res_mode: int = L.RES_NONE, act_code: Optional[int] = None) -> Var:
        self.mark_step(tape)
        act = self.act_code if act_code is None else act_code
        if self.depthwise:
            if res is not None:
                raise NotImplementedError("depth-wise Conv with a fused residual")
            y = tape.dwconv_bn_act(x, self, act)
            return y if out is None else tape.copy(y, out)
        if getattr(self, "_fused", None) is not None and not tape.train:
            y = tape.conv_bias_act(x, self, act, res=res, res_mode=res_mode)
            return y if out is None else tape.copy(y, out)
        if (x.ext_src is not None and x.real is None and not x.need and config.stem_s2d() and res is None and self.s > 1
                and self.k % self.s == 0 and self.p % self.s == 0 and self.c1 * self.s * self.s <= 16
                and x.H % self.s == 0 and x.W % self.s == 0):
            # stem on the raw input: conv(k, s, p) == conv(k/s, 1, p/s) over the space-to-depth form (K = 9*16, not 36*8)
            ad = _S2DStem(self)
            return tape.conv_bn_act(tape.input_s2d(x.ext_src, self.s), ad, 1, ad.p, act, out=out)
        return tape.conv_bn_act(x, self, self.s, self.p, act, out=out, res=res, res_mode=res_mode)

    # -- inference-time Conv+BN folding (models/common.py:61-64, utils/torch_utils.py:248-269, models/yolo.py:140-148) --
    def fuse(self) -> "Conv":
        """fold the BatchNorm running statistics into the convolution: eval-mode forward becomes act(conv(x, w') + b')"""
        if self.depthwise:
            raise NotImplementedError("fuse() of a depth-wise Conv")
        from .checkpoint import fuse_conv_and_bn
        wf, bf = fuse_conv_and_bn(self.conv, self.bn)
        cp = round_up(self.c2, 8)
        bias = torch.zeros(cp, dtype=torch.float32, device=wf.device)
        bias[:self.c2] = bf
        self._fused = {"w": wf.permute(0, 2, 3, 1).contiguous(), "bias": bias, "ones": torch.ones(cp, dtype=torch.float32, device=wf.device),
                       "cw": {}}
        return self

    def unfuse(self) -> "Conv":
        self._fused = None
        return self

    def forward_fuse(self, x):
        if getattr(self, "_fused", None) is None:
            self.fuse()
        return self.forward(x)

    def _fused_weights(self, tape: Tape):
        f = self._fused
        cw = f["cw"].get(tape.dname)
        if cw is None:
            kk = self.k * self.k
            w = torch.empty((self.c2, kk, round_up(self.c1, 8)), dtype=tape.tdt, device=f["w"].device)
            wt = torch.empty((self.c1, kk, round_up(self.c2, 8)), dtype=tape.tdt, device=f["w"].device)
            L.call("ydl_weight_prep", tape.dt, _p(f["w"]), _p(w), _p(wt), self.c2, kk, self.c1, _stream())
            cw = f["cw"][tape.dname] = (w, wt)
        return cw


class _S2DStem:
    """View of a stem ``Conv`` (k and p multiples of the stride s) as the stride-1 conv it equals over the space-to-depth
    input (ydl_nchw_to_s2d): k/s taps per side, s*s*c1 input channels.  Duck-types the part of ``Conv`` the tape uses;
    weights and gradients go through ydl_weight_prep_s2d / ydl_wgrad_unpack_s2d (same index map as the activation)."""

    def __init__(self, conv: "Conv"):
        self.m = conv
        self.k, self.s, self.p = conv.k // conv.s, 1, conv.p // conv.s
        self.c1, self.c2 = conv.c1 * conv.s * conv.s, conv.c2
        self.bn, self.act_code = conv.bn, conv.act_code

    def splittable(self) -> bool:
        return False

    def compute_weights(self, tape: Tape):
        m = self.m
        key = m._wkey(tape)
        c = m._wcache
        if c.get("s2d_key") != key or c.get("s2d_w") is None:
            if c.get("s2d_w") is None or c["s2d_w"].dtype != tape.tdt:
                c["s2d_w"] = torch.empty((self.c2, self.k * self.k, round_up(self.c1, 8)), dtype=tape.tdt,
                                         device=m.conv.weight.device)
            L.call("ydl_weight_prep_s2d", tape.dt, _p(m._master_krsc()), _p(c["s2d_w"]), m.c2, m.k, m.s, m.c1, _stream())
            c["s2d_key"] = key
        return c["s2d_w"], c["s2d_w"]          # no dgrad: the input is external and needs no gradient

    def coeffs(self, device):
        return self.m.coeffs(device)

    def grad_slot(self, tape: Tape, which: str):
        return self.m.grad_slot(tape, which)

    def touch_bn(self) -> None:
        self.m.touch_bn()

    def trainable(self):
        return self.m.trainable()

    def wgrad(self, tape: Tape, gp, x: Var, dy: Var, st, col0: int = 0, final: bool = True) -> None:
        m = self.m
        p = m.conv.weight
        g = m._grad_of(p)
        gk = g.permute(0, 2, 3, 1)
        tmp = zero_(torch.empty((self.c2, self.k * self.k, round_up(self.c1, 8)), dtype=torch.float32, device=g.device), st)
        _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), _p(tmp), st)
        if gk.is_contiguous():
            L.call("ydl_wgrad_unpack_s2d", _p(tmp), _p(gk), m.c2, m.k, m.s, m.c1, 1, st)
        else:                                           # exotic grad layout: let torch place it (cold path)
            k2, s_ = self.k, m.s
            t6 = tmp[:, :, :self.c1].view(m.c2, k2, k2, s_, s_, m.c1).permute(0, 1, 3, 2, 4, 5).reshape(m.c2, m.k, m.k, m.c1)
            g.add_(t6.permute(0, 3, 1, 2))
        config.mark_touched(p)


class _FusedPair:
    """Two 1x1 Convs that read the SAME input (cv1/cv2 of a CSP block) run as ONE convolution with concatenated output
    channels: the input is read once in forward and wgrad, and dgrad writes the input gradient once instead of
    write + read-modify-write.  Possible without any copy when the two modules' parameters, gradients and BN buffers
    are adjacent in memory — which is how yolo_dual_amd.optim.FlatSGDEMA lays the arenas out; otherwise ``make``
    returns None and the block falls back to two convolutions.  Duck-types the part of ``Conv`` the tape uses."""

    def __init__(self, a: "Conv", b: "Conv"):
        self.a, self.b = a, b
        self.c1, self.c2, self.k, self.s, self.p = a.c1, a.c2 + b.c2, a.k, a.s, a.p
        self.act_code = a.act_code
        self._wcache = {}
        self._views = None

    @staticmethod
    def _adjacent(t1: torch.Tensor, t2: torch.Tensor) -> bool:
        return (t1 is not None and t2 is not None and t1.dtype == t2.dtype and
                t1.data_ptr() + t1.numel() * t1.element_size() == t2.data_ptr())

    @classmethod
    def make(cls, a: "Conv", b: "Conv") -> Optional["_FusedPair"]:
        if (a.k, a.s, a.p, a.c1, a.act_code) != (b.k, b.s, b.p, b.c1, b.act_code) or a.k != 1:
            return None
        if a.c2 % 8 or b.c2 % 8:
            return None
        pairs = [(a.conv.weight, b.conv.weight), (a.bn.weight, b.bn.weight), (a.bn.bias, b.bn.bias),
                 (a.bn.running_mean, b.bn.running_mean), (a.bn.running_var, b.bn.running_var)]
        if not all(cls._adjacent(x.detach(), y.detach()) for x, y in pairs):
            return None
        if not (a.conv.weight.detach().permute(0, 2, 3, 1).is_contiguous() and
                b.conv.weight.detach().permute(0, 2, 3, 1).is_contiguous()):
            return None
        return cls(a, b)

    def still_valid(self) -> bool:
        a, b = self.a, self.b
        key = (a.conv.weight.data_ptr(), b.conv.weight.data_ptr(), a.bn.running_mean.data_ptr())
        if self._views is not None and self._views[0] == key:
            return True
        if _FusedPair.make(a, b) is None:
            return False
        c = self.c2

        def cat1(x, y):
            return torch.as_strided(x.detach(), (c,), (1,))
        w = torch.as_strided(a.conv.weight.detach().permute(0, 2, 3, 1), (c, self.k, self.k, self.c1),
                             (self.k * self.k * self.c1, self.k * self.c1, self.c1, 1))
        bn = _FusedBN()
        bn.weight, bn.bias = cat1(a.bn.weight, b.bn.weight), cat1(a.bn.bias, b.bn.bias)
        bn.running_mean, bn.running_var = cat1(a.bn.running_mean, b.bn.running_mean), cat1(a.bn.running_var, b.bn.running_var)
        bn.eps, bn.momentum = a.bn.eps, a.bn.momentum
        self._views = (key, w, bn)
        return True

    @property
    def bn(self):
        return self._views[2]

    def _master_krsc(self) -> torch.Tensor:
        return self._views[1]

    def _wkey(self, tape: Tape):
        wa, wb = self.a.conv.weight, self.b.conv.weight
        return (tape.dname, wa.data_ptr(), wa._version, wb._version, config.weight_epoch())

    _wbuffers = None        # bound below (shared implementation with Conv)
    compute_weights = None
    coeffs = None

    def mark_step(self, tape: Tape) -> None:
        self.a.mark_step(tape)
        self.b.mark_step(tape)

    def _grads(self, which: str):
        if which == "w":
            return self.a.conv.weight, self.b.conv.weight
        return (self.a.bn.weight, self.b.bn.weight) if which == "gamma" else (self.a.bn.bias, self.b.bn.bias)

    def grads_adjacent(self) -> bool:
        for which in ("w", "gamma", "beta"):
            p1, p2 = self._grads(which)
            if p1.grad is None or p2.grad is None or not self._adjacent(p1.grad, p2.grad):
                return False
        g = self.a.conv.weight.grad
        return g.permute(0, 2, 3, 1).is_contiguous() and self.b.conv.weight.grad.permute(0, 2, 3, 1).is_contiguous()

    def grad_slot(self, tape: Tape, which: str):
        p1, p2 = self._grads(which)
        return torch.as_strided(p1.grad, (self.c2,), (1,)), 1

    def touch_bn(self) -> None:
        for which in ("gamma", "beta"):
            for p in self._grads(which):
                if p.requires_grad:
                    config.mark_touched(p)

    def trainable(self):
        return self.a.trainable()

    def uniform_trainable(self) -> bool:
        """both halves frozen or both trainable (one launch writes both halves' gradients)"""
        return self.a.trainable() == self.b.trainable()

    def splittable(self) -> bool:
        return self.c1 % 8 == 0

    def wgrad(self, tape: Tape, gp, x: Var, dy: Var, st, col0: int = 0, final: bool = True, fuse=None) -> bool:
        p1, p2 = self._grads("w")
        gk = p1.grad.permute(0, 2, 3, 1)
        done = _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), ctypes.c_void_p(gk.data_ptr() + 4 * col0), st, fuse)
        if final:
            config.mark_touched(p1)
            config.mark_touched(p2)
        return done


class _FusedBN:
    pass


_FusedPair._wbuffers = Conv._wbuffers
_FusedPair.compute_weights = Conv.compute_weights
_FusedPair.coeffs = Conv.coeffs


def _csp_forward(blk, tape: Tape, x: Var, add: bool) -> Var:
    """shared by C3 / C3Common / C3k2: cv3(cat(m(cv1 x), cv2 x)) (+x) with cv1|cv2 fused when the memory layout allows.
    Buffer T has 3*c_ channels [a | m_out | right]: the fused conv's two halves are activated into T[0:c_] and
    T[2c_:3c_], the last layer of ``m`` writes T[c_:2c_], and cv3 reads the contiguous slice T[c_:3c_] (no copy)."""
    c_, n = blk.c_, len(blk.m)
    pair = getattr(blk, "_pair", None)
    if pair is None:
        pair = blk._pair = _FusedPair(blk.cv1, blk.cv2)
    fused = (config.fuse_siblings() and x.aligned() and not x.lazy and pair.still_valid() and pair.uniform_trainable() and
             (not tape.record or pair.grads_adjacent()))
    if not fused:
        cat = tape.new(x.N, 2 * c_, x.H, x.W)
        left, right = cat.slice(0, c_), cat.slice(c_, 2 * c_)
        a = blk.cv1._fwd(tape, x, out=left if n == 0 else None)
        for i, mm in enumerate(blk.m):
            a = mm._fwd(tape, a, out=left if i == n - 1 else None)
        blk.cv2._fwd(tape, x, out=right)
    else:
        pair.mark_step(tape)
        if n == 0:
            cat = tape.conv_bn_act(x, pair, pair.s, pair.p, pair.act_code)
        else:
            T = tape.new(x.N, 3 * c_, x.H, x.W)
            cat = T.slice(c_, 3 * c_)                 # what cv3 reads: [m_out | right]
            a = T.slice(0, c_)
            mslot, right = cat.slice(0, c_), cat.slice(c_, 2 * c_)   # nested slices: they see cat's gradient state
            tape.conv_bn_act(x, pair, pair.s, pair.p, pair.act_code, out=[a, right])
            for i, mm in enumerate(blk.m):
                a = mm._fwd(tape, a, out=mslot if i == n - 1 else None)
    if add:
        return blk.cv3._fwd(tape, cat, res=x, res_mode=L.RES_AFTER_ACT)
    return blk.cv3._fwd(tape, cat)


# ----------------------------------------------------------------------------------------------------------
# CSP blocks
# ----------------------------------------------------------------------------------------------------------
class C3(YdlModule):
    """Seg-script C3 (seg_diceloss_yolov5.py:416-428): cv3(cat(m(cv1 x), cv2 x)) (+ x), m = n plain 3x3 Convs."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Conv(c_, c_, 3, 1, g=g) for _ in range(n)))
        self.add = shortcut and c1 == c2
        self.c_ = c_

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return _csp_forward(self, tape, x, self.add)


class C3k2(C3):
    """yolo9 C3k2 (seg_diceloss_yolov9.py:451-472) = script C3; its crop-align branch can never trigger because both
    branches are stride-1 'same' convolutions of the same input."""


class Bottleneck(YdlModule):
    """models/common.py:115-125."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def _fwd(self, tape: Tape, x: Var, out: Optional[Var] = None) -> Var:
        h = self.cv1._fwd(tape, x)
        if self.add:
            return self.cv2._fwd(tape, h, out=out, res=x, res_mode=L.RES_AFTER_ACT)
        return self.cv2._fwd(tape, h, out=out)


class C3Common(YdlModule):
    """models/common.py:161-172: m = n Bottlenecks (e=1.0), no outer residual.  Exposed to parse_model as ``C3``."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)))
        self.c_ = c_

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return _csp_forward(self, tape, x, False)


class C2f(YdlModule):
    """yolov8/seg_jaccardloss_yolov8.py:401-414.  cv1 writes the first two chunks of the concat buffer directly."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Conv(self.c, self.c, 3, 1, g=g) for _ in range(n))
        self.add = shortcut and c1 == c2

    def _fwd(self, tape: Tape, x: Var) -> Var:
        c, n = self.c, len(self.m)
        cat = tape.new(x.N, (2 + n) * c, x.H, x.W)
        self.cv1._fwd(tape, x, out=cat.slice(0, 2 * c))
        last = cat.slice(c, 2 * c)
        for i, mm in enumerate(self.m):
            last = mm._fwd(tape, last, out=cat.slice((2 + i) * c, (3 + i) * c))
        if self.add:
            return self.cv2._fwd(tape, cat, res=x, res_mode=L.RES_AFTER_ACT)
        return self.cv2._fwd(tape, cat)


class GAM(YdlModule):
    """yolo9 global-aggregation channel attention (unet-lite/yolo9-seg/seg_diceloss_yolov9.py:475-510):
    x * sigmoid(conv2(avgpool(conv1 x)) + conv3(maxpool(conv1 x))).  conv1 runs twice like the reference (its BN
    running statistics advance twice per step).  Must be built as ``GAM(c)``: the shipped yaml's ``GAM [512]`` binds
    k=512 and cannot be constructed."""

    def __init__(self, c, k=1, s=1, e=0.25):
        super().__init__()
        c_ = int(c * e)
        self.conv1 = Conv(c, c_, k, s)
        self.conv2 = Conv(c_, c, k, s, act=False)
        self.conv3 = Conv(c_, c, k, s, act=False)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        y1 = tape.global_pool(self.conv1._fwd(tape, x), "avg")
        y1 = self.conv2._fwd(tape, y1)
        y2 = tape.global_pool(self.conv1._fwd(tape, x), "max")
        y2 = self.conv3._fwd(tape, y2)
        return tape.gate_mul(x, y1, y2)


class SPPF(YdlModule):
    """seg_diceloss_yolov5.py:468-481 / models/common.py:223-238: three chained 5x5/s1 max-pools written straight into
    the 4-way concat buffer."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k
        self.c_ = c_

    def _fwd(self, tape: Tape, x: Var) -> Var:
        c_ = self.c_
        cat = tape.new(x.N, 4 * c_, x.H, x.W)
        s0 = self.cv1._fwd(tape, x, out=cat.slice(0, c_))
        tape.sppf_pools(s0, self.k, [cat.slice(c_, 2 * c_), cat.slice(2 * c_, 3 * c_), cat.slice(3 * c_, 4 * c_)])
        return self.cv2._fwd(tape, cat)


class Concat(YdlModule):
    """Auto-aligning Concat (seg_diceloss_yolov5.py:484-507); with equal sizes it is models/common.py:310-317."""
    takes_list = True

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension
        if dimension != 1:
            raise NotImplementedError("Concat along the channel dimension only")

    def _fwd(self, tape: Tape, xs: Sequence[Var]) -> Var:
        if len(xs) == 1:
            return xs[0]
        return tape.concat(xs, align=True)


class Upsample(YdlModule):
    """nn.Upsample restated for the taped path (nearest, or bilinear with either align_corners convention)."""

    def __init__(self, size=None, scale_factor=None, mode="nearest", align_corners=None):
        super().__init__()
        self.size = size
        self.scale_factor = scale_factor
        self.mode = mode
        self.align_corners = align_corners
        if mode not in ("nearest", "bilinear"):
            raise NotImplementedError(f"Upsample mode {mode}")

    def _out_size(self, x: Var):
        if self.size is not None:
            return (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        sf = self.scale_factor
        sh, sw = (sf, sf) if not isinstance(sf, (tuple, list)) else sf
        return int(math.floor(x.H * sh)), int(math.floor(x.W * sw))

    def _fwd(self, tape: Tape, x: Var, out: Optional[Var] = None) -> Var:
        if self.mode == "nearest" and self.size is None and out is None and config.lazy_upsample():
            sf = self.scale_factor
            sh, sw = (sf, sf) if not isinstance(sf, (tuple, list)) else sf
            if float(sh).is_integer() and float(sw).is_integer() and sh >= 1 and sw >= 1:
                return tape.upsample_lazy(x, int(sh), int(sw))     # no memory traffic; see tape.Var.rep
        x = tape.materialize(x)
        Ho, Wo = self._out_size(x)
        if self.mode == "nearest":
            # ATen passes 1/scale_factor as the index scale when a scale_factor was given
            if self.size is None:
                sf = self.scale_factor
                sh, sw = (sf, sf) if not isinstance(sf, (tuple, list)) else sf
                return tape.resize(x, Ho, Wo, L.RESIZE_NEAREST, 1.0 / sh, 1.0 / sw, out=out)
            return tape.resize(x, Ho, Wo, L.RESIZE_NEAREST, out=out)
        mode = L.RESIZE_BILINEAR_AC if self.align_corners else L.RESIZE_BILINEAR
        if self.size is None and not self.align_corners:
            sf = self.scale_factor
            sh, sw = (sf, sf) if not isinstance(sf, (tuple, list)) else sf
            return tape.resize(x, Ho, Wo, mode, 1.0 / sh, 1.0 / sw, out=out)
        return tape.resize(x, Ho, Wo, mode, out=out)

    def extra_repr(self):
        return f"size={self.size}, scale_factor={self.scale_factor}, mode={self.mode}"


# ----------------------------------------------------------------------------------------------------------
# ResNet blocks + multi-scale SegmentHead (segment/train.py:74-210, Resnet18/seg_diceloss_resnet18.py:216-349)
# ----------------------------------------------------------------------------------------------------------
class BasicBlock(YdlModule):
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv(in_channels, out_channels, 3, stride, 1, act=True)
        self.conv2 = Conv(out_channels, out_channels, 3, 1, 1, act=False)
        self.downsample = downsample
        self.act = nn.ReLU(inplace=True)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        out = self.conv1._fwd(tape, x)
        idt = x if self.downsample is None else self.downsample._fwd(tape, x)
        # relu(bn(conv2(out)) + identity) fused into conv2's apply kernel
        return self.conv2._fwd(tape, out, res=idt, res_mode=L.RES_BEFORE_ACT, act_code=L.ACT_RELU)


class BottleneckBlock(YdlModule):
    expansion = 4

    def __init__(self, in_channels, mid_channels, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv(in_channels, mid_channels, 1, 1, 0, act=True)
        self.conv2 = Conv(mid_channels, mid_channels, 3, stride, 1, act=True)
        self.conv3 = Conv(mid_channels, mid_channels * self.expansion, 1, 1, 0, act=False)
        self.downsample = downsample
        self.act = nn.ReLU(inplace=True)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        out = self.conv2._fwd(tape, self.conv1._fwd(tape, x))
        idt = x if self.downsample is None else self.downsample._fwd(tape, x)
        return self.conv3._fwd(tape, out, res=idt, res_mode=L.RES_BEFORE_ACT, act_code=L.ACT_RELU)


class MaxPool2d(YdlModule):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.k, self.s, self.p = kernel_size, stride or kernel_size, padding

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return tape.maxpool(x, self.k, self.s, self.p)


class SegmentHead(YdlModule):
    """segment/train.py:159-210: lateral 1x1 -> 128, bilinear(align_corners=True) x2^i, cat, 3x3 -> 256, 1x1 -> nc."""
    takes_list = True

    def __init__(self, num_classes: int = 12, in_channels: List[int] = [256, 512, 1024]):
        super().__init__()
        self.num_classes = num_classes
        self.lateral_convs = nn.ModuleList()
        self.up_samples = nn.ModuleList()
        for i, c in enumerate(in_channels):
            self.lateral_convs.append(Conv(c, 128, 1, 1))
            self.up_samples.append(Upsample(scale_factor=2 ** i, mode="bilinear", align_corners=True))
        self.final_conv = nn.Sequential(Conv(128 * len(in_channels), 256, 3, 1), Conv(256, num_classes, 1, 1, act=False))

    def _fwd(self, tape: Tape, feats: Sequence[Var]) -> Var:
        if len(feats) != len(self.lateral_convs):
            raise ValueError(f"feature count mismatch: expected {len(self.lateral_convs)}, got {len(feats)}")
        feats = [tape.materialize(f) for f in feats]
        H, W = feats[0].H, feats[0].W
        cat = tape.new(feats[0].N, 128 * len(feats), H, W)
        for i, (f, lat, up) in enumerate(zip(feats, self.lateral_convs, self.up_samples)):
            sl = cat.slice(128 * i, 128 * (i + 1))
            if (f.H, f.W) == (H, W):
                lat._fwd(tape, f, out=sl)
                continue
            f = lat._fwd(tape, f)
            Ho, Wo = up._out_size(f)
            if (Ho, Wo) == (H, W):
                up._fwd(tape, f, out=sl)
            else:                                   # F.interpolate(size=target, align_corners=True) fallback
                f = up._fwd(tape, f)
                tape.resize(f, H, W, L.RESIZE_BILINEAR_AC, out=sl)
        return self.final_conv[1]._fwd(tape, self.final_conv[0]._fwd(tape, cat))


# ----------------------------------------------------------------------------------------------------------
# DCNv3 module and its YOLO wiring (models/ops_dcnv3/build/.../modules/dcnv3.py:50-136, "common and yolo.py":2-38)
# ----------------------------------------------------------------------------------------------------------
class Linear(YdlModule):
    """``nn.Linear`` over the channel dimension of an NHWC tensor (state_dict: weight [out, in], bias [out]) = a 1x1
    convolution with bias on the implicit-GEMM kernels; the bias gradient is a deterministic per-channel sum."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)
        self._wcache = {}

    def _wkey(self, tape: Tape):
        return (tape.dname, self.weight.data_ptr(), self.weight._version, config.weight_epoch())

    def compute_weights(self, tape: Tape):
        key = self._wkey(tape)
        c = self._wcache
        if c.get("key") != key or c.get("w") is None or c["w"].dtype != tape.tdt:
            dev = self.weight.device
            if c.get("w") is None or c["w"].dtype != tape.tdt or c["w"].device != dev:
                c["w"] = torch.empty((self.out_features, 1, round_up(self.in_features, 8)), dtype=tape.tdt, device=dev)
                c["wt"] = torch.empty((self.in_features, 1, round_up(self.out_features, 8)), dtype=tape.tdt, device=dev)
            master = self.weight.detach()
            if not master.is_contiguous():
                master = master.contiguous()
            L.call("ydl_weight_prep", tape.dt, _p(master), _p(c["w"]), _p(c["wt"]), self.out_features, 1, self.in_features,
                   _stream())
            c["key"] = key
        return c["w"], c["wt"]

    def bias_coeffs(self, device):
        """(ones, bias) padded to a multiple of 8 channels for ydl_bn_act_fwd (scale = 1, shift = bias)"""
        cp = round_up(self.out_features, 8)
        key = (self.bias.data_ptr(), self.bias._version, config.weight_epoch())
        c = self._wcache
        if c.get("ones") is None or c["ones"].device != device:       # created once (a refresh below only rewrites bpad's contents)
            c["ones"] = torch.ones(cp, dtype=torch.float32, device=device)
            c["bpad"] = torch.zeros(cp, dtype=torch.float32, device=device)
            c["bkey"] = None
        if c.get("bkey") != key:
            L.call("ydl_copy2d", L.YDL_F32, _p(self.bias.detach()), self.out_features, _p(c["bpad"]), cp, 1, self.out_features, 0,
                   _stream())
            c["bkey"] = key
        return c["ones"], c["bpad"]

    def _grad_of(self, p: nn.Parameter) -> torch.Tensor:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def wgrad(self, tape: Tape, gp, x: Var, dy: Var, st) -> None:
        if not self.weight.requires_grad:          # frozen: no gradient kernel, never marked touched
            return
        g = self._grad_of(self.weight)
        cin_p = round_up(self.in_features, 8)
        if cin_p == self.in_features and g.is_contiguous():
            _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), _p(g), st)
        else:
            tmp = zero_(torch.empty((self.out_features, 1, cin_p), dtype=torch.float32, device=g.device), st)
            _launch_wgrad(tape, gp, _p(x.t), _p(dy.t), _p(tmp), st)
            L.call("ydl_wgrad_unpad", _p(tmp), _p(g), self.out_features, 1, self.in_features, 1, st)
        config.mark_touched(self.weight)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return tape.linear(x, self)


def _lanes_per_group_channel_block(gc: int) -> int:
    """lanes one (pixel, group) item of the DCNv3 kernels occupies: the next power of two >= the group's channel count, at most a
    wavefront (csrc/dcnv3.hip: fill_args) — the idle share of those lanes is what a non-power-of-two head width costs here"""
    seg = 1
    while seg < gc and seg < 64:
        seg <<= 1
    return seg


class DCNv3(YdlModule):
    """modules/dcnv3.py:50-136.  Input and output are channels-last in the reference ((N, H, W, C)); inside a taped model
    every activation already is, so the permutes of ``DCNV3_YoLo.forward`` vanish.  Stand-alone calls take the
    reference's (N, H, W, C) tensor."""

    def __init__(self, channels=64, kernel_size=3, stride=1, pad=1, dilation=1, group=4, offset_scale=1.0,
                 act_layer="GELU", norm_layer="LN"):
        super().__init__()
        if channels % group != 0:
            raise ValueError(f"channels must be divisible by group, but got {channels} and {group}")
        gc = channels // group
        seg = _lanes_per_group_channel_block(gc)
        lanes = -(-gc // seg) * seg                # lanes x passes an item takes
        if lanes != gc:
            # (the reference warns at the same place, modules/dcnv3.py:75-79, about its own kernel; here the cost is idle lanes)
            import warnings
            warnings.warn(f"DCNv3: {gc} channels per group run as {lanes // seg} pass(es) of {seg}-lane segments in the HIP sampling "
                          f"kernels ({lanes - gc} of {lanes} lane slots idle); a power of two up to 64, or a multiple of 64, wastes none.")
        self.offset_scale = offset_scale
        self.channels = channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = 1                 # the reference ignores its dilation argument (modules/dcnv3.py:82)
        self.pad = pad
        self.group = group
        self.group_channels = channels // group
        self.dw_conv = Conv(channels, channels, kernel_size, g=channels)
        self.offset = Linear(channels, group * kernel_size * kernel_size * 2)
        self.mask = Linear(channels, group * kernel_size * kernel_size)
        self.input_proj = Linear(channels, channels)
        self.output_proj = Linear(channels, channels)
        self._reset_parameters()

    def _reset_parameters(self):
        with torch.no_grad():
            self.offset.weight.zero_(); self.offset.bias.zero_()
            self.mask.weight.zero_(); self.mask.bias.zero_()
            nn.init.xavier_uniform_(self.input_proj.weight); self.input_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight); self.output_proj.bias.zero_()

    def forward(self, x, *a, **kw):
        if isinstance(x, Var):
            return self._fwd(x.tape, x)
        # stand-alone: (N, H, W, C) in, (N, H, W, C) out like the reference module
        return run_region(self, [x.permute(0, 3, 1, 2)]).permute(0, 2, 3, 1)

    def _fwd(self, tape: Tape, inp: Var) -> Var:
        x = self.input_proj._fwd(tape, inp)
        x1 = self.dw_conv._fwd(tape, inp)
        offset = self.offset._fwd(tape, x1)
        P = self.kernel_size * self.kernel_size
        mask = tape.group_softmax(self.mask._fwd(tape, x1), self.group, P)
        y = tape.dcnv3(x, offset, mask, self.kernel_size, self.stride, self.pad, self.dilation, self.group,
                       self.group_channels, float(self.offset_scale))
        return self.output_proj._fwd(tape, y)


class DCNV3_YoLo(YdlModule):
    """"common and yolo.py":2-14: Conv(inc, ouc, k=1) -> DCNv3(ouc, kernel_size=k, stride=s, group=g, dilation=d)"""

    def __init__(self, inc, ouc, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = Conv(inc, ouc, k=1)
        self.dcnv3 = DCNv3(ouc, kernel_size=k, stride=s, group=g, dilation=d)

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return self.dcnv3._fwd(tape, self.conv._fwd(tape, x))


class Bottleneck_DCNV3(YdlModule):
    """"common and yolo.py":16-25"""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = DCNV3_YoLo(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def _fwd(self, tape: Tape, x: Var, out: Optional[Var] = None) -> Var:
        y = self.cv2._fwd(tape, self.cv1._fwd(tape, x))
        if self.add:
            y = tape.add(x, y)
        return y if out is None else tape.copy(y, out)


class C3_DCNV3(YdlModule):
    """"common and yolo.py":27-38: models/common.py C3 with Bottleneck_DCNV3 inner blocks (no outer residual)"""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck_DCNV3(c_, c_, shortcut, g, e=1.0) for _ in range(n)))
        self.c_ = c_

    def _fwd(self, tape: Tape, x: Var) -> Var:
        return _csp_forward(self, tape, x, False)
