"""The MI355X-native runtime under the nn.Module facade: a static reverse-mode *tape* over NHWC device buffers.

One forward pass of a model (or of a single block used on its own) records a list of backward closures; every
primitive launches hand-written HIP kernels through the C ABI (include/ydl.h) on torch's current stream.  torch is
plumbing only (allocator, streams, the autograd edge at the region boundary): there is no ATen compute inside a
taped region.  Gradient fan-in is resolved by accumulate flags on the kernels (first writer overwrites, later
writers add), concatenation is free (producers write channel slices of one buffer), and branches that never
receive a gradient are skipped, which reproduces autograd's "grad is None" for the reference's dead head layers
(SURVEY T4).
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib as L

_DT = {"f32": (L.YDL_F32, torch.float32), "bf16": (L.YDL_BF16, torch.bfloat16)}


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> ctypes.c_void_p:
    """hipStream_t of torch's current stream (the raw accessor is ~20x cheaper than building a Stream object, and this is
    called once per kernel launch)"""
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE = {}
_NO_SLAB = __import__("os").environ.get("YDL_SLAB", "1") == "0"


def side_stream(device) -> "torch.cuda.Stream":
    """second HIP stream per device: weight-gradient kernels run here, concurrently with the input-gradient chain on
    the main stream (both only read dy; small layers do not fill 256 CUs on their own)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=device)
    return st


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_wait(waiter: "torch.cuda.Stream", on: "torch.cuda.Stream") -> None:
    """``waiter`` waits for everything enqueued on ``on`` so far.  Every cross-stream edge of the taped step goes through here so
    that a launch-list recording (yolo_dual_amd.replay) sees it."""
    waiter.wait_stream(on)
    rec = L.recorder()
    if rec is not None:
        rec.edge(on.cuda_stream, waiter.cuda_stream)


_ZDT = {torch.float32: L.YDL_F32, torch.bfloat16: L.YDL_BF16, torch.float16: L.YDL_BF16}


def zero_(t: torch.Tensor, st=None) -> torch.Tensor:
    """t[...] = 0 through the C ABI (ydl_fill_zero / ydl_zero2d) on stream ``st`` (default: torch's current stream): dense
    tensors in any dimension order, or a channel slice of an NHWC buffer seen as (N, C, H, W).  No ATen kernel runs inside a
    taped region — a launch list replays only what went through the C ABI."""
    if t.numel() == 0:
        return t
    if t.device.type != "cuda":         # host-side arenas (the data-parallel rehearsals on CPU): no device work to record
        return t.zero_()
    st = _stream() if st is None else st
    es = t.element_size()
    dense = t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())
    if dense:
        L.call("ydl_fill_zero", _p(t), t.numel() * es, st)
        return t
    if t.dim() == 4 and t.stride(1) == 1 and t.dtype in _ZDT:
        N, C, H, W = t.shape
        ld = t.stride(3)
        if t.stride(2) == W * ld and (N == 1 or t.stride(0) == H * W * ld):
            L.call("ydl_zero2d", _ZDT[t.dtype], _p(t), ld, N * H * W, C, st)
            return t
    raise RuntimeError(f"zero_: unsupported view (shape {tuple(t.shape)}, strides {t.stride()}, {t.dtype})")


class Var:
    """A NHWC activation: ``t`` is a logical (N,C,H,W) torch view whose memory is [N][H][W][ld] with c fastest."""
    __slots__ = ("t", "N", "C", "H", "W", "ld", "g", "gset", "need", "parent", "c0", "children", "tape", "dt", "rep", "alias",
                 "cat_parts", "real", "ext_src", "nuse", "wcount", "bn_src", "bnred")

    def __init__(self, tape: "Tape", t: torch.Tensor, ld: int, need: bool, parent: Optional["Var"] = None, c0: int = 0):
        self.tape = tape
        self.dt = L.YDL_F32 if t.dtype == torch.float32 else L.YDL_BF16
        self.t = t
        self.N, self.C, self.H, self.W = t.shape
        self.ld = ld
        self.g: Optional[torch.Tensor] = None
        self.gset = False
        self.need = need
        self.parent = parent
        self.c0 = c0
        self.children: List["Var"] = []
        # lazy nearest up-sampling: the LOGICAL tensor is this (stored) one replicated rep[0] x rep[1] times.
        # Point-wise ops (1x1 conv, BN, activation, channel softmax) commute with it and run at the stored size;
        # gradients of a lazy Var are kept as the SUM over the replicas, which makes every backward formula of
        # those ops identical to the un-replicated one.
        self.rep = (1, 1)
        self.alias = False      # full-range view of ``parent`` (lazy / materialise views): shares its gradient state
        # virtual channel concat (Tape.concat): no memory behind ``t`` (a zero-stride placeholder of the right shape);
        # cat_parts = [(source Var, first channel, needs bilinear resize)].  A 1x1 convolution consumes the parts
        # directly (conv over a concat = sum of convs over the sources, and a 1x1 conv commutes with the resize);
        # every other consumer goes through Tape.materialize, which builds the real tensor once (``real``).
        self.cat_parts = None
        self.real = None
        # fused BatchNorm-backward reduce (Tape._try_bnred): on the TOP buffer, nuse = pending consumers (a hint) and wcount = the
        # write log (the validation), both as absolute channel ranges; on a conv+BN output, bn_src = what a consumer's dgrad epilogue
        # needs to run this layer's reduce pass, bnred = (sums, length of the write log right after that dgrad's write) once one has
        self.nuse = None          # top buffer: channel ranges of the recorded ops that will still write into its gradient
        self.wcount = None        # top buffer: channel ranges written so far, in order
        self.bn_src = None
        self.bnred = None
        # lazily converted region input: the caller's (N,C,H,W) f32 tensor; the NHWC copy is made by Tape.materialize on
        # first use — or never, when the first layer is a stem conv that reads a space-to-depth conversion instead
        self.ext_src = None

    def root(self) -> "Var":
        v = self
        while v.alias:
            v = v.parent
        return v

    def top(self) -> "Var":
        """the Var that owns the buffer (slices and alias views share their parent's gradient storage)"""
        v = self
        while v.parent is not None:
            v = v.parent
        return v

    def abs_range(self) -> tuple:
        """[c0, c1) of this Var in its top buffer's channels"""
        c0, v = 0, self
        while v.parent is not None:
            if not v.alias:
                c0 += v.c0
            v = v.parent
        return (c0, c0 + self.C)

    @property
    def npix(self) -> int:
        return self.N * self.H * self.W

    @property
    def LH(self) -> int:
        return self.H * self.rep[0]

    @property
    def LW(self) -> int:
        return self.W * self.rep[1]

    @property
    def lazy(self) -> bool:
        return self.rep != (1, 1)

    @property
    def virtual(self) -> bool:
        return self.cat_parts is not None or self.ext_src is not None

    def is_set(self) -> bool:
        v = self.root()
        return v.gset or (v.parent is not None and v.parent.is_set())

    def aligned(self) -> bool:
        """16-byte channel alignment of a slice (whole buffers always are): required by the vector kernels"""
        return self.parent is None or self.alias and self.parent.aligned() or (self.c0 % 8 == 0 and self.C % 8 == 0 and self.parent.aligned())

    def slice(self, c0: int, c1: int) -> "Var":
        assert not self.virtual, "materialize a virtual Var before slicing it"
        v = Var(self.tape, self.t[:, c0:c1], self.ld, self.need, parent=self, c0=c0)
        self.children.append(v)
        return v


def _alloc(N: int, C: int, H: int, W: int, tdtype: torch.dtype, device, zero: bool = False) -> (torch.Tensor, int):
    ld = round_up(C, 8)
    buf = torch.empty((N, H, W, ld), dtype=tdtype, device=device)
    if zero or ld != C:
        zero_(buf)
    return buf.permute(0, 3, 1, 2)[:, :C], ld


class LazyOutput:
    """side channel between a region whose output is a nearest-replicated tensor and SegmentationLoss:
    ``low`` = the (N, C, H, W) tensor the (N, C, H*rh, W*rw) output replicates, ``version`` = the output's version
    counter when it was produced (an in-place edit by the caller invalidates the shortcut), ``dlow`` = replica-summed
    gradient handed back by the loss backward, ``dummy`` = the zero-stride placeholder gradient it returned."""
    __slots__ = ("low", "rep", "version", "dlow", "dummy", "claimed")

    def __init__(self, low: torch.Tensor, rep):
        self.low, self.rep, self.version, self.dlow, self.dummy, self.claimed = low, tuple(rep), -1, None, None, False


class Tape:
    def __init__(self, dtype: str, device, train: bool, record: bool):
        self.dname = dtype
        self.dt, self.tdt = _DT[dtype]
        self.V = 4 if dtype == "f32" else 8
        self.device = device
        self.train = train
        self.record = record
        self.bw: List[Callable[[], None]] = []
        self.touched_params: List[torch.nn.Parameter] = []
        self._side_used = False
        self._keep: List[Var] = []
        self._pending_wgrad: List[Callable] = []   # weight-gradient launches waiting for the next fork of the side stream
        self._bw_left = 0
        self.ext = None      # (Var, tensor) of a region whose external output was written directly (softmax head)
        self._slab = None    # zero-initialised f32 scratch of the region: accumulator rows the kernels add into (BN replica sums)
        self._slab_off = 0
        self._slab_total = 0         # floats handed out so far (the owner module remembers it as next step's slab size)
        self._slab_hint = 0
        self._main = torch.cuda.current_stream() if torch.device(device).type == "cuda" else None

    # ------------------------------------------------------------------ buffers
    def new(self, N: int, C: int, H: int, W: int, need: bool = True, zero: bool = False, f32: bool = False) -> Var:
        t, ld = _alloc(N, C, H, W, torch.float32 if f32 else self.tdt, self.device, zero)
        return Var(self, t, ld, need)

    def zeroed(self, nfloats: int) -> torch.Tensor:
        """``nfloats`` f32 zeros (16-byte aligned) from the region's slab: ONE memset per slab instead of one per accumulator row.
        A slab is cleared when it is created, before any kernel that adds into it is enqueued; rows are handed out once."""
        n = round_up(nfloats, 4)
        if _NO_SLAB:                      # debugging switch: every row its own buffer and memset
            return zero_(torch.empty(nfloats, dtype=torch.float32, device=self.device))
        self._slab_total += n
        if self._slab is None or self._slab_off + n > self._slab.numel():
            cap = max(1 << 18, self._slab_hint, 2 * n, 2 * (self._slab.numel() if self._slab is not None else 0))
            self._slab = zero_(torch.empty(cap, dtype=torch.float32, device=self.device))
            self._slab_off = 0
            # the memset went to the CURRENT stream; rows of this slab may be used on the region's other stream too (dead head
            # branch, deferred work): order it behind the memset.  Normally the slab is sized from the previous step's need and
            # created once, by the first layer, before any fork.
            # (Only when this region has forked the side stream already: waiting on a stream that is not part of the step
            #  would pull it into a HIP-graph capture and leave it unjoined.)
            cur = torch.cuda.current_stream()
            side = _SIDE.get(torch.cuda.current_device())
            if side is not None and (self._side_used or getattr(self, "_side_fwd", False)):
                for other in (self._main, side):
                    if other is not None and other.cuda_stream != cur.cuda_stream:
                        stream_wait(other, cur)
        out = self._slab[self._slab_off:self._slab_off + nfloats]
        self._slab_off += n
        return out

    def new_like(self, v: Var) -> Var:
        return self.new(v.N, v.C, v.H, v.W, v.need, f32=(v.dt == L.YDL_F32))

    def _gbuf(self, v: Var) -> torch.Tensor:
        """gradient buffer of v (a slice shares its parent's buffer)"""
        v = v.root()
        if v.g is None:
            if v.parent is not None:
                pg = self._gbuf(v.parent)
                v.g = pg[:, v.c0:v.c0 + v.C]
            else:
                v.g, _ = _alloc(v.N, v.C, v.H, v.W, v.t.dtype, self.device)
        return v.g

    def _alias_grad(self, v: Var, o: Var, dout: torch.Tensor) -> bool:
        """A residual joined after the activation has d/dv = dout: when v's gradient has no buffer and no writer yet, and both are
        whole buffers of the same layout, v adopts dout's storage (later writers accumulate into it; dout's consumer, the BN
        backward that calls this, reads it before anything else is enqueued).  Saves one write pass over the tensor."""
        v, o = v.root(), o.root()
        if (v.g is not None or v.is_set() or v.parent is not None or o.parent is not None or v.children or o.children
                or v.lazy or o.lazy or v.ld != o.ld or (v.N, v.C, v.H, v.W) != (o.N, o.C, o.H, o.W) or v.t.dtype != o.t.dtype
                or dout.shape != o.t.shape):
            return False
        v.g = dout
        v.gset = True
        self._note_write(v)
        return True

    def _try_bnred(self, x: Var, gp, dy: Var, wt: torch.Tensor, gx: torch.Tensor, acc: int, st) -> bool:
        """the input gradient of a convolution with the BatchNorm-backward REDUCE pass of the layer(s) that produced ``x`` in its
        epilogue (ydl_conv_dgrad_bnred) — when this is, as far as the forward pass could tell, the last write of that gradient
        (no pending consumer on the buffer), the producers are plain conv+BN(+SiLU) outputs covering 8-aligned channel ranges, and
        the geometry runs on a kernel that has the fused epilogue.  The producers find (sums, write count) on their output Var and
        skip their own reduce launch if nothing wrote the buffer in between (Tape.conv_bn_act.bw); a wrong guess only wastes the
        epilogue work."""
        from . import config as _cfg
        if not _cfg.bn_bwd_fuse() or self.dt != L.YDL_BF16:
            return False
        xr = x.root()
        tp = xr.top()
        pend = tp.nuse or ()

        def last_writer(v: Var) -> bool:
            a, b = v.abs_range()
            return not any(r[0] < b and a < r[1] for r in pend)
        segs = []
        if xr.bn_src is not None:
            if last_writer(xr):
                segs.append((0, xr))
        else:
            for c in xr.children:
                if c.bn_src is not None and not c.alias and last_writer(c):
                    segs.append((c.c0, c))
        if not segs or len(segs) > 2:
            return False
        segs.sort(key=lambda e: e[0])
        if len(segs) == 2 and segs[0][0] + segs[0][1].C > segs[1][0]:
            return False
        if not L.lib().ydl_conv_dgrad_bnred_supported(gp, self.dt):
            return False
        red = L.BnRed()
        red.nseg = len(segs)
        keep = []
        for i, (c0, v) in enumerate(segs):
            yptr, ldy, cf, co, cw, act, ykeep = v.bn_src
            cp = round_up(cw, 8)
            sums = self.zeroed(L.BN_REPLICAS * 2 * cp)
            red.c0[i], red.c1[i], red.ldy[i], red.cp[i], red.act[i] = c0, c0 + cw, ldy, cp, act
            red.y[i] = yptr
            red.scale[i], red.shift[i] = cf["scale"][co:].data_ptr(), cf["shift"][co:].data_ptr()
            red.mean[i], red.invstd[i] = cf["mean"][co:].data_ptr(), cf["invstd"][co:].data_ptr()
            red.sums[i] = sums.data_ptr()
            keep.append((v, sums))
        L.call("ydl_conv_dgrad_bnred", gp, self.dt, _p(dy.t), _p(wt), _p(gx), acc, ctypes.byref(red), st)
        wc = len(tp.wcount)
        for v, sums in keep:
            v.bnred = (sums, wc)
        self._keep.append(red)
        return True

    def _use(self, v: Optional[Var]) -> None:
        """forward-time note: a recorded op will later write d/dv (hint for Tape._try_bnred; a missing note only costs a fallback)"""
        if v is not None and self.record:
            v = v.root()
            tp = v.top()
            if tp.nuse is None:
                tp.nuse = []
            tp.nuse.append(v.abs_range())

    def _unwritten_since(self, o: Var, n: int) -> bool:
        """no gradient write has touched o's channels since the write log of its buffer had n entries"""
        o = o.root()
        a, b = o.abs_range()
        log = o.top().wcount or ()
        return not any(r[0] < b and a < r[1] for r in log[n:])

    def _note_write(self, v: Var) -> None:
        tp = v.top()
        r = v.abs_range()
        if tp.wcount is None:
            tp.wcount = []
        tp.wcount.append(r)
        if tp.nuse:
            try:
                tp.nuse.remove(r)
            except ValueError:
                pass

    def grad_target(self, v: Var) -> (torch.Tensor, int):
        """(buffer, accumulate) for a kernel about to write d/dv; marks v as set."""
        v = v.root()
        self._note_write(v)
        g = self._gbuf(v)
        acc = 1 if v.is_set() else 0
        if not acc and v.children and any(c.gset for c in v.children):
            # a slice already holds a gradient but the whole buffer does not: zero the rest, then accumulate
            for c in v.children:
                if not c.gset:
                    zero_(self._gbuf(c))
            covered = sorted((c.c0, c.c0 + c.C) for c in v.children)
            pos = 0
            for a, b in covered:
                if a > pos:
                    zero_(g[:, pos:a])
                pos = max(pos, b)
            if pos < v.C:
                zero_(g[:, pos:])
            acc = 1
        v.gset = True
        return g, acc

    # ------------------------------------------------------------------ region boundary
    def input_nchw(self, x: torch.Tensor, lazy: bool = False) -> Var:
        """external (N,C,H,W) f32 tensor (any strides) -> internal NHWC compute dtype, channels zero padded.
        ``lazy``: return a virtual Var and convert on first use (see Var.ext_src)"""
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.contiguous().float()
        N, C, H, W = x.shape
        if lazy:
            ph = torch.empty(1, dtype=self.tdt, device=self.device)
            v = Var(self, ph.expand(N, C, H, W), round_up(C, 8), False)
            v.ext_src = x
            return v
        v = self.new(N, C, H, W, need=False, zero=True)
        L.call("ydl_nchw_to_nhwc", self.dt, _p(x), _p(v.t), v.ld, N, C, H, W, _stream())
        return v

    def input_s2d(self, src: torch.Tensor, s: int) -> Var:
        """space-to-depth conversion of an external (N,C,H,W) tensor: (N, s*s*C, H/s, W/s) NHWC Var (ydl_nchw_to_s2d)"""
        N, C, H, W = src.shape
        v = self.new(N, C * s * s, H // s, W // s, need=False)
        L.call("ydl_nchw_to_s2d", self.dt, _p(src), _p(v.t), v.ld, N, C, H, W, s, _stream())
        return v

    # ------------------------------------------------------------------ lazy nearest up-sampling
    def upsample_lazy(self, x: Var, fh: int, fw: int) -> Var:
        """nearest up-sampling by integer factors without touching memory (see Var.rep)"""
        if x.virtual:
            x = self.materialize(x)
        v = Var(self, x.t, x.ld, x.need, parent=x, c0=0)
        v.alias = True
        v.rep = (x.rep[0] * fh, x.rep[1] * fw)
        return v

    def materialize(self, x: Var, out: Optional[Var] = None) -> Var:
        """turn a lazy Var into a real (LH, LW) tensor (nearest replication kernel; backward sums the replicas);
        a virtual concat becomes the real concatenated tensor"""
        if x.virtual:
            if x.real is None:
                if x.ext_src is not None:
                    x.real = self.input_nchw(x.ext_src)
                else:
                    x.real = self._concat_real([v for (v, _c0, _rs) in x.cat_parts], True)
            x = x.real
        if not x.lazy:
            return x if out is None else self.copy(x, out)
        fh, fw = x.rep
        plain = Var(self, x.t, x.ld, x.need, parent=x, c0=0)
        plain.alias = True
        return self.resize(plain, x.H * fh, x.W * fw, L.RESIZE_NEAREST, 1.0 / fh, 1.0 / fw, out=out)

    def export_nchw(self, v: Var) -> torch.Tensor:
        v = self.materialize(v)
        out = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=self.device)
        L.call("ydl_nhwc_to_nchw", v.dt, _p(v.t), v.ld, _p(out), v.N, v.C, v.H, v.W, 0, _stream())
        return out

    def seed_grad_nchw(self, v: Var, g: torch.Tensor) -> None:
        g = g.detach()
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.contiguous().float()
        buf, acc = self.grad_target(v)
        assert acc == 0
        L.call("ydl_nchw_to_nhwc", v.dt, _p(g), _p(buf), v.ld, v.N, v.C, v.H, v.W, _stream())

    def grad_nchw(self, v: Var) -> Optional[torch.Tensor]:
        if not v.is_set():
            return None
        out = torch.empty((v.N, v.C, v.H, v.W), dtype=torch.float32, device=self.device)
        L.call("ydl_nhwc_to_nchw", v.dt, _p(self._gbuf(v)), v.ld, _p(out), v.N, v.C, v.H, v.W, 0, _stream())
        return out

    def _defer_wgrad(self, launch: Callable, keep=()) -> None:
        """queue a weight-gradient launch for the side stream.  One fork (event record + stream wait) per ``wgrad_batch`` launches
        instead of one per layer: forks cost host time in eager mode and a cross-branch edge each in a captured graph.  The last
        layers of the sweep fork at once, so the tail of the backward pass does not wait for a batch to fill."""
        from . import config as _cfg
        self._pending_wgrad.append(launch)
        self._keep.extend(keep)
        if len(self._pending_wgrad) >= _cfg.wgrad_batch() or self._bw_left <= 2:
            self._flush_wgrads()

    def _flush_wgrads(self) -> None:
        if not self._pending_wgrad:
            return
        side = side_stream(self.device)
        stream_wait(side, torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = _stream()
            for launch in self._pending_wgrad:
                launch(st)
        self._pending_wgrad.clear()
        self._side_used = True

    def run_backward(self) -> None:
        n = len(self.bw)
        for i, fn in enumerate(reversed(self.bw)):
            self._bw_left = n - 1 - i
            fn()
        self.bw.clear()
        self._flush_wgrads()
        if self._side_used:
            stream_wait(torch.cuda.current_stream(), side_stream(self.device))   # parameter grads complete
            self._side_used = False
        self._keep.clear()

    # ------------------------------------------------------------------ conv + BN + act (+ residual)
    def conv_bn_act(self, x: Var, m, s: int, p: int, act: int, out=None,
                    res: Optional[Var] = None, res_mode: int = L.RES_NONE) -> Var:
        """out = act(bn(conv(x))) [+ res].  ``m`` is a yolo_dual_amd.modules.Conv (parameter holder) or a fused sibling
        pair.  ``out`` may be a Var (typically a concat slice) or a LIST of Vars that split the output channels (fused
        siblings: each part is activated into its own destination).
        Reference: Conv.forward seg_diceloss_yolov5.py:403-409."""
        k = m.k
        Cout, Cin = m.c2, m.c1
        if x.C != Cin:
            raise RuntimeError(f"Conv layer input channel mismatch: got {x.C}, weight expects {Cin}")
        outs = list(out) if isinstance(out, (list, tuple)) else None
        rep = 1
        virt = None
        if x.virtual and x.cat_parts is None:
            x = self.materialize(x)
        if x.virtual:
            # conv1x1(cat(a.., bilinear_up(b))) = sum_a conv1x1_a(a) + bilinear_up(conv1x1_b(b)): worth it when the
            # up-sampled source is much wider than the output (the 512-channel 1/16-scale map of the yolov5 head)
            rs = [v for (v, _c0, r_) in x.cat_parts if r_]
            if (k == 1 and s == 1 and p == 0 and res is None and m.splittable() and Cin % 8 == 0
                    and rs[0].C >= 2 * Cout and (outs is None or all(o.aligned() for o in outs))
                    and (outs is not None or out is None or out.aligned())):
                virt = [(self.materialize(v) if (v.lazy or not v.aligned()) else v, c0_, r_) for (v, c0_, r_) in x.cat_parts]
                virt = [(v if v.aligned() else self.copy(v, self.new(v.N, v.C, v.H, v.W, need=v.need)), c0_, r_)
                        for (v, c0_, r_) in virt]
            else:
                x = self.materialize(x)
        if x.lazy:
            if k == 1 and s == 1 and p == 0 and res is None and out is None:
                rep = x.rep[0] * x.rep[1]           # point-wise: run at the stored size, stay lazy
            else:
                x = self.materialize(x)
        if res is not None and (res.lazy or res.virtual):
            res = self.materialize(res)
        if not x.aligned():                       # odd channel split: stage through an aligned buffer (cold path)
            x = self.copy(x, self.new(x.N, x.C, x.H, x.W, need=x.need))
        if res is not None and not res.aligned():
            res = self.copy(res, self.new(res.N, res.C, res.H, res.W, need=res.need))
        if outs is None and out is not None and not out.aligned():
            tmp = self.conv_bn_act(x, m, s, p, act, None, res, res_mode)
            return self.copy(tmp, out)
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        y = self.new(x.N, Cout, Ho, Wo)
        if outs is None:
            if out is None:
                out = self.new(x.N, Cout, Ho, Wo)
                out.rep = x.rep
                if virt is None and not x.need and not any(m.trainable()) and (res is None or not res.need):
                    out.need = False      # frozen layer on a prefix that needs no gradient: consumers skip their dgrad into it
            elif (out.N, out.C, out.H, out.W) != (x.N, Cout, Ho, Wo):
                raise RuntimeError("conv_bn_act: output slice has the wrong shape")
            outs = [out]
        else:
            if sum(o.C for o in outs) != Cout or any(o.C % 8 or not o.aligned() for o in outs) or res is not None:
                raise RuntimeError("conv_bn_act: split outputs must be aligned channel groups summing to Cout")
        parts = []                                 # (channel offset, width, destination Var)
        c0 = 0
        for o in outs:
            parts.append((c0, o.C, o))
            c0 += o.C
        st = _stream()
        w, wt = m.compute_weights(self)
        npix = x.N * Ho * Wo
        cf = m.coeffs(self.device)                   # dict of f32 [Cp] tensors: mean, invstd, scale, shift
        es = w.element_size()
        Cin_p, Cout_p = round_up(Cin, 8), round_up(Cout, 8)
        if virt is None:
            gkey = (x.N, x.H, x.W, Cin, Cout, k, s, p, x.ld, y.ld, self.dt, L.debug_epoch())
            gc = getattr(m, "_geom_cache", None)
            if gc is None or gc[0] != gkey:
                geom = L.ConvGeom(x.N, x.H, x.W, Cin, Ho, Wo, Cout, k, s, p, x.ld, y.ld, 0)
                gpp = ctypes.byref(geom)
                gc = (gkey, geom, L.lib().ydl_conv_fwd_stats_ws_bytes(gpp, self.dt), L.lib().ydl_conv_fwd_grid_m(gpp, self.dt),
                      L.lib().ydl_conv_fwd_block_m(gpp, self.dt))
                try:
                    m._geom_cache = gc
                except Exception:
                    pass
            geom = gc[1]
            subs = None
        else:
            # one sub-convolution per source, on column block [c0, c0+C) of the weight matrix (row stride = Cin_p)
            subs = []
            for (v, c0_, r_) in virt:
                gv = L.ConvGeom(v.N, v.H, v.W, v.C, v.H, v.W, Cout, 1, 1, 0, v.ld, y.ld, Cin_p)
                subs.append((v, c0_, r_, gv))
            subs.sort(key=lambda e: not e[2])        # the resized source first: it initialises y
            geom = subs[-1][3]                       # the launch that writes the BN partials
        gp = ctypes.byref(geom)

        from . import config as _cfg
        sums_mode = self.train and _cfg.bn_sums(self.dname)      # replica sums instead of partial rows + finalize / merge launches
        fwd_entry = "ydl_conv_fwd_sums" if sums_mode else "ydl_conv_fwd"

        def conv_fwd(ws_ptr):
            if subs is None:
                L.call(fwd_entry if ws_ptr is not None else "ydl_conv_fwd", gp, self.dt, _p(x.t), _p(w), _p(y.t), ws_ptr, 0, st)
                return
            if sums_mode and ws_ptr is not None and _cfg.resize_last() and sum(1 for e in subs if e[2]) == 1 and len(subs) >= 2:
                # replica-sums mode: the plain sub-convolutions first (the first one overwrites y, no statistics), the resized
                # source LAST through ydl_resize_acc_sums — a streaming read-modify-write that also adds the statistics of the sum.
                # (The other order costs a full write of the resized tensor plus the point-wise kernel's register-layout
                # read-modify-write with statistics: 42 + 94 us against 48 + ~45 us for the 128-channel 160^2 case of config 2.)
                first = True
                for (v, c0_, r_, gv) in subs:
                    if r_:
                        continue
                    wv = ctypes.c_void_p(w.data_ptr() + c0_ * es)
                    L.call("ydl_conv_fwd", ctypes.byref(gv), self.dt, _p(v.t), wv, _p(y.t), None, 0 if first else 1, st)
                    first = False
                for (v, c0_, r_, gv) in subs:
                    if not r_:
                        continue
                    wv = ctypes.c_void_p(w.data_ptr() + c0_ * es)
                    z = self.new(v.N, Cout, v.H, v.W)
                    gz = L.ConvGeom(v.N, v.H, v.W, v.C, v.H, v.W, Cout, 1, 1, 0, v.ld, z.ld, Cin_p)
                    L.call("ydl_conv_fwd", ctypes.byref(gz), self.dt, _p(v.t), wv, _p(z.t), None, 0, st)
                    L.call("ydl_resize_acc_sums", self.dt, L.RESIZE_BILINEAR, _p(z.t), z.ld, _p(y.t), y.ld, v.N, v.H, v.W, Ho, Wo,
                           Cout, 0.0, 0.0, ws_ptr, Cout_p, st)
                return
            first = True
            for i, (v, c0_, r_, gv) in enumerate(subs):
                wv = ctypes.c_void_p(w.data_ptr() + c0_ * es)
                last = i == len(subs) - 1
                if r_:
                    z = self.new(v.N, Cout, v.H, v.W)
                    gz = L.ConvGeom(v.N, v.H, v.W, v.C, v.H, v.W, Cout, 1, 1, 0, v.ld, z.ld, Cin_p)
                    L.call("ydl_conv_fwd", ctypes.byref(gz), self.dt, _p(v.t), wv, _p(z.t), None, 0, st)
                    L.call("ydl_resize_fwd", self.dt, L.RESIZE_BILINEAR, _p(z.t), z.ld, _p(y.t), y.ld, v.N, v.H, v.W, Ho, Wo,
                           Cout, 0.0, 0.0, st)
                else:
                    L.call(fwd_entry if (last and ws_ptr is not None) else "ydl_conv_fwd", ctypes.byref(gv), self.dt, _p(v.t), wv, _p(y.t),
                           ws_ptr if last else None, 0 if first else 1, st)
                first = False

        sums = None
        if sums_mode:
            sums = self.zeroed(L.BN_REPLICAS * 2 * Cout_p)
            conv_fwd(_p(sums))
        elif self.train:
            if subs is None:
                nbytes, grid_m, block_m = gc[2], gc[3], gc[4]
            else:
                nbytes = L.lib().ydl_conv_fwd_stats_ws_bytes(gp, self.dt)
                grid_m, block_m = L.lib().ydl_conv_fwd_grid_m(gp, self.dt), L.lib().ydl_conv_fwd_block_m(gp, self.dt)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            conv_fwd(_p(ws))
            L.call("ydl_bn_finalize", _p(ws), grid_m, block_m,
                   npix, Cout, _p(m.bn.weight), _p(m.bn.bias), m.bn.eps, m.bn.momentum,
                   _p(m.bn.running_mean), _p(m.bn.running_var), _p(cf["mean"]), _p(cf["invstd"]),
                   _p(cf["scale"]), _p(cf["shift"]), rep, st)
        else:
            conv_fwd(None)
            L.call("ydl_bn_eval_coeffs", Cout, _p(m.bn.weight), _p(m.bn.bias), _p(m.bn.running_mean),
                   _p(m.bn.running_var), m.bn.eps, _p(cf["scale"]), _p(cf["shift"]), st)
        single = len(parts) == 1
        for (co, cw, o) in parts:
            cp = round_up(cw, 8)
            if sums is not None:
                # statistics, coefficients (kept for the backward) and running statistics of this channel group in the apply launch
                L.call("ydl_bn_act_fwd_sums", self.dt, _p(y.t if single else y.t[:, co:co + cw]), y.ld, _p(sums[co:]), Cout_p, npix,
                       _p(m.bn.weight[co:]), _p(m.bn.bias[co:]), m.bn.eps, m.bn.momentum, _p(m.bn.running_mean[co:]),
                       _p(m.bn.running_var[co:]), _p(cf["mean"][co:]), _p(cf["invstd"][co:]), _p(cf["scale"][co:]), _p(cf["shift"][co:]),
                       rep, _p(res.t) if res is not None else None, res.ld if res is not None else 0, res_mode, act,
                       _p(o.t), o.ld, npix, cw, cp, st)
                continue
            L.call("ydl_bn_act_fwd", self.dt, _p(y.t if single else y.t[:, co:co + cw]), y.ld,
                   _p(cf["scale"][co:]), _p(cf["shift"][co:]),
                   _p(res.t) if res is not None else None, res.ld if res is not None else 0, res_mode, act,
                   _p(o.t), o.ld, npix, cp, st)
        ret = outs[0] if single else None
        if not self.record:
            return ret
        if not self.train:
            raise RuntimeError("backward through eval-mode BatchNorm is not supported")

        train_w, train_g, train_b = m.trainable()
        if subs is None and x.need:
            self._use(x)
        if res is not None and res.need and res_mode in (L.RES_BEFORE_ACT, L.RES_AFTER_ACT):
            self._use(res)
        if (sums_mode and _cfg.bn_bwd_fuse() and self.dt == L.YDL_BF16 and rep == 1 and act in (L.ACT_NONE, L.ACT_SILU)
                and res_mode in (L.RES_NONE, L.RES_AFTER_ACT)):
            # what the dgrad of this output's LAST consumer needs to run this layer's reduce pass in its epilogue
            es2 = y.t.element_size()
            for (co, cw, o) in parts:
                if cw % 8 == 0 and o.aligned():
                    o.bn_src = (y.t.data_ptr() + co * es2, y.ld, cf, co, cw, act, y)

        def bw():
            if not any(o.is_set() for (_c, _w, o) in parts):
                return                                   # dead branch: parameters keep grad None
            if not (x.need or train_w or train_g or train_b) and subs is None and (res is None or not res.need):
                return                                   # frozen layer on a frozen prefix: nothing upstream wants a gradient
                                                         # (a residual operand that does is served by the BN-backward pass below)
            st2 = _stream()
            dy = self.new(x.N, Cout, Ho, Wo)
            # frozen BN parameters (requires_grad False): the sums still exist (dy needs them) but land in a scratch row
            gw, accw = m.grad_slot(self, "gamma") if train_g else (torch.empty(Cout_p, dtype=torch.float32, device=self.device), 0)
            gb, accb = m.grad_slot(self, "beta") if train_b else (torch.empty(Cout_p, dtype=torch.float32, device=self.device), 0)
            if accb != accw:                              # one accumulate flag serves both rows: give the scratch row defined contents
                zero_(gw if not train_g else gb)
                accw = 1
            for (co, cw, o) in parts:
                cp = round_up(cw, 8)
                dyv = dy.t if single else dy.t[:, co:co + cw]
                if not o.is_set():                        # this half feeds nothing that reaches the loss
                    zero_(dyv)
                    continue
                dout = self._gbuf(o)
                # gradient of the residual branch, written (or added) by the same kernel pass: dz for a residual joined before the
                # activation, dout itself for one joined after it
                dres_t, dres_ld, rmode = None, 0, res_mode
                if res is not None and res.need and res_mode == L.RES_AFTER_ACT and self._alias_grad(res, o, dout):
                    pass          # d/dres IS dout and nothing else has written it yet: res takes dout's buffer, no copy pass
                elif res is not None and res.need and res_mode in (L.RES_BEFORE_ACT, L.RES_AFTER_ACT):
                    gbuf, racc = self.grad_target(res)
                    dres_t, dres_ld = gbuf, res.ld
                    if racc:
                        rmode = res_mode | L.RES_GRAD_ACCUMULATE
                if sums_mode:
                    red = o.bnred
                    o.bnred = None
                    if red is not None and self._unwritten_since(o, red[1]):
                        # the reduce pass ran in the epilogue of the dgrad that wrote dout last (nothing has written since)
                        entry, sums_b = "ydl_bn_act_bwd_apply_sums", red[0]
                    else:
                        entry, sums_b = "ydl_bn_act_bwd_sums", self.zeroed(L.BN_REPLICAS * 2 * cp)
                    L.call(entry, self.dt, _p(y.t if single else y.t[:, co:co + cw]), y.ld, _p(dout), o.ld,
                           _p(o.t), o.ld, _p(cf["mean"][co:]), _p(cf["invstd"][co:]), _p(cf["scale"][co:]), _p(cf["shift"][co:]),
                           rmode, act, _p(dyv), dy.ld, _p(dres_t), dres_ld, _p(gw[co:]), _p(gb[co:]), accw,
                           _p(sums_b), npix, cw, cp, st2)
                    continue
                nws = L.lib().ydl_bn_bwd_ws_bytes(npix, cp) // 4
                ws2 = torch.empty(nws, dtype=torch.float32, device=self.device)
                L.call("ydl_bn_act_bwd", self.dt, _p(y.t if single else y.t[:, co:co + cw]), y.ld, _p(dout), o.ld,
                       _p(o.t), o.ld, _p(m.bn.weight[co:]), _p(cf["mean"][co:]), _p(cf["invstd"][co:]),
                       _p(cf["scale"][co:]), _p(cf["shift"][co:]), rmode, act, _p(dyv), dy.ld, _p(dres_t), dres_ld,
                       _p(gw[co:]), _p(gb[co:]), accw, _p(ws2), npix, cw, cp, st2)
            m.touch_bn()          # dgamma / dbeta kernels are enqueued: the DP hook may now reduce their bucket
            # weight gradient (f32, KRSC) accumulated into the parameter's grad storage; on the side stream when the
            # input gradient is needed too, so wgrad and dgrad of a layer overlap
            from . import config as _cfg
            if subs is not None:
                self._bw_split(m, subs, dy, wt, Cout_p, Ho, Wo, st2)
                return
            if (train_w and x.need and not _cfg.bn_bwd_fuse() and _cfg.fuse_pw_backward() and not _cfg.deterministic(self.dname)
                    and L.lib().ydl_conv_bwd_pw_supported(gp, self.dt)):
                # HBM-bound 1x1 layer: input and weight gradient in ONE pass over dy, on the main stream (ydl_conv_bwd_pw)
                gx, acc = self.grad_target(x)
                if m.wgrad(self, gp, x, dy, st2, fuse=(_p(wt), _p(gx), x.ld, acc)):
                    _keep = (geom,)
                    return
                L.call("ydl_conv_dgrad", gp, self.dt, _p(dy.t), _p(wt), _p(gx), acc, st2)     # (the weight gradient ran alone)
                return
            if not train_w:
                pass                                       # frozen weight: no weight-gradient launch, never marked touched
            elif x.need and _cfg.overlap_wgrad():
                # dy was allocated on the main stream and would die with this closure while the side stream still reads
                # it: park the reference until the streams are joined at the end of run_backward (graph-capture safe,
                # unlike Tensor.record_stream)
                self._defer_wgrad(lambda sst: m.wgrad(self, gp, x, dy, sst), keep=(dy, geom))
            else:
                m.wgrad(self, gp, x, dy, st2)
            if x.need:
                gx, acc = self.grad_target(x)
                if not self._try_bnred(x, gp, dy, wt, gx, acc, st2):
                    L.call("ydl_conv_dgrad", gp, self.dt, _p(dy.t), _p(wt), _p(gx), acc, st2)
            _keep = (geom,)   # keep the ctypes struct alive for the closure

        self.bw.append(bw)
        return ret

    def conv_bias_act(self, x: Var, m, act: int, res: Optional[Var] = None, res_mode: int = L.RES_NONE) -> Var:
        """eval-only folded Conv (``Conv.forward_fuse``, models/common.py:61-64): act(conv(x, w') + b') [+ res]"""
        if self.record:
            raise RuntimeError("a fused (BN-folded) Conv is inference-only")
        x = self._flat(x)
        if res is not None:
            res = self._flat(res)
        k, s, p = m.k, m.s, m.p
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        out = self.new(x.N, m.c2, Ho, Wo)
        w, _wt = m._fused_weights(self)
        geom = L.ConvGeom(x.N, x.H, x.W, m.c1, Ho, Wo, m.c2, k, s, p, x.ld, out.ld, 0)
        st = _stream()
        L.call("ydl_conv_fwd", ctypes.byref(geom), self.dt, _p(x.t), _p(w), _p(out.t), None, 0, st)
        f = m._fused
        L.call("ydl_bn_act_fwd", self.dt, _p(out.t), out.ld, _p(f["ones"]), _p(f["bias"]), _p(res.t) if res is not None else None,
               res.ld if res is not None else 0, res_mode, act, _p(out.t), out.ld, x.N * Ho * Wo, round_up(m.c2, 8), st)
        return out

    def _bw_split(self, m, subs, dy: Var, wt: torch.Tensor, Cout_p: int, Ho: int, Wo: int, st2) -> None:
        """backward of the commuted conv-over-concat: per source one wgrad into its column block of the weight gradient
        and one dgrad with its row block of wt; the resized source sees d z = bilinear_up^T (dy)."""
        from . import config as _cfg
        es = wt.element_size()
        jobs = []
        for (v, c0_, r_, gv) in subs:
            d = dy
            if r_:
                d = self.new(v.N, dy.C, v.H, v.W)
                L.call("ydl_resize_bwd", self.dt, L.RESIZE_BILINEAR, _p(dy.t), dy.ld, _p(d.t), d.ld, 0,
                       v.N, v.H, v.W, Ho, Wo, dy.C, 0.0, 0.0, st2)
                gv = L.ConvGeom(v.N, v.H, v.W, v.C, v.H, v.W, dy.C, 1, 1, 0, v.ld, d.ld, gv.ldw)
            jobs.append((v, c0_, gv, d))
        overlap = _cfg.overlap_wgrad() and any(v.need for (v, _c, _g, _d) in jobs)
        fused = set()
        if (m.trainable()[0] and not _cfg.bn_bwd_fuse() and _cfg.fuse_pw_backward() and not _cfg.deterministic(self.dname)):
            # sources whose column block is an HBM-bound 128 -> 128 layer of its own: both gradients in one pass (ydl_conv_bwd_pw)
            todo = [i for i, (v, c0_, gv, d) in enumerate(jobs)
                    if v.need and L.lib().ydl_conv_bwd_pw_supported(ctypes.byref(gv), self.dt)]
            rest = [i for i in range(len(jobs)) if i not in todo]
            for i in todo:
                v, c0_, gv, d = jobs[i]
                gx, acc = self.grad_target(v)
                wtv = ctypes.c_void_p(wt.data_ptr() + c0_ * Cout_p * es)
                if m.wgrad(self, ctypes.byref(gv), v, d, st2, col0=c0_, final=(not rest and i == todo[-1]), fuse=(wtv, _p(gx), v.ld, acc)):
                    fused.add(i)
                else:
                    L.call("ydl_conv_dgrad", ctypes.byref(gv), self.dt, _p(d.t), wtv, _p(gx), acc, st2)
                    fused.add(i)
            if fused:
                self._keep.append(tuple(g for (_v, _c, g, _d) in jobs))
                jobs = [jobs[i] for i in rest]
                if not jobs:
                    return
        if not m.trainable()[0]:
            pass                                           # frozen weight
        elif overlap:
            def launch(sst, jobs=jobs):
                for i, (v, c0_, gv, d) in enumerate(jobs):
                    m.wgrad(self, ctypes.byref(gv), v, d, sst, col0=c0_, final=(i == len(jobs) - 1))
            self._defer_wgrad(launch, keep=[d for (_v, _c, _g, d) in jobs] + [dy])
        else:
            for i, (v, c0_, gv, d) in enumerate(jobs):
                m.wgrad(self, ctypes.byref(gv), v, d, st2, col0=c0_, final=(i == len(jobs) - 1))
        for (v, c0_, gv, d) in jobs:
            if v.need:
                gx, acc = self.grad_target(v)
                wtv = ctypes.c_void_p(wt.data_ptr() + c0_ * Cout_p * es)
                L.call("ydl_conv_dgrad", ctypes.byref(gv), self.dt, _p(d.t), wtv, _p(gx), acc, st2)
        self._keep.append(tuple(g for (_v, _c, g, _d) in jobs))      # ctypes structs outlive the enqueue

    # ------------------------------------------------------------------ pooling / resize / copies
    def maxpool(self, x: Var, k: int, s: int, p: int, out: Optional[Var] = None) -> Var:
        x = self.materialize(x)
        Ho = (x.H + 2 * p - k) // s + 1
        Wo = (x.W + 2 * p - k) // s + 1
        if not x.aligned():
            x = self.copy(x, self.new(x.N, x.C, x.H, x.W, need=x.need, f32=(x.dt == L.YDL_F32)))
        if out is not None and not out.aligned():
            return self.copy(self.maxpool(x, k, s, p), out)
        if out is None:
            out = self.new(x.N, x.C, Ho, Wo, f32=(x.dt == L.YDL_F32))
        Cp = round_up(x.C, 4 if x.dt == L.YDL_F32 else 8)
        idx = torch.empty((x.N * Ho * Wo * Cp,), dtype=torch.uint8, device=self.device) if self.record else None
        L.call("ydl_maxpool_fwd", x.dt, _p(x.t), x.ld, _p(out.t), out.ld, _p(idx), x.N, x.H, x.W, Ho, Wo, x.C,
               k, s, p, _stream())
        if self.record:
            self._use(x)
            def bw():
                if not out.is_set() or not x.need:
                    return
                gx, acc = self.grad_target(x)
                L.call("ydl_maxpool_bwd", x.dt, _p(self._gbuf(out)), out.ld, _p(idx), _p(gx), x.ld, acc,
                       x.N, x.H, x.W, Ho, Wo, x.C, k, s, p, _stream())
            self.bw.append(bw)
        return out

    def sppf_pools(self, x: Var, k: int, outs: Sequence[Var]) -> Sequence[Var]:
        """SPPF's chain y1 = mp(x), y2 = mp(y1), y3 = mp(y2) (k x k, stride 1, pad k//2; seg_diceloss_yolov5.py:468-481) into the
        three given destinations (concat slices): one LDS-resident launch per direction when the plane fits
        (ydl_sppf_pool_fwd / _bwd, bit-identical to the chain), otherwise three ``maxpool`` calls."""
        o1, o2, o3 = outs
        x = self.materialize(x)
        fused = (k % 2 == 1 and x.aligned() and all(o.aligned() and o.ld == o1.ld and o.dt == x.dt for o in outs)
                 and all((o.N, o.C, o.H, o.W) == (x.N, x.C, x.H, x.W) for o in outs)
                 and L.lib().ydl_sppf_pool_supported(x.dt, x.H, x.W, x.C, k))
        if not fused:
            s1 = self.maxpool(x, k, 1, k // 2, out=o1)
            s2 = self.maxpool(s1, k, 1, k // 2, out=o2)
            s3 = self.maxpool(s2, k, 1, k // 2, out=o3)
            return s1, s2, s3
        Cp = round_up(x.C, 4 if x.dt == L.YDL_F32 else 8)
        n = x.N * x.H * x.W * Cp
        idx = [torch.empty((n,), dtype=torch.uint8, device=self.device) for _ in range(3)] if self.record else [None] * 3
        L.call("ydl_sppf_pool_fwd", x.dt, _p(x.t), x.ld, _p(o1.t), _p(o2.t), _p(o3.t), o1.ld, _p(idx[0]), _p(idx[1]), _p(idx[2]),
               x.N, x.H, x.W, x.C, k, _stream())
        if self.record:
            self._use(x)
            from . import config as _cfg
            hook = _cfg.sppf_argmax_hook()
            if hook is not None:          # test instrument: read or replace the three arg-max planes the backward will route by
                hook(idx)

            def bw():
                st = _stream()
                if all(o.is_set() for o in outs) and x.need:
                    g1, g2, g3 = (self._gbuf(o) for o in outs)
                    gx, acc = self.grad_target(x)
                    L.call("ydl_sppf_pool_bwd", x.dt, _p(g1), _p(g2), _p(g3), o1.ld, _p(idx[0]), _p(idx[1]), _p(idx[2]),
                           _p(gx), x.ld, acc, x.N, x.H, x.W, x.C, k, st)
                    return
                # a slice without a gradient (dead consumer): the chain link by link, as three maxpool closures would run it
                for src, dst, ix in ((o2, o3, idx[2]), (o1, o2, idx[1]), (x, o1, idx[0])):
                    if not dst.is_set() or not src.need:
                        continue
                    gx, acc = self.grad_target(src)
                    L.call("ydl_maxpool_bwd", x.dt, _p(self._gbuf(dst)), dst.ld, _p(ix), _p(gx), src.ld, acc,
                           x.N, x.H, x.W, x.H, x.W, x.C, k, 1, k // 2, st)
            self.bw.append(bw)
        return o1, o2, o3

    def resize(self, x: Var, Ho: int, Wo: int, mode: int, scale_h: float = 0.0, scale_w: float = 0.0,
               out: Optional[Var] = None) -> Var:
        """mode: L.RESIZE_NEAREST / RESIZE_BILINEAR (align_corners=False) / RESIZE_BILINEAR_AC (True)."""
        x = self.materialize(x)
        if not x.aligned():
            x = self.copy(x, self.new(x.N, x.C, x.H, x.W, need=x.need, f32=(x.dt == L.YDL_F32)))
        if out is not None and not out.aligned():
            return self.copy(self.resize(x, Ho, Wo, mode, scale_h, scale_w), out)
        if out is None:
            out = self.new(x.N, x.C, Ho, Wo, f32=(x.dt == L.YDL_F32))
        L.call("ydl_resize_fwd", x.dt, mode, _p(x.t), x.ld, _p(out.t), out.ld, x.N, x.H, x.W, Ho, Wo, x.C,
               scale_h, scale_w, _stream())
        if self.record:
            def bw():
                if not out.is_set() or not x.need:
                    return
                gx, acc = self.grad_target(x)
                L.call("ydl_resize_bwd", x.dt, mode, _p(self._gbuf(out)), out.ld, _p(gx), x.ld, acc,
                       x.N, x.H, x.W, Ho, Wo, x.C, scale_h, scale_w, _stream())
            self.bw.append(bw)
        return out

    def copy(self, x: Var, out: Var) -> Var:
        """out[:] = x (channel-slice aware); backward adds d(out) into d(x)."""
        if x.lazy or x.virtual:
            return self.materialize(x, out=out)
        L.call("ydl_copy2d", x.dt, _p(x.t), x.ld, _p(out.t), out.ld, x.npix, x.C, 0, _stream())
        if self.record:
            def bw():
                if not out.is_set() or not x.need:
                    return
                gx, acc = self.grad_target(x)
                L.call("ydl_copy2d", x.dt, _p(self._gbuf(out)), out.ld, _p(gx), x.ld, x.npix, x.C, acc, _stream())
            self.bw.append(bw)
        return out

    def add(self, a: Var, b: Var) -> Var:
        """out = a + b (element-wise, same shape); backward hands d(out) to both"""
        a, b = self._flat(a) if (a.lazy or a.virtual) else a, self._flat(b) if (b.lazy or b.virtual) else b
        out = self.new(a.N, a.C, a.H, a.W, f32=(a.dt == L.YDL_F32))
        st = _stream()
        L.call("ydl_copy2d", a.dt, _p(a.t), a.ld, _p(out.t), out.ld, a.npix, a.C, 0, st)
        L.call("ydl_copy2d", b.dt, _p(b.t), b.ld, _p(out.t), out.ld, b.npix, b.C, 1, st)
        if self.record:
            def bw():
                if not out.is_set():
                    return
                for v in (a, b):
                    if v.need:
                        gv, acc = self.grad_target(v)
                        L.call("ydl_copy2d", v.dt, _p(self._gbuf(out)), out.ld, _p(gv), v.ld, v.npix, v.C, acc, _stream())
            self.bw.append(bw)
        return out

    def concat(self, xs: Sequence[Var], align: bool = True) -> Var:
        """Concat along channels with the reference's auto-align (bilinear, align_corners=False, to the first
        input's size) — seg_diceloss_yolov5.py:484-507.
        When exactly one source has to be up-sampled and everything is 16-byte aligned the result is VIRTUAL (see
        Var.cat_parts): the typical consumer is a 1x1 convolution, which then never needs the up-sampled tensor."""
        from . import config as _cfg
        xs = [self.materialize(v) if (v.virtual and v.ext_src is not None) else v for v in xs]
        H, W = xs[0].LH, xs[0].LW
        ups = [v for v in xs if (v.LH, v.LW) != (H, W)]
        if (_cfg.commute_concat() and align and len(ups) == 1 and ups[0].LH < H and ups[0].LW < W
                and all(v.C % 8 == 0 and not v.virtual for v in xs) and len({v.dt for v in xs}) == 1):
            Ct = sum(v.C for v in xs)
            ph = torch.empty(1, dtype=self.tdt if xs[0].dt != L.YDL_F32 else torch.float32, device=self.device)
            cat = Var(self, ph.expand(xs[0].N, Ct, H, W), round_up(Ct, 8), any(v.need for v in xs))
            parts, c0 = [], 0
            for v in xs:
                parts.append((v, c0, (v.LH, v.LW) != (H, W)))
                c0 += v.C
            cat.cat_parts = parts
            return cat
        return self._concat_real(xs, align)

    def _concat_real(self, xs: Sequence[Var], align: bool) -> Var:
        """each source is written (copied / bilinearly resized) straight into its channel slice"""
        H, W = xs[0].LH, xs[0].LW
        Ct = sum(v.C for v in xs)
        cat = self.new(xs[0].N, Ct, H, W, f32=(xs[0].dt == L.YDL_F32))
        c0 = 0
        for v in xs:
            sl = cat.slice(c0, c0 + v.C)
            if (v.LH, v.LW) == (H, W):
                self.copy(v, sl)
            elif align:
                self.resize(v, H, W, L.RESIZE_BILINEAR, out=sl)
            else:
                raise RuntimeError("Concat: spatial sizes differ")
            c0 += v.C
        return cat

    # ------------------------------------------------------------------ softmax head
    def softmax(self, x: Var) -> Var:
        """nn.Softmax(1) into an f32 NHWC Var (probabilities stay f32 even in bf16 mode)."""
        if x.virtual:
            x = self.materialize(x)
        out = self.new(x.N, x.C, x.H, x.W, zero=True, f32=True)
        out.rep = x.rep                                   # point-wise: stays lazy
        sn, sc, sh, sw = out.t.stride()
        L.call("ydl_softmax_fwd", x.dt, _p(x.t), x.ld, _p(out.t), sn, sc, sh, sw, x.N, x.H, x.W, x.C, 1, 1, _stream())
        if self.record:
            def bw():
                if not out.is_set() or not x.need:
                    return
                dp = self._gbuf(out)
                gx, acc = self.grad_target(x)
                assert acc == 0, "softmax input must have a single consumer"
                L.call("ydl_softmax_bwd", x.dt, _p(out.t), _p(dp), sn, sc, sh, sw, _p(gx), x.ld, x.N, x.H, x.W, x.C, 1, 1,
                       _stream())
            self.bw.append(bw)
        return out

    def softmax_nchw(self, x: Var) -> torch.Tensor:
        """nn.Softmax(1) producing the region's external output directly: f32 NCHW contiguous (no extra pass).
        The region owner must call softmax_nchw_backward with the incoming gradient.
        When x is a lazily up-sampled tensor the stored-resolution probabilities are kept as well (``self.lazy_out``):
        SegmentationLoss uses them to evaluate the loss and its gradient per stored pixel (ydl_seg_loss_rep_*)."""
        assert not x.virtual
        p = torch.empty((x.N, x.C, x.LH, x.LW), dtype=torch.float32, device=self.device)
        sn, sc, sh, sw = p.stride()
        L.call("ydl_softmax_fwd", x.dt, _p(x.t), x.ld, _p(p), sn, sc, sh, sw, x.N, x.H, x.W, x.C, x.rep[0], x.rep[1],
               _stream())
        self.lazy_out = None
        if x.rep != (1, 1) and self.record:
            plow = torch.empty((x.N, x.C, x.H, x.W), dtype=torch.float32, device=self.device)
            sn, sc, sh, sw = plow.stride()
            L.call("ydl_softmax_fwd", x.dt, _p(x.t), x.ld, _p(plow), sn, sc, sh, sw, x.N, x.H, x.W, x.C, 1, 1, _stream())
            self.lazy_out = LazyOutput(plow, x.rep)
        return p

    def softmax_nchw_backward(self, x: Var, p: torch.Tensor, dp: torch.Tensor) -> None:
        dp = dp.detach()
        gx, acc = self.grad_target(x)
        assert acc == 0
        lazy = getattr(self, "lazy_out", None)
        dlow = None
        if lazy is not None:
            dlow, lazy.dlow = lazy.dlow, None
        if dlow is not None:
            if all(s == 0 for s in dp.stride()) and getattr(lazy, "dummy", None) is not None:
                # the loss already summed its gradient over the replicas: soft-max backward at the stored resolution
                lazy.dummy = None
                sn, sc, sh, sw = lazy.low.stride()
                L.call("ydl_softmax_bwd", x.dt, _p(lazy.low), _p(dlow), sn, sc, sh, sw, _p(gx), x.ld, x.N, x.H, x.W, x.C,
                       1, 1, _stream())
                return
            # pred had other consumers besides the loss: fold the summed part into one replica of the dense gradient
            dp = dp.float().contiguous().clone()
            dp.view(x.N, x.C, x.H, x.rep[0], x.W, x.rep[1])[:, :, :, 0, :, 0] += dlow
        if dp.dtype != torch.float32 or dp.stride() != p.stride():
            dp = dp.float().contiguous()
        sn, sc, sh, sw = p.stride()
        L.call("ydl_softmax_bwd", x.dt, _p(p), _p(dp), sn, sc, sh, sw, _p(gx), x.ld, x.N, x.H, x.W, x.C, x.rep[0], x.rep[1],
               _stream())

    # ------------------------------------------------------------------ GAM pieces
    def global_pool(self, x: Var, kind: str) -> Var:
        """AdaptiveAvgPool2d(1) / AdaptiveMaxPool2d(1) -> (N, C, 1, 1)"""
        x = self.materialize(x)
        avg = self.new(x.N, x.C, 1, 1, zero=True)
        mx = self.new(x.N, x.C, 1, 1, zero=True)
        amax = torch.empty((x.N, round_up(x.C, 8)), dtype=torch.int32, device=self.device)
        HW = x.H * x.W
        L.call("ydl_global_pool_fwd", x.dt, _p(x.t), x.ld, _p(avg.t), avg.ld, _p(mx.t), mx.ld, _p(amax), x.N, HW, x.C,
               _stream())
        out = avg if kind == "avg" else mx
        if self.record:
            def bw():
                if not out.is_set() or not x.need:
                    return
                g = self._gbuf(out)
                gx, acc = self.grad_target(x)
                L.call("ydl_global_pool_bwd", x.dt, _p(g) if kind == "avg" else None, out.ld,
                       _p(g) if kind == "max" else None, out.ld, _p(amax), _p(gx), x.ld, acc, x.N, HW, x.C, _stream())
            self.bw.append(bw)
        return out

    def gate_mul(self, x: Var, a: Var, b: Var) -> Var:
        """out = x * sigmoid(a + b) with a, b of shape (N, C, 1, 1) (GAM: the bilinear expansion of a 1x1 map is constant)"""
        x = self.materialize(x)
        gate = torch.empty((x.N, x.C), dtype=torch.float32, device=self.device)
        L.call("ydl_gate_fwd", x.dt, _p(a.t), a.ld, _p(b.t), b.ld, _p(gate), x.N, x.C, _stream())
        out = self.scale_channels(x, gate)
        if self.record:
            def bw():
                if not out.is_set():
                    return
                dout = self._gbuf(out)
                st = _stream()
                HW = x.H * x.W
                dgate = torch.empty((x.N, x.C), dtype=torch.float32, device=self.device)
                L.call("ydl_channel_dot", x.dt, _p(dout), out.ld, _p(x.t), x.ld, _p(dgate), x.N, HW, x.C, st)
                ga, acca = self.grad_target(a)
                gb, accb = self.grad_target(b)
                L.call("ydl_gate_bwd", x.dt, _p(gate), _p(dgate), _p(ga), a.ld, acca, _p(gb), b.ld, accb, x.N, x.C, st)
                if x.need:
                    if x.is_set():
                        tmp = self.new_like(x)
                        L.call("ydl_scale_channels", x.dt, _p(dout), out.ld, _p(gate), _p(tmp.t), tmp.ld, x.N, HW, x.C, st)
                        gx, acc = self.grad_target(x)
                        L.call("ydl_copy2d", x.dt, _p(tmp.t), tmp.ld, _p(gx), x.ld, x.npix, x.C, 1, st)
                    else:
                        gx, _ = self.grad_target(x)
                        L.call("ydl_scale_channels", x.dt, _p(dout), out.ld, _p(gate), _p(gx), x.ld, x.N, HW, x.C, st)
            self.bw.append(bw)
        return out

    # ------------------------------------------------------------------ DCNv3 module pieces (NHWC is the tape's native layout)
    def _flat(self, x: Var) -> Var:
        x = self.materialize(x)
        if not x.aligned():
            x = self.copy(x, self.new(x.N, x.C, x.H, x.W, need=x.need, f32=(x.dt == L.YDL_F32)))
        return x

    def linear(self, x: Var, lin) -> Var:
        """nn.Linear on the channel dimension of an NHWC tensor (modules/dcnv3.py:92-100) = 1x1 convolution + bias.
        ``lin`` holds weight [out, in] (KRSC of a 1x1 conv is the same memory) and bias [out]."""
        x = self._flat(x)
        Cin, Cout = lin.in_features, lin.out_features
        if x.C != Cin:
            raise RuntimeError(f"Linear input channel mismatch: got {x.C}, weight expects {Cin}")
        w, wt = lin.compute_weights(self)
        out = self.new(x.N, Cout, x.H, x.W)
        geom = L.ConvGeom(x.N, x.H, x.W, Cin, x.H, x.W, Cout, 1, 1, 0, x.ld, out.ld, 0)
        gp = ctypes.byref(geom)
        st = _stream()
        L.call("ydl_conv_fwd", gp, self.dt, _p(x.t), _p(w), _p(out.t), None, 0, st)
        if lin.bias is not None:
            cp = round_up(Cout, 8)
            ones, bias = lin.bias_coeffs(self.device)
            L.call("ydl_bn_act_fwd", self.dt, _p(out.t), out.ld, _p(ones), _p(bias), None, 0, L.RES_NONE, L.ACT_NONE,
                   _p(out.t), out.ld, x.npix, cp, st)
        if self.record:
            def bw():
                if not out.is_set():
                    return
                st2 = _stream()
                dout = self._gbuf(out)
                dov = Var(self, dout, out.ld, False)
                if lin.bias is not None and lin.bias.requires_grad:
                    gb = lin._grad_of(lin.bias)
                    ws = torch.empty(L.lib().ydl_channel_sum_ws_bytes(Cout) // 4, dtype=torch.float32, device=self.device)
                    L.call("ydl_channel_sum", self.dt, _p(dout), out.ld, _p(gb), _p(ws), x.npix, Cout, 1, st2)
                    from . import config as _cfg
                    _cfg.mark_touched(lin.bias)
                lin.wgrad(self, gp, x, dov, st2)
                if x.need:
                    gx, acc = self.grad_target(x)
                    L.call("ydl_conv_dgrad", gp, self.dt, _p(dout), _p(wt), _p(gx), acc, st2)
                self._keep.append(geom)
            self.bw.append(bw)
        return out

    def dwconv_bn_act(self, x: Var, m, act: int) -> Var:
        """depth-wise ``Conv(c, c, k, g=c)`` = conv -> train-mode BN -> SiLU (modules/dcnv3.py:89, :35-39)"""
        x = self._flat(x)
        C, k, p = m.c1, m.k, m.p
        y = self.new(x.N, C, x.H, x.W)
        out = self.new(x.N, C, x.H, x.W)
        st = _stream()
        wm = m.master_dw()                                   # f32 [C][k*k]
        L.call("ydl_dwconv_fwd", self.dt, _p(x.t), x.ld, _p(wm), _p(y.t), y.ld, x.N, x.H, x.W, C, k, p, st)
        cf = m.coeffs(self.device)
        npix = x.npix
        cp = round_up(C, 8)
        if self.train:
            ws = torch.empty(L.lib().ydl_bn_stats_ws_bytes(npix, C) // 4, dtype=torch.float32, device=self.device)
            bm = L.lib().ydl_bn_stats_block_m()
            L.call("ydl_bn_stats", self.dt, _p(y.t), y.ld, _p(ws), npix, C, st)
            L.call("ydl_bn_finalize", _p(ws), (npix + bm - 1) // bm, bm, npix, C, _p(m.bn.weight), _p(m.bn.bias), m.bn.eps,
                   m.bn.momentum, _p(m.bn.running_mean), _p(m.bn.running_var), _p(cf["mean"]), _p(cf["invstd"]),
                   _p(cf["scale"]), _p(cf["shift"]), 1, st)
        else:
            L.call("ydl_bn_eval_coeffs", C, _p(m.bn.weight), _p(m.bn.bias), _p(m.bn.running_mean), _p(m.bn.running_var),
                   m.bn.eps, _p(cf["scale"]), _p(cf["shift"]), st)
        L.call("ydl_bn_act_fwd", self.dt, _p(y.t), y.ld, _p(cf["scale"]), _p(cf["shift"]), None, 0, L.RES_NONE, act,
               _p(out.t), out.ld, npix, cp, st)
        if self.record:
            if not self.train:
                raise RuntimeError("backward through eval-mode BatchNorm is not supported")

            def bw():
                if not out.is_set():
                    return
                st2 = _stream()
                dy = self.new(x.N, C, x.H, x.W)
                train_w, train_g, train_b = m.trainable()
                gw, accw = m.grad_slot(self, "gamma") if train_g else (torch.empty(cp, dtype=torch.float32, device=self.device), 0)
                gb, accb = m.grad_slot(self, "beta") if train_b else (torch.empty(cp, dtype=torch.float32, device=self.device), 0)
                if accb != accw:
                    zero_(gw if not train_g else gb)
                    accw = 1
                ws2 = torch.empty(L.lib().ydl_bn_bwd_ws_bytes(npix, cp) // 4, dtype=torch.float32, device=self.device)
                L.call("ydl_bn_act_bwd", self.dt, _p(y.t), y.ld, _p(self._gbuf(out)), out.ld, _p(out.t), out.ld,
                       _p(m.bn.weight), _p(cf["mean"]), _p(cf["invstd"]), _p(cf["scale"]), _p(cf["shift"]), L.RES_NONE, act,
                       _p(dy.t), dy.ld, None, 0, _p(gw), _p(gb), accw, _p(ws2), npix, C, cp, st2)
                m.touch_bn()
                if train_w:
                    gk = m.grad_dw()
                    ws3 = torch.empty(L.lib().ydl_dwconv_wgrad_ws_bytes(C, k) // 4, dtype=torch.float32, device=self.device)
                    L.call("ydl_dwconv_wgrad", self.dt, _p(x.t), x.ld, _p(dy.t), dy.ld, _p(gk), _p(ws3), x.N, x.H, x.W, C, k, p, st2)
                    from . import config as _cfg
                    _cfg.mark_touched(m.conv.weight)
                if x.need:
                    gx, acc = self.grad_target(x)
                    L.call("ydl_dwconv_dgrad", self.dt, _p(dy.t), dy.ld, _p(wm), _p(gx), x.ld, acc, x.N, x.H, x.W, C, k, p, st2)
            self.bw.append(bw)
        return out

    def group_softmax(self, x: Var, G: int, P: int) -> Var:
        """softmax over the P sampling points of each group (modules/dcnv3.py:122-123)"""
        x = self._flat(x)
        assert x.C == G * P
        out = self.new(x.N, x.C, x.H, x.W)
        L.call("ydl_group_softmax_fwd", self.dt, _p(x.t), x.ld, _p(out.t), out.ld, x.npix, G, P, _stream())
        if self.record:
            def bw():
                if not out.is_set() or not x.need:
                    return
                gx, acc = self.grad_target(x)
                L.call("ydl_group_softmax_bwd", self.dt, _p(out.t), out.ld, _p(self._gbuf(out)), out.ld, _p(gx), x.ld, acc,
                       x.npix, G, P, _stream())
            self.bw.append(bw)
        return out

    def dcnv3(self, x: Var, offset: Var, mask: Var, k: int, s: int, pad: int, dil: int, G: int, Gc: int, scale: float) -> Var:
        """the deformable sampling op itself (ydl_dcnv3_fwd / _bwd; functions/dcnv3_func.py:19-61).  The op wants dense
        rows: channel counts that are not multiples of 8 (offset 18*G, mask 9*G) are packed first."""
        def dense(v: Var) -> torch.Tensor:
            v = self.materialize(v)
            if v.ld == v.C and v.parent is None:
                return v.t.permute(0, 2, 3, 1)
            buf = torch.empty((v.N, v.H, v.W, v.C), dtype=v.t.dtype, device=self.device)
            L.call("ydl_copy2d", v.dt, _p(v.t), v.ld, _p(buf), v.C, v.npix, v.C, 0, _stream())
            return buf
        xd, od, md = dense(x), dense(offset), dense(mask)
        N, H, W, C = xd.shape
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // s + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // s + 1
        out = self.new(N, C, Ho, Wo)
        assert out.ld == C, "DCNv3: channels must be a multiple of 8"
        L.call("ydl_dcnv3_fwd", self.dt, _p(xd), _p(od), _p(md), _p(out.t), k, k, s, s, pad, pad, dil, dil, G, Gc,
               ctypes.c_float(scale), N, H, W, Ho, Wo, _stream())
        if self.record:
            def bw():
                if not out.is_set():
                    return
                st2 = _stream()
                gin = zero_(torch.empty(xd.shape, dtype=torch.float32, device=self.device))
                goff = torch.empty(od.shape, dtype=torch.float32, device=self.device)
                gmsk = torch.empty(md.shape, dtype=torch.float32, device=self.device)
                L.call("ydl_dcnv3_bwd", self.dt, _p(xd), _p(od), _p(md), _p(self._gbuf(out)), _p(gin), _p(goff), _p(gmsk),
                       k, k, s, s, pad, pad, dil, dil, G, Gc, ctypes.c_float(scale), N, H, W, Ho, Wo, st2)
                for v, g32 in ((x, gin), (offset, goff), (mask, gmsk)):
                    if v.need:
                        gv, acc = self.grad_target(v)
                        L.call("ydl_cast_f32", v.dt, _p(g32), g32.shape[-1], _p(gv), v.ld, v.npix, v.C, acc, st2)
            self.bw.append(bw)
        return out

    def scale_channels(self, x: Var, gate: torch.Tensor) -> Var:
        x = self.materialize(x)
        out = self.new_like(x)
        L.call("ydl_scale_channels", x.dt, _p(x.t), x.ld, _p(gate), _p(out.t), out.ld, x.N, x.H * x.W, x.C, _stream())
        return out
