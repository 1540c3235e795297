"""The training step as a recorded launch list, re-issued from C with one call per step (include/ydl.h "launch-list replay").

The eager path issues ~330 entry-point calls per step from Python through ctypes — 6-7 ms of host time against ~7 ms of GPU time,
so the step is host-bound as soon as the host is busy, and kernel gains stop registering.  A single-stream HIP graph removes the
host time but also the two-stream overlap of the eager step (weight gradients and the dead head branch run beside the main chain),
and capturing both streams into one graph replays slower still.  This module keeps eager's exact launch structure and
drops its host cost instead:

  * RECORD: one real training step runs with ``_lib.call`` mirroring every entry-point call (name, arguments, stream) and every
    cross-stream edge (``tape.stream_wait``) into a C-side list (csrc/replay.cpp).  The step allocates from a PRIVATE allocator
    pool (torch.cuda.MemPool) and backward runs on the recording thread, so every address the list names belongs to this object
    and the allocator hands the same blocks to nobody else.
  * REPLAY: ``ydl_replay_run`` walks the list — the same C entry points with the same arguments on the same two streams with
    the same event edges.  Host-side bookkeeping the kernels do not do (BatchNorm ``num_batches_tracked``, the EMA counter and
    learning-rate vector, the weight epoch) is advanced per replay, as graph.py does for HIP graphs.

What must hold (checked where it can be): no device work of the step bypasses the C ABI (the taped region issues no ATen kernel;
``tape.zero_`` replaces fills), inputs are static device tensors (copy each batch into ``imgs`` / ``targets``), shapes and the set
of parameters with gradients do not change.  ``poison()`` overwrites every free byte of the private pool; a replay that still
matches eager afterwards proves nothing in the step depended on un-replayed writes (tests/test_gpu_replay.py)."""
from __future__ import annotations

import ctypes
import time
import struct
from typing import List, Optional

import torch

from . import _lib as L
from . import config
from .modules import _BNHolder
from .optim import FlatSGDEMA

_G = L._G


class Recorder:
    """mirror of the entry-point calls of one step into a C-side ydl_replay (see _lib.call / tape.stream_wait)"""

    def __init__(self):
        lib = L.lib()
        self.lib = lib
        self.h = ctypes.c_void_p(lib.ydl_replay_create())
        self.fn = {lib.ydl_replay_fn_name(i).decode(): i for i in range(lib.ydl_replay_fn_count())}
        self.handles: List[int] = []        # raw hipStream_t per slot
        self._slot = {}
        self.n_events = 0
        self.calls = 0

    def close(self) -> None:
        if self.h is not None:
            self.lib.ydl_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def slot(self, handle) -> int:
        handle = int(handle or 0)
        s = self._slot.get(handle)
        if s is None:
            s = self._slot[handle] = len(self.handles)
            self.handles.append(handle)
        return s

    @staticmethod
    def _ptr(a) -> int:
        if a is None:
            return 0
        if isinstance(a, int):
            return a
        v = getattr(a, "value", None)
        if v is not None or isinstance(a, ctypes.c_void_p):
            return int(v or 0)
        raise TypeError(f"cannot record pointer argument {a!r}")

    def add(self, name: str, args) -> None:
        fn = self.fn.get(name)
        if fn is None:
            raise RuntimeError(f"{name} is not a recordable entry point (its last parameter must be the stream)")
        argtypes = L.SIGNATURES[name][1]
        if len(args) != len(argtypes):
            raise TypeError(f"{name}: {len(args)} arguments for {len(argtypes)} parameters")
        vals = []
        for a, t in zip(args[:-1], argtypes[:-1]):
            if t is _G or t is L._R:
                obj = getattr(a, "_obj", None)
                if obj is None:
                    obj = a.contents
                vals.append(ctypes.addressof(obj))
            elif t is L._vp or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
                vals.append(self._ptr(a))
            elif t is L._f:
                vals.append(struct.unpack("<q", struct.pack("<d", float(getattr(a, "value", a))))[0])
            else:
                vals.append(int(getattr(a, "value", a)))
        n = len(vals)
        arr = (ctypes.c_int64 * max(n, 1))(*vals)
        L.check(self.lib.ydl_replay_add_call(self.h, fn, arr, n, self.slot(self._ptr(args[-1]))), "ydl_replay_add_call")
        self.calls += 1

    def edge(self, src_handle, dst_handle) -> None:
        """everything enqueued so far on ``src`` happens before what ``dst`` gets next"""
        ev = self.n_events
        self.n_events += 1
        L.check(self.lib.ydl_replay_add_event_record(self.h, ev, self.slot(src_handle)), "ydl_replay_add_event_record")
        L.check(self.lib.ydl_replay_add_event_wait(self.h, ev, self.slot(dst_handle)), "ydl_replay_add_event_wait")

    def size(self) -> int:
        return int(self.lib.ydl_replay_size(self.h))

    def run(self, first: int, last: int, handles=None) -> None:
        """re-issue operations [first, last) on the recorded streams, or on ``handles`` (one raw stream per slot: the list only names
        slots, any set of distinct streams preserves its dependencies)"""
        hs = self.handles if handles is None else handles
        arr = (ctypes.c_void_p * len(hs))(*[ctypes.c_void_p(h) for h in hs])
        L.check(self.lib.ydl_replay_run(self.h, first, last, arr, len(hs)), "ydl_replay_run")


class ReplayedTrainStep:
    """``step()`` = zero_grad + forward + loss + backward + fused SGD/EMA step of ``model`` on the static batch tensors, issued as ONE
    C call.  Same interface as graph.GraphedTrainStep (``step()`` returns the device-resident [total, ce, overlap] scalars).

    With a data-parallel wrapper (``dp``) the list is cut where the eager step launches a gradient bucket: a replay alternates
    "segment, bucket collective" exactly like eager mode, collectives overlapping the remaining backward segments."""

    def __init__(self, model, criterion, optimizer: FlatSGDEMA, imgs: torch.Tensor, targets: torch.Tensor, dp=None, warmup: int = 2,
                 prioritize: bool = False):
        if getattr(criterion, "sync", False):
            raise ValueError("a launch list needs a loss that does not sync: SegmentationLoss(..., sync=False)")
        self.model, self.criterion, self.opt, self.dp = model, criterion, optimizer, dp
        self.imgs, self.targets = imgs, targets
        self.multi = dp is not None and dp.reducer.world > 1
        self._bns: List[_BNHolder] = [m for m in model.modules() if isinstance(m, _BNHolder)]
        dev = imgs.device
        # persistent state lives OUTSIDE the private pool: a tensor that is born inside it during the recording pass can land on an
        # address an earlier kernel of the step used as scratch, and every replay would then overwrite it (found the hard way: the
        # optimizer's hyper-parameter vector)
        self._one = torch.ones((), dtype=torch.float32, device=dev)        # gradient seed of loss.backward
        self._loss_out = torch.zeros(3, dtype=torch.float32, device=dev)   # [total, ce, overlap] of the last step
        optimizer.ensure_hyper()
        for _ in range(max(warmup, 1)):        # lazily created caches (compute weights, descriptors, bias rows), momentum flags settle
            self._eager_step()
        optimizer.ensure_runs_table()
        torch.cuda.synchronize()
        self.main = torch.cuda.current_stream()
        self.pool = torch.cuda.MemPool()
        self.rec = Recorder()
        self.rec.slot(self.main.cuda_stream)                                 # slot 0 = the stream the step is issued on
        self._cuts: List[tuple] = []        # (list position, bucket, stream current there): where a bucket collective is launched
        nbt0 = [bn._nbt_pending for bn in self._bns]
        scale = 1.0
        red = dp.reducer if self.multi else None
        with torch.cuda.use_mem_pool(self.pool), torch.autograd.set_multithreading_enabled(False):
            if red is not None:
                red.begin_step()
                red.capture_cb = self._cut
            L.set_recorder(self.rec)
            try:
                self._fwd_bwd()
            finally:
                L.set_recorder(None)
                if red is not None:
                    red.capture_cb = None
            self._n_fb = self.rec.size()
            if self.multi:
                scale = dp.finish()
            optimizer.prepare_step(scale)
            L.set_recorder(self.rec)
            try:
                optimizer.step_device_hyper()
            finally:
                L.set_recorder(None)
        self._n_all = self.rec.size()
        # optional: replay slot 0 (the main chain: forward, BN backward, input gradients, optimizer) on a HIGH-priority stream and
        # the other slots (weight gradients, dead head branch) on low-priority ones, so that the hardware scheduler hands CUs to
        # the critical path first and the side work fills what is left
        self._handles = None
        if prioritize and not self.multi:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            self._prio_streams = [torch.cuda.Stream(priority=hi)] + [torch.cuda.Stream(priority=lo) for _ in self.rec.handles[1:]]
            self._handles = [st.cuda_stream for st in self._prio_streams]
            self.main = self._prio_streams[0]
        self._nbt_per_replay = [bn._nbt_pending - a for bn, a in zip(self._bns, nbt0)]
        self._runs_tab = getattr(optimizer, "_runs_dev", None)      # the recorded list holds the raw pointer of this table: keep it alive
        self.host_run_s = 0.0
        self.host_launch_s = 0.0          # data parallel: host time inside the bucket launches between the segments ...
        self.host_finish_s = 0.0          # ... and inside the wait for the in-flight buckets at the end of backward
        torch.cuda.synchronize()

    # ------------------------------------------------------------------
    def _cut(self, bucket: int) -> None:
        """called by the reducer where the eager step launches bucket ``bucket``: remember the list position and the stream that is
        current there (weight-gradient batches are flushed inside the side-stream context), then launch it — the recording pass is a
        real step and its gradients must be reduced like any other's"""
        self._cuts.append((self.rec.size(), bucket, torch.cuda.current_stream()))
        L.set_recorder(None)                 # the collective's own staging kernels are issued per step by the reducer, not replayed
        try:
            self.dp.reducer._launch(bucket)
        finally:
            L.set_recorder(self.rec)

    def _eager_step(self):
        self.opt.zero_grad()
        if self.dp:
            self.dp.begin()
        out = self.model(self.imgs)
        loss, items = self.criterion(out, self.targets)
        loss.backward(self._one)
        scale = self.dp.finish() if self.dp else 1.0
        self.opt.step(grad_scale=scale)
        return items

    def _fwd_bwd(self):
        self.opt.zero_grad()
        out = self.model(self.imgs)
        loss, items = self.criterion(out, self.targets)
        loss.backward(self._one)
        # the loss vector lives in the private pool (later steps' early kernels may use its block as scratch): the last entry of the
        # forward/backward list copies it to a persistent buffer
        from .tape import _p, _stream
        vec = getattr(items[0], "_base", None)
        if vec is None or vec.numel() != 3:
            raise RuntimeError("ReplayedTrainStep expects SegmentationLoss(..., sync=False): loss items that are views of one 3-float vector")
        L.call("ydl_copy2d", L.YDL_F32, _p(vec), 3, _p(self._loss_out), 3, 1, 3, 0, _stream())
        self.loss_items = [self._loss_out[0], self._loss_out[1], self._loss_out[2]]
        return self.loss_items

    # ------------------------------------------------------------------
    def poison(self, byte: int = 0xFF) -> None:
        """test hook: overwrite every FREE block of the private pool (everything the recorded step allocated and released) so that
        a value a replay would read without having written it is garbage, not a leftover of the recording pass"""
        torch.cuda.synchronize()
        n = 0
        for seg in self.pool.snapshot():
            for b in seg["blocks"]:
                if b["state"] == "inactive":
                    _memset(b["address"], byte, b["size"])
                    n += b["size"]
        torch.cuda.synchronize()
        return n

    @property
    def launches(self) -> int:
        return self.rec.calls

    def step(self):
        """one training step; returns the (device-resident) [total, ce, overlap] loss scalars.  ``self.host_run_s`` accumulates the
        host time spent inside ``ydl_replay_run`` alone (what re-issuing the list costs, net of the optimizer's hyper-parameter ring
        wait and of the collectives between the segments)"""
        cur = torch.cuda.current_stream()
        foreign = cur.cuda_stream != self.main.cuda_stream
        if foreign:
            self.main.wait_stream(cur)                 # whatever the caller enqueued (the batch) precedes the list
        scale = 1.0
        pc = time.perf_counter
        if self.multi:
            red = self.dp.reducer
            red.begin_step()
            pos = 0
            for cut, bi, st in self._cuts:
                t0 = pc()
                self.rec.run(pos, cut)
                self.host_run_s += pc() - t0
                t0 = pc()
                with torch.cuda.stream(st):
                    red._launch(bi)                    # overlaps with the remaining segments
                self.host_launch_s += pc() - t0
                pos = cut
            t0 = pc()
            self.rec.run(pos, self._n_fb)
            self.host_run_s += pc() - t0
            t0 = pc()
            scale = self.dp.finish()
            self.host_finish_s += pc() - t0
            self.opt.prepare_step(scale)               # H2D of {lr, momentum, wd, scale, EMA decay} on the CURRENT stream ...
            if foreign:
                self.main.wait_stream(cur)             # ... which the optimizer segment on the main slot must see
            t0 = pc()
            self.rec.run(self._n_fb, self._n_all)
            self.host_run_s += pc() - t0
        else:
            self.opt.prepare_step(scale)               # H2D of {lr, momentum, wd, scale, EMA decay} on the CURRENT stream
            if foreign or self._handles is not None:
                self.main.wait_stream(cur)             # the list's optimizer kernel reads that vector: order the main slot behind the copy
            t0 = pc()
            self.rec.run(0, self._n_all, self._handles)
            self.host_run_s += pc() - t0
        for bn, k in zip(self._bns, self._nbt_per_replay):
            bn._nbt_pending += k
        config.bump_weight_epoch()
        if cur.cuda_stream != self.main.cuda_stream:
            cur.wait_stream(self.main)
        return self.loss_items


def _memset(addr: int, byte: int, nbytes: int) -> None:
    """hipMemset on a raw device address (test hook of ReplayedTrainStep.poison)"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemset.restype = ctypes.c_int
    rc = hip.hipMemset(ctypes.c_void_p(addr), byte, nbytes)
    if rc != 0:
        raise RuntimeError(f"hipMemset failed ({rc})")
