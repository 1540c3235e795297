"""Batch-dimension data parallelism: one process per GPU, RCCL sum-all-reduce of the flat gradient arena over xGMI,
overlapped with backward.

The reference never creates a process group for segmentation (only nn.DataParallel; BN statistics stay
per-GPU) — this is the new capability modelled on utils/torch_utils.py:55-63.  Design for the 8-GPU xGMI mesh:
  * gradients already live in ONE contiguous f32 arena (yolo_dual_amd.optim.FlatSGDEMA), laid out in module order, so
    backward completes it from the tail to the head;
  * the arena is cut into a few large buckets (default 16 MiB: with 7 point-to-point links per GPU a ring step moves
    1/8 of the bucket per link, so buckets must be MBs to reach link bandwidth) and a bucket's all-reduce is launched
    on RCCL's stream as soon as the last gradient inside it has been enqueued (hook from the wgrad / BN-backward
    launches), while earlier layers are still computing;
  * parameters that get no gradient (dead head layers, ResNet layer4) are known after the first step and are
    excluded: a bucket never waits for them, and their (zero) ranges are not sent;
  * averaging (1/world) is folded into the optimizer kernel's ``grad_scale``.
BatchNorm stays per-GPU exactly like the reference's DataParallel (its SyncBN branch is dead code).

Two knobs exist for the xGMI mesh (no 8-GPU node has run this code yet — there is no scaling curve):
  * ``algo="rs_ag"``: a hand-rolled reduce-scatter + all-gather in which every rank exchanges 1/world of the bucket with
    EVERY peer at once (point-to-point sends to all 7 neighbours, so all 7 links carry traffic, where a ring is bound by one
    link per direction), sums the received pieces locally in a fixed rank order (bitwise reproducible), and sends its reduced
    piece back to every peer.  The comparator for RCCL's own all-reduce (``algo="allreduce"``, the default).
  * ``wire="bf16"``: gradients cross the links as bf16 (31 MB instead of 62 MB for BASELINE config 2).  With ``rs_ag`` the
    local sum stays f32 (one rounding on the way out, one on the way back); with ``allreduce`` RCCL sums in bf16."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import config
from .optim import FlatSGDEMA


def _local_ops(t: torch.Tensor):
    """(cast f32 -> wire, cast wire -> f32 [accumulate], sum of chunks -> f32) for the device the arena lives on: HIP kernels on
    the GPU; plain tensor ops when the reducer is rehearsed on CPU tensors (gloo tests — collectives plumbing only)"""
    if t.is_cuda:
        from . import _lib as L
        from .tape import _p, _stream

        def to_wire(src, dst):
            if dst.dtype == torch.float32:
                L.call("ydl_copy2d", L.YDL_F32, _p(src), src.numel(), _p(dst), src.numel(), 1, src.numel(), 0, _stream())
            else:
                L.call("ydl_cast_f32", L.YDL_BF16, _p(src), src.numel(), _p(dst), src.numel(), 1, src.numel(), 0, _stream())

        def from_wire(src, dst):
            L.call("ydl_cast_to_f32", L.YDL_F32 if src.dtype == torch.float32 else L.YDL_BF16, _p(src), _p(dst), src.numel(), 0, _stream())

        def sum_chunks(src, dst, nchunks):
            L.call("ydl_reduce_chunks", L.YDL_F32 if src.dtype == torch.float32 else L.YDL_BF16, _p(src), _p(dst), dst.numel(), nchunks,
                   _stream())
        return to_wire, from_wire, sum_chunks

    def to_wire(src, dst):
        dst.copy_(src)

    def from_wire(src, dst):
        dst.copy_(src)

    def sum_chunks(src, dst, nchunks):
        acc = torch.zeros_like(dst)
        for r in range(nchunks):                      # fixed rank order, f32 accumulation
            acc += src.view(nchunks, -1)[r].float()
        dst.copy_(acc)
    return to_wire, from_wire, sum_chunks


def pin_rank_to_cores(local_rank: int, local_world: int) -> list:
    """give each rank of a node its own slice of the CPUs this process may run on (os.sched_setaffinity): eight ranks that each
    issue a few hundred launches per step otherwise migrate across cores and NUMA nodes.  Returns the cores chosen ([] when the
    platform has no affinity call or there are fewer cores than ranks: nothing is changed then)."""
    import os
    if not hasattr(os, "sched_getaffinity") or local_world <= 1:
        return []
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return []
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


class GradBucketReducer:
    def __init__(self, opt: FlatSGDEMA, bucket_bytes: int = 16 << 20, group=None, algo: str = "allreduce", wire: str = "f32"):
        if algo not in ("allreduce", "rs_ag") or wire not in ("f32", "bf16"):
            raise ValueError("algo must be 'allreduce' or 'rs_ag', wire 'f32' or 'bf16'")
        self.opt = opt
        self.group = group
        self.algo, self.wire = algo, wire
        self._wdt = torch.float32 if wire == "f32" else torch.bfloat16
        self._stage: Dict[int, dict] = {}
        self._post: List = []
        self._phase1: List = []      # rs_ag: buckets whose reduce-scatter is in flight (handles, post2, fin2)
        self._phase2: List = []      # rs_ag: buckets whose all-gather is in flight (handles, fin2)
        self.overlap_phase2 = True   # False: both phases of every bucket in _drain (the round-3 form; kept as the tests' reference)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = max(bucket_bytes // 4, 1)
        # RCCL's send/recv are ordered on the stream.  The gloo transport (rehearsals of the N>1 path on one GPU or on the CPU) takes
        # the raw pointer of a device tensor and copies from the host side with no stream ordering at all: there the stream has to
        # be drained before a point-to-point batch is posted
        self._host_p2p = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._plan: Optional[List[dict]] = None
        self._slot_of: Dict[int, int] = {id(p): i for i, (p, *_r) in enumerate(opt._slots)}
        self._handles: List = []
        self._pending: Dict[int, int] = {}
        self._live: Optional[List[bool]] = None
        self._launched: List[Tuple[int, int]] = []
        self.enabled = True          # switched off while a HIP graph of the step is being captured
        self.capture_cb = None       # segmented graph capture: called with the bucket index instead of launching the collective
        config.add_grad_hook(self._on_grad)

    def close(self) -> None:
        config.remove_grad_hook(self._on_grad)

    # ------------------------------------------------------------------ planning
    def _build_plan(self, live: List[bool]) -> None:
        """buckets = contiguous arena ranges covering only live parameters, filled from the arena tail (the order in
        which backward completes them)"""
        slots = self.opt._slots
        buckets: List[dict] = []
        cur: Optional[dict] = None
        for i in range(len(slots) - 1, -1, -1):
            if not live[i]:
                cur = None
                continue
            _p_, off, n, _g = slots[i]
            if cur is not None and cur["a"] == off + n and (cur["b"] - off) <= self.bucket_elems:
                cur["a"] = off
                cur["slots"].append(i)
            else:
                cur = {"a": off, "b": off + n, "slots": [i]}
                buckets.append(cur)
        self._plan = buckets
        self._bucket_of = {}
        for bi, b in enumerate(buckets):
            for si in b["slots"]:
                self._bucket_of[si] = bi
        self._live = list(live)

    def begin_step(self) -> None:
        self._handles = []
        self._launched = []
        if self._plan is not None:
            self._pending = {bi: len(b["slots"]) for bi, b in enumerate(self._plan)}

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p) -> None:
        if self.world == 1 or self._plan is None or not self.enabled:
            return
        si = self._slot_of.get(id(p))
        if si is None:
            return
        bi = self._bucket_of.get(si)
        if bi is None:
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            if self.capture_cb is not None:
                self.capture_cb(bi)          # the step is being captured: the graph is cut here, the bucket goes out at replay
            else:
                self._launch(bi)

    def _launch(self, bi: int) -> None:
        b = self._plan[bi]
        self._reduce_range(b["a"], b["b"], key=bi)
        self._launched.append((b["a"], b["b"]))

    def _reduce_range(self, a: int, b: int, key=None) -> None:
        """start the sum over ranks of grads_arena[a:b]; completion + write-back happen in ``_drain``"""
        view = self.opt.grads_arena[a:b]
        if view.is_cuda:        # weight gradients are produced on the side stream (tape.side_stream)
            # (a bucket completed by a weight-gradient batch is launched from inside the side-stream context, which the flush
            #  ordered behind the main stream: the collective then waits for that stream only and the main chain is not held up;
            #  launched from the main stream it waits for the side stream first)
            from .tape import side_stream, stream_wait
            cur, side = torch.cuda.current_stream(), side_stream(view.device)
            if cur.cuda_stream != side.cuda_stream:
                stream_wait(cur, side)
        if self.algo == "allreduce" and self.wire == "f32":
            self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        to_wire, from_wire, sum_chunks = _local_ops(view)
        n = b - a
        W = self.world
        chunk = (n + W - 1) // W
        st = self._stage.get(key) if key is not None else None
        if st is None or st["n"] != n:
            st = {"n": n, "send": torch.zeros(W * chunk, dtype=self._wdt, device=view.device),
                  "recv": torch.zeros(W * chunk, dtype=self._wdt, device=view.device),
                  "own": torch.zeros(chunk, dtype=torch.float32, device=view.device)}
            if key is not None:
                self._stage[key] = st
        to_wire(view, st["send"][:n])
        if self.algo == "allreduce":                       # bf16 wire, RCCL sums in bf16
            h = dist.all_reduce(st["send"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._post.append(([h], lambda st=st, view=view, n=n: from_wire(st["send"][:n], view)))
            return
        # reduce-scatter by hand: piece r of my bucket goes to rank r, pieces from every peer come back — all peers at once
        me = self.rank
        ops = []
        for r in range(W):
            if r == me:
                continue
            ops.append(dist.P2POp(dist.isend, st["send"][r * chunk:(r + 1) * chunk], self._peer(r), self.group))
            ops.append(dist.P2POp(dist.irecv, st["recv"][r * chunk:(r + 1) * chunk], self._peer(r), self.group))
        st["recv"][me * chunk:(me + 1) * chunk].copy_(st["send"][me * chunk:(me + 1) * chunk])
        if self._host_p2p and view.is_cuda:
            torch.cuda.current_stream().synchronize()
        hs = dist.batch_isend_irecv(ops) if ops else []

        def post2(st=st, view=view, chunk=chunk):
            """phase 2 of a bucket, posting half: sum the received pieces, send my reduced piece to every peer (all-gather)"""
            sum_chunks(st["recv"], st["own"], W)                                   # f32 sum in rank order
            to_wire(st["own"], st["send"][me * chunk:(me + 1) * chunk])
            ops2 = []
            for r in range(W):
                if r == me:
                    continue
                ops2.append(dist.P2POp(dist.isend, st["send"][me * chunk:(me + 1) * chunk], self._peer(r), self.group))
                ops2.append(dist.P2POp(dist.irecv, st["send"][r * chunk:(r + 1) * chunk], self._peer(r), self.group))
            if self._host_p2p and view.is_cuda:
                torch.cuda.current_stream().synchronize()
            return dist.batch_isend_irecv(ops2) if ops2 else []

        def fin2(st=st, view=view, n=n):
            from_wire(st["send"][:n], view)                                        # every rank holds the same reduced bucket

        # The all-gather of bucket k is posted when bucket k + 1 is launched (its reduce-scatter has had a bucket's worth of backward
        # to complete), not at the end of backward: phase 2 of every bucket but the last overlaps the remaining backward.  The point
        # is a fixed place in the program, NOT "whenever is_completed() turns true": point-to-point messages between two ranks are
        # matched in posting order, so every rank must post in the same order.  The order that runs (the reduce-scatter of bucket
        # k + 1 was posted a few lines up, BEFORE this call completes bucket k's phase 1 and posts its phase 2):
        #     P1(0)  P1(1) P2(0)  P1(2) P2(1)  ...  P1(n-1) P2(n-2)  | _drain: P2(n-1)
        # — the same on every rank because the bucket plan and the launch points are.  Hardware status: exercised on gloo (world 2,
        # 4, 8) and with two ranks on one GPU only; RCCL has not seen this path with more than one rank, so
        # ``overlap_phase2 = False`` stays selectable from the command line (bench.py / train_seg.py --dp-serial-phase2).
        if self.overlap_phase2:
            self._advance_phase1()
        self._phase1.append((hs, post2, fin2))

    def _advance_phase1(self) -> None:
        """wait for the reduce-scatter of the buckets launched so far and post their all-gathers"""
        for hs, post2, fin2 in self._phase1:
            for h in hs:
                h.wait()              # (RCCL: a stream dependency, not a host wait)
            self._phase2.append((post2(), fin2))
        self._phase1 = []

    def _peer(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _drain(self) -> None:
        for h in self._handles:
            h.wait()
        self._handles = []
        for hs, fn in self._post:
            for h in hs:
                h.wait()
            fn()
        self._post = []
        if self.overlap_phase2:
            self._advance_phase1()                 # the last bucket's all-gather
            for hs2, fin2 in self._phase2:
                for h in hs2:
                    h.wait()
                fin2()
        else:                                      # serial form (reference for the tests): each bucket's phases back to back
            for hs, post2, fin2 in self._phase1:
                for h in hs:
                    h.wait()
                for h in post2():
                    h.wait()
                fin2()
            self._phase1 = []
        self._phase2 = []
        assert not self._phase1 and not self._phase2 and not self._handles and not self._post, "a bucket exchange is still pending"

    # ------------------------------------------------------------------ end of backward
    def finish(self) -> float:
        """Wait for the in-flight buckets and return the grad_scale (1/world) for ``FlatSGDEMA.step``.
        First step (or whenever the set of live parameters changes): ranges not yet reduced are reduced now and
        the bucket plan is rebuilt for the following steps."""
        if self.world == 1:
            return 1.0
        slots = self.opt._slots
        live = [bool(getattr(p, "_ydl_touched", False)) for p, *_r in slots]
        if self._plan is None or live != self._live:
            covered = sorted(self._launched)
            def is_covered(a, b):
                return any(ca <= a and b <= cb for ca, cb in covered)
            # coalesce the uncovered live slots into ranges and reduce them synchronously-launched
            todo = []
            for (p_, off, n, _g), lv in zip(slots, live):
                if lv and not is_covered(off, off + n):
                    if todo and todo[-1][1] == off:
                        todo[-1][1] = off + n
                    else:
                        todo.append([off, off + n])
            for a, b in todo:
                self._reduce_range(a, b)
            self._build_plan(live)
            self._stage = {}
        self._drain()
        self._launched = []
        return 1.0 / self.world


class DataParallel:
    """Thin training-step helper: ``dp = DataParallel(model, opt)``; per step ``dp.begin(); loss.backward();
    scale = dp.finish(); opt.step(grad_scale=scale)``."""

    def __init__(self, model, opt: FlatSGDEMA, bucket_bytes: int = 16 << 20, group=None, broadcast: bool = True,
                 algo: str = "allreduce", wire: str = "f32"):
        self.model, self.opt = model, opt
        self.reducer = GradBucketReducer(opt, bucket_bytes, group, algo=algo, wire=wire)
        if self.reducer.world > 1:
            # a bucket goes out when its last gradient launch has been ENQUEUED: weight gradients batched eight layers deep before
            # a fork of the side stream would hold every bucket back to the end of backward (the batching exists for the eager
            # path's host time only)
            config.set_wgrad_batch(1)
        if broadcast and dist.is_initialized() and self.reducer.world > 1:
            dist.broadcast(opt.params_arena, src=0, group=group)     # identical replicas (params + BN buffers)
            if opt.ema_arena is not None:
                opt.ema_arena.copy_(opt.params_arena)
            config.bump_weight_epoch()

    def begin(self) -> None:
        self.reducer.begin_step()

    def finish(self) -> float:
        return self.reducer.finish()

    def reduce_now(self) -> float:
        """all-reduce every live gradient range right now (used after a captured forward/backward graph, where the
        per-bucket launch hooks did not fire); returns the grad_scale for the optimizer"""
        r = self.reducer
        if r.world == 1:
            return 1.0
        r.begin_step()
        live = [bool(getattr(p, "_ydl_touched", False)) for p, *_x in r.opt._slots]
        if r._plan is None or live != r._live:
            r._build_plan(live)
        for bi in range(len(r._plan)):
            r._launch(bi)
        r._drain()
        r._launched = []
        return 1.0 / r.world
