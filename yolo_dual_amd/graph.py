"""HIP-graph capture of the training step ("HIP streams and graphs instead of a tracing compiler").

The eager path issues ~700 kernel launches per step through Python + ctypes; on a slow or shared host that costs more
than the GPU work.  ``GraphedTrainStep`` captures
    G1 = zero_grad + forward + loss + backward        (both HIP streams: the fork/join of the side stream is captured)
    G2 = fused SGD-nesterov + EMA step                (hyper-parameters read from a device vector)
with torch.cuda.CUDAGraph (plumbing: capture/replay API and the graph-private memory pool) and replays them; between the
two graphs the data-parallel all-reduce of the gradient arena runs eagerly.  With more than one rank G1 is captured in SEGMENTS:
the capture is cut wherever the eager path would launch a gradient bucket (the reducer's launch hook fires during capture), so a
replay alternates "graph segment, bucket collective" and the collectives overlap with the remaining backward segments exactly
as in eager mode (single compute stream inside the segments; backward runs on the capturing thread for this).  Host-side bookkeeping that the captured
kernels cannot do (BatchNorm ``num_batches_tracked``, EMA update counter, learning-rate schedule) is advanced per replay.
Inputs are static device tensors: copy each new batch into ``imgs`` / ``targets`` before ``step()``."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import config
from .modules import _BNHolder
from .optim import FlatSGDEMA


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer: FlatSGDEMA, imgs: torch.Tensor, targets: torch.Tensor,
                 dp=None, warmup: int = 3, segmented: Optional[bool] = None):
        self.model, self.criterion, self.opt, self.dp = model, criterion, optimizer, dp
        if segmented is None:
            segmented = dp is not None and dp.reducer.world > 1
        self.segmented = bool(segmented) and dp is not None and dp.reducer.world > 1
        self._segments: List[torch.cuda.CUDAGraph] = []
        self._bucket_after: List[int] = []
        self.imgs, self.targets = imgs, targets
        self._bns: List[_BNHolder] = [m for m in model.modules() if isinstance(m, _BNHolder)]
        if getattr(criterion, "sync", False):
            raise ValueError("graph capture needs a loss that does not sync: SegmentationLoss(..., sync=False)")
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):                       # warm-up off the default stream (torch's capture recipe)
            for _ in range(warmup):
                self._fwd_bwd()
                scale = dp.finish() if dp else 1.0
                optimizer.prepare_step(scale)
                optimizer.step_device_hyper()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        nbt0 = [bn._nbt_pending for bn in self._bns]
        self.g1 = torch.cuda.CUDAGraph()
        if self.segmented and dp.reducer._plan is not None:
            self._capture_segments()
        else:
            self.segmented = False
            if dp is not None:
                dp.reducer.enabled = False               # no collective launches from the gradient hooks while capturing
            try:
                with torch.cuda.graph(self.g1):
                    self._fwd_bwd(capturing=True)
            finally:
                if dp is not None:
                    dp.reducer.enabled = True
        self._nbt_per_replay = [bn._nbt_pending - a for bn, a in zip(self._bns, nbt0)]
        self.g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g2, pool=(self._segments[0] if self.segmented else self.g1).pool()):
            optimizer.step_device_hyper()
        # the captured pass did not execute: undo its host-side counters, and make eager code re-derive the compute
        # weights (the capture marked them fresh without running the re-layout kernel)
        for bn, a in zip(self._bns, nbt0):
            bn._nbt_pending = a
        config.bump_weight_epoch()

    def _capture_segments(self) -> None:
        """G1 as a chain of graphs cut at the bucket-launch points of the reducer's plan (all in one private pool, replayed in
        capture order).  Everything runs on ONE stream and one thread: the side stream is switched off and autograd's device
        thread is bypassed, because a stream capture has to be ended by the thread that began it."""
        import gc
        red = self.dp.reducer
        overlap_was = config.overlap_wgrad()
        config.set_overlap_wgrad(False)
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        state = {"g": None}

        def cut(bi: int) -> None:
            state["g"].capture_end()
            self._segments.append(state["g"])
            self._bucket_after.append(bi)
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=self._segments[0].pool())
            state["g"] = g

        try:
            with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
                g = torch.cuda.CUDAGraph()
                g.capture_begin()
                state["g"] = g
                red.begin_step()
                red.capture_cb = cut
                try:
                    self._fwd_bwd(capturing=True)
                finally:
                    red.capture_cb = None
                    state["g"].capture_end()
                    self._segments.append(state["g"])
        finally:
            config.set_overlap_wgrad(overlap_was)
        torch.cuda.current_stream().wait_stream(stream)

    def _fwd_bwd(self, capturing: bool = False):
        self.opt.zero_grad()
        if self.dp and not capturing:
            self.dp.begin()
        out = self.model(self.imgs)
        loss, items = self.criterion(out, self.targets)
        loss.backward()
        self.loss_items = items
        return items

    def step(self):
        """one training step; returns the (device-resident) [total, ce, overlap] loss scalars"""
        scale = 1.0
        if self.segmented:
            red = self.dp.reducer
            red.begin_step()
            for k, g in enumerate(self._segments):
                g.replay()
                if k < len(self._bucket_after):
                    red._launch(self._bucket_after[k])     # overlaps with the remaining segments
            for bn, k in zip(self._bns, self._nbt_per_replay):
                bn._nbt_pending += k
            scale = self.dp.finish()
        else:
            self.g1.replay()
            for bn, k in zip(self._bns, self._nbt_per_replay):
                bn._nbt_pending += k
            if self.dp is not None:
                scale = self.dp.reduce_now()   # gradients were produced inside the graph: no per-bucket hooks fired
        self.opt.prepare_step(scale)
        self.g2.replay()
        config.bump_weight_epoch()
        return self.loss_items
