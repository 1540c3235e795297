"""CPU, world_size 2, gloo: the data-parallel bucket reducer (yolo_dual_amd.parallel) — plan building from the set of
live parameters, hook-triggered bucket launches in backward order, exclusion of dead parameters, liveness changes
between steps, and the broadcast of the flat parameter arena.  Gradients are produced by writing into the arena and
calling the same ``config.mark_touched`` hook the HIP wgrad / BN-backward launches call."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import yolo_dual_amd as ydl
    torch.manual_seed(0)
    return nn.Sequential(ydl.Conv(8, 16, 3, 1), ydl.C3(16, 16, 1), ydl.Conv(16, 8, 1, 1), ydl.Conv(8, 8, 1, 1))


def _worker(rank, world, port, q, algo="allreduce", wire="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolo_dual_amd import config
        from yolo_dual_amd.optim import FlatSGDEMA
        from yolo_dual_amd.parallel import DataParallel
        net = _build()
        if rank == 1:                      # replicas start different: the broadcast must equalise them
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        opt = FlatSGDEMA(net, lr=0.1)
        dp = DataParallel(net, opt, bucket_bytes=4096, algo=algo, wire=wire)      # tiny buckets -> several of them
        ref0 = [torch.zeros_like(opt.params_arena) for _ in range(world)]
        dist.all_gather(ref0, opt.params_arena)
        assert torch.equal(ref0[0], ref0[1]), "broadcast did not equalise the replicas"

        slots = opt._slots
        dead = {id(p) for p in net[3].parameters()}           # last block never gets a gradient ("dead head layer")
        expect_scale = 1.0 / world
        for step in range(3):
            opt.zero_grad()
            dp.begin()
            if step == 2:
                dead = set()                                   # liveness changes: everything is live now
            # backward order = reverse of module order; each rank writes rank-dependent gradients
            for p, off, n, _g in reversed(slots):
                if id(p) in dead:
                    continue
                opt.grads_arena[off:off + n] = (rank + 1) * (1.0 + 0.001 * torch.arange(n, dtype=torch.float32) + step)
                config.mark_touched(p)
            scale = dp.finish()
            assert abs(scale - expect_scale) < 1e-12
            for p, off, n, _g in slots:
                got = opt.grads_arena[off:off + n]
                if id(p) in dead:
                    assert float(got.abs().max()) == 0.0       # dead ranges are neither written nor reduced
                else:
                    want = sum(r + 1 for r in range(world)) * (1.0 + 0.001 * torch.arange(n, dtype=torch.float32) + step)
                    # bf16 wire: one rounding on the way out and one on the way back (rs_ag sums in f32 in between; RCCL-style
                    # all_reduce sums in bf16): element-wise within 2e-2, and as a GRADIENT (relative L2 against the exact f32 sum)
                    # within 6e-3 — two to log2(world)+1 roundings of 2^-9 each
                    assert torch.allclose(got, want, rtol=1e-6 if wire == "f32" else 2e-2), (step, off)
                    rel = float((got - want).norm() / want.norm())
                    assert rel < (1e-6 if wire == "f32" else 6e-3), (step, off, rel)
            if step == 0:
                plan = dp.reducer._plan
                assert plan is not None and len(plan) >= 2
                covered = sorted((b["a"], b["b"]) for b in plan)
                live_elems = sum(n for p, off, n, _g in slots if id(p) not in dead)
                assert sum(b - a for a, b in covered) == live_elems
            if step == 1:
                # second step: every bucket was launched from the hooks before finish()
                assert all(v == 0 for v in dp.reducer._pending.values())
            # every rank ends with bit-identical gradients (the replicas must not drift apart)
            allg = [torch.zeros_like(opt.grads_arena) for _ in range(world)]
            dist.all_gather(allg, opt.grads_arena)
            assert all(torch.equal(allg[0], t) for t in allg[1:]), "ranks disagree on the reduced gradients"
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(world, algo="allreduce", wire="f32"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, algo, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_bucket_reducer_two_ranks():
    _run(2)


def test_bucket_reducer_four_ranks():
    _run(4)


def test_bucket_reducer_eight_ranks():
    """the rank count of one MI355X node (8 processes on the CPU here: the collective plumbing, not the links)"""
    _run(8)


def test_rank_pinning_splits_the_cpu_set():
    import os
    from yolo_dual_amd.parallel import pin_rank_to_cores
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before = sorted(os.sched_getaffinity(0))
    try:
        if len(before) >= 2:
            a = pin_rank_to_cores(0, 2)
            assert a == before[:len(before) // 2] and sorted(os.sched_getaffinity(0)) == a
            os.sched_setaffinity(0, before)
            b = pin_rank_to_cores(1, 2)
            assert b == before[len(before) // 2:2 * (len(before) // 2)] and not set(a) & set(b)
        os.sched_setaffinity(0, before)
        assert pin_rank_to_cores(0, 1) == [] and pin_rank_to_cores(0, 10 * len(before)) == []
        assert sorted(os.sched_getaffinity(0)) == before
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.parametrize("world,wire", [(2, "f32"), (4, "f32"), (4, "bf16"), (8, "bf16")])
def test_reduce_scatter_all_gather_over_all_peers(world, wire):
    """the hand-rolled comparator of RCCL's all-reduce: every rank trades 1/world of each bucket with every peer at
    once, sums in f32 in rank order, and returns its reduced piece to every peer"""
    _run(world, algo="rs_ag", wire=wire)


def test_bf16_wire_all_reduce():
    _run(2, algo="allreduce", wire="bf16")


def _overlap_worker(rank, world, port, q, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from yolo_dual_amd import config
        from yolo_dual_amd.optim import FlatSGDEMA
        from yolo_dual_amd.parallel import DataParallel
        net = _build()
        opt = FlatSGDEMA(net, lr=0.1)
        dp = DataParallel(net, opt, bucket_bytes=4096, algo="rs_ag", wire=wire)
        gen = torch.Generator().manual_seed(100 + rank)
        results = {}
        for overlap in (True, False):
            dp.reducer.overlap_phase2 = overlap
            gen.manual_seed(100 + rank)
            outs = []
            for step in range(3):
                opt.zero_grad()
                dp.begin()
                seen2 = 0
                for p, off, n, _g in reversed(opt._slots):
                    opt.grads_arena[off:off + n] = torch.randn(n, generator=gen)
                    config.mark_touched(p)
                    seen2 = max(seen2, len(dp.reducer._phase2))
                if step >= 1:          # the plan exists from the first finish() on: buckets go out from the hooks
                    nb = len(dp.reducer._plan)
                    assert nb >= 3
                    if overlap:        # all-gathers of all buckets but the last were posted DURING backward
                        assert seen2 == nb - 1 and len(dp.reducer._phase1) == 1, (seen2, nb)
                    else:
                        assert seen2 == 0 and len(dp.reducer._phase1) == nb
                dp.finish()
                outs.append(opt.grads_arena.clone())
            results[overlap] = outs
        for a, b in zip(results[True], results[False]):
            assert torch.equal(a, b), "overlapped all-gather changed the reduced gradients"
        allg = [torch.zeros_like(opt.grads_arena) for _ in range(world)]
        dist.all_gather(allg, results[True][-1])
        assert all(torch.equal(allg[0], t) for t in allg[1:])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["f32", "bf16"])
def test_overlapped_all_gather_equals_the_serial_form(wire):
    """rs_ag posts the all-gather of bucket k when bucket k + 1 is launched (during backward), in the same order on every rank; the
    reduced gradients equal, bit for bit, those of the serial form that runs both phases of every bucket after backward (world 4)"""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_flat_arena_layout_and_state_dict_roundtrip():
    """parameters become views of one arena; state_dict keys/shapes are untouched and load_state_dict writes through"""
    from yolo_dual_amd.optim import FlatSGDEMA
    net = _build()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    opt = FlatSGDEMA(net, lr=0.1)
    a0 = opt.params_arena.data_ptr()
    for p, off, n, g in opt._slots:
        assert p.data_ptr() == a0 + 4 * off and p.grad.data_ptr() == opt.grads_arena.data_ptr() + 4 * off
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    sd1 = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in sd0.items()}
    net.load_state_dict(sd1)
    w = net[0].conv.weight
    assert torch.equal(w.detach(), sd1["0.conv.weight"])
    # KRSC physical layout survives the load (copy_ preserves the destination strides)
    assert w.permute(0, 2, 3, 1).is_contiguous()
    flat = opt.params_arena[opt._slots[0][1]:opt._slots[0][1] + opt._slots[0][2]]
    assert torch.equal(flat.view(16, 3, 3, 8), sd1["0.conv.weight"].permute(0, 2, 3, 1))
    ema = opt.ema_state_dict()
    assert set(ema) == set(sd0)
