"""GPU, 2 ranks sharing cuda:0 over gloo (RCCL needs one device per rank; the box has one): the data-parallel training
step with the real kernels — bucketed all-reduce from the wgrad/BN-backward hooks (eager) and the post-graph reduction
(HIP-graph mode) keep the replicas identical, and the averaged gradient is what the optimizer applies."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

pytestmark = pytest.mark.gpu
CFG = os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, use_graph, q, algo="allreduce", wire="f32", use_replay=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import yolo_dual_amd as ydl
        from oracle.fill import fill_state_dict
        from yolo_dual_amd.graph import GraphedTrainStep
        from yolo_dual_amd.parallel import DataParallel
        torch.cuda.set_device(0)
        ydl.set_compute_dtype("bf16")
        cfg = yaml.safe_load(open(os.path.join(CFG, "yolov5_seg.yaml")))
        for sec in ("backbone", "head"):
            for l in cfg[sec]:
                if l[2] == "C3_DCN":
                    l[2] = "C3"
        m = ydl.YOLOv5Seg(cfg)
        m.img_size = [64, 64]
        sd = m.state_dict()
        fill_state_dict(sd, 11 + rank, bn_stats=False)          # replicas start different: broadcast must fix it
        m.load_state_dict(sd)
        m = m.cuda().train()
        opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4, ema=(rank == 0))
        dp = DataParallel(m, opt, bucket_bytes=1 << 20, algo=algo, wire=wire)
        cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
        crit = ydl.SegmentationLoss(12, 0.0, cw, "dice", sync=False)
        gen = torch.Generator("cuda").manual_seed(100 + rank)      # different data per rank
        x = torch.rand(2, 3, 64, 64, device="cuda", generator=gen)
        t = torch.randint(0, 12, (2, 64, 64), device="cuda", generator=gen)

        def eager_step():
            opt.zero_grad()
            dp.begin()
            total, items = crit(m(x), t)
            total.backward()
            opt.step(grad_scale=dp.finish())
            return items

        if use_replay:
            from yolo_dual_amd.replay import ReplayedTrainStep
            r = ReplayedTrainStep(m, crit, opt, x, t, dp=dp, warmup=2)
            # the list is cut where the eager step launches a gradient bucket: collectives overlap the remaining segments
            assert r.multi and len(r._cuts) >= 2 and len(r.rec.handles) == 2, (len(r._cuts), r.rec.handles)
            for _ in range(3):
                items = r.step()
        elif use_graph:
            g = GraphedTrainStep(m, crit, opt, x, t, dp=dp, warmup=2)
            # more than one rank: the forward/backward graph is cut at every bucket-launch point, the collectives of a replay
            # overlap with the remaining segments
            assert g.segmented and len(g._segments) == len(g._bucket_after) + 1 and len(g._segments) >= 3, len(g._segments)
            for _ in range(2):
                items = g.step()
        else:
            for _ in range(3):
                items = eager_step()
        torch.cuda.synchronize()
        # replicas identical (parameters; BN buffers are per-rank by design, like nn.DataParallel replicas)
        n = opt.n_params
        mine = opt.params_arena[:n].clone()
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        same = bool(torch.equal(mine, ref))
        if not same:
            bad = []
            for p_, off, n_, g_ in opt._slots:
                d = float((mine[off:off + n_] - ref[off:off + n_]).abs().max())
                if d > 0:
                    bad.append((off, n_, g_, d))
            print("DIVERGED slots:", len(bad), "of", len(opt._slots), bad[:6], bad[-3:], flush=True)
        finite = bool(torch.isfinite(mine).all()) and bool(torch.isfinite(torch.stack([i.float() for i in items])).all())
        q.put((rank, "ok" if (same and finite) else f"fail same={same} finite={finite}"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,algo,wire", [(False, "allreduce", "f32"), (True, "allreduce", "f32"), (False, "rs_ag", "f32"),
                                                 (False, "rs_ag", "bf16"), (True, "rs_ag", "bf16")])
def test_two_rank_data_parallel_on_one_gpu(use_graph, algo, wire):
    """two ranks on one MI355X (gloo transport): RCCL-style all-reduce, and the hand-rolled reduce-scatter + all-gather over all
    peers with f32 / bf16 wire format (the HIP cast / chunk-sum kernels run here), eager and replayed from HIP graphs"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, q, algo, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("algo,wire", [("allreduce", "f32"), ("rs_ag", "bf16")])
def test_two_rank_data_parallel_launch_list(algo, wire):
    """the launch-list replay under data parallelism: two HIP streams per rank, the recorded list cut at the bucket launches, the
    replicas stay identical"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, False, q, algo, wire, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


_CLI_ARGS = ["--batch-size", "2", "--imgsz", "64", "--steps-per-epoch", "5", "--epochs", "2", "--dtype", "f32"]


def _cli_worker(rank, world, port, save_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    try:
        import train_seg
        fit = train_seg.train(train_seg.parse_opt(_CLI_ARGS + ["--save-dir", save_dir, "--dist-backend", "gloo", "--one-gpu"]))
        import yolo_dual_amd as ydl
        ok = 0.0 <= fit <= 1.0
        if rank == 0:
            ck = ydl.load_checkpoint(os.path.join(save_dir, "last.pt"))
            ok = ok and ck["epoch"] == 1 and ck["optimizer"] is not None
        q.put((rank, "ok" if ok else "fail", dict(train_seg.LAST_RUN)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + traceback.format_exc(), {}))


def _emu_worker(save_dir, q):
    """one process playing both ranks in turn (train_seg.py --emulate-world 2): same micro-batches, gradients summed, 1/2 in the step"""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        os.environ.pop(k, None)
    try:
        import train_seg
        train_seg.train(train_seg.parse_opt(_CLI_ARGS + ["--save-dir", save_dir, "--emulate-world", "2"]))
        q.put((0, "ok", dict(train_seg.LAST_RUN)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((0, "fail: " + repr(e) + traceback.format_exc(), {}))


def test_train_cli_runs_data_parallel(tmp_path):
    """train_seg.py under two ranks (what torch.distributed.run sets up): the nominal-batch scaling uses the TOTAL batch as the
    reference does (seg_diceloss_yolov5.py:970-972, :1001: 2 ranks x bs 2 -> accumulate 16, weight decay x 4 x 16 / 64), one exchange
    per optimizer step, rank 0 validates and saves; the ranks end with identical parameters, and those equal — to f32 rounding — the
    parameters of ONE process that plays both ranks' micro-batches in turn (sum of the gradients, 1/2 in the step)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(tmp_path / "dp"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    facts = {r[0]: r[2] for r in res}
    for f in facts.values():
        assert f["world"] == 2 and f["total_batch"] == 4 and f["accumulate"] == 16, f
        assert abs(f["weight_decay"] - 0.0005 * 4 * 16 / 64) < 1e-12, f
    assert facts[0]["param_sum"] == facts[1]["param_sum"] and facts[0]["param_abs_sum"] == facts[1]["param_abs_sum"], facts
    pe = ctx.Process(target=_emu_worker, args=(str(tmp_path / "emu"), q))
    pe.start()
    re_ = q.get(timeout=280)
    pe.join(timeout=60)
    assert re_[1] == "ok", re_
    emu = re_[2]
    assert emu["total_batch"] == 4 and emu["accumulate"] == 16, emu
    # same products, different summation order (ranks' sums meet in the all-reduce instead of one arena): f32 rounding over two epochs
    assert abs(emu["param_abs_sum"] - facts[0]["param_abs_sum"]) < 1e-5 * facts[0]["param_abs_sum"], (emu, facts[0])
    assert abs(emu["param_sum"] - facts[0]["param_sum"]) < 1e-5 * facts[0]["param_abs_sum"], (emu, facts[0])
