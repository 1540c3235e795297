#!/usr/bin/env python3
"""dev tool: the 2-rank HIP-graph rs_ag/bf16 data-parallel case of tests/test_gpu_dp.py with per-step NaN diagnostics"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp, yaml


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import yolo_dual_amd as ydl
    from oracle.fill import fill_state_dict
    from yolo_dual_amd.graph import GraphedTrainStep
    from yolo_dual_amd.parallel import DataParallel
    torch.cuda.set_device(0)
    ydl.set_compute_dtype("bf16")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            if l[2] == "C3_DCN":
                l[2] = "C3"
    m = ydl.YOLOv5Seg(cfg); m.img_size = [64, 64]
    sd = m.state_dict(); fill_state_dict(sd, 11 + rank, bn_stats=False); m.load_state_dict(sd)
    m = m.cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4, ema=(rank == 0))
    dp = DataParallel(m, opt, bucket_bytes=1 << 20, algo=sys.argv[1] if len(sys.argv) > 1 else "rs_ag", wire=sys.argv[2] if len(sys.argv) > 2 else "bf16")
    cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
    crit = ydl.SegmentationLoss(12, 0.0, cw, "dice", sync=False)
    gen = torch.Generator("cuda").manual_seed(100 + rank)
    x = torch.rand(2, 3, 64, 64, device="cuda", generator=gen)
    t = torch.randint(0, 12, (2, 64, 64), device="cuda", generator=gen)
    g = GraphedTrainStep(m, crit, opt, x, t, dp=dp, warmup=2)
    names = {id(p): k for k, p in m.named_parameters()}

    def report(tag):
        torch.cuda.synchronize()
        badg = [names[id(p)] for p, off, n, _g in opt._slots if not torch.isfinite(opt.grads_arena[off:off + n]).all()]
        badp = [names[id(p)] for p, off, n, _g in opt._slots if not torch.isfinite(opt.params_arena[off:off + n]).all()]
        badb = [k for k, v in m.state_dict().items() if "running" in k and not torch.isfinite(v).all()]
        print(f"[rank {rank}] {tag}: segments {len(g._segments)} nan grads {len(badg)} {badg[:4]} | nan params {len(badp)} {badp[:4]} | nan running {len(badb)} {badb[:3]}", flush=True)
    report("after capture")
    for i in range(3):
        items = g.step()
        report(f"after graph step {i} loss {[float(v) for v in items]}")
    dist.destroy_process_group()
    q.put(rank)


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]; [p.join(240) for p in ps]
