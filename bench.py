#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec of one training step (forward + loss + backward + SGD/EMA step) of the
YOLOv5-backbone segmentation model (BASELINE config 2: C3_DCN -> C3, 640x640, bs=16 per GPU, Dice loss, 12 classes)
on synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = whole-job images/sec with inputs resident in HBM.  `roofline` is the
dominant kernel family measured with events on the launch stream in an instrumented pass after the timed region;
`cpu_baseline` is the CPU oracle (oracle/ref_cpu.py, kind "port") timed on this host on a bounded sample."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_IMG = 129.3e9          # cfg2: algorithmic FLOP / image, fwd+bwd, live graph, convs only
PEAK_BF16 = 2500.0       # TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32 = 157.3         # TFLOP/s f32 MFMA
PEAK_HBM = 8000.0        # GB/s


CW = [1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1]          # unet-lite/yolo5-seg/weight.yaml:3-14

# BASELINE.json configs[1..4] (configs[0] is the CPU plumbing case).  flop_img: conv FLOP per image of one training step
# (forward + dgrad + wgrad; forward figures from BASELINE.md §2, the stem's input gradient is never needed), None = take the
# executed conv FLOPs of the instrumented pass.
WORKLOADS = {
    "cfg2": dict(model="YOLOv5Seg", yaml="yolov5_seg.yaml", swap={"C3_DCN": "C3"}, loss="dice", cw=True, bs=16, size=640, flop_img=129.3e9,
                 desc="YOLOv5-backbone (C3+SPPF) + UNet-lite SegmentHead", base="BASELINE configs[1]"),
    "cfg3": dict(model="ResNet50Seg", loss="dice", cw=False, bs=32, size=640, flop_img=3 * 101.91e9 - 1.93e9,
                 desc="ResNet50 + multi-scale SegmentHead", base="BASELINE configs[2]"),
    "cfg4": dict(model="YOLOv8Seg", yaml="yolov8_seg.yaml", swap={"C2f_DCN": "C2f"}, loss="jaccard", cw=True, bs=8, size=1024,
                 flop_img=3 * 131.47e9 - 3.62e9, desc="YOLOv8 backbone (C2f) + UNet-lite SegmentHead", base="BASELINE configs[3]"),
    "cfg5": dict(model="YOLOv9Seg", yaml="yolov9_seg.yaml", swap={}, loss="dice", cw=True, bs=16, size=640, flop_img=3 * 89.69e9 - 0.35e9,
                 desc="YOLOv9 backbone (C3k2 + GAM + SPPF) + UNet-lite SegmentHead (the yaml as shipped: no DCN block)",
                 base="BASELINE configs[4], reference yaml"),
    "cfg5dcn": dict(model="YOLOv9Seg", yaml="yolov9_dcnv3_seg.yaml", swap={}, loss="dice", cw=True, bs=16, size=640, flop_img=None,
                    desc="YOLOv9 backbone with C3_DCNV3 (models/ops_dcnv3) + UNet-lite SegmentHead",
                    base="BASELINE configs[4], C3-DCN wired through the reference's C3_DCNV3 note"),
}


def load_cfg(name="yolov5_seg.yaml", swap=None):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", name)))
    swap = {"C3_DCN": "C3"} if swap is None else swap
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = swap.get(l[2], l[2])
    return cfg


def cpu_baseline(wl: dict, bs: int, size: int, steps: int):
    """the CPU oracle's training step (fwd + CE+0.5*Dice|Jaccard + bwd + SGD-nesterov) on all host cores"""
    import torch
    from oracle import ref_cpu as R
    from oracle.fill import fill_state_dict
    from tests.model_shapes import resnet_seg_state_shapes, script_model_state_shapes
    if wl["model"] == "ResNet50Seg":
        cfg, shapes = None, resnet_seg_state_shapes("bottleneck", 12)
    else:
        cfg = load_cfg(wl["yaml"], wl["swap"])
        shapes = script_model_state_shapes(cfg)
    fam = {"YOLOv5Seg": "v5", "YOLOv8Seg": "v8", "YOLOv9Seg": "v9"}.get(wl["model"], "v5")
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64))
          for k, s in shapes.items()}
    fill_state_dict(sd, 1, bn_stats=False)
    pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]
    g = torch.Generator().manual_seed(0)
    x = torch.rand(bs, 3, size, size, generator=g)
    t_hw = 640 if (cfg is None or size == 1024) else size      # hard-coded 640x640 outputs (segment/train.py:209; T7)
    t = torch.randint(0, 12, (bs, t_hw, t_hw), generator=g)
    cw = torch.tensor(CW, dtype=torch.float32) if wl["cw"] else None
    bufs = {}
    times = []
    for st in range(steps + 1):
        t0 = time.perf_counter()
        ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
        run = dict(sd)
        run.update(ps)
        if cfg is None:
            out = R.resnet_seg_forward(run, x, "bottleneck", out_size=(640, 640))
        else:
            out = R.script_model_forward(run, cfg, x, (t_hw, t_hw), family=fam)
        total, _, _ = R.seg_loss(out, t, cw, wl["loss"])
        total.backward()
        for k in pnames:
            if ps[k].grad is not None:
                wd = 5e-4 if k.endswith("conv.weight") else 0.0
                bufs[k] = R.sgd_nesterov_step(sd[k], ps[k].grad, bufs.get(k), 0.01, 0.937, wd)
        for k in sd:
            if k not in ps:
                sd[k] = run[k]
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / max(len(times) - 1, 1)
    return {"value": bs / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (1 warm-up) of the same model/loss/optimizer at bs={bs}, {size}x{size}, fp32"}


def spawn_ranks(n: int) -> int:
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print(f"[bench] {n}-rank launch failed: exit code {proc.returncode}, {len(lines)} result lines", file=sys.stderr)
        return proc.returncode or 1
    print(lines[0])
    return 0


def parity_leg(ydl, args, wl, dev, imgs, tgts, out_hw):
    """A bounded leg of the SAME workload in parity mode (f32 storage and arithmetic, deterministic weight gradients): the mode every
    1e-4 claim against the reference is made in.  Fresh model and optimizer, eager launches (26 ms of GPU time per step hide the host),
    timed like the main region."""
    import torch
    ydl.set_compute_dtype("f32")
    try:
        torch.manual_seed(0)
        if wl["model"] == "ResNet50Seg":
            model = ydl.ResNet50Seg({"nc": 12}).to(dev).train()
        else:
            model = getattr(ydl, wl["model"])(load_cfg(wl["yaml"], wl["swap"])).to(dev).train()
            model.img_size = [out_hw, out_hw]
        cw = torch.tensor(CW, dtype=torch.float32) if wl["cw"] else None
        crit = ydl.SegmentationLoss(12, 0.0, cw, wl["loss"], sync=False)
        opt = ydl.FlatSGDEMA(model, lr=0.01, momentum=0.937, weight_decay=5e-4 * args.bs / 64.0)

        def step():
            opt.zero_grad()
            out = model(imgs)
            loss, items = crit(out, tgts)
            loss.backward()
            opt.step(grad_scale=1.0)
            return items

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.parity_steps):
            items = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ips = args.bs * args.parity_steps / dt
        flop_img = wl["flop_img"] if (wl["flop_img"] and args.size == wl["size"]) else None
        return {"dtype": "f32", "value": ips, "unit": "images/sec", "steps": args.parity_steps, "ms_per_step": dt / args.parity_steps * 1e3,
                "launch_mode": "eager", "loss": float(items[0]),
                "end_to_end_mfma_frac": (ips * flop_img / 1e12 / PEAK_F32) if flop_img else None,
                "note": "same workload, f32 storage + exact-f32 MFMA, deterministic weight gradients; peak = 157.3 TFLOP/s f32 matrix"}
    finally:
        ydl.set_compute_dtype(args.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="cfg2 (default) = BASELINE.json's metric config; cfg3/cfg4/cfg5/cfg5dcn = the other BASELINE configs")
    ap.add_argument("--bs", type=int, default=0, help="images per GPU (default: the workload's)")
    ap.add_argument("--size", type=int, default=0, help="input size (default: the workload's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dcn-offset-std", type=float, default=0.0,
                    help="DCNv3 workloads: re-draw the offset projections so that the sampling offsets have about this standard deviation "
                         "in pixels (the module's own initialisation is zero offsets — every point on the grid, the cheapest case)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="keep wgrad on the main stream")
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python (no launch-list replay, no HIP graph)")
    ap.add_argument("--prioritize", type=int, default=0, help="replay the main chain on a high-priority stream, side work on low-priority ones")
    ap.add_argument("--try-hipgraph", action="store_true", help="also time the single-stream HIP-graph capture of the step")
    ap.add_argument("--profile-json", default="", help="dump the per-launch event records of the instrumented pass")
    ap.add_argument("--graph-overlap", action="store_true", help="capture the side stream (wgrad / dead branch) into the graphs too")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--dp-algo", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="gradient exchange: RCCL all-reduce (default) or the hand-rolled reduce-scatter + all-gather over all peers")
    ap.add_argument("--dp-wire", default="f32", choices=["f32", "bf16"], help="wire format of the gradients")
    ap.add_argument("--dp-serial-phase2", action="store_true",
                    help="rs_ag: run each bucket's all-gather at the end of backward instead of overlapping it (the round-3 form)")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the bounded f32 (parity-mode) leg after the timed region")
    ap.add_argument("--parity-steps", type=int, default=5)
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.bs = args.bs or wl["bs"]
    args.size = args.size or wl["size"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly: start the N ranks ourselves (one process per GPU under torch.distributed.run, the way the reference starts
        # its DDP runs, utils/torch_utils.py:55-63) BEFORE anything here touches the GPU, pass rank 0's JSON line through, and fail
        # if any worker fails.  The child is a fresh interpreter, never an exec of this one.
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd.parallel import DataParallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        from yolo_dual_amd.parallel import pin_rank_to_cores
        pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))      # each rank on its own slice of the host's cores
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(0 if args.one_gpu else local)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    ydl.set_compute_dtype(args.dtype)
    if args.no_overlap:
        ydl.config.set_overlap_wgrad(False)

    torch.manual_seed(0)
    if wl["model"] == "ResNet50Seg":
        model = ydl.ResNet50Seg({"nc": 12}).to(dev).train()
        out_hw = 640                               # segment/train.py:209: the head's output is hard-coded to 640x640
    else:
        model = getattr(ydl, wl["model"])(load_cfg(wl["yaml"], wl["swap"])).to(dev).train()
        # the yaml models resize to img_size (hard-coded 640 in the reference, T7); the benchmark keeps 640 for the 1024 input of
        # cfg4 like the reference does and follows --size otherwise
        out_hw = 640 if args.size == 1024 else args.size
        model.img_size = [out_hw, out_hw]
    if args.dcn_offset_std > 0:
        gen_o = torch.Generator(device="cpu").manual_seed(77)
        with torch.no_grad():
            for mod in model.modules():
                off = getattr(mod, "offset", None)
                if off is not None and hasattr(mod, "mask") and hasattr(off, "weight") and off.weight.dim() == 2:
                    # offset = W x + b over ~unit-variance features: per-element std of W x is about sqrt(in_features) * std(W) * rms(x)
                    sw = args.dcn_offset_std / (off.weight.shape[1] ** 0.5) / 0.7
                    off.weight.copy_((torch.randn(off.weight.shape, generator=gen_o) * sw).to(off.weight.device))
                    off.bias.copy_((torch.randn(off.bias.shape, generator=gen_o) * 0.5 * args.dcn_offset_std).to(off.bias.device))
    cw = torch.tensor(CW, dtype=torch.float32) if wl["cw"] else None
    crit = ydl.SegmentationLoss(12, 0.0, cw, wl["loss"], sync=False)
    # reference hyper-parameters (seg_diceloss_yolov5.py:970-972): lr0 0.01, momentum 0.937, weight_decay 5e-4 * bs*accumulate/64.
    # The throughput run steps the optimizer every batch (accumulate = 1; the reference would accumulate
    # round(64/16) = 4 batches at bs 16), which is what the formula is evaluated with here.
    opt = ydl.FlatSGDEMA(model, lr=0.01, momentum=0.937, weight_decay=5e-4 * args.bs * world / 64.0, ema=(rank == 0))
    dp = DataParallel(model, opt, algo=args.dp_algo, wire=args.dp_wire) if world > 1 else None
    if dp and args.dp_serial_phase2:
        dp.reducer.overlap_phase2 = False

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.rand(args.bs, 3, args.size, args.size, device=dev, generator=g)
    tgts = torch.randint(0, 12, (args.bs, out_hw, out_hw), device=dev, generator=g)

    def step():
        opt.zero_grad()
        if dp:
            dp.begin()
        out = model(imgs)
        loss, items = crit(out, tgts)
        loss.backward()
        scale = dp.finish() if dp else 1.0
        opt.step(grad_scale=scale)
        return items

    for _ in range(args.warmup):
        step()
    mode = "eager"

    def quick_ms(fn, n=6):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    rstep = None
    if not args.eager:
        # Launch-mode selection (outside the timed region).  Three ways to issue the same kernels:
        #   eager   : ~330 entry-point calls per step from Python on two HIP streams (weight gradients + dead head branch overlap the
        #             main chain); 6-7 ms of host time per step on an idle CPU
        #   replay  : the same calls, streams and event edges recorded once and re-issued from C with one call per step
        #             (yolo_dual_amd/replay.py): eager's overlap at about a millisecond of host time
        #   hipgraph: (--try-hipgraph) the step captured into HIP graphs on one stream: no host time, no overlap
        # Each is measured for a few steps and the fastest runs the timed region; a failed recording / capture falls back to eager.
        # With several ranks the choice must be the SAME everywhere: timings are max-reduced, a failure on any rank disables the mode.
        def agree(vals):
            if world == 1:
                return vals
            tt = torch.tensor(vals, device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return [float(v) for v in tt.tolist()]

        BIG = 1e9
        t_eager = quick_ms(step)
        rstep, t_replay, failed = None, BIG, 0.0
        try:
            from yolo_dual_amd.replay import ReplayedTrainStep
            rstep = ReplayedTrainStep(model, crit, opt, imgs, tgts, dp=dp, warmup=1, prioritize=bool(args.prioritize))
        except Exception as e:          # pragma: no cover
            failed = 1.0
            print(f"[bench] launch-list recording unavailable on rank {rank}: {e!r}", file=sys.stderr)
            torch.cuda.synchronize()
        if not agree([failed])[0]:
            rstep.host_run_s = rstep.host_launch_s = rstep.host_finish_s = 0.0
            t_replay = quick_ms(rstep.step)
            if world > 1 and rank == 0:        # where a replayed data-parallel step spends its host time (rehearsal diagnostics)
                print(f"[bench] replayed step, host ms/step over 6 steps: list segments {rstep.host_run_s / 6 * 1e3:.2f}, bucket launches "
                      f"{rstep.host_launch_s / 6 * 1e3:.2f}, wait for the buckets {rstep.host_finish_s / 6 * 1e3:.2f}", file=sys.stderr)
        gstep, t_graph = None, BIG
        if args.try_hipgraph:
            failed = 0.0
            try:
                from yolo_dual_amd.graph import GraphedTrainStep
                ydl.config.set_overlap_wgrad(bool(args.graph_overlap) and not args.no_overlap)
                gstep = GraphedTrainStep(model, crit, opt, imgs, tgts, dp=dp, warmup=2)
            except Exception as e:          # pragma: no cover
                failed = 1.0
                print(f"[bench] graph capture unavailable on rank {rank}: {e!r}", file=sys.stderr)
                torch.cuda.synchronize()
            finally:
                ydl.config.set_overlap_wgrad(not args.no_overlap)
            if not agree([failed])[0]:
                t_graph = quick_ms(gstep.step)
        t_eager, t_replay, t_graph = agree([t_eager, t_replay, t_graph])
        best = min(t_eager, t_replay, t_graph)
        if best == t_replay and t_replay < BIG:
            eager_step, step, mode = step, rstep.step, "replay"
            gstep = None
        elif best == t_graph and t_graph < BIG:
            eager_step, step, mode = step, gstep.step, "hipgraph"
            rstep = None
        else:
            # drop the recorded list / captured graphs and their private memory pools, then re-warm the eager path: its first steps
            # after that re-grow the caching allocator's per-stream pools (hundreds of ms of hipMalloc on the big workloads)
            gstep = rstep = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            for _ in range(3):
                step()
        if rank == 0:
            fmt = lambda v: "n/a" if v >= BIG else f"{v:.2f}"
            print(f"[bench] ms/step: eager {fmt(t_eager)}, replay {fmt(t_replay)}, hipgraph {fmt(t_graph)} -> {mode}", file=sys.stderr)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    if rstep is not None:
        rstep.host_run_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = step()
    t_enq = time.perf_counter() - t0          # host time to enqueue the K steps (no sync inside the loop); with the launch-list replay this
    fence()                                   # includes the optimizer's wait on its 8-deep hyper-parameter ring, i.e. the host being
    dt = time.perf_counter() - t0             # throttled to the GPU's pace — host_replay_ms_per_step is the time inside ydl_replay_run alone
    host_replay_ms = rstep.host_run_s / args.steps * 1e3 if rstep is not None else None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(items[0])
    ips = args.bs * world * args.steps / dt

    roof = None
    if rank == 0 and not args.no_roofline:
        # instrumented pass: every C-ABI launch bracketed by events on the stream it is launched on.  The second
        # stream is switched off here so that a kernel's duration is its own (in the timed region above wgrad and the
        # dead head branch overlap with the main chain, which inflates per-kernel durations but shortens the step)
        overlap_was = ydl.config.overlap_wgrad()
        ydl.config.set_overlap_wgrad(False)
        # rank-local: the other ranks are already past the timed region, so no collective may be issued here
        if dp:
            dp.reducer.enabled = False

        def step():          # per-kernel events need individual launches (also when the timed region replayed graphs)
            opt.zero_grad()
            out = model(imgs)
            loss, _items = crit(out, tgts)
            loss.backward()
            opt.step(grad_scale=1.0)

        step()
        L.profile_begin()
        for _ in range(3):
            step()
        rec = L.profile_end()
        ydl.config.set_overlap_wgrad(overlap_was)
        if args.profile_json:
            with open(args.profile_json, "w") as fh:
                json.dump(rec, fh)
        fam = {}
        for r in rec:
            f = fam.setdefault(r["name"], {"ms": 0.0, "flops": 0.0, "n": 0, "bytes": 0.0})
            f["ms"] += r["ms"]; f["flops"] += r["flops"]; f["n"] += 1; f["bytes"] += r.get("bytes", 0.0)
        tot_ms = sum(f["ms"] for f in fam.values())
        # headline = the family with the largest GPU time in THIS run, whichever it is (every family's own figure is in "by_kernel_roofline")
        dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
        name, f = dom
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.json")))     # the newest round's counter summary
        tpath = cands[-1] if cands else ""
        if tpath and args.workload == "cfg2":      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this same command
            try:
                tj = json.load(open(tpath))
                ig = [k for k in tj["kernels"] if "igemm" in k["kernel"] or "pw_kernel" in k["kernel"]]
                n_l = sum(k["launches_per_step"] for k in ig)
                traffic = {"unit": "MB per launch (conv fwd+dgrad launches: igemm / igemm2 / pw kernels; rocprofv3 --pmc FETCH_SIZE x2 per the "
                                   "gfx950 note + WRITE_SIZE)",
                           "source": "committed profile profiles/" + os.path.basename(tpath) + " (separate --pmc passes of this same command, "
                                     "collected by tools/collect_profiles.sh; NOT measured in this run)",
                           "value": sum(k["fetch_MB"] + k["write_MB"] for k in ig) / max(n_l, 1)}
            except Exception:
                traffic = None
        peak = PEAK_BF16 if args.dtype == "bf16" else PEAK_F32
        if f["flops"] > 0:
            ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": traffic, "launches": f["n"], "avg_launch_ms": f["ms"] / f["n"],
                    "share_of_gpu_time": f["ms"] / tot_ms}
            if traffic is not None:         # the same launches' algorithmic bytes (x, y, w once), for comparison with the PMC figure
                cf = [fam[k] for k in ("ydl_conv_fwd", "ydl_conv_dgrad") if k in fam]
                traffic["algorithmic_MB_per_launch"] = sum(c["bytes"] for c in cf) / max(sum(c["n"] for c in cf), 1) / 1e6
        else:
            # HBM-bound family: algorithmic bytes (every input read once, every output written once) over its time
            ach = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f.get("bytes") else None
            roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": PEAK_HBM, "unit": "GB/s",
                    "frac": ach / PEAK_HBM if ach else None,
                    "traffic": None, "launches": f["n"], "avg_launch_ms": f["ms"] / f["n"],
                    "share_of_gpu_time": f["ms"] / tot_ms}
        roof["by_kernel_ms_per_step"] = {k: round(v["ms"] / 3, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        # every family above 3 % of the GPU time against its own roofline: MFMA (dense peak of the dtype) where it counts FLOPs,
        # HBM (algorithmic bytes: every input read once, every output written once) otherwise
        bk = {}
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
            if v["ms"] < 0.03 * tot_ms:
                continue
            if v["flops"] > 0:
                a_ = v["flops"] / (v["ms"] * 1e-3) / 1e12
                bk[k] = {"bound": "mfma", "achieved": round(a_, 1), "unit": "TFLOP/s", "frac": round(a_ / peak, 4)}
            elif v.get("bytes"):
                a_ = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                bk[k] = {"bound": "hbm", "achieved": round(a_, 1), "unit": "GB/s", "frac": round(a_ / PEAK_HBM, 4)}
        roof["by_kernel_roofline"] = bk
        # conv FLOP per image of one step: the canonical figure of the workload, or the executed conv FLOPs of the instrumented pass
        exec_flop_img = sum(f2["flops"] for f2 in fam.values()) / 3.0 / args.bs
        flop_img = wl["flop_img"] if (wl["flop_img"] and args.size == wl["size"]) else exec_flop_img
        roof["flop_per_image"] = flop_img
        roof["flop_per_image_executed"] = exec_flop_img
        roof["end_to_end_mfma_frac"] = ips / world * flop_img / 1e12 / peak

    parity = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_parity_leg:
        rstep = gstep = None                # the recorded step's private pool is not needed any more
        torch.cuda.synchronize()
        try:
            parity = parity_leg(ydl, args, wl, dev, imgs, tgts, out_hw)
        except Exception as e:              # pragma: no cover  (the leg must never take the headline down with it)
            parity = {"dtype": "f32", "error": repr(e)}
            torch.cuda.synchronize()

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(wl, 2, args.size, args.cpu_steps if args.size <= 640 else 1)

    if rank == 0:
        print(json.dumps({
            "metric": f"images/sec at {args.size}x{args.size} bs={args.bs}/GPU", "value": ips, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic", "launch_mode": mode, "loss": loss_val, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "host_replay_ms_per_step": host_replay_ms,
            "config": {"workload": f"{wl['desc']}, fwd+bwd+SGD/EMA step, {args.size}x{args.size}, bs={args.bs}/GPU, "
                                   f"CE+0.5*{wl['loss'].capitalize()}, 12 classes ({wl['base']})", "name": args.workload,
                       "global_batch": args.bs * world, "parallelism": f"dp{world}"},
            "roofline": roof, "parity_mode": parity, "cpu_baseline": base}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
