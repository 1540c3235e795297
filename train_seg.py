#!/usr/bin/env python3
"""Training entry point with the reference's ``--cfg / --weights`` surface on the MI355X path.

Mirrors the argument list and the loop of unet-lite/yolo5-seg/seg_diceloss_yolov5.py (:1235-1287 CLI, :940-952 intersect-load of
``--weights``, :966-980 accumulate / weight-decay scaling / LambdaLR, :1084-1103 hot loop, :1204-1212 checkpoint dict, :1229
strip_optimizer).  What the reference does around the hot path — JSON datasets, augmentation, TensorBoard, early stopping — is outside this
repository's scope: batches are synthetic "blobby" masks generated on the device, validation is the
confusion-matrix mIoU of val_diceloss.py:37-75 on held-out synthetic batches.

Multi-GPU: where the reference wraps the model in nn.DataParallel (:988-992; its SyncBN branch is dead code), this entry runs one
process per GPU under torch.distributed.run and averages gradients with yolo_dual_amd.parallel.DataParallel (RCCL all-reduce of the
flat gradient arena, bucketed, launched from the backward hooks).  ``--batch-size`` is per GPU, the nominal-batch scaling of the
reference (accumulate = round(64 / total batch), weight decay x total batch x accumulate / 64, seg_diceloss_yolov5.py:970-972) uses the
total over all ranks, as the reference does with its batch_size before splitting it (:1001); BatchNorm statistics stay per
replica and rank 0's are the ones validated and saved, as with nn.DataParallel, whose replica 0 owns the buffers.

    python train_seg.py --cfg yolo_dual_amd/cfg/yolov5_seg.yaml --weights '' --epochs 2 --batch-size 16 --imgsz 640
    python train_seg.py --weights runs/train-seg/last.pt --resume
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_seg.py --batch-size 16 ...
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_CW = [1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1]            # unet-lite/yolo5-seg/weight.yaml:3-14
ARCH = {"yolov5": ("YOLOv5Seg", "yolov5_seg.yaml", "dice"), "yolov8": ("YOLOv8Seg", "yolov8_seg.yaml", "jaccard"),
        "yolov9": ("YOLOv9Seg", "yolov9_seg.yaml", "dice"), "resnet18": ("ResNet18Seg", None, "dice"),
        "resnet50": ("ResNet50Seg", None, "dice")}


LAST_RUN: dict = {}      # facts of the last train() call of this process (tests): world, total batch, accumulate, weight decay, parameter sums


def parse_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--weights", type=str, default="", help="initial weights (.pt with a state_dict under 'model'); '' = from scratch")
    p.add_argument("--cfg", type=str, default="", help="model yaml (default: the architecture's yaml under yolo_dual_amd/cfg)")
    p.add_argument("--arch", default="yolov5", choices=sorted(ARCH), help="which of the reference's seg scripts to mirror")
    p.add_argument("--data", type=str, default="synthetic", help="ignored: batches are synthetic (no dataset ships with the reference)")
    p.add_argument("--epochs", type=int, default=300)
    p.add_argument("--batch-size", type=int, default=4)
    p.add_argument("--imgsz", "--img", "--img-size", type=int, default=640)
    p.add_argument("--device", default="", help="cuda device index (the HIP path has no CPU fallback)")
    p.add_argument("--freeze", nargs="+", type=int, default=[0], help="freeze the first N backbone layers (or the listed ones)")
    p.add_argument("--cos-lr", action="store_true")
    p.add_argument("--resume", nargs="?", const=True, default=False)
    p.add_argument("--class-weights", type=str, default="", help="comma separated list, a yaml file with a 'weights' list, or '' for weight.yaml's values")
    p.add_argument("--save-dir", type=str, default="runs/train-seg")
    p.add_argument("--noval", action="store_true")
    p.add_argument("--nosave", action="store_true")
    p.add_argument("--optimizer", type=str, default="SGD")
    p.add_argument("--label-smoothing", type=float, default=0.0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="compute dtype of the HIP kernels")
    p.add_argument("--steps-per-epoch", type=int, default=50, help="synthetic batches per epoch")
    p.add_argument("--lr0", type=float, default=0.01)
    p.add_argument("--lrf", type=float, default=0.2)
    p.add_argument("--momentum", type=float, default=0.937)
    p.add_argument("--weight-decay", type=float, default=0.0005)
    p.add_argument("--raw-size", type=str, default="", help="WxH: synthetic samples are generated as uint8 arrays of this size and go "
                   "through the GPU letterbox (yolo_dual_amd.data.LetterboxGPU = the dataset's _resize_and_pad + /255)")
    p.add_argument("--dist-backend", default="nccl", help="torch.distributed backend when WORLD_SIZE > 1 (nccl = RCCL; gloo for rehearsals)")
    p.add_argument("--dp-algo", default="allreduce", choices=["allreduce", "rs_ag"])
    p.add_argument("--dp-serial-phase2", action="store_true", help="rs_ag: all-gathers at the end of backward, not overlapped with it")
    p.add_argument("--dp-wire", default="f32", choices=["f32", "bf16"])
    p.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --dist-backend gloo)")
    p.add_argument("--emulate-world", type=int, default=0, help="debugging aid (single process): play N data-parallel ranks in turn — rank "
                   "r's batches, gradients summed over the ranks and averaged in the step; parameters then equal an N-rank run's to rounding")
    return p.parse_args(argv)


def class_weights(spec: str, nc: int):
    import torch
    import yaml
    if not spec:
        vals = DEFAULT_CW
    elif os.path.exists(spec):
        d = yaml.safe_load(open(spec))
        vals = d["weights"] if isinstance(d, dict) and "weights" in d else list(d.values())[0] if isinstance(d, dict) else d
    else:
        vals = [float(v) for v in spec.split(",")]
    if len(vals) != nc:
        raise ValueError(f"class weights: expected {nc} values, got {len(vals)}")
    return torch.tensor(vals, dtype=torch.float32)


def blobby_batch(gen, n: int, size: int, nc: int, device, palette, letterbox=None, raw=None):
    """8x8 random class grid nearest-upsampled to size x size; the image is a class colour plus noise.
    With ``letterbox`` the samples are made as uint8 HWC arrays of the raw size and prepared like the reference's dataset does."""
    import torch
    if letterbox is not None:
        rw, rh = raw
        grid = torch.randint(0, nc - 1, (n, 8, 8), device=device, generator=gen)
        tgt = grid.repeat_interleave((rh + 7) // 8, 1).repeat_interleave((rw + 7) // 8, 2)[:, :rh, :rw]
        img = palette[tgt] * 0.8 + 0.2 * torch.rand(n, rh, rw, 3, device=device, generator=gen)
        u8 = (img * 255).to(torch.uint8)
        return letterbox.batch([u8[i] for i in range(n)], [tgt[i].to(torch.uint8) for i in range(n)])
    grid = torch.randint(0, nc - 1, (n, 8, 8), device=device, generator=gen)
    rep = (size + 7) // 8
    tgt = grid.repeat_interleave(rep, 1).repeat_interleave(rep, 2)[:, :size, :size].contiguous()
    img = palette[tgt].permute(0, 3, 1, 2) * 0.8 + 0.2 * torch.rand(n, 3, size, size, device=device, generator=gen)
    return img.contiguous(), tgt


def build_model(opt):
    import yaml
    import yolo_dual_amd as ydl
    cls_name, default_yaml, loss_kind = ARCH[opt.arch]
    if default_yaml is None:
        return getattr(ydl, cls_name)({"nc": 12}), loss_kind
    path = opt.cfg or os.path.join(ROOT, "yolo_dual_amd", "cfg", default_yaml)
    cfg = yaml.safe_load(open(path))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            if l[2] in ("C3_DCN", "C2f_DCN"):          # torchvision DeformConv2d blocks: parity unpinned, substituted like the benchmark
                print(f"[train_seg] {l[2]} -> {l[2][:-4]} (torchvision deformable conv is not part of the reference; SURVEY §8c)")
                l[2] = l[2][:-4]
    model = getattr(ydl, cls_name)(cfg)
    model.img_size = [opt.imgsz, opt.imgsz]
    return model, loss_kind


def train(opt) -> float:
    import torch
    import yolo_dual_amd as ydl
    from torch.optim import lr_scheduler

    # one process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE; the launcher starts before any GPU call)
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        from yolo_dual_amd.parallel import pin_rank_to_cores
        pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if not torch.cuda.is_available():
        raise RuntimeError("train_seg.py runs on the GPU only (yolo_dual_amd has no CPU fallback)")
    device = torch.device("cuda", 0 if opt.one_gpu else (local if world > 1 else (int(opt.device) if str(opt.device).isdigit() else 0)))
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if not dist.is_initialized():
            dist.init_process_group(opt.dist_backend, rank=rank, world_size=world)
    main = rank == 0
    torch.manual_seed(opt.seed)
    ydl.set_compute_dtype(opt.dtype)
    os.makedirs(opt.save_dir, exist_ok=True)
    last, best = os.path.join(opt.save_dir, "last.pt"), os.path.join(opt.save_dir, "best.pt")
    if opt.resume and not opt.weights:
        opt.weights = last

    model, loss_kind = build_model(opt)
    nc = model.num_classes
    ckpt = None
    if opt.weights.endswith(".pt"):                                   # seg_diceloss_yolov5.py:944-952
        ckpt = ydl.load_checkpoint(opt.weights)
        n, tot = ydl.load_weights(model, ckpt)
        if main:
            print(f"[train_seg] loaded weights: {n}/{tot} entries match")
    model = model.to(device)
    freeze = [f"backbone.{x}." for x in (opt.freeze if len(opt.freeze) > 1 else range(opt.freeze[0]))]   # :955-959
    for k, v in model.named_parameters():
        v.requires_grad = not any(x in k for x in freeze)

    bs, epochs = opt.batch_size, opt.epochs
    emu = max(int(opt.emulate_world), 0) if world == 1 else 0           # debugging aid: one process plays the ranks of a DP run in turn
    tb = bs * (emu or world)                                            # the reference's batch_size is the TOTAL over the replicas (:1001)
    nbs = 64
    accumulate = max(round(nbs / tb), 1)                               # :970-972, on the total batch
    wd = opt.weight_decay * tb * accumulate / nbs
    LAST_RUN.clear()
    LAST_RUN.update(world=world, total_batch=tb, accumulate=accumulate, weight_decay=wd)
    optimizer = ydl.smart_optimizer(model, opt.optimizer, opt.lr0, opt.momentum, wd, ema=main)       # EMA is fused into the step
    dp = None
    if world > 1:                                                                        # :988-992, as processes instead of threads
        from yolo_dual_amd.parallel import DataParallel
        dp = DataParallel(model, optimizer, algo=opt.dp_algo, wire=opt.dp_wire)          # broadcasts rank 0's parameters and buffers
        if opt.dp_serial_phase2:
            dp.reducer.overlap_phase2 = False
    if opt.cos_lr:
        lf = lambda x: ((1 - math.cos(x * math.pi / epochs)) / 2) * (opt.lrf - 1) + 1     # one_cycle(1, lrf, epochs)
    else:
        lf = lambda x: (1 - x / epochs) * (1.0 - opt.lrf) + opt.lrf
    scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)
    best_fitness, start_epoch = 0.0, 0
    if ckpt is not None and opt.resume:
        best_fitness, start_epoch, epochs = ydl.smart_resume(ckpt, optimizer, optimizer if main else None, opt.weights, epochs, True)
        scheduler.last_epoch = start_epoch - 1

    cw = class_weights(opt.class_weights, nc).to(device)
    criterion = (ydl.SegmentationLoss if loss_kind == "dice" else ydl.JaccardSegmentationLoss)(nc, opt.label_smoothing, cw)
    gen = torch.Generator(device=device).manual_seed(1000 + opt.seed + 7919 * rank)     # every rank draws its own batches
    emu_gens = [torch.Generator(device=device).manual_seed(1000 + opt.seed + 7919 * r) for r in range(emu)]
    palette = torch.rand(nc, 3, device=device, generator=torch.Generator(device=device).manual_seed(7))
    val_gen = torch.Generator(device=device).manual_seed(99)
    lb, raw = None, None
    if opt.raw_size:
        raw = tuple(int(v) for v in opt.raw_size.lower().split("x"))
        lb = ydl.LetterboxGPU(opt.imgsz, num_classes=nc, device=device)
    val_batches = [blobby_batch(val_gen, bs, opt.imgsz, nc, device, palette, lb, raw) for _ in range(2)]

    t0 = time.time()
    for epoch in range(start_epoch, epochs):
        model.train()
        mloss = torch.zeros(3)
        optimizer.zero_grad()
        for i in range(opt.steps_per_epoch):
            stepping = (i + 1) % accumulate == 0 or i == opt.steps_per_epoch - 1       # :1095-1103
            if dp is not None:
                # the arena accumulates local gradients over the micro-batches; ranks exchange them once, on the step that applies them
                dp.reducer.enabled = stepping
                if stepping:
                    dp.begin()
            for g_r in (emu_gens if emu else [gen]):                                    # (emulation: the ranks' micro-batches in turn)
                imgs, targets = blobby_batch(g_r, bs, opt.imgsz, nc, device, palette, lb, raw)
                pred = model(imgs)                                                      # :1084-1092
                loss, items_r = criterion(pred, targets)
                loss.backward()
                if g_r is (emu_gens[0] if emu else gen):
                    loss_items = items_r
            if stepping:
                optimizer.step(grad_scale=dp.finish() if dp is not None else (1.0 / emu if emu else 1.0))
                optimizer.zero_grad()
            mloss = (mloss * i + torch.tensor(loss_items)) / (i + 1)
        scheduler.step()
        final_epoch = epoch == epochs - 1
        miou = 0.0
        if main and (not opt.noval or final_epoch):
            # validation on the EMA weights (validate.run(model=ema.ema), :1155): swap them in, evaluate, swap back
            live = {k: v.clone() for k, v in model.state_dict().items()}
            model.load_state_dict(optimizer.ema_state_dict())
            ydl.config.bump_weight_epoch()
            model.eval()
            cm = ydl.ConfusionMatrix(nc, ignore_index=nc - 1)
            with torch.no_grad():
                for xv, tv in val_batches:
                    cm.process_batch(model(xv), tv)
            miou, _ = cm.compute_iou()
            model.load_state_dict(live)
            ydl.config.bump_weight_epoch()
        if main:
            print(f"epoch {epoch + 1}/{epochs}  loss {mloss[0]:.4f} (ce {mloss[1]:.4f}, {loss_kind} {mloss[2]:.4f})  mIoU {miou:.4f}  "
                  f"lr {optimizer.param_groups[1]['lr']:.5f}  {time.time() - t0:.0f}s" + (f"  [{world} ranks]" if world > 1 else ""), flush=True)
        fi = miou                                                          # fitness = mIoU (:1197)
        if fi > best_fitness:
            best_fitness = fi
        if main and ((not opt.nosave) or final_epoch):
            ema_sd = optimizer.ema_state_dict()
            ydl.save_checkpoint(last, ema_sd, optimizer, epoch, best_fitness)           # :1204-1209
            if fi == best_fitness:
                ydl.save_checkpoint(best, ema_sd)                                       # :1211
    if main and os.path.exists(best):
        mb = ydl.strip_optimizer(best)                                                   # :1229
        print(f"[train_seg] best model saved to {best} ({mb:.1f} MB, optimizer stripped)")
    pa = optimizer.params_arena[:optimizer.n_params].double()
    LAST_RUN.update(param_sum=float(pa.sum()), param_abs_sum=float(pa.abs().sum()))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if dp is not None:
            dp.reducer.close()
    return best_fitness


if __name__ == "__main__":
    train(parse_opt())
