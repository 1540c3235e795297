#!/usr/bin/env python3
"""Generate tests/golden/train_curve_yolov5seg_128.npz: the CPU oracle's training run that the GPU training-parity test replays.

TEST INFRASTRUCTURE (build container; ~2 minutes on 8 cores).  YOLOv5Seg (script C3 blocks, yolov5_seg.yaml) from a seeded
fill, 128x128 synthetic blobby masks, batch 8 cycling over 16 batches, CE + 0.5*Dice with weight.yaml's class
weights, SGD-nesterov lr 0.02 decaying linearly to 0.001 / momentum 0.937, 600 steps, then the mIoU of val_diceloss.py:37-75 (eval-mode BN) on a
held-out batch.  Stored: per-step losses, the validation mIoU every 25 steps, the final per-class IoUs.  The oracle
(oracle/ref_cpu.py) is pinned to the reference's own classes by tests/test_oracle_golden.py."""
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402
from oracle.fill import fill_state_dict  # noqa: E402
from tests.model_shapes import script_model_state_shapes  # noqa: E402

S, BS, STEPS, LR, NB = 128, 8, 600, 0.02, 16
NVAL = 8            # held-out batches of the final evaluation (8 x 8 = 64 images, seeds 2..9)
LRF = 0.05          # linear decay of the learning rate to LRF * LR over the run (the reference's LambdaLR shape, :976-980)
CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)


def blobby(seed, n):
    """8x8 random class grid nearest-upsampled to SxS; the image is a class-dependent colour plus noise"""
    rs = np.random.RandomState(seed)
    grid = torch.from_numpy(rs.randint(0, 11, size=(n, 8, 8)).astype(np.int64))
    tgt = grid.repeat_interleave(S // 8, 1).repeat_interleave(S // 8, 2)
    pal = torch.from_numpy(np.random.RandomState(1234).rand(12, 3).astype(np.float32))
    img = pal[tgt].permute(0, 3, 1, 2) * 0.8 + 0.2 * torch.from_numpy(rs.rand(n, 3, S, S).astype(np.float32))
    return img.contiguous(), tgt.contiguous()


def cfg_v5():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = "C3" if l[2] == "C3_DCN" else l[2]
    return cfg


def run(eps: float, verbose: bool = True):
    """one training run; ``eps`` is added to the images of the first batch (the run is chaotic: a 1e-6 perturbation moves the
    final mIoU by several points, so the fixture stores a small ensemble)"""
    cfg = cfg_v5()
    shapes = script_model_state_shapes(cfg)
    sd = {k: (torch.zeros(s) if not k.endswith("num_batches_tracked") else torch.zeros((), dtype=torch.int64)) for k, s in shapes.items()}
    fill_state_dict(sd, 77, bn_stats=False)
    pnames = [k for k in sd if k.endswith(".weight") or k.endswith(".bias")]
    xv, tv = blobby(2, BS)
    bufs, losses, mious = {}, [], []
    held_out = [blobby(2 + i, BS) for i in range(NVAL)]          # NVAL x BS held-out images (seeds never used for training)
    for st in range(STEPS):
        x, t = blobby(100 + st % NB, BS)
        if st % NB == 0:
            x = x + eps
        ps = {k: sd[k].detach().clone().requires_grad_(True) for k in pnames}
        run = dict(sd)
        run.update(ps)
        out = R.script_model_forward(run, cfg, x, (S, S))
        total, _, _ = R.seg_loss(out, t, CW, "dice")
        total.backward()
        losses.append(float(total.detach()))
        for k in pnames:
            if ps[k].grad is not None:
                bufs[k] = R.sgd_nesterov_step(sd[k], ps[k].grad, bufs.get(k), LR * (1.0 - (1.0 - LRF) * st / STEPS), 0.937, 0.0)
        for k in sd:
            if k not in ps:
                sd[k] = run[k]
        if st % 25 == 24:
            with torch.no_grad():
                pv = R.script_model_forward({k: v.clone() for k, v in sd.items()}, cfg, xv, (S, S), train=False)
            miou, ious = R.miou_from_confusion(R.confusion_matrix(pv.argmax(1), tv, 12))
            mious.append(miou)
            if verbose:
                print(f"eps {eps:g} step {st + 1}: loss {losses[-1]:.4f}  val mIoU {miou:.4f}", flush=True)
    # final evaluation on the large held-out set: one confusion matrix over all NVAL batches (val_diceloss.py:216-256 accumulates
    # the matrix over the loader the same way)
    cm = None
    with torch.no_grad():
        for xh, th in held_out:
            ph = R.script_model_forward({k: v.clone() for k, v in sd.items()}, cfg, xh, (S, S), train=False)
            c = R.confusion_matrix(ph.argmax(1), th, 12)
            cm = c if cm is None else cm + c
    miou64, ious64 = R.miou_from_confusion(cm)
    if verbose:
        print(f"eps {eps:g}: final val mIoU on {NVAL * BS} held-out images {miou64:.4f} (first batch alone {mious[-1]:.4f})", flush=True)
    return np.array(losses), np.array(mious), np.array(ious), float(miou64), np.array(ious64)


EPS = (0.0, 1e-6, -1e-6, 2e-6)


def main():
    """``--member i`` runs ensemble member i into a part file (so that members can run in parallel processes); without
    arguments the parts are merged into the fixture (missing members are computed first)"""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--member", type=int, default=-1)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 8))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    part = lambda i: os.path.join("/tmp", f"train_curve_part{i}.npz")
    if a.member >= 0:
        l, m, io, m64, io64 = run(EPS[a.member])
        np.savez(part(a.member), losses=l, mious=m, ious=io, miou64=m64, ious64=io64)
        return
    L, M, I, M64, I64 = [], [], [], [], []
    for i, e in enumerate(EPS):
        if not os.path.exists(part(i)):
            l, m, io, m64, io64 = run(e)
            np.savez(part(i), losses=l, mious=m, ious=io, miou64=m64, ious64=io64)
        z = np.load(part(i))
        L.append(z["losses"]); M.append(z["mious"]); I.append(z["ious"]); M64.append(float(z["miou64"])); I64.append(z["ious64"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_curve_yolov5seg_128.npz"), losses=L[0], mious=M[0],
                        final_ious=I[0], ens_eps=np.array(EPS), ens_mious=np.stack(M), ens_final=np.array([m[-1] for m in M]),
                        ens_final64=np.array(M64), ens_ious64=np.stack(I64), hyp=np.array([S, BS, STEPS, LR, NB, LRF, NVAL]))
    print("ensemble final mIoU (8 held-out images):", [round(float(m[-1]), 4) for m in M])
    print("ensemble final mIoU (64 held-out images):", [round(v, 4) for v in M64])


if __name__ == "__main__":
    main()
