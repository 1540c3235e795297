#!/usr/bin/env python3
"""dev tool (GPU): every zero-fill entry-point call of one eager training step of BASELINE config 2, with its size and the Python
line that asked for it — to see which ones could be merged or moved off the critical stream."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_dual_amd as ydl
from yolo_dual_amd import _lib as L
import bench

wl = bench.WORKLOADS["cfg2"]
model = getattr(ydl, wl["model"])(bench.load_cfg(wl["yaml"], wl["swap"])).cuda().train()
model.img_size = [640, 640]
crit = ydl.SegmentationLoss(12, 0.0, torch.tensor(bench.CW, dtype=torch.float32), wl["loss"], sync=False)
opt = ydl.FlatSGDEMA(model, lr=0.01, momentum=0.937, weight_decay=5e-4 * 16 / 64.0, ema=True)
x = torch.rand(16, 3, 640, 640, device="cuda")
t = torch.randint(0, 12, (16, 640, 640), device="cuda")


def step():
    opt.zero_grad()
    loss, _ = crit(model(x), t)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
orig = L.call
log = []


def call(name, *args):
    if name in ("ydl_fill_zero", "ydl_zero2d"):
        fr = [f for f in traceback.extract_stack()[:-1] if "yolo_dual_amd" in f.filename][-3:]
        size = args[1] if name == "ydl_fill_zero" else args[3] * args[4] * 2
        log.append((name, int(size), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)), int(args[-1].value or 0) if hasattr(args[-1], "value") else 0))
    return orig(name, *args)


L.call = call
import yolo_dual_amd.tape as T, yolo_dual_amd.optim as O, yolo_dual_amd.modules as M
for mod in (T, O, M):
    if hasattr(mod, "L"):
        mod.L.call = call
step()
torch.cuda.synchronize()
for e in log:
    print(f"{e[0]:14s} {e[1] / 1e6:9.3f} MB  stream {e[3]:#x}  {e[2]}")
