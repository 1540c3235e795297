#!/usr/bin/env python3
"""dev tool (GPU): every call of ONE entry point during an eager training step of a bench workload, with the Python lines that asked
for it — to see which launches could be avoided.   usage: python tools/entry_calls.py <workload> <entry point> [more entry points]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_dual_amd as ydl
from yolo_dual_amd import _lib as L
import bench

wname, names = sys.argv[1], set(sys.argv[2:])
wl = bench.WORKLOADS[wname]
model = getattr(ydl, wl["model"])(bench.load_cfg(wl["yaml"], wl["swap"])).cuda().train()
S = wl["size"]
model.img_size = [S, S]
crit = ydl.SegmentationLoss(12, 0.0, torch.tensor(bench.CW, dtype=torch.float32), wl["loss"], sync=False)
opt = ydl.FlatSGDEMA(model, lr=0.01, momentum=0.937, weight_decay=5e-4, ema=True)
x = torch.rand(wl["bs"], 3, S, S, device="cuda")
t = torch.randint(0, 12, (wl["bs"], S, S), device="cuda")


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
    if name in names:
        fr = [f for f in traceback.extract_stack()[:-1] if "yolo_dual_amd" in f.filename][-4:]
        ints = [a for a in args if isinstance(a, int)]
        log.append((name, ints[:8], " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr))))
    return orig(name, *args)


L.call = call
import yolo_dual_amd.tape as T, yolo_dual_amd.optim as O, yolo_dual_amd.modules as M
for mod in (T, O, M):
    if hasattr(mod, "L"):
        mod.L.call = call
step()
torch.cuda.synchronize()
for e in log:
    print(f"{e[0]:16s} ints {e[1]}  {e[2]}")
