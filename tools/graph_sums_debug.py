import os, sys
sys.path.insert(0, "/root/repo")
import torch, yaml
import yolo_dual_amd as ydl
from oracle.fill import fill_state_dict
from yolo_dual_amd.graph import GraphedTrainStep
ydl.set_compute_dtype("bf16")
cfg = yaml.safe_load(open("/root/repo/yolo_dual_amd/cfg/yolov5_seg.yaml"))
for sec in ("backbone", "head"):
    for l in cfg[sec]:
        if l[2] == "C3_DCN": l[2] = "C3"
m = ydl.YOLOv5Seg(cfg); m.img_size = [64, 64]
sd = m.state_dict(); fill_state_dict(sd, 11, bn_stats=False); m.load_state_dict(sd)
m = m.cuda().train()
opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
cw = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
crit = ydl.SegmentationLoss(12, 0.0, cw, "dice", sync=False)
gen = torch.Generator("cuda").manual_seed(100)
x = torch.rand(2, 3, 64, 64, device="cuda", generator=gen); t = torch.randint(0, 12, (2, 64, 64), device="cuda", generator=gen)
g = GraphedTrainStep(m, crit, opt, x, t, warmup=2)
for i in range(4):
    items = g.step(); torch.cuda.synchronize()
    print("graph step", i, [float(v) for v in items], "nan running:", sum(1 for k, v in m.state_dict().items() if "running" in k and not torch.isfinite(v).all()))
