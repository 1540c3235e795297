"""dev tool: eager vs HIP-graph loss trajectories of the yolov5 model at 128x128 (prints both)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import yolo_dual_amd as ydl
from yolo_dual_amd.graph import GraphedTrainStep
from oracle.fill import fill_state_dict

CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)
cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", "yolov5_seg.yaml")))
for sec in ("backbone", "head"):
    for l in cfg[sec]:
        l[2] = "C3" if l[2] == "C3_DCN" else l[2]
ydl.set_compute_dtype(os.environ.get("DT", "bf16"))
for mode in ("eager", "graph"):
    m = ydl.YOLOv5Seg(cfg)
    m.img_size = [128, 128]
    sd = m.state_dict()
    fill_state_dict(sd, 5, bn_stats=False)
    m.load_state_dict(sd)
    m = m.cuda().train()
    opt = ydl.FlatSGDEMA(m, lr=0.01, momentum=0.937, weight_decay=5e-4)
    crit = ydl.SegmentationLoss(12, 0.0, CW, "dice", sync=False)
    gen = torch.Generator("cuda").manual_seed(3)
    x = torch.rand(4, 3, 128, 128, device="cuda", generator=gen)
    t = torch.randint(0, 12, (4, 128, 128), device="cuda", generator=gen)
    rec = []
    if mode == "eager":
        for _ in range(6):
            opt.zero_grad()
            total, items = crit(m(x), t)
            total.backward()
            opt.step()
            rec.append(float(items[0]))
    else:
        g = GraphedTrainStep(m, crit, opt, x, t, warmup=2)
        rec = [None, None]
        for _ in range(4):
            items = g.step()
            rec.append(float(items[0]))
    print(mode, rec, "params finite:", bool(torch.isfinite(opt.params_arena).all()), flush=True)
