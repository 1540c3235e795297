#!/usr/bin/env python3
"""Dev tool (GPU): per-parameter relative-L2 distance of the bf16 (throughput) gradients from the f32 (parity) gradients of one
forward/backward of a model at a realistic size — the data behind the per-layer bounds of tests/test_gpu_model.py::test_bf16_tracks_f32.
usage: python tools/bf16_layer_err.py [yolov5|yolov9] [size] [bs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yaml

import yolo_dual_amd as ydl
from oracle.fill import fill_state_dict

CW = torch.tensor([1, 2, 25, 2, 10, 3, 25, 10, 5, 15, 25, 1], dtype=torch.float32)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "yolov5"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfgf = {"yolov5": "yolov5_seg.yaml", "yolov9": "yolov9_seg.yaml"}[arch]
    cfg = yaml.safe_load(open(os.path.join(ROOT, "yolo_dual_amd", "cfg", cfgf)))
    for sec in ("backbone", "head"):
        for l in cfg[sec]:
            l[2] = {"C3_DCN": "C3", "C2f_DCN": "C2f"}.get(l[2], l[2])
    res = {}
    for mode in ("f32", "bf16"):
        ydl.set_compute_dtype(mode)
        m = getattr(ydl, {"yolov5": "YOLOv5Seg", "yolov9": "YOLOv9Seg"}[arch])(cfg)
        m.img_size = [size, size]
        sd = m.state_dict()
        fill_state_dict(sd, 99, bn_stats=False)
        m.load_state_dict(sd)
        m = m.cuda().train()
        crit = ydl.SegmentationLoss(12, 0.0, CW, "dice")
        gen = torch.Generator("cuda").manual_seed(5)
        x = torch.rand(bs, 3, size, size, device="cuda", generator=gen)
        t = torch.randint(0, 12, (bs, size, size), device="cuda", generator=gen)
        out = m(x)
        total, items = crit(out, t)
        total.backward()
        res[mode] = (out.detach().float().cpu(), items, {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()
                                                           if getattr(p, "_ydl_touched", False)})
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    print(f"probabilities l2 {l2(res['bf16'][0], res['f32'][0]):.3e}  loss {res['bf16'][1][0]:.6f} vs {res['f32'][1][0]:.6f}")
    for k in res["f32"][2]:
        print(f"{k:48s} {l2(res['bf16'][2][k], res['f32'][2][k]):.3e}")


if __name__ == "__main__":
    main()
